#!/usr/bin/env python3
"""SpMV timing on the cfg3 matrix (run on the GPU box): python tools/spmv_probe.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from matrixextra_amd import device as D, synth
from oracle import oracle as O
p, j, x = synth.csr_fixed(1_000_000, 100_000, 32)
A = D.DeviceCSR.from_host(p, j, x, 100_000)
vh = synth.dense_normal(100_000, 1).reshape(-1)
v = torch.from_numpy(vh).cuda()
y = D.spmv(A, v); torch.cuda.synchronize()
ref = O.matmul_csr_dvec_numeric(p, j, x, vh)
print("max rel err", float(np.max(np.abs(y.cpu().numpy() - ref)) / np.max(np.abs(ref))))
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): D.spmv(A, v)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20
    print(f"spmv {t:.4f} ms  {396.8e6 / (t * 1e-3) / 1e9:.0f} GB/s of algorithmic bytes")
