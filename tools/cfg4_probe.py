#!/usr/bin/env python3
"""BASELINE configs[3] timing: CSR + CSR and CSR * CSR on 2M x 2M, 50/row (nnz 1e8 each, ~50 % shared pattern), operands
resident in HBM (run on the GPU box): python tools/cfg4_probe.py"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from matrixextra_amd import _lib, device as D, synth
m = K = 2_000_000
t0 = time.time()
p1, j1, x1 = synth.csr_fixed(m, K, 50)
p2, j2, x2 = synth.csr_overlapping(p1, j1, K, 50)
print(f"generated in {time.time() - t0:.0f} s", flush=True)
A, B = D.DeviceCSR.from_host(p1, j1, x1, K), D.DeviceCSR.from_host(p2, j2, x2, K)
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"          # the one-pass kernel instead of count -> scan -> fill
for name, op in (("add", _lib.MX_OP_ADD), ("sub", _lib.MX_OP_SUB), ("mul", _lib.MX_OP_MUL)):
    R = D.csr_elemwise(op, A, B, two_pass=not fused); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): R = D.csr_elemwise(op, A, B, two_pass=not fused)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 5 * 1e-3
    byts = 2 * (12 * A.nnz + 4 * (m + 1)) + 12 * R.nnz + 4 * (m + 1)
    print(f"{name}: {t * 1e3:.3f} ms, nnz_out {R.nnz}, {byts / t / 1e9:.0f} GB/s of algorithmic bytes, {(A.nnz + B.nnz) / t / 1e9:.1f} G input-nnz/s")
