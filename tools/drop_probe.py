#!/usr/bin/env python3
"""remove_zero_valued_csr at BASELINE configs[3] size on device-resident arrays (2M rows, 1e8 entries, 30 % zeros):
time per call and per kernel.   [MXGPU_DROP_G=4|8|16|32|64] python tools/drop_probe.py [nnz_row]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matrixextra_amd import _lib, device as D, synth
per_row = int(sys.argv[1]) if len(sys.argv) > 1 else 50
m = K = 2_000_000
p, j, x = synth.device_csr_fixed(m, K, per_row)
gen = torch.Generator(device="cuda"); gen.manual_seed(11)
xz = torch.where(torch.rand(x.numel(), device="cuda", generator=gen) < 0.3, torch.zeros_like(x), x)
A = D.DeviceCSR(p, j, xz, m, K, int(j.numel()))
R = D.csr_drop_zeros(A)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); D.csr_drop_zeros(A); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
byts = 2 * 4 * (m + 1) + 12 * A.nnz + 12 * R.nnz
t = min(ts)
print(f"G={os.environ.get('MXGPU_DROP_G', 'auto')} per_row={per_row}: {t:.3f} ms per call (min of 10), {byts / t / 1e6:.0f} GB/s algorithmic, kept {R.nnz} of {A.nnz}")
