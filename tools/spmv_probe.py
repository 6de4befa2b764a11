#!/usr/bin/env python3
"""SpMV kernel comparison on the GPU box: lane-group kernel (1) vs LDS-panel tile kernel (2), several shapes.
python tools/spmv_probe.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from matrixextra_amd import _lib, device as D, synth


def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (m, K, k) in ((1_000_000, 100_000, 32), (1_000_000, 16_000, 32), (1_000_000, 200_000, 64), (2_000_000, 390_000, 16),
                  (4_000_000, 100_000, 8)):
    p, j, x = synth.csr_fixed(m, K, k)
    A = D.DeviceCSR.from_host(p, j, x, K)
    byts = 4 * (m + 1) + 12 * A.nnz + 8 * K + 8 * m
    for dt in (torch.float64, torch.float32):
        v = torch.randn(K, dtype=dt, device="cuda")
        r = {}
        for algo in (1, 2, 3):
            y = D.spmv(A, v, algo=algo)
            r[algo] = (timeit(lambda: D.spmv(A, v, algo=algo)), y)
        err = (r[1][1].double() - r[3][1].double()).abs().max().item()
        print(f"{m}x{K} {k}/row {str(dt)[6:]}: group {r[1][0]*1e3:.1f} us  tile {r[2][0]*1e3:.1f} us  flat {r[3][0]*1e3:.1f} us "
              f"({byts / r[3][0] / 1e6:.0f} GB/s)  maxdiff {err:.2e}", flush=True)
p, j, x = synth.csr_skewed(300_000, 100_000, 32, seed=11, sigma=1.4)
A = D.DeviceCSR.from_host(p, j, x, 100_000)
v = torch.randn(100_000, dtype=torch.float64, device="cuda")
for algo in (1, 2, 3):
    print("skewed sigma 1.4, algo", algo, f"{timeit(lambda: D.spmv(A, v, algo=algo))*1e3:.1f} us", "nnz", A.nnz, "longest row", int(np.diff(p).max()))
