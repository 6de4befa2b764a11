#!/usr/bin/env python3
"""Kernel-only timing of the planned sweep on cfg2 / cfg5-shard without any parity check (for A/B builds whose results
are deliberately wrong, e.g. a plan stream confined to cache):  MXGPU_LIB=... python tools/sweep_time.py [cfg2|cfg5|skew]"""
import sys, ctypes
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from matrixextra_amd import _lib, device as D, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
m, K, n, r, dt = (1_000_000, 100_000, 128, 32, torch.float64) if cfg == "cfg2" else (1_000_000, 200_000, 256, 64, torch.float32)
if cfg == "skew":                                                 # cfg2's shape, log-normal row lengths (sigma 1, same mean)
    m, K, n, r, dt = 1_000_000, 100_000, 128, 32, torch.float64
    hp, hj, hx = synth.csr_skewed_fast(m, K, 32, seed=synth.SEED_A, sigma=1.0)
    A = D.DeviceCSR.from_host(hp, hj, hx, K)
else:
    p, j, x = synth.device_csr_fixed(m, K, r)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
B = torch.randn(K, n, dtype=dt, device="cuda")
C = torch.empty(n, m, dtype=dt, device="cuda")
lib = _lib.load()
for _ in range(int(os.environ.get('SWEEP_SETUP', 12))): D.spmm(A, B, out=C, colmajor=True)
torch.cuda.synchronize()
lib.mxd_spmm_kernel_timing(1)
for _ in range(int(os.environ.get('SWEEP_STEPS', 20))): D.spmm(A, B, out=C, colmajor=True)
torch.cuda.synchronize()
kt = (ctypes.c_float * 256)(); kc = ctypes.c_int(0)
lib.mxd_spmm_kernel_times(kt, 256, ctypes.byref(kc))
t = np.array(kt[:kc.value])
print(cfg, lib.mxd_spmm_last_kernel().decode(), "kernel avg %.4f ms min %.4f ms" % (t.mean(), t.min()))
