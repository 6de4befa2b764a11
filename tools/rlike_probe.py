#!/usr/bin/env python3
"""Export-level calls with host memory as R provides it: numpy's own MADV_HUGEPAGE switched off, so operands and results
are plain malloc / mmap memory (4-KiB pages where THP is in madvise mode).  CSR + CSR with ~1 GB of results and the
headline product:   [MXGPU_HUGEPAGE=0] python tools/rlike_probe.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
try:
    from numpy._core.multiarray import _set_madvise_hugepage
except ImportError:
    from numpy.core.multiarray import _set_madvise_hugepage
_set_madvise_hugepage(False)
from matrixextra_amd import _lib, exports as G, synth
lib = _lib.load()
m = K = 1_000_000
p1, j1, x1 = synth.csr_fixed(m, K, 50)
p2, j2, x2 = synth.csr_overlapping(p1, j1, K, 50)
for i in range(3):
    lib.mx_cache_invalidate(None)
    t0 = time.perf_counter(); r = G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False); t = time.perf_counter() - t0
    print(f"CSR + CSR (nnz 5e7 each -> {r['indices'].size}): {t * 1e3:.1f} ms", flush=True)
    del r
m, K, n = 1_000_000, 100_000, 128
p, j, x = synth.csr_fixed(m, K, 32)
Y = np.asfortranarray(synth.dense_normal(K, n).T)
for i in range(3):
    lib.mx_cache_invalidate(None)
    t0 = time.perf_counter(); out = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1); t = time.perf_counter() - t0
    print(f"cfg2 product, cold: {t * 1e3:.1f} ms", flush=True)
    del out
