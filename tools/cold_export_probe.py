#!/usr/bin/env python3
"""Spread of the COLD export-level call (cfg2, CSR not on the device, freshly allocated 1 GB result) inside one process,
with the page-size facts that explain it.  `malloc` mode hands the library buffers from plain libc malloc, as R does
(no MADV_HUGEPAGE: on a THP=madvise machine they are 4-KiB pages), `numpy` mode numpy arrays (numpy advises huge pages).
  [MXGPU_HUGEPAGE=1] [MXGPU_TRACE=1] python tools/cold_export_probe.py [numpy|malloc] [calls]"""
import ctypes as C, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matrixextra_amd import _lib, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "numpy"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for f in ("enabled", "defrag"):
    try:
        print("THP", f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except OSError as e:
        print(f, e)
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
m, K, n = 1_000_000, 100_000, 128
p, j, x = synth.csr_fixed(m, K, 32)
Y = np.asfortranarray(synth.dense_normal(K, n).T)


def as_malloc(a):
    q = libc.malloc(a.nbytes)
    C.memmove(q, a.ctypes.data, a.nbytes)
    return q


if mode == "malloc":
    pp, pj, px, pY = (as_malloc(a) for a in (p, j, x, Y))
else:
    pp, pj, px, pY = (a.ctypes.data for a in (p, j, x, Y))
lib = _lib.load()
fn = lib.mx_tcrossprod_csr_dense_numeric
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]


def anon_huge():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("AnonHugePages"):
            return int(line.split()[1]) // 1024


def call():
    if mode == "malloc":
        out = libc.malloc(8 * m * n)
        keep = None
    else:
        keep = np.empty((m, n), order="F")
        out = keep.ctypes.data
    t0 = time.perf_counter()
    _lib.check(fn(pp, pj, px, m, pY, n, K, 1, out))
    t = time.perf_counter() - t0
    first = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_double)), shape=(4,)).copy()
    if mode == "malloc":
        libc.free(out)
    return t, first


call()                                                        # warm-up: scratch, streams, pool
ts = []
for i in range(calls):
    lib.mx_cache_invalidate(None)
    h0 = anon_huge()
    t, first = call()
    ts.append(t * 1e3)
    print(f"{mode} cold call {i}: {t * 1e3:.1f} ms   AnonHugePages before {h0} MiB, C[0:2] = {first[:2]}", flush=True)
print("%s: min %.1f  median %.1f  max %.1f ms" % (mode, min(ts), sorted(ts)[len(ts) // 2], max(ts)))
