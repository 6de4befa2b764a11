#!/usr/bin/env python3
"""Stress of the transfer engine: many uploads / downloads of freshly allocated host buffers of varying size and
alignment (numpy re-uses freed address ranges: registrations of overlapping ranges follow each other closely)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from matrixextra_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
d = C.c_void_p()
_lib.check(lib.mx_dev_malloc(C.byref(d), C.c_size_t(200 << 20)))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    n = int(rng.integers(16 << 20, 120 << 20))
    off = int(rng.integers(0, 64))
    src = np.frombuffer(rng.bytes(4096), dtype=np.uint8)
    buf = np.empty(n + 64, dtype=np.uint8)[off:off + n]
    buf[:] = np.resize(src, n)
    _lib.check(lib.mx_upload(d, C.c_void_p(buf.ctypes.data), C.c_size_t(n)))
    out = np.empty(n + 64, dtype=np.uint8)[off:off + n]
    _lib.check(lib.mx_download(C.c_void_p(out.ctypes.data), d, C.c_size_t(n)))
    assert np.array_equal(buf, out), it
    del buf, out
    if it % 50 == 0:
        print("iteration", it, flush=True)
print("xfer stress ok")
