#!/usr/bin/env python3
"""mx_upload / mx_download of large buffers (downloads into freshly allocated, untouched pages, as an R result is):
  [MXGPU_LIB=other.so] [MXGPU_XFER=0: plain hipMemcpy] python tools/download_probe.py [MiB ...]"""
import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from matrixextra_amd import _lib
lib = _lib.load()
SIZES = [int(float(a) * (1 << 20)) for a in sys.argv[1:]] or [1 << 30, 600 << 20, 200 << 20]
for nbytes in SIZES:
    d = C.c_void_p()
    _lib.check(lib.mx_dev_malloc(C.byref(d), C.c_size_t(nbytes)))
    src = np.random.default_rng(0).integers(0, 256, size=nbytes, dtype=np.uint8)
    tu = []
    for _ in range(4):
        t0 = time.perf_counter()
        _lib.check(lib.mx_upload(d, C.c_void_p(src.ctypes.data), C.c_size_t(nbytes)))
        tu.append((time.perf_counter() - t0) * 1e3)
    print(round(nbytes / 2**20, 2), "MiB upload: ms", [round(t, 3) for t in tu], "GB/s best", round(nbytes / min(tu) / 1e6, 1))
    ts = []
    for _ in range(6):
        dst = np.empty(nbytes, dtype=np.uint8)
        t0 = time.perf_counter()
        _lib.check(lib.mx_download(C.c_void_p(dst.ctypes.data), d, C.c_size_t(nbytes)))
        ts.append((time.perf_counter() - t0) * 1e3)
        assert np.array_equal(dst[::4097], src[::4097]) and dst[-1] == src[-1]
        del dst
    print(round(nbytes / 2**20, 2), "MiB download into fresh pages: ms", [round(t, 3) for t in ts], "GB/s best", round(nbytes / min(ts) / 1e6, 1))
    lib.mx_dev_free(d)
