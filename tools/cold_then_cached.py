import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from matrixextra_amd import _lib, synth
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
m, K, n = 1_000_000, 100_000, 128
p, j, x = synth.csr_fixed(m, K, 32)
Y = np.asfortranarray(synth.dense_normal(K, n).T)
lib = _lib.load()
fn = lib.mx_tcrossprod_csr_dense_numeric
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
def call():
    out = libc.malloc(8 * m * n); t0 = time.perf_counter()
    _lib.check(fn(p.ctypes.data, j.ctypes.data, x.ctypes.data, m, Y.ctypes.data, n, K, 1, out)); t = time.perf_counter() - t0
    libc.free(out); buf = C.create_string_buffer(512); lib.mx_last_call_phases(buf, C.c_size_t(512)); return t * 1e3, buf.value.decode()
for _ in range(3): lib.mx_cache_invalidate(None); call()
for i in range(4):
    lib.mx_cache_invalidate(None)
    t, ph = call(); print("cold   %.1f %s" % (t, ph))
    t, ph = call(); print("cached %.1f %s" % (t, ph))
    t, ph = call(); print("cached %.1f %s" % (t, ph))
