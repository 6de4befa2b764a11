import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from matrixextra_amd import _lib
lib = _lib.load()
_lib.device_count()
for mb in (64, 388, 1024):
    a = np.ones(mb * (1 << 20) // 8)          # touched pages
    t0 = time.perf_counter(); rc = lib.mx_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)); t1 = time.perf_counter()
    lib.mx_host_unregister(C.c_void_p(a.ctypes.data)); t2 = time.perf_counter()
    print(f"{mb} MB touched: register {1e3*(t1-t0):.1f} ms rc={rc}, unregister {1e3*(t2-t1):.1f} ms")
    b = np.empty(mb * (1 << 20) // 8)         # fresh (untouched) pages
    t0 = time.perf_counter(); rc = lib.mx_host_register(C.c_void_p(b.ctypes.data), C.c_size_t(b.nbytes)); t1 = time.perf_counter()
    lib.mx_host_unregister(C.c_void_p(b.ctypes.data)); t2 = time.perf_counter()
    print(f"{mb} MB fresh:   register {1e3*(t1-t0):.1f} ms rc={rc}, unregister {1e3*(t2-t1):.1f} ms")
# first-touch cost of 1 GB with 1 thread, and memcpy rate
b = np.empty(1 << 27); t0 = time.perf_counter(); b[::512] = 1.0; print("first touch 1 GB (1 thread):", round(1e3*(time.perf_counter()-t0),1), "ms")
c = np.empty(1 << 27); c[::512] = 0; t0 = time.perf_counter(); np.copyto(c, b); print("memcpy 1 GB warm (1 thread):", round(1e3*(time.perf_counter()-t0),1), "ms")
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
