import ctypes as C, sys
sys.path.insert(0, ".")
from matrixextra_amd import _lib
import numpy as np
lib = _lib.load()
for n in (16, 4096, 1 << 20, 64 << 20, 1 << 30):
    d = C.c_void_p()
    lib.mx_dev_malloc(C.byref(d), C.c_size_t(n))
    print("dev", n, hex(d.value))
a = np.empty(1 << 20); b = np.empty(40 << 20, dtype=np.uint8); c = np.empty(5 << 20, dtype=np.uint8)
print("host small", hex(a.ctypes.data), "host 40MB", hex(b.ctypes.data), "host 5MB", hex(c.ctypes.data))
import torch
t = torch.empty(1 << 20, device="cuda"); print("torch", hex(t.data_ptr()))
