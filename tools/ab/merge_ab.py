"""same-box A/B of libmxgpu.so builds on the cfg4 merges (rows of equal length): MXGPU_LIB=<lib> python tools/ab/merge_ab.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import torch
from matrixextra_amd import device as D, synth, _lib
from auto_map import timeit
lib = _lib.load()
m = K = 2_000_000
p1, j1, x1 = synth.device_csr_fixed(m, K, 50)
p2, j2, x2 = synth.device_csr_overlapping(j1, m, K, 50)
A1 = D.DeviceCSR(p1, j1, x1, m, K, int(j1.numel())); A2 = D.DeviceCSR(p2, j2, x2, m, K, int(j2.numel()))
ta = min(timeit(lambda: D.csr_elemwise(_lib.MX_OP_ADD, A1, A2), reps=8) for _ in range(3))
tm = min(timeit(lambda: D.csr_elemwise(_lib.MX_OP_MUL, A1, A2), reps=8) for _ in range(3))
# count pass alone (device-level call, size read back)
ws = torch.empty(lib.mxd_merge_workspace_bytes(m), dtype=torch.uint8, device="cuda")
op_ = torch.empty(m + 1, dtype=torch.int32, device="cuda")
n_out = C.c_int64(0)
def count(op):
    _lib.check(lib.mxd_csr_merge_count(C.c_int(op), C.c_int(m), D._dp(A1.indptr), D._dp(A1.indices), C.c_int64(A1.nnz), D._dp(A2.indptr), D._dp(A2.indices),
                                       C.c_int64(A2.nnz), D._dp(op_), D._dp(ws), C.byref(n_out), D._stream()))
tc = min(timeit(lambda: count(_lib.MX_OP_ADD), reps=10) for _ in range(3))
print(f"{os.path.basename(os.environ.get('MXGPU_LIB', 'in-tree')):28s} add {ta:.4f}  mul {tm:.4f}  count-pass(+) {tc:.4f} ms", flush=True)
