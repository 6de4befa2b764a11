"""Single-process, multi-GPU SpMM with the RCCL all-gather behind the C-ABI (csrc/sharded.hip, include/mxgpu.h
mx_spmm_sharded_*): the row blocks of one CSR stay on their devices with AUTO's kept plan, every product runs on all
devices at once and one in-place ncclAllGather leaves the full row-major C on EVERY device.  This is the host-side mirror
a C++ / R caller would write; `distributed.py` is the one-process-per-GPU form under torch.distributed.

Reference: the product is tcrossprod_csr_dense (src/matmul.cpp:316-343 -> gemm_csr_drm_as_drm :118-142, one row block per
device); BASELINE.json north_star names the sharding and the collective."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MX_F32, MX_F64, check, ptr


def layout(indptr, ndev: int, dense_cols: int = 128, dense_bytes: int = 8, equal_rows: bool = False):
    """(cuts[ndev + 1], slot_rows) the sharded product uses for this matrix and device count — host arithmetic only."""
    p = np.ascontiguousarray(indptr, dtype=np.int32)
    cuts = (C.c_int * (ndev + 1))()
    slot = C.c_int(0)
    check(_lib.load().mx_spmm_sharded_layout(ptr(p), C.c_int(p.size - 1), C.c_int(ndev), C.c_int(dense_cols), C.c_int(dense_bytes),
                                             C.c_int(int(equal_rows)), cuts, C.byref(slot)))
    return list(cuts), slot.value


class ShardedSpMM:
    """One CSR (host arrays) cut into one row block per device of `devices`; products against any number of B."""

    def __init__(self, devices, indptr, indices, values, K: int, equal_rows: bool = False):
        lib = _lib.load()
        lib.mx_spmm_sharded_kernel.restype = C.c_char_p
        self._p = np.ascontiguousarray(indptr, dtype=np.int32)
        self._j = np.ascontiguousarray(indices, dtype=np.int32)
        self._x = np.ascontiguousarray(values, dtype=np.float64)
        self.m, self.K = self._p.size - 1, int(K)
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        check(lib.mx_spmm_sharded_create(devs, C.c_int(len(devices)), C.c_int(self.m), C.c_int(self.K), ptr(self._p), ptr(self._j),
                                         ptr(self._x), C.c_int(int(equal_rows)), C.byref(self._h)))
        ns, slot, uses, ver = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        cuts = (C.c_int * (len(devices) + 1))()
        check(lib.mx_spmm_sharded_info(self._h, C.byref(ns), C.byref(slot), cuts, C.byref(uses), C.byref(ver)))
        self.nshards, self.slot_rows, self.cuts = ns.value, slot.value, list(cuts)
        self.uses_rccl, self.rccl_version = bool(uses.value), ver.value
        self.devices = list(devices)

    def run(self, B: np.ndarray, want_host: bool = True, colmajor: bool = False):
        """B: (K, n) row-major float64 / float32 on the host.  Returns C (m, n) from the first device (None if not wanted)."""
        assert B.ndim == 2 and B.shape[0] == self.K and B.flags.c_contiguous and B.dtype in (np.float64, np.float32)
        n = B.shape[1]
        dt = MX_F64 if B.dtype == np.float64 else MX_F32
        out = None
        if want_host:
            out = np.empty((self.m, n), dtype=B.dtype, order="F" if colmajor else "C")
        check(_lib.load().mx_spmm_sharded_run(self._h, C.c_int(n), C.c_int(dt), ptr(B), C.c_size_t(n), ptr(out),
                                              C.c_size_t(self.m if colmajor else n), C.c_int(int(colmajor))))
        return out

    def run_dev(self, B_ptrs, n: int, dtype, ldb: int | None = None, asynchronous: bool = False):
        """B_ptrs[k]: device address of the (K, n) row-major operand on shard k's device."""
        dt = MX_F64 if np.dtype(dtype) == np.float64 else MX_F32
        arr = (C.c_void_p * self.nshards)(*B_ptrs)
        check(_lib.load().mx_spmm_sharded_run_dev(self._h, C.c_int(n), C.c_int(dt), arr, C.c_size_t(ldb or n), C.c_int(int(asynchronous))))

    def sync(self):
        check(_lib.load().mx_spmm_sharded_sync(self._h))

    def result_ptr(self, k: int):
        """(device address, n, dtype, device) of the last product's gathered buffer on shard k's device."""
        p, n, dt, dev = C.c_void_p(), C.c_int(0), C.c_int(0), C.c_int(0)
        check(_lib.load().mx_spmm_sharded_result(self._h, C.c_int(k), C.byref(p), C.byref(n), C.byref(dt), C.byref(dev)))
        return p.value, n.value, (np.float64 if dt.value == MX_F64 else np.float32), dev.value

    def gathered(self, k: int) -> np.ndarray:
        """The full C (m, n) as shard k's device holds it after the all-gather, read back block by block (checks / tests)."""
        addr, n, dtype, _ = self.result_ptr(k)
        sz = np.dtype(dtype).itemsize
        out = np.empty((self.m, n), dtype=dtype)
        lib = _lib.load()
        for r in range(self.nshards):
            r0, r1 = self.cuts[r], self.cuts[r + 1]
            if r1 > r0:
                check(lib.mx_download(ptr(out[r0:r1]), C.c_void_p(addr + r * self.slot_rows * n * sz), C.c_size_t((r1 - r0) * n * sz)))
        return out

    def kernel(self, k: int) -> str:
        return _lib.load().mx_spmm_sharded_kernel(self._h, C.c_int(k)).decode()

    def close(self):
        if self._h:
            check(_lib.load().mx_spmm_sharded_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
