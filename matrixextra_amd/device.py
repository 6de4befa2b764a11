"""Device-resident operands and mxd_* launches on torch tensors.

torch is plumbing here: it owns the HBM allocations, the stream and (in
distributed.py) the RCCL communicator; every kernel that runs is one of
libmxgpu.so's hand-written HIP kernels, launched on torch's current stream
through the device-level C-ABI (include/mxgpu.h, mxd_*).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import MX_F32, MX_F64, MX_LGL, MX_NONE, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


@dataclass
class DeviceCSR:
    """CSR arrays resident in HBM: int32 indptr[m+1], int32 indices[nnz], values (f64 / int32 logical / None)."""
    indptr: torch.Tensor
    indices: torch.Tensor
    values: torch.Tensor | None
    m: int
    K: int
    nnz: int
    _sorted: bool | None = None
    _plan: object = None
    _plan_panels: int = -1
    _plan_ready: bool = True
    _plan_limited: bool = False            # the kept plan was built under AUTO's padding limit (auto_plan) / without one (plan)
    _plan_key: tuple = ()                  # (data_ptr, _version) of the three tensors when the kept plan / flags were derived
    _spmv_plan: object = None
    _profile: object = None

    def invalidate(self):
        """Forget everything derived from the arrays (plans, the sortedness flag): call after changing indptr / indices /
        values in place.  The arrays of a DeviceCSR are otherwise taken as immutable, like an R object's slots."""
        lib = _lib.load()
        if self._plan is not None:
            lib.mxd_spmm_plan_destroy(self._plan)
        self._plan, self._plan_panels, self._plan_ready, self._sorted, self._plan_limited = None, -1, True, None, False
        self._profile = None
        self.drop_spmv_plan()

    def _key(self):
        """what the kept plans depend on: the tensors' storage and torch's in-place version counters — an `A.values.mul_(2)`
        or an assignment of another tensor changes it (a kept plan must not outlive the arrays it regroups)"""
        return tuple((t.data_ptr(), t._version) for t in (self.indptr, self.indices, self.values) if t is not None)

    def _check_unchanged(self):
        k = self._key()
        if self._plan_key and self._plan_key != k:
            self.invalidate()
        self._plan_key = k

    def auto_plan(self, npanels: int = 0):
        """The plan AUTO products keep on the matrix (mxd_spmm_plan_create_auto: AUTO's padding limit applies); None when
        the plan would pad too much — the caller then runs the row-split kernel, as AUTO does.  A plan kept by plan()
        (no limit) is rebuilt under the limit first; a matrix whose tensors changed since the plan was built gets a new one."""
        lib = _lib.load()
        self._check_unchanged()
        if self._plan is None or self._plan_panels != npanels or not self._plan_limited:
            handle = self._plan if self._plan is not None else C.c_void_p()
            ready = C.c_int(0)
            check(lib.mxd_spmm_plan_create_auto(C.c_int(self.m), C.c_int(self.K), _dp(self.indptr), _dp(self.indices),
                                                _dp(self.values), C.c_int(npanels), _stream(), C.byref(handle),
                                                C.byref(ready)))
            self._plan, self._plan_panels, self._plan_ready, self._plan_limited = handle, npanels, bool(ready.value), True
        return self._plan if self._plan_ready else None

    def plan(self, npanels: int = 0, rebuild: bool = False):
        """Device-resident SpMM plan (mxd_spmm_plan_create); cached per DeviceCSR, buffers re-used on rebuild."""
        lib = _lib.load()
        self._check_unchanged()
        if self._plan is None or rebuild or self._plan_panels != npanels or not self._plan_ready:
            handle = self._plan if self._plan is not None else C.c_void_p()
            check(lib.mxd_spmm_plan_create(C.c_int(self.m), C.c_int(self.K), _dp(self.indptr), _dp(self.indices),
                                           _dp(self.values), C.c_int(npanels), _stream(), C.byref(handle)))
            self._plan, self._plan_panels, self._plan_ready, self._plan_limited = handle, npanels, True, False
        return self._plan

    def plan_info(self):
        lib = _lib.load()
        P, padded = C.c_int(0), C.c_int64(0)
        check(lib.mxd_spmm_plan_info(self._plan, C.byref(P), C.byref(padded)))
        cv = C.c_double(0.0)
        check(lib.mxd_spmm_plan_octet_cv(self._plan, C.byref(cv)))
        return dict(npanels=P.value, padded_entries=padded.value, octet_length_cv=round(cv.value, 4))

    def spmv_plan(self):
        """Planned-SpMV plan (mxd_spmv_plan_create): built on first use, cached on the DeviceCSR."""
        self._check_unchanged()
        if self._spmv_plan is None:
            handle = C.c_void_p()
            check(_lib.load().mxd_spmv_plan_create(C.c_int(self.m), C.c_int(self.K), _dp(self.indptr), _dp(self.indices),
                                                   _dp(self.values), _stream(), C.byref(handle)))
            self._spmv_plan = handle
        return self._spmv_plan

    def drop_spmv_plan(self):
        if self._spmv_plan is not None:
            _lib.load().mxd_spmv_plan_destroy(self._spmv_plan)
            self._spmv_plan = None

    def __del__(self):
        try:
            if self._plan is not None:
                _lib.load().mxd_spmm_plan_destroy(self._plan)
            if self._spmv_plan is not None:
                _lib.load().mxd_spmv_plan_destroy(self._spmv_plan)
        except Exception:
            pass

    def profile(self):
        """mxd_csr_profile (csrc/profile.hip): mass of the hottest columns + row-length statistics, what AUTO's cost model
        needs beyond the sizes; computed once per matrix (~40 us), cached like rows_sorted().  A ctypes float[40]."""
        self._check_unchanged()
        if self._profile is None:
            lib = _lib.load()
            ws = torch.empty(int(lib.mxd_csr_profile_workspace_bytes(C.c_int(self.K))) + 64, dtype=torch.uint8, device=self.indptr.device)
            prof = (C.c_float * 40)()
            check(lib.mxd_csr_profile(C.c_int(self.m), C.c_int(self.K), C.c_int64(self.nnz), _dp(self.indptr), _dp(self.indices), prof,
                                      _dp(ws), _stream()))
            self._profile = prof
        return self._profile

    def rows_sorted(self) -> bool:
        """check_is_sorted per row on the device (src/misc.cpp:118-128); cached."""
        self._check_unchanged()
        if self._sorted is None:
            lib = _lib.load()
            ws = torch.empty(4, dtype=torch.int32, device=self.indptr.device)
            flag = C.c_int(0)
            check(lib.mxd_csr_rows_sorted(C.c_int(self.m), _dp(self.indptr), _dp(self.indices), _dp(ws),
                                          C.byref(flag), _stream()))
            self._sorted = bool(flag.value)
        return self._sorted

    @classmethod
    def from_host(cls, indptr, indices, values, K, device="cuda"):
        p = torch.from_numpy(np.ascontiguousarray(indptr, dtype=np.int32)).to(device)
        j = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int32)).to(device)
        x = None if values is None else torch.from_numpy(np.ascontiguousarray(values)).to(device)
        return cls(p, j, x, int(p.numel() - 1), int(K), int(j.numel()))

    def to_host(self):
        return (self.indptr.cpu().numpy(), self.indices.cpu().numpy(),
                None if self.values is None else self.values.cpu().numpy())


def spmm(A: DeviceCSR, B: torch.Tensor, out: torch.Tensor | None = None, colmajor: bool = False,
         algo: int = 0, npanels: int = 0, wg_per_cu: int = 0, keep_plan: bool = True):
    """C = A @ B with B (K x n) row-major in HBM.  colmajor=False: C row-major (m x n) —
    gemm_csr_drm_as_drm layout; colmajor=True: C column-major (what tcrossprod_csr_dense returns to R),
    stored as a row-major (n x m) tensor and returned as its transposed view.
    algo: 0 auto, 1 row-wave kernel, 2 slab/panel kernel, 3 planned (rebuilt per call), 4 row-split kernel — npanels is
    then its number of column panels and wg_per_cu the segments per row (1, 2, 4, 8; -1 = the row-group form: several
    rows per wavefront), 0 = chosen from the shape (include/mxgpu.h mx_spmm_algo).
    keep_plan (algo 0 only): when AUTO picks the planned kernel, the plan — a regrouping of A's entries that depends on A
    alone — is built once and kept on the DeviceCSR (like rows_sorted()); keep_plan=False is the C-ABI's own AUTO, which
    rebuilds the plan from plain CSR inside every call."""
    lib = _lib.load()
    assert B.is_cuda and B.dim() == 2 and B.stride(1) == 1 and B.shape[0] == A.K
    n = int(B.shape[1])
    dt = MX_F64 if B.dtype == torch.float64 else MX_F32
    assert B.dtype in (torch.float64, torch.float32)
    if colmajor:
        if out is None:
            out = torch.empty((n, A.m), dtype=B.dtype, device=B.device)
        assert out.shape == (n, A.m) and out.is_contiguous()
        ldc = A.m
    else:
        if out is None:
            out = torch.empty((A.m, n), dtype=B.dtype, device=B.device)
        assert out.shape == (A.m, n) and out.is_contiguous()
        ldc = n
    if A.nnz == 0:                       # reference early-out (matmul.cpp:128-129,160-161): all zeros
        out.zero_()
        return out.t() if colmajor else out
    if algo == 0 and keep_plan and wg_per_cu == 0:
        pick = C.c_int(0)
        check(lib.mxd_spmm_auto_algo3(C.c_int(A.m), C.c_int(n), C.c_int(A.K), C.c_int64(A.nnz), C.c_int(1), C.c_int(dt), _dp(B),
                                      C.c_size_t(B.stride(0)), _dp(out), C.c_size_t(ldc), C.c_int(1 if colmajor else 0),
                                      A.profile(), C.byref(pick)))
        if pick.value == 3:
            plan = A.auto_plan(npanels)
            if plan is not None:
                imb = C.c_double(0.0)
                check(lib.mxd_spmm_plan_imbalance(plan, C.c_int(n), C.c_int(dt), C.byref(imb)))
                if imb.value > lib.mxd_spmm_plan_imbalance_limit(C.c_int(dt)):   # one octet outlasts the sweep (spmm_plan.hip plan_imbalance)
                    plan = None
            if plan is not None:
                check(lib.mxd_spmm_plan_run(plan, C.c_int(n), _dp(B), C.c_size_t(B.stride(0)), _dp(out), C.c_size_t(ldc),
                                            C.c_int(dt), C.c_int(1 if colmajor else 0), C.c_int(0), C.c_int(-1), _stream()))
                return out.t() if colmajor else out
            algo = 4                                                 # too much padding: AUTO's fallback (row-split kernel)
        elif pick.value != 3:
            algo = pick.value                                        # the choice made WITH a kept plan in mind stands
    sorted_rows = A.rows_sorted() if algo in (2, 5) else False         # the slab kernel's panels and the tile kernel's sweep need it
    # (keep_plan callers also keep the matrix's profile: AUTO's choice and the row-split kernel's panels follow the real column
    # popularity and row-length skew; keep_plan=False is the bare C-ABI AUTO: sizes only)
    prof = A.profile() if keep_plan and A.nnz > 0 else None
    check(lib.mxd_spmm_csr_dense_ex3(C.c_int(A.m), C.c_int(n), C.c_int(A.K), C.c_int64(A.nnz), _dp(A.indptr), _dp(A.indices),
                                     _dp(A.values), _dp(B), C.c_size_t(B.stride(0)), _dp(out), C.c_size_t(ldc),
                                     C.c_int(dt), C.c_int(1 if colmajor else 0), C.c_int(algo),
                                     C.c_int(int(sorted_rows)), C.c_int(npanels), C.c_int(wg_per_cu), prof, _stream()))
    return out.t() if colmajor else out


def spmm_planned(A: DeviceCSR, B: torch.Tensor, out: torch.Tensor | None = None, colmajor: bool = False,
                 npanels: int = 0, wg_per_cu: int = 0, sync_mode: int = -1, rebuild_plan: bool = False):
    """C = A @ B through the planned panel-sweep kernel (v3).  The plan is built on first use (or every call with
    rebuild_plan=True, which is what a one-shot product from plain CSR costs) and cached on the DeviceCSR."""
    lib = _lib.load()
    assert B.is_cuda and B.dim() == 2 and B.stride(1) == 1 and B.shape[0] == A.K
    n = int(B.shape[1])
    dt = MX_F64 if B.dtype == torch.float64 else MX_F32
    if colmajor:
        if out is None:
            out = torch.empty((n, A.m), dtype=B.dtype, device=B.device)
        ldc = A.m
    else:
        if out is None:
            out = torch.empty((A.m, n), dtype=B.dtype, device=B.device)
        ldc = n
    if A.nnz == 0:
        out.zero_()
        return out.t() if colmajor else out
    plan = A.plan(npanels, rebuild=rebuild_plan)
    check(lib.mxd_spmm_plan_run(plan, C.c_int(n), _dp(B), C.c_size_t(B.stride(0)), _dp(out), C.c_size_t(ldc),
                                C.c_int(dt), C.c_int(1 if colmajor else 0), C.c_int(wg_per_cu), C.c_int(sync_mode),
                                _stream()))
    return out.t() if colmajor else out


def spmv(A: DeviceCSR, v: torch.Tensor, v_dtype=None, out=None, algo: int = 0):
    """y = A @ v (matmul_csr_dvec).  v float64 / float32 / int32 (v_dtype MX_I32 or MX_LGL for int32).
    algo: 0 auto, 1 lane-group kernel, 2 LDS-panel tile kernel (include/mxgpu.h mx_spmv_algo)."""
    lib = _lib.load()
    if v_dtype is None:
        v_dtype = {torch.float64: MX_F64, torch.float32: MX_F32}[v.dtype]
    odt = torch.float32 if v_dtype == MX_F32 else torch.float64
    if out is None:
        out = torch.empty(A.m, dtype=odt, device=v.device)
    assert int(v.numel()) == A.K and v.is_contiguous()
    # (AUTO reads the matrix profile — cached on A — for very long rows, once the product is large enough for that to matter)
    prof = A.profile() if algo == 0 and A.nnz >= (1 << 20) else None
    check(lib.mxd_spmv_csr_dvec_ex2(C.c_int(A.m), C.c_int(A.K), C.c_int64(A.nnz), _dp(A.indptr), _dp(A.indices),
                                    _dp(A.values), _dp(v), C.c_int(v_dtype), _dp(out), C.c_int(algo), prof, _stream()))
    return out


def spmv_planned(A: DeviceCSR, v: torch.Tensor, v_dtype=None, out=None):
    """y = A @ v through the planned kernel (v's panels staged in LDS); the plan is built on first use and kept on A."""
    lib = _lib.load()
    if v_dtype is None:
        v_dtype = {torch.float64: MX_F64, torch.float32: MX_F32}[v.dtype]
    odt = torch.float32 if v_dtype == MX_F32 else torch.float64
    if out is None:
        out = torch.empty(A.m, dtype=odt, device=v.device)
    assert int(v.numel()) == A.K and v.is_contiguous()
    check(lib.mxd_spmv_plan_run(A.spmv_plan(), _dp(v), C.c_int(v_dtype), _dp(out), _stream()))
    return out


def csr_elemwise(op, A: DeviceCSR, B: DeviceCSR, two_pass: bool = True):
    """CSR (+) CSR on device-resident operands.  Default (two_pass=True): count -> scan -> (host round trip for nnz) ->
    fill into exactly sized arrays.  two_pass=False: ONE pass (mxd_csr_merge_fused) into arrays sized for the upper
    bound of the result (nnz1 + nnz2, or min for the intersection), returned as views of their first nnz_out entries."""
    lib = _lib.load()
    assert A.m == B.m
    dev = A.indptr.device
    logical = op in (_lib.MX_OP_OR, _lib.MX_OP_XOR, _lib.MX_OP_AND)
    bound = min(A.nnz, B.nnz) if op in (_lib.MX_OP_MUL, _lib.MX_OP_AND) else A.nnz + B.nnz
    if not two_pass and bound < 2 ** 31:
        ws = torch.empty(lib.mxd_merge_fused_workspace_bytes(A.m), dtype=torch.uint8, device=dev)
        out_p = torch.empty(A.m + 1, dtype=torch.int32, device=dev)
        out_j = torch.empty(bound, dtype=torch.int32, device=dev)
        out_x = torch.empty(bound, dtype=torch.int32 if logical else torch.float64, device=dev)
        nnz_out = C.c_int64(0)
        check(lib.mxd_csr_merge_fused(C.c_int(op), C.c_int(A.m), _dp(A.indptr), _dp(A.indices), _dp(A.values),
                                      C.c_int64(A.nnz), _dp(B.indptr), _dp(B.indices), _dp(B.values), C.c_int64(B.nnz),
                                      _dp(out_p), _dp(out_j), _dp(out_x), _dp(ws), C.byref(nnz_out), _stream()))
        n = int(nnz_out.value)
        return DeviceCSR(out_p, out_j[:n], out_x[:n], A.m, A.K, n)
    # rows of uneven length (the cached matrix profiles' cv): one lane-group width up (mxd_csr_merge_rows_uneven)
    uneven = A.nnz + B.nnz >= (1 << 21) and max(A.profile()[32], B.profile()[32]) > 0.3
    lib.mxd_csr_merge_rows_uneven(C.c_int(int(uneven)))
    ws = torch.empty(lib.mxd_merge_workspace_bytes(A.m), dtype=torch.uint8, device=dev)
    out_p = torch.empty(A.m + 1, dtype=torch.int32, device=dev)
    nnz_out = C.c_int64(0)
    check(lib.mxd_csr_merge_count(C.c_int(op), C.c_int(A.m), _dp(A.indptr), _dp(A.indices), C.c_int64(A.nnz),
                                  _dp(B.indptr), _dp(B.indices), C.c_int64(B.nnz), _dp(out_p), _dp(ws),
                                  C.byref(nnz_out), _stream()))
    out_j = torch.empty(nnz_out.value, dtype=torch.int32, device=dev)
    out_x = torch.empty(nnz_out.value, dtype=torch.int32 if logical else torch.float64, device=dev)
    check(lib.mxd_csr_merge_fill(C.c_int(op), C.c_int(A.m), _dp(A.indptr), _dp(A.indices), _dp(A.values),
                                 C.c_int64(A.nnz), _dp(B.indptr), _dp(B.indices), _dp(B.values), C.c_int64(B.nnz),
                                 _dp(out_p), _dp(out_j), _dp(out_x), _stream()))
    lib.mxd_csr_merge_rows_uneven(C.c_int(0))
    return DeviceCSR(out_p, out_j, out_x, A.m, A.K, int(nnz_out.value))


def csr_drop_zeros(A: DeviceCSR, remove_NAs: bool = False):
    """remove_zero_valued_csr on a device-resident matrix (src/misc.cpp:553-664): count -> scan -> (host round trip for
    nnz) -> ordered compaction.  Returns A itself when the reference's first scan would find nothing to remove."""
    lib = _lib.load()
    dev = A.indptr.device
    vd = MX_F64 if A.values.dtype == torch.float64 else MX_LGL
    lib.mxd_csr_drop_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(lib.mxd_csr_drop_workspace_bytes(C.c_int(A.m)), dtype=torch.uint8, device=dev)
    out_p = torch.empty(A.m + 1, dtype=torch.int32, device=dev)
    nnz_out, dirty = C.c_int64(0), C.c_int(0)
    check(lib.mxd_csr_drop_count(C.c_int(A.m), C.c_int64(A.nnz), _dp(A.indptr), _dp(A.values), C.c_int(vd),
                                 C.c_int(1 if remove_NAs else 0), _dp(out_p), _dp(ws), C.byref(nnz_out), C.byref(dirty), _stream()))
    if not dirty.value:
        return A
    out_j = torch.empty(nnz_out.value, dtype=torch.int32, device=dev)
    out_x = torch.empty(nnz_out.value, dtype=A.values.dtype, device=dev)
    check(lib.mxd_csr_drop_fill(C.c_int(A.m), C.c_int64(A.nnz), _dp(A.indptr), _dp(A.indices), _dp(A.values), C.c_int(vd),
                                C.c_int(1 if remove_NAs else 0), _dp(out_p), _dp(out_j), _dp(out_x), _stream()))
    return DeviceCSR(out_p, out_j, out_x, A.m, A.K, int(nnz_out.value))


def csr_gather_rows(A: DeviceCSR, rows: torch.Tensor, one_launch: bool = True):
    """A[rows, :] on device (copy_csr_rows).  rows int32, 0-based.
    one_launch (default): mxd_csr_gather_fused into arrays sized for 1.25x the expected result (r mean row lengths) — one
    kernel, the size read back once behind it; a selection that does not fit (very uneven rows) is copied again into
    exactly sized arrays by the fill kernel.  one_launch=False: count -> (host round trip) -> fill."""
    lib = _lib.load()
    dev = A.indptr.device
    r = int(rows.numel())
    nnz_out = C.c_int64(0)
    if A.values is None:
        vd, vdt = MX_NONE, None
    else:
        vd, vdt = (MX_F64 if A.values.dtype == torch.float64 else MX_LGL), A.values.dtype
    counted = False
    if one_launch and A.m > 0 and r > 0:
        avg = A.nnz / A.m
        cap = min(int(1.25 * r * avg) + 1024, 2 ** 31 - 1) & ~3
        # one allocation for the three result arrays (values first: 8-byte aligned), no workspace (the look-back state lives
        # in the library): the call is short enough for every torch.empty to show
        vb = 0 if vdt is None else (8 if vdt == torch.float64 else 4)
        buf = torch.empty(cap * (vb + 4) + 4 * (r + 1), dtype=torch.uint8, device=dev)
        new_x = None if vdt is None else buf[:cap * vb].view(vdt)
        new_j = buf[cap * vb:cap * (vb + 4)].view(torch.int32)
        new_p = buf[cap * (vb + 4):].view(torch.int32)
        check(lib.mxd_csr_gather_fused(C.c_int(r), _dp(A.indptr), _dp(A.indices), _dp(A.values), _dp(rows), _dp(new_p),
                                       _dp(new_j), _dp(new_x), C.c_int(vd), C.c_int64(cap), C.c_double(avg), None,
                                       C.byref(nnz_out), _stream()))
        n = int(nnz_out.value)
        if n <= cap:
            return DeviceCSR(new_p, new_j[:n], None if new_x is None else new_x[:n], r, A.K, n)
        counted = True                                               # new_p is complete; only the copy has to be redone
    if not counted:
        ws = torch.empty(lib.mxd_gather_workspace_bytes(r), dtype=torch.uint8, device=dev)
        new_p = torch.empty(r + 1, dtype=torch.int32, device=dev)
        check(lib.mxd_csr_gather_count(C.c_int(r), _dp(A.indptr), _dp(rows), _dp(new_p), _dp(ws),
                                       C.byref(nnz_out), _stream()))
    new_j = torch.empty(nnz_out.value, dtype=torch.int32, device=dev)
    new_x = None if vdt is None else torch.empty(nnz_out.value, dtype=vdt, device=dev)
    check(lib.mxd_csr_gather_fill(C.c_int(r), _dp(A.indptr), _dp(A.indices), _dp(A.values), _dp(rows), _dp(new_p),
                                  _dp(new_j), _dp(new_x), C.c_int(vd), C.c_int64(nnz_out.value), _stream()))
    return DeviceCSR(new_p, new_j, new_x, r, A.K, int(nnz_out.value))
