"""include/mxgpu.h says the device level (mxd_*) enqueues on the caller's stream without allocating or synchronising
"(graph-capture safe) except where a size must come back to the host".  Here the entry points a device-resident
optimisation loop is made of (the vignette's L-BFGS: products with one matrix, Rmd:452-470) are captured into ONE HIP graph
on a side stream — the planned SpMM with its repack of B, the row-split and the LDS-tile SpMM with the sizes passed, the
flat SpMV, the fill passes of CSR + CSR and X[rows, ] (their count passes bring a size back: they run before the capture) —
and the graph is replayed three times on fresh B, v and values; every replay is checked against the oracle.

Not capturable, by contract (mxgpu.h "Graph capture"): every *_count, the *_fused forms, mxd_spmm_plan_create /
mxd_spmv_plan_create, mxd_csr_rows_sorted, mxd_check_is_seq, MX_SPMM_AUTO / _PLANNED through mxd_spmm_csr_dense_ex* (the plan is
sized on the host), and MX_SPMM_ROWSPLIT / _TILE with nnz = -1 (indptr[m] is read back)."""
import ctypes as C

import numpy as np
import pytest
import torch

from matrixextra_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def test_device_level_sequence_in_one_hip_graph(gpu):
    lib = _lib.load()
    check = _lib.check
    m, K, n, npr = 40_000, 6_000, 128, 24
    p, j, x0 = synth.csr_fixed(m, K, npr, seed=3)
    p2, j2, x20 = synth.csr_overlapping(p, j, K, npr, seed=4)
    rows = synth.rows_with_replacement(5_000, m, seed=5)
    nnz, nnz2, r = int(p[-1]), int(p2[-1]), rows.size
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dp, dj, dx, dp2, dj2, dx2, drows = dev(p), dev(j), dev(x0), dev(p2), dev(j2), dev(x20), dev(rows)
    B = torch.empty((K, n), dtype=torch.float64, device="cuda")
    v = torch.empty(K, dtype=torch.float64, device="cuda")
    Cc = torch.empty((n, m), dtype=torch.float64, device="cuda")           # column-major C of the planned product
    Cr = torch.empty((m, n), dtype=torch.float64, device="cuda")           # row-major C of the row-split product
    Ct = torch.empty((m, n), dtype=torch.float64, device="cuda")           # ... and of the tile kernel
    y = torch.empty(m, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = C.c_void_p(side.cuda_stream)
        # ---- what brings a size back to the host runs BEFORE the capture: the plan, the two count passes
        plan = C.c_void_p()
        check(lib.mxd_spmm_plan_create(C.c_int(m), C.c_int(K), _p(dp), _p(dj), _p(dx), C.c_int(0), st, C.byref(plan)))
        ws = torch.empty(int(lib.mxd_merge_workspace_bytes(C.c_int(m))) + 16, dtype=torch.uint8, device="cuda")
        out_p = torch.empty(m + 1, dtype=torch.int32, device="cuda")
        nout = C.c_int64(-1)
        check(lib.mxd_csr_merge_count(C.c_int(_lib.MX_OP_ADD), C.c_int(m), _p(dp), _p(dj), C.c_int64(nnz), _p(dp2), _p(dj2), C.c_int64(nnz2),
                                      _p(out_p), _p(ws), C.byref(nout), st))
        out_j = torch.empty(nout.value, dtype=torch.int32, device="cuda")
        out_x = torch.empty(nout.value, dtype=torch.float64, device="cuda")
        gws = torch.empty(int(lib.mxd_gather_workspace_bytes(C.c_int(r))) + 16, dtype=torch.uint8, device="cuda")
        new_p = torch.empty(r + 1, dtype=torch.int32, device="cuda")
        ng = C.c_int64(-1)
        check(lib.mxd_csr_gather_count(C.c_int(r), _p(dp), _p(drows), _p(new_p), _p(gws), C.byref(ng), st))
        new_j = torch.empty(ng.value, dtype=torch.int32, device="cuda")
        new_x = torch.empty(ng.value, dtype=torch.float64, device="cuda")

        def sequence():
            check(lib.mxd_spmm_plan_run(plan, C.c_int(n), _p(B), C.c_size_t(n), _p(Cc), C.c_size_t(m), C.c_int(_lib.MX_F64), C.c_int(1),
                                        C.c_int(0), C.c_int(-1), st))
            check(lib.mxd_spmm_csr_dense_ex2(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(nnz), _p(dp), _p(dj), _p(dx), _p(B), C.c_size_t(n),
                                             _p(Cr), C.c_size_t(n), C.c_int(_lib.MX_F64), C.c_int(0), C.c_int(4), C.c_int(1), C.c_int(2),
                                             C.c_int(1), st))                      # row-split kernel, 2 column panels
            check(lib.mxd_spmm_csr_dense_ex2(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(nnz), _p(dp), _p(dj), _p(dx), _p(B), C.c_size_t(n),
                                             _p(Ct), C.c_size_t(n), C.c_int(_lib.MX_F64), C.c_int(0), C.c_int(5), C.c_int(0), C.c_int(0),
                                             C.c_int(0), st))                      # tile kernel with its sortedness pass
            check(lib.mxd_spmv_csr_dvec_ex(C.c_int(m), C.c_int(K), C.c_int64(nnz), _p(dp), _p(dj), _p(dx), _p(v), C.c_int(_lib.MX_F64), _p(y),
                                           C.c_int(3), st))                        # MX_SPMV_FLAT
            check(lib.mxd_csr_merge_fill(C.c_int(_lib.MX_OP_ADD), C.c_int(m), _p(dp), _p(dj), _p(dx), C.c_int64(nnz), _p(dp2), _p(dj2), _p(dx2),
                                         C.c_int64(nnz2), _p(out_p), _p(out_j), _p(out_x), st))
            check(lib.mxd_csr_gather_fill(C.c_int(r), _p(dp), _p(dj), _p(dx), _p(drows), _p(new_p), _p(new_j), _p(new_x), C.c_int(_lib.MX_F64),
                                          C.c_int64(ng.value), st))
        B.normal_()
        v.normal_()
        sequence()                                     # once uncaptured: the library's grow-only per-thread scratch exists
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            sequence()
    torch.cuda.synchronize()
    rng = np.random.default_rng(11)
    for it in range(3):
        Bh, vh = rng.normal(size=(K, n)), rng.normal(size=K)
        xh, x2h = rng.uniform(-1, 1, size=nnz), rng.uniform(-1, 1, size=nnz2)
        B.copy_(torch.from_numpy(Bh))
        v.copy_(torch.from_numpy(vh))
        dx.copy_(torch.from_numpy(xh))
        dx2.copy_(torch.from_numpy(x2h))
        for t in (Cc, Cr, Ct, y, out_x, new_x):
            t.fill_(float("nan"))                      # every replay must write everything again
        graph.replay()
        torch.cuda.synchronize()
        rows_chk = 3000
        pc, e = p[:rows_chk + 1], int(p[rows_chk])
        Y = np.asfortranarray(Bh.T)
        ref_plan = O.tcrossprod_csr_dense(pc, j[:e], x0[:e], Y, 1, False)          # (the plan keeps the values it was built from)
        got = Cc[:, :rows_chk].t().cpu().numpy()
        assert np.max(np.abs(got - ref_plan)) <= 1e-12 * np.max(np.abs(ref_plan))
        ref_now = O.tcrossprod_csr_dense(pc, j[:e], xh[:e], Y, 1, True)
        assert np.max(np.abs(Cr[:rows_chk].cpu().numpy() - ref_now)) <= 1e-12 * np.max(np.abs(ref_now))
        np.testing.assert_array_equal(Ct[:rows_chk].cpu().numpy(), ref_now)        # the tile kernel: the FMA chain bit for bit
        assert bool(torch.isfinite(Cc).all()) and bool(torch.isfinite(Cr).all()) and bool(torch.isfinite(Ct).all())
        np.testing.assert_array_equal(y.cpu().numpy(), O.matmul_csr_dvec_numeric(p, j, xh, vh))
        ref_add = O.add_csr_elemwise(p, p2, j, j2, xh, x2h, False)
        np.testing.assert_array_equal(out_j.cpu().numpy(), ref_add["indices"])
        np.testing.assert_array_equal(out_x.cpu().numpy(), ref_add["values"])
        ref_g = O.copy_csr_rows_numeric(p, j, xh, rows)
        np.testing.assert_array_equal(new_j.cpu().numpy(), ref_g["indices"])
        np.testing.assert_array_equal(new_x.cpu().numpy(), ref_g["values"])
    lib.mxd_spmm_plan_destroy(plan)
