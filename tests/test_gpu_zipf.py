"""Inputs that look like real dgRMatrix data (the vignette's own application is LibSVM real-sim,
vignettes/Introducing_MatrixExtra.Rmd:442-502): power-law columns and log-normal row lengths (synth.csr_zipf /
device_csr_zipf).  The matrix profile AUTO's cost model reads (csrc/profile.hip) against numpy, AUTO's choices with it, the
SpMM kernels at cfg2's size on such a matrix (checksum, linearity, sampled rows against the oracle), and the merge and the
gather at cfg4 / cfg3 size with skewed rows (their lane-group width is chosen from the MEAN row length)."""
import ctypes as C

import numpy as np
import pytest
import torch

from matrixextra_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _mass_exact(j, K, top):
    cnt = np.sort(np.bincount(j, minlength=K))[::-1]
    return cnt[:top].sum() / max(1, j.size)


def test_profile_against_numpy(gpu):
    from matrixextra_amd import device as D
    # power-law columns: the sampled, split-half estimate of mass(top) is within a few points of the exact one
    p, j, x = synth.csr_zipf(60_000, 30_000, 40, alpha=1.0, sigma=1.0, seed=5)
    A = D.DeviceCSR.from_host(p, j, x, 30_000)
    prof = A.profile()
    for i in (4, 8, 10, 12, 14):
        assert abs(prof[i] - _mass_exact(j, 30_000, 1 << i)) <= 0.04, (i, prof[i], _mass_exact(j, 30_000, 1 << i))
    assert prof[15] == 1.0 or prof[15] > 0.97
    lens = np.diff(p)
    assert abs(prof[32] - lens.std() / lens.mean()) <= 1e-3 and abs(prof[34] - lens.mean()) <= 1e-2
    assert abs(prof[33] - lens.max() / lens.mean()) <= 1e-2
    # uniform columns, many more columns than the sample can rank: no winner's curse — mass(top) stays top / K
    pu, ju, xu = synth.csr_fixed(200_000, 1_000_000, 16, seed=6)
    Au = D.DeviceCSR.from_host(pu, ju, xu, 1_000_000)
    pf = Au.profile()
    for i in (10, 13, 15, 17):
        assert abs(pf[i] - (1 << i) / 1e6) <= 0.02 + 0.25 * (1 << i) / 1e6, (i, pf[i])
    assert pf[32] == 0.0


def test_auto_reads_the_profile(gpu):
    """real-sim's shape, n = 128: sizes alone (uniform columns: an XCD's L2 serves 20 % of the gather) say PLANNED, the
    profile (80 %) says ROWSPLIT — which is what runs faster there (tools/zipf_map.py: 0.163 ms against 0.19 kept / 0.257
    rebuilt); many short skewed rows against a 16-column B: the lockstep price of the row-group form turns the choice to the
    kept plan"""
    from matrixextra_amd import device as D
    lib = _lib.load()
    al = C.c_void_p(256)

    def pick(A, n, keep, prof):
        out = C.c_int(0)
        _lib.check(lib.mxd_spmm_auto_algo3(C.c_int(A.m), C.c_int(n), C.c_int(A.K), C.c_int64(A.nnz), C.c_int(keep), C.c_int(_lib.MX_F64), al,
                                           C.c_size_t(n), al, C.c_size_t(n), C.c_int(0), prof, C.byref(out)))
        return out.value
    p, j, x = synth.device_csr_zipf(72_309, 20_958, 51, seed=21)
    A = D.DeviceCSR(p, j, x, 72_309, 20_958, int(j.numel()))
    assert pick(A, 128, 1, None) == 3 and pick(A, 128, 1, A.profile()) == 4
    B = torch.randn((20_958, 128), dtype=torch.float64, device="cuda")
    D.spmm(A, B)                                                   # keep_plan: the DeviceCSR hands its profile to AUTO
    assert lib.mxd_spmm_last_kernel() == b"spmm_rowsplit_kernel"
    D.spmm(A, B, keep_plan=False)                                  # the bare C-ABI AUTO profiles the matrix itself (nnz >= 2^21)
    assert lib.mxd_spmm_last_kernel() == b"spmm_rowsplit_kernel"
    p2, j2, x2 = synth.device_csr_zipf(1_000_000, 10_000, 10, sigma=0.5, seed=21)
    A2 = D.DeviceCSR(p2, j2, x2, 1_000_000, 10_000, int(j2.numel()))
    assert pick(A2, 16, 1, None) == 4 and pick(A2, 16, 1, A2.profile()) == 3


@pytest.fixture(scope="module")
def cfg2_zipf(gpu):
    from matrixextra_amd import device as D
    m, K, n = 1_000_000, 100_000, 128
    p, j, x = synth.csr_zipf(m, K, 40, alpha=1.0, sigma=1.0, seed=1)         # ~33 entries per row after the hot columns collide
    A = D.DeviceCSR.from_host(p, j, x, K)
    return dict(D=D, m=m, K=K, n=n, p=p, j=j, x=x, A=A)


def test_cfg2_zipf_spmm_properties(cfg2_zipf):
    """cfg2's shape with power-law columns and log-normal rows: AUTO (kept plan / one-shot), the planned sweep, the row-split
    kernel and the row-wave kernel — column checksum over the whole product, exact rows at both ends and around the longest row"""
    D, m, K, n, p, j, x, A = (cfg2_zipf[k] for k in ("D", "m", "K", "n", "p", "j", "x", "A"))
    lib = _lib.load()
    B1 = synth.dense_normal(K, n, seed=2)
    tB1 = torch.from_numpy(B1).cuda()
    w = np.bincount(j, weights=x, minlength=K)
    expect = w @ B1
    scale = np.abs(x).sum() * np.abs(B1).max()
    longest = int(np.argmax(np.diff(p)))
    starts = sorted({0, m - 256, max(0, min(m - 256, longest - 128))})
    runs = (("auto kept", lambda: D.spmm(A, tB1, colmajor=True)), ("auto one-shot", lambda: D.spmm(A, tB1, colmajor=True, keep_plan=False)),
            ("planned", lambda: D.spmm_planned(A, tB1, colmajor=True)), ("row-split", lambda: D.spmm(A, tB1, colmajor=False, algo=4)),
            ("row-wave", lambda: D.spmm(A, tB1, colmajor=False, algo=1)))
    for name, fn in runs:
        C1 = fn()
        kern = lib.mxd_spmm_last_kernel().decode()
        got = C1.sum(dim=0).cpu().numpy()
        assert np.max(np.abs(got - expect)) <= 1e-12 * scale, (name, kern)
        for r0 in starts:
            pp = (p[r0:r0 + 257] - p[r0]).astype(np.int32)
            ref = np.zeros(256 * n)
            O.gemm_csr_drm_as_drm(256, n, pp, j[p[r0]:p[r0 + 256]].copy(), x[p[r0]:p[r0 + 256]].copy(), B1.reshape(-1), n, ref, n, 4, True)
            ref = ref.reshape(256, n)
            g = C1[r0:r0 + 256].cpu().numpy()
            if name == "row-wave":
                np.testing.assert_array_equal(g, ref)
            else:
                assert np.max(np.abs(g - ref)) <= 1e-12 * max(1.0, np.abs(ref).max()), (name, kern, r0)
    assert bool(torch.isfinite(C1).all())


def test_merge_and_gather_with_skewed_rows(gpu):
    """CSR + CSR, CSR * CSR and X[rows, ] on log-normal row lengths (sigma 1.2: rows of 0 .. several thousand entries, mean 50 —
    the lane-group widths are chosen from the mean): structure bit-exact, values exact, at a fifth of cfg4's size through
    the exports, and the gather of 200k random rows of the cfg2-sized Zipf matrix"""
    from matrixextra_amd import exports as G
    m, K = 400_000, 400_000
    p1, j1, x1 = synth.csr_zipf(m, K, 50, alpha=0.9, sigma=1.2, seed=11)
    p2, j2, x2 = synth.csr_zipf(m, K, 50, alpha=0.9, sigma=1.2, seed=12)     # same popularity law: hot columns coincide, tails do not
    got = G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)
    ref = O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)
    for k in ("indptr", "indices", "values"):
        np.testing.assert_array_equal(got[k], ref[k])
    got = G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2)
    ref = O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2)
    for k in ("indptr", "indices", "values"):
        np.testing.assert_array_equal(got[k], ref[k])
    rows = synth.rows_with_replacement(200_000, m, seed=3)
    got = G.copy_csr_rows_numeric(p1, j1, x1, rows)
    ref = O.copy_csr_rows_numeric(p1, j1, x1, rows)
    for k in ("indptr", "indices", "values"):
        np.testing.assert_array_equal(got[k], ref[k])


def test_export_path_profiles_the_host_arrays(gpu):
    """the export-level product chooses its kernel family before the CSR is on the device: a host-side profile of the
    caller's arrays (csrc/api.hip host_csr_profile, the device pass's estimator) makes the same choice the DeviceCSR path
    makes — real-sim's shape with power-law columns, n = 128: the row-split kernel, where a matrix of the same sizes with
    uniform columns still gets the planned sweep; both results against the oracle"""
    from matrixextra_amd import exports as G
    lib = _lib.load()
    m, K, n = 72_309, 20_958, 128
    Y = np.asfortranarray(synth.dense_normal(n, K, seed=3))
    pz, jz, xz = synth.csr_zipf(m, K, 51, alpha=1.0, sigma=1.0, seed=21)
    got = G.tcrossprod_csr_dense_numeric(pz, jz, xz, Y, 1)
    assert lib.mxd_spmm_last_kernel() == b"spmm_rowsplit_kernel", lib.mxd_spmm_last_kernel()
    rows = 3000
    e = int(pz[rows])
    ref = O.tcrossprod_csr_dense_numeric(pz[:rows + 1], jz[:e], xz[:e], Y, 4)
    np.testing.assert_allclose(got[:rows], ref, rtol=1e-12, atol=1e-12 * float(np.abs(ref).max()))
    pu, ju, xu = synth.csr_fixed(m, K, 40, seed=22)
    got = G.tcrossprod_csr_dense_numeric(pu, ju, xu, Y, 1)
    assert lib.mxd_spmm_last_kernel() == b"spmm_plan_kernel", lib.mxd_spmm_last_kernel()
    e = int(pu[rows])
    ref = O.tcrossprod_csr_dense_numeric(pu[:rows + 1], ju[:e], xu[:e], Y, 4)
    np.testing.assert_allclose(got[:rows], ref, rtol=1e-12, atol=1e-12 * float(np.abs(ref).max()))
