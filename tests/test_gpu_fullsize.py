"""Full-size runs of BASELINE.json's configs on the GPU, checked through size-independent properties
(the CPU oracle would need minutes per case at these sizes):

  cfg2  SpMM 1M x 100k, 32/row, f64 x dense 100k x 128:
        column checksum  1^T (A B) == (A^T 1)^T B   (O(nnz + K n) on the host)
        linearity        A (B1 + 2 B2) == A B1 + 2 A B2
        sampled rows     the first and last 512 rows against the oracle (bitwise for the CSR-order kernels,
                         1e-12 for the planned kernel, which regroups the sum by column panel)
  cfg3  SpMV + gather of 200k random rows (with replacement):
        y == (A B)[:, 0] of a one-column SpMM; sum(y) checksum;  gathered row sums == row sums[rows];
        indptr of the result == cumsum of the picked row lengths (bit-exact)
  cfg5  one GPU's row block of configs[4]: 1M x 200k, 64/row (CSR values f64) x dense 200k x 256 FLOAT32, both layouts,
        AUTO and the explicitly planned kernel: column checksum (f64 sum of the f32 result), sampled rows against
        mxo_gemm_csr_drm_as_drm_f32 at 1e-5, linearity  (alpha narrowed per nonzero: matmul.cpp:53-57,361-375)
  cfg4  CSR + CSR, CSR - CSR, CSR * CSR on 2M x 2M, 50/row (nnz 1e8 each, ~50 % shared pattern):
        rows sorted & unique; nnz(A+B) + nnz(A*B) == nnz(A) + nnz(B) (inclusion-exclusion);
        sum(values(A+B)) == sum(A) + sum(B); (A-B) has the same structure as (A+B); A + A == 2 A exactly
        (general path); sampled rows bit-exact against the oracle
"""
import numpy as np
import pytest
import torch

from matrixextra_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg2(gpu):
    from matrixextra_amd import device as D
    m, K, n = 1_000_000, 100_000, 128
    p, j, x = synth.csr_fixed(m, K, 32)
    A = D.DeviceCSR.from_host(p, j, x, K)
    return dict(D=D, m=m, K=K, n=n, p=p, j=j, x=x, A=A)


def test_cfg2_spmm_properties(cfg2):
    D, m, K, n, p, j, x, A = (cfg2[k] for k in ("D", "m", "K", "n", "p", "j", "x", "A"))
    B1 = synth.dense_normal(K, n, seed=2)
    B2 = synth.dense_normal(K, n, seed=22)
    tB1, tB2 = torch.from_numpy(B1).cuda(), torch.from_numpy(B2).cuda()
    for colmajor, algo in ((True, 0), (False, 0), (True, 1), (True, 2)):      # AUTO (planned), row-wave, slab
        C1 = D.spmm(A, tB1, colmajor=colmajor, algo=algo)
        # column checksum: w = A^T 1 (length K), 1^T C = w^T B
        w = np.bincount(j, weights=x, minlength=K)
        expect = w @ B1
        got = C1.sum(dim=0).cpu().numpy()
        scale = np.abs(x).sum() * np.abs(B1).max()
        assert np.max(np.abs(got - expect)) <= 1e-12 * scale
        # exact rows at both ends (same summation order + FMA as the oracle => bitwise)
        for r0 in (0, m - 512):
            rows = slice(r0, r0 + 512)
            ref = np.zeros(512 * n)
            pp = (p[r0:r0 + 513] - p[r0]).astype(np.int32)
            O.gemm_csr_drm_as_drm(512, n, pp, j[p[r0]:p[r0 + 512]].copy(), x[p[r0]:p[r0 + 512]].copy(),
                                  B1.reshape(-1), n, ref, n, 4, True)
            got = C1[rows].cpu().numpy()
            if algo in (1, 2):        # CSR-order summation with FMA: bitwise equal to the FMA oracle
                np.testing.assert_array_equal(got, ref.reshape(512, n))
            else:                     # planned kernel: per-panel partial sums added in panel order
                np.testing.assert_allclose(got, ref.reshape(512, n), rtol=1e-12, atol=1e-13)
    # linearity
    C1 = D.spmm(A, tB1)
    C2 = D.spmm(A, tB2)
    C12 = D.spmm(A, tB1 + 2.0 * tB2)
    err = (C12 - (C1 + 2.0 * C2)).abs().max().item()
    assert err <= 1e-11 * (C1.abs().max().item() + 2 * C2.abs().max().item())


def test_cfg3_spmv_and_gather(cfg2):
    D, m, K, p, j, x, A = (cfg2[k] for k in ("D", "m", "K", "p", "j", "x", "A"))
    v = synth.dense_normal(K, 1, seed=5).reshape(-1)
    tv = torch.from_numpy(v).cuda()
    y = D.spmv(A, tv)
    y_mm = D.spmm(A, tv.reshape(K, 1).contiguous())          # independent kernel (row-wave, VEC=1 path)
    assert (y - y_mm[:, 0]).abs().max().item() <= 1e-12 * np.abs(x).max() * np.abs(v).max() * 32
    w = np.bincount(j, weights=x, minlength=K)
    assert abs(y.sum().item() - w @ v) <= 1e-9 * np.abs(x).sum() * np.abs(v).max()
    # float32 / integer kinds on the full matrix against the f64 result
    yf = D.spmv(A, tv.float())
    assert yf.dtype == torch.float32 and (yf.double() - y).abs().max().item() <= 1e-4 * y.abs().max().item()
    vi = torch.randint(-5, 6, (K,), dtype=torch.int32, device="cuda")
    yi = D.spmv(A, vi, v_dtype=_lib.MX_I32)
    yi_ref = D.spmv(A, vi.double())
    assert torch.equal(yi, yi_ref)
    # gather of 200k random rows
    rows = synth.rows_with_replacement(200_000, m)
    G = D.csr_gather_rows(A, torch.from_numpy(rows).cuda())
    gp, gj, gx = G.to_host()
    lens = (p[1:] - p[:-1])[rows]
    np.testing.assert_array_equal(gp, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32))
    assert G.nnz == int(lens.sum()) == 6_400_000
    take = np.random.default_rng(0).integers(0, rows.size, size=2000)
    for t in take:
        r = rows[t]
        np.testing.assert_array_equal(gj[gp[t]:gp[t + 1]], j[p[r]:p[r + 1]])
        np.testing.assert_array_equal(gx[gp[t]:gp[t + 1]], x[p[r]:p[r + 1]])
    rowsum = np.add.reduceat(x, p[:-1])
    np.testing.assert_allclose(np.add.reduceat(gx, gp[:-1]), rowsum[rows], rtol=0, atol=0)


def test_cfg5_shard_spmm_f32(gpu):
    from matrixextra_amd import device as D
    m, K, n, k = 1_000_000, 200_000, 256, 64
    p, j, x = synth.csr_fixed(m, K, k)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B1 = synth.dense_normal(K, n, seed=2, dtype=np.float32)
    B2 = synth.dense_normal(K, n, seed=22, dtype=np.float32)
    tB1, tB2 = torch.from_numpy(B1).cuda(), torch.from_numpy(B2).cuda()
    w = np.bincount(j, weights=x, minlength=K)                    # A^T 1 in f64
    expect = w @ B1.astype(np.float64)
    scale = np.abs(x).sum() * np.abs(B1).max()
    blocks = {}
    for r0 in (0, 499_744, m - 256):                              # oracle rows (f32 arithmetic, FMA like the GPU)
        pp = (p[r0:r0 + 257] - p[r0]).astype(np.int32)
        ref = np.zeros(256 * n, dtype=np.float32)
        O.gemm_csr_drm_as_drm(256, n, pp, j[p[r0]:p[r0 + 256]].copy(), x[p[r0]:p[r0 + 256]].copy(), B1.reshape(-1), n,
                              ref, n, 4, True)
        blocks[r0] = ref.reshape(256, n)
    for colmajor in (True, False):
        for planned in (False, True):
            C1 = D.spmm_planned(A, tB1, colmajor=colmajor) if planned else D.spmm(A, tB1, colmajor=colmajor)
            assert C1.dtype == torch.float32 and C1.shape == (m, n)
            if not planned:
                assert _lib.load().mxd_spmm_last_kernel().decode() == "spmm_plan_kernel"   # what AUTO picks at this size
            got = C1.double().sum(dim=0).cpu().numpy()
            assert np.max(np.abs(got - expect)) <= 1e-8 * scale
            for r0, ref in blocks.items():
                np.testing.assert_allclose(C1[r0:r0 + 256].cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    C1, C2 = D.spmm(A, tB1), D.spmm(A, tB2)
    C12 = D.spmm(A, tB1 + 2.0 * tB2)
    err = (C12.double() - (C1.double() + 2.0 * C2.double())).abs().max().item()
    assert err <= 2e-5 * (C1.abs().max().item() + 2 * C2.abs().max().item())
    print("cfg5 shard plan:", A.plan_info())


def test_cfg4_merge_properties(gpu):
    from matrixextra_amd import device as D
    m = K = 2_000_000
    p1, j1, x1 = synth.csr_fixed(m, K, 50)
    p2, j2, x2 = synth.csr_overlapping(p1, j1, K, 50)
    A, B = D.DeviceCSR.from_host(p1, j1, x1, K), D.DeviceCSR.from_host(p2, j2, x2, K)
    S = D.csr_elemwise(_lib.MX_OP_ADD, A, B)
    Dm = D.csr_elemwise(_lib.MX_OP_SUB, A, B)
    M = D.csr_elemwise(_lib.MX_OP_MUL, A, B)
    assert S.nnz + M.nnz == A.nnz + B.nnz                       # inclusion-exclusion on the patterns
    for R, op in ((S, _lib.MX_OP_ADD), (M, _lib.MX_OP_MUL)):    # the one-pass kernel == count -> scan -> fill (default)
        R2 = D.csr_elemwise(op, A, B, two_pass=False)
        assert R2.nnz == R.nnz and torch.equal(R2.indptr, R.indptr) and torch.equal(R2.indices, R.indices) \
            and torch.equal(R2.values, R.values)
        del R2
    assert torch.equal(S.indptr, Dm.indptr) and torch.equal(S.indices, Dm.indices)
    assert abs(S.values.sum().item() - (x1.sum() + x2.sum())) <= 1e-9 * (np.abs(x1).sum() + np.abs(x2).sum())
    assert abs(Dm.values.sum().item() - (x1.sum() - x2.sum())) <= 1e-9 * (np.abs(x1).sum() + np.abs(x2).sum())
    for R in (S, M):                                            # rows sorted, unique
        assert D.DeviceCSR(R.indptr, R.indices, None, m, K, R.nnz).rows_sorted()
        d = R.indices[1:] - R.indices[:-1]
        starts = torch.zeros(R.nnz, dtype=torch.bool, device="cuda")
        starts[R.indptr[1:-1].long().clamp(max=R.nnz - 1)] = True
        assert bool(((d > 0) | starts[1:]).all())
    # A + A' (same pattern, different buffers => general path) == 2 A exactly
    A2 = D.DeviceCSR(A.indptr.clone(), A.indices.clone(), A.values.clone(), m, K, A.nnz)
    T = D.csr_elemwise(_lib.MX_OP_ADD, A, A2)
    assert torch.equal(T.indptr, A.indptr) and torch.equal(T.indices, A.indices) and torch.equal(T.values, 2 * A.values)
    # sampled row blocks bit-exact against the oracle
    sp, sj, sx = S.to_host()
    mp_, mj, mx_ = M.to_host()
    for r0 in (0, 777_777, m - 1000):
        r1 = r0 + 1000
        a = (p1[r0:r1 + 1] - p1[r0]).astype(np.int32), j1[p1[r0]:p1[r1]].copy(), x1[p1[r0]:p1[r1]].copy()
        b = (p2[r0:r1 + 1] - p2[r0]).astype(np.int32), j2[p2[r0]:p2[r1]].copy(), x2[p2[r0]:p2[r1]].copy()
        o = O.add_csr_elemwise(a[0], b[0], a[1], b[1], a[2], b[2], False)
        np.testing.assert_array_equal(sp[r0:r1 + 1] - sp[r0], o["indptr"])
        np.testing.assert_array_equal(sj[sp[r0]:sp[r1]], o["indices"])
        np.testing.assert_array_equal(sx[sp[r0]:sp[r1]], o["values"])
        o = O.multiply_csr_elemwise(a[0], b[0], a[1], b[1], a[2], b[2])
        np.testing.assert_array_equal(mj[mp_[r0]:mp_[r1]], o["indices"])
        np.testing.assert_array_equal(mx_[mp_[r0]:mp_[r1]], o["values"])


def test_auto_spmm_on_skewed_rows_large(gpu):
    """AUTO on a big matrix with log-normal row lengths (some rows thousands of entries long, some empty): whichever
    kernel AUTO settles on (planned, or row-wave when the interleave would pad too much) must match the oracle."""
    from matrixextra_amd import device as D
    m, K, n = 300_000, 120_000, 128
    p, j, x = synth.csr_skewed(m, K, 20, seed=11, sigma=1.4)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = synth.dense_normal(K, n, seed=3)
    C = D.spmm(A, torch.from_numpy(B).cuda(), colmajor=True)
    Cp = D.spmm_planned(A, torch.from_numpy(B).cuda(), colmajor=True)
    w = np.bincount(j, weights=x, minlength=K)
    scale = np.abs(x).sum() * np.abs(B).max()
    for got in (C, Cp):
        assert np.max(np.abs(got.sum(dim=0).cpu().numpy() - w @ B)) <= 1e-12 * scale
    long_row = int(np.argmax(np.diff(p)))
    for r0 in (0, max(0, long_row - 100), m - 400):
        rows = 400
        pp = (p[r0:r0 + rows + 1] - p[r0]).astype(np.int32)
        ref = np.zeros(rows * n)
        O.gemm_csr_drm_as_drm(rows, n, pp, j[p[r0]:p[r0 + rows]].copy(), x[p[r0]:p[r0 + rows]].copy(), B.reshape(-1), n,
                              ref, n, 4, True)
        for got in (C, Cp):
            np.testing.assert_allclose(got[r0:r0 + rows].cpu().numpy(), ref.reshape(rows, n), rtol=1e-11, atol=1e-11)
    print("kernel picked by AUTO:", _lib.load().mxd_spmm_last_kernel().decode(), "plan:", A.plan_info(), "nnz", A.nnz)


def _direct_rows(p_host, j, x, B, rows):
    out = []
    for i in rows:
        s, e = int(p_host[i]), int(p_host[i + 1])
        out.append((x[s:e, None].to(torch.float64) * B[j[s:e].long()].to(torch.float64)).sum(dim=0).cpu().numpy())
    return out


@pytest.mark.parametrize("what", ["result", "dense"])
def test_operands_above_4_gib(gpu, what):
    """32-bit offset bugs: a 6.1 GB result (6M x 128 f64, both layouts) and a 5.1 GB dense operand (K = 5M), every SpMM
    kernel, rows around the 2^32-byte marks against a direct evaluation (maximum sizes: SURVEY §4 edge cases)."""
    from matrixextra_amd import device as D
    if what == "result":
        m, K, n, r, algos, layouts = 6_000_000, 50_000, 128, 8, (0, 1), (True, False)
        rows = [0, 1, 4_194_303, 4_194_304, 4_500_000, m - 2, m - 1]      # 4 GiB / (128 * 8 B) = row 4,194,304
    else:
        m, K, n, r, algos, layouts = 300_000, 5_000_000, 128, 16, (0, 1, 2, 3), (True,)
        rows = [0, 1, 150_000, m - 1]
    p, j, x = synth.device_csr_fixed(m, K, r)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
    B = torch.randn(K, n, dtype=torch.float64, device="cuda")
    refs = _direct_rows(p.cpu().numpy(), j, x, B, rows)
    if what == "dense":
        assert int(j.max()) * n * 8 > 2 ** 32                             # rows of B beyond the 4 GiB mark are read
    for colmajor in layouts:
        for algo in algos:
            C = torch.full((n, m) if colmajor else (m, n), float("nan"), dtype=torch.float64, device="cuda")
            if algo == 3:
                D.spmm_planned(A, B, out=C, colmajor=colmajor)
            else:
                D.spmm(A, B, out=C, colmajor=colmajor, algo=algo)
            torch.cuda.synchronize()
            for i, ref in zip(rows, refs):
                got = (C[:, i] if colmajor else C[i]).cpu().numpy()
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
            assert bool(torch.isfinite(C[:, ::4099] if colmajor else C[::4099]).all())   # nothing left unwritten
            del C


def test_results_beyond_int32_are_refused(gpu):
    """A union of 2 x 1.1e9 disjoint entries and a gather of 3e9 entries do not fit R's int32 index vectors: the count
    passes must come back with an error (the reference would overflow its int counters, operators.cpp:402-406,
    slice.cpp:234-235), not with a wrapped size."""
    from matrixextra_amd import device as D
    m, per = 11_000_000, 100
    indptr = (torch.arange(m + 1, dtype=torch.int64, device="cuda") * per).to(torch.int32)
    base = (torch.arange(m * per, dtype=torch.int32, device="cuda") % per)
    A = D.DeviceCSR(indptr, base, None, m, 2 * per, m * per)
    B = D.DeviceCSR(indptr, base + per, None, m, 2 * per, m * per)
    A.values = B.values = torch.empty(0, dtype=torch.float64, device="cuda")     # the count pass never reads values
    with pytest.raises(_lib.MxError, match="int32"):
        D.csr_elemwise(_lib.MX_OP_ADD, A, B)
    del A, B, base
    # gather: 30M copies of 100-entry rows
    small = D.DeviceCSR(indptr[:1001].clone(), (torch.arange(1000 * per, dtype=torch.int32, device="cuda") % per),
                        torch.ones(1000 * per, dtype=torch.float64, device="cuda"), 1000, per, 1000 * per)
    rows = (torch.arange(30_000_000, dtype=torch.int32, device="cuda") % 1000)
    with pytest.raises(_lib.MxError, match="int32"):
        D.csr_gather_rows(small, rows)


def test_kernels_at_the_int32_limit(gpu):
    """nnz = 2.1e9 (just under R's int32 limit; values 16.8 GB): every SpMV kernel bit for bit, the sortedness check, the
    gather, row-wave and planned SpMM, and merges whose union has 1.575e9 entries (offsets beyond 2^33 bytes) — the
    maximum sizes the boundary admits.  tools/maxnnz_probe.py, run as a child process (it takes ~60 GB of HBM)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "maxnnz_probe.py")], cwd=root, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0 and "max nnz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_published_workload_full_size(gpu):
    """The one workload the reference publishes a number for, at its size (vignettes/Introducing_MatrixExtra.Rmd:247-251):
    dense 100 x 1e4 %*% CSC 1e4 x 1e4, density .05, through the export (matmul_dense_csc_numeric, src/matmul.cpp:188-235) —
    AUTO = the LDS-tile kernel (round 5; the row-split kernel with column panels before) — against the WHOLE oracle product
    (60 ms on the host), BIT FOR BIT its storage-order FMA chain; and the row-split kernel at device level for every panel
    count, bit for bit the same chain (row-major C, one segment)."""
    from matrixextra_amd import device as D, exports as G
    m, K, n = 10_000, 10_000, 100
    p, j, x = synth.csr_fixed(m, K, 500, seed=7)
    X = np.asfortranarray(synth.dense_normal(n, K, seed=8))            # Y_dense, column-major 100 x 1e4
    got = G.matmul_dense_csc_numeric(X, p, j, x, 1)
    assert _lib.load().mxd_spmm_last_kernel() == b"spmm_tile_kernel"
    ref = O.matmul_dense_csc(X, p, j, x, O.max_threads(), True)        # FMA chain in storage order
    np.testing.assert_array_equal(got, ref)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()             # K x n row-major
    for P in (1, 3, 4, 7):
        C = D.spmm(A, B, colmajor=False, algo=4, npanels=P, wg_per_cu=1).cpu().numpy()     # m x n = t(result)
        np.testing.assert_array_equal(C, ref.T)


def test_rowsplit_panels_mid_size_properties(gpu):
    """100k x 10k, 128 per row, n = 128, column-major C (the shape where four column panels beat the kept plan):
    column checksum and linearity over the whole product, sampled rows against the oracle."""
    from matrixextra_amd import device as D
    m, K, n = 100_000, 10_000, 128
    p, j, x = synth.csr_fixed(m, K, 128, seed=3)
    A = D.DeviceCSR.from_host(p, j, x, K)
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    B1 = torch.randn((K, n), dtype=torch.float64, device="cuda", generator=g)
    B2 = torch.randn((K, n), dtype=torch.float64, device="cuda", generator=g)

    def prod(Bt):
        out = D.spmm(A, Bt, colmajor=True)                             # AUTO
        return out.clone()
    C1 = prod(B1)
    assert _lib.load().mxd_spmm_last_kernel() == b"spmm_rowsplit_kernel"
    colsum_A = np.zeros(K)
    np.add.at(colsum_A, j, x)
    expect = torch.from_numpy(colsum_A).cuda() @ B1
    scale = float(expect.abs().max())
    assert float((C1.sum(dim=0) - expect).abs().max()) <= 1e-9 * scale
    C12 = prod(B1 + 2.0 * B2)
    C2 = prod(B2)
    assert float((C12 - (C1 + 2.0 * C2)).abs().max()) <= 1e-10 * float(C12.abs().max())
    rows = 256
    for r0 in (0, m - rows):
        ref = np.zeros(rows * n)
        lo, hi = int(p[r0]), int(p[r0 + rows])
        O.gemm_csr_drm_as_drm(rows, n, (p[r0:r0 + rows + 1] - p[r0]).astype(np.int32), j[lo:hi].copy(), x[lo:hi].copy(),
                              B1.cpu().numpy().reshape(-1), n, ref, n, 1, True)
        np.testing.assert_allclose(C1[r0:r0 + rows].cpu().numpy(), ref.reshape(rows, n), rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_planned_kernel_same_bits_run_to_run_on_skewed_rows(gpu, dtype):
    """Log-normal row lengths put octets in the plan's dealt layout: long rows are shared by several lane groups of the
    wavefront that owns the octet and folded with LDS atomics (spmm_plan.hip).  Only that one wavefront touches the sums,
    so the result is the same bits on every run — with the plan kept, with the plan rebuilt, with other kernels running
    in between (include/mxgpu.h, 'Determinism')."""
    from matrixextra_amd import device as D
    m, K, n = 400_000, 100_000, 128
    p, j, x = synth.csr_skewed_fast(m, K, 24, seed=21, sigma=1.3)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.randn((K, n), dtype=dtype, device="cuda")
    bits = torch.int64 if dtype == torch.float64 else torch.int32
    for colmajor in (False, True):
        ref = D.spmm_planned(A, B, colmajor=colmajor).clone()
        assert A.plan_info()["padded_entries"] < 1.25 * A.nnz          # dealt octets (the bundle layout pads ~1.8x here)
        for r in range(12):
            if r % 3 == 0:
                _ = torch.randn((2048, 2048), device="cuda") @ torch.randn((2048, 2048), device="cuda")
            got = D.spmm_planned(A, B, colmajor=colmajor, rebuild_plan=(r % 2 == 1))
            assert torch.equal(got.view(bits), ref.view(bits))
