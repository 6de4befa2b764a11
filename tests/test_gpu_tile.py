"""The LDS-tile SpMM kernel (csrc/spmm_tile.hip, MX_SPMM_TILE) against the oracle's gemm_csr_drm_as_drm /
gemm_csr_drm_as_dcm (src/matmul.cpp:118-185): every geometry (rows per lane group, compute wavefronts, 256- and 512-byte
slabs, 64 KB tiles with 32-entry windows and 32 KB tiles with 16-entry windows), both layouts of C, f64 and f32, rows
sorted by column (the LDS sweep), rows that are not (summed whole from global memory, flagged by the kernel's own
sortedness pass), repeated column ids, empty rows, rows denser than a window per tile, K not a multiple of the tile,
n not a multiple of the slab, NaN / Inf / signed zeros.

Every sum is the reference's storage-order FMA chain BIT FOR BIT, in both layouts: a row is summed by one lane group in
CSR order, and the padding steps of a group that has fewer entries in a tile than its neighbours add -0.0 * 0.0.
The one exception (round 6): a matrix whose profile shows rows several tile-windows long has THOSE rows cut into 2 / 4 / 8
interleaved parts, summed side by side and added in order — the same bits on every run, regrouped against the chain
(1e-13 in f64; north_star's tolerance is 1e-6); every other row keeps the chain."""
import numpy as np
import pytest

from conftest import rand_csr
from devmem import spmm_device
from matrixextra_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TILE = 5


def _oracle(p, j, x, B):
    return O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, True)        # the FMA chain


def variant(cpl=0, rg=0, small=False):
    return cpl + 4 * rg + (32 if small else 0)


def _run(p, j, x, B, colmajor, var=0, nw=0, rows_sorted=False):
    return spmm_device(p, j, x, B, colmajor, TILE, rows_sorted, npanels=nw, wg_per_cu=var)


def _same(got, ref):
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_array_equal(got[ok], ref[ok])
    np.testing.assert_array_equal(np.signbit(got[ok]), np.signbit(ref[ok]))


SHAPES = [                      # m, K, n, density
    (300, 3000, 100, 0.05),     # the vignette's shape in small: 12 tiles, ~13 entries per row and tile, n = 100 (a 4-column last slab)
    (37, 700, 32, 0.3),         # rows denser than a 32-entry window per tile: the wavefront goes round its rows again
    (1, 5000, 128, 0.5),        # one very long row
    (100, 50, 20, 0.4),         # test-matmul.R:108-114
    (1000, 257, 64, 0.02),      # K one past a tile
    (65, 512, 2, 0.2), (513, 1000, 48, 0.1), (90, 200, 96, 0.2), (10, 12, 300, 0.5), (241, 900, 130, 0.1),
]


@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", SHAPES)
def test_tile_is_the_storage_order_fma_chain_f64(gpu, m, K, n, dens, colmajor):
    p, j, x = rand_csr(m, K, dens, seed=m + 13 * n, sorted_cols=True, empty_rows=(0, m // 2, m - 1) if m > 4 else ())
    B = np.random.default_rng(n).normal(size=(K, n))
    ref = _oracle(p, j, x, B)
    for rows_sorted in (True, False):                  # vouched for / checked by the kernel's own pass
        _same(_run(p, j, x, B, colmajor, rows_sorted=rows_sorted), ref)


@pytest.mark.parametrize("small", [False, True])
@pytest.mark.parametrize("cpl", [1, 2])
@pytest.mark.parametrize("rg", [1, 2, 3, 4])
def test_tile_every_geometry(gpu, rg, cpl, small):
    m, K, n = 530, 1500, 72
    p, j, x = rand_csr(m, K, 0.08, seed=rg + 10 * cpl, sorted_cols=True, empty_rows=(3, 200))
    B = np.random.default_rng(5).normal(size=(K, n))
    ref = _oracle(p, j, x, B)
    for nw in (1, 4, 7, 15):
        for colmajor in (False, True):
            if colmajor and small and cpl == 2 and nw * 4 * rg > 128:
                continue                                  # (the column-major epilogue's LDS: refused by the library)
            _same(_run(p, j, x, B, colmajor, variant(cpl, rg, small), nw, True), ref)


def test_tile_refuses_a_geometry_whose_epilogue_does_not_fit(gpu):
    p, j, x = rand_csr(100, 300, 0.1, seed=1)
    B = np.zeros((300, 64))
    with pytest.raises(_lib.MxError, match="column-major epilogue"):
        _run(p, j, x, B, True, variant(2, 4, True), 15, True)


@pytest.mark.parametrize("colmajor", [False, True])
def test_tile_unsorted_rows_and_repeated_columns(gpu, colmajor):
    """some rows sorted, some shuffled (flagged by the sortedness pass and summed whole from global memory, in storage
    order), some with repeated column ids (non-decreasing: the LDS sweep takes them as they come; SpMM accumulates
    duplicates)"""
    m, K, n = 190, 2000, 100
    p, j, x = rand_csr(m, K, 0.1, seed=12, sorted_cols=True)
    rng = np.random.default_rng(5)
    for r in range(0, m, 3):
        s, e = p[r], p[r + 1]
        perm = rng.permutation(e - s)
        j[s:e], x[s:e] = j[s:e][perm], x[s:e][perm]
    for r in range(1, m, 7):
        s, e = p[r], p[r + 1]
        if e - s > 4:
            j[s + 2] = j[s + 1]
    B = rng.normal(size=(K, n))
    ref = _oracle(p, j, x, B)
    for var in (0, variant(1, 2), variant(2, 3), variant(1, 4, True)):
        _same(_run(p, j, x, B, colmajor, var, 0, False), ref)


@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", SHAPES[:7] + [(300, 500, 256, 0.05)])
def test_tile_f32(gpu, m, K, n, dens, colmajor):
    """B and C f32, the CSR values f64 narrowed per entry (src/matmul.cpp:53-57): the f32 FMA chain bit for bit"""
    if n % 4:
        pytest.skip("rows of B must be whole 16-byte vectors")
    p, j, x = rand_csr(m, K, dens, seed=m + 5 * n, sorted_cols=True, empty_rows=(0,) if m > 4 else ())
    B = np.random.default_rng(n).normal(size=(K, n)).astype(np.float32)
    ref = _oracle(p, j, x, B)
    for var in (0, variant(2, 2), variant(1, 3, True)):
        got = _run(p, j, x, B, colmajor, var, 0, False)
        assert got.dtype == np.float32
        _same(got, ref)


def test_tile_nonfinite_values_and_signed_zeros(gpu):
    """NaN / Inf in A or B reach exactly the cells they reach in the reference's loop — the padding steps of a lane group
    (fewer entries in the tile than its neighbours) multiply -0.0 with a row of zeros, never with a row of B — and a sum
    that is -0.0 stays -0.0 (x + -0.0 = x for every x)"""
    m, K, n = 150, 600, 24
    p, j, x = rand_csr(m, K, 0.1, seed=3, sorted_cols=True)
    B = np.random.default_rng(4).normal(size=(K, n))
    B[0, :] = np.inf
    B[j[5], 3] = np.nan
    B[17, :] = 0.0
    x[7] = np.inf
    x[p[40]:p[41]] = -1e-200                     # products that underflow to -0.0: the chain's running sum becomes -0.0 ...
    B[j[p[40]:p[41]], 5] = 1e-200                # ... and must not be flipped to +0.0 by a padding step
    ref = _oracle(p, j, x, B)
    assert np.signbit(ref[40, 5]) and ref[40, 5] == 0.0
    for colmajor in (False, True):
        for var in (0, variant(1, 1), variant(2, 4)):
            _same(_run(p, j, x, B, colmajor, var, 0, True), ref)


def test_tile_refuses_unaligned_rows_of_b(gpu):
    p, j, x = rand_csr(50, 300, 0.1, seed=1)
    with pytest.raises(_lib.MxError, match="16-byte"):
        _run(p, j, x, np.zeros((300, 33)), False)


def test_tile_at_the_published_shape_fifth(gpu):
    """a fifth of dense 100 x 1e4 %*% CSC 1e4 x 1e4, d = .05 (vignette Rmd:247-251): 2,000 rows of 500 entries, one round of
    workgroups; bitwise the FMA chain, and run-to-run identical"""
    p, j, x = synth.csr_fixed(2000, 10_000, 500, seed=7)
    B = synth.dense_normal(10_000, 100, seed=8)
    ref = _oracle(p, j, x, B)
    got = _run(p, j, x, B, False, 0, 0, True)
    _same(got, ref)
    np.testing.assert_array_equal(got, _run(p, j, x, B, False, 0, 0, False))
    assert _lib.load().mxd_spmm_last_kernel() == b"spmm_tile_kernel"


def test_auto_takes_the_tile_kernel_for_dense_ish_operands(gpu):
    """AUTO's cost model (csrc/spmm.hip spmm_auto_cost + csrc/spmm_tile.hip tile_est_us): TILE for the published product and
    for the reference's test densities once there are enough rows, never for the headline's sparse operands; and
    DeviceCSR.spmm really runs it (sortedness cached on the matrix: no flag pass), bit for bit the chain"""
    import ctypes as C
    import torch
    from matrixextra_amd import device as D
    lib = _lib.load()
    pick = C.c_int(0)
    al = C.c_void_p(256)
    for (m, n, K, nnz, colmajor, keep, want) in ((10_000, 100, 10_000, 5_000_000, 0, 0, 5), (10_000, 256, 10_000, 20_000_000, 1, 1, 5),
                                                 (100_000, 100, 1000, 40_000_000, 0, 1, 5), (1_000_000, 128, 100_000, 32_000_000, 1, 1, 3),
                                                 (100_000, 128, 10_000, 12_800_000, 1, 1, 4), (10_000, 101, 10_000, 5_000_000, 0, 0, 4)):
        _lib.check(lib.mxd_spmm_auto_algo2(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(nnz), C.c_int(keep), C.c_int(_lib.MX_F64), al,
                                           C.c_size_t(n), al, C.c_size_t(m if colmajor else n), C.c_int(colmajor), C.byref(pick)))
        assert pick.value == want, (m, n, K, nnz, pick.value)
    p, j, x = synth.csr_fixed(6000, 4000, 400, seed=3)                 # density .1
    B = synth.dense_normal(4000, 96, seed=4)
    A = D.DeviceCSR.from_host(p, j, x, 4000)
    for colmajor in (False, True):
        got = D.spmm(A, torch.from_numpy(B).cuda(), colmajor=colmajor).cpu().numpy()
        assert lib.mxd_spmm_last_kernel() == b"spmm_tile_kernel"
        _same(np.ascontiguousarray(got), _oracle(p, j, x, B))


@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_rows_dealt_by_length_keep_every_bit(gpu, monkeypatch, colmajor, dtype):
    """Rows of uneven length (round 6): the rows of a block are ranked by length and dealt to the lane groups — four rows of
    nearly the same length per visit, the visits in snake order over the wavefronts (tile_deal_rows_kernel).  A row is still
    summed by ONE group in storage order: bit for bit the oracle's FMA chain, in both layouts, with the map forced on
    (MXGPU_TILE_DEAL=1), forced off, and chosen from the matrix profile; every geometry the dealing can meet: a last block with
    fewer rows than slots, empty rows, a giant row, rows that are not sorted by column."""
    rng = np.random.default_rng(21)
    for m, K, n in ((1000, 2000, 100), (333, 900, 64), (5, 400, 32)):
        lens = np.minimum(rng.lognormal(mean=3.0, sigma=1.4, size=m).astype(np.int64), K)
        lens[rng.integers(0, m, size=max(1, m // 50))] = 0
        lens[m // 3] = K                                             # one full row
        p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
        j = np.concatenate([np.sort(rng.choice(K, size=int(L), replace=False)) for L in lens]).astype(np.int32)
        j[p[2]:p[3]] = j[p[2]:p[3]][::-1]                            # one row not sorted by column: summed whole, flagged
        x = rng.uniform(-1, 1, size=j.size)
        B = rng.normal(size=(K, n)).astype(dtype)
        ref = _oracle(p, j, x, B)
        outs = []
        monkeypatch.setenv("MXGPU_TILE_SPLIT", "0")                  # (long rows cut into parts: the next test)
        for deal in ("1", "0"):
            monkeypatch.setenv("MXGPU_TILE_DEAL", deal)
            for greedy in (("2", "0") if deal == "1" else ("0",)):    # row blocks in one pass (by units and weight) / cut every R rows
                monkeypatch.setenv("MXGPU_TILE_GREEDY", greedy)
                for var, nw in ((0, 0), (variant(1, 3), 5), (variant(2, 2), 14)):
                    got = _run(p, j, x, B, colmajor, var, nw, rows_sorted=False)
                    _same(got, ref)
            outs.append(got)
        monkeypatch.delenv("MXGPU_TILE_GREEDY")
        assert np.array_equal(outs[0], outs[1])
        monkeypatch.delenv("MXGPU_TILE_DEAL")
        # through the device layer with the matrix profile in scope (cv ~ 2: dealt) — the same bits again
        import torch
        from matrixextra_amd import device as D
        A = D.DeviceCSR.from_host(p, j, x, K)
        assert A.profile()[32] > 0.15
        Bd = torch.from_numpy(B).cuda()
        got = D.spmm(A, Bd, colmajor=colmajor, algo=TILE).cpu().numpy()
        _same(got, ref)
        # the cuts and the slot -> row map are kept for the next product with the same row pointers; they must not survive the
        # release of the thread's workspaces (a freed buffer may come back at the same address)
        _same(D.spmm(A, Bd, colmajor=colmajor, algo=TILE).cpu().numpy(), ref)
        _lib.check(_lib.load().mxd_release_workspaces())
        _same(D.spmm(A, Bd, colmajor=colmajor, algo=TILE).cpu().numpy(), ref)
        monkeypatch.delenv("MXGPU_TILE_SPLIT")


@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_long_rows_cut_into_parts(gpu, monkeypatch, colmajor, dtype):
    """Long rows (round 6): with the rows dealt by length, a row several windows per tile long is cut into 2 / 4 / 8 interleaved
    parts that take slots of their own and are added up in order by tile_combine_kernel — forced (MXGPU_TILE_SPLIT=1) and chosen
    from the matrix profile.  Those rows are the chain regrouped (1e-13 f64 / 1e-5 f32 of the row's largest term sum); rows of
    at most 40 entries are never cut (a part is 40 entries per tile at least) and keep every bit; a long row that is not sorted
    by column stays whole — every bit again; the same bits on every run, with a kept map and after the workspaces' release."""
    import torch
    from matrixextra_amd import device as D
    rng = np.random.default_rng(33)
    tol = 1e-13 if dtype == np.float64 else 2e-5
    for m, K, n in ((1000, 2000, 100), (333, 900, 64), (64, 6000, 132), (5, 400, 32)):
        lens = np.minimum(rng.lognormal(mean=3.0, sigma=1.4, size=m).astype(np.int64), K)
        lens[rng.integers(0, m, size=max(1, m // 50))] = 0
        lens[m // 3] = K; lens[m // 2] = K // 2; lens[1] = K - 1     # long rows; row 1 will not be sorted by column
        p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
        j = np.concatenate([np.sort(rng.choice(K, size=int(L), replace=False)) for L in lens]).astype(np.int32)
        j[p[1]:p[2]] = j[p[1]:p[2]][::-1]
        x = rng.uniform(-1, 1, size=j.size)
        B = rng.normal(size=(K, n)).astype(dtype)
        ref = _oracle(p, j, x, B)
        scale = np.abs(_oracle(p, j, np.abs(x), np.abs(B)))          # sum of |terms| per element
        short = lens <= 40
        short[1] = True                                              # (not sorted by column: whole)

        def check(got):
            assert np.all(np.abs(got.astype(np.float64) - ref) <= tol * np.maximum(scale, 1e-300))
            _same(got[short], ref[short])

        monkeypatch.setenv("MXGPU_TILE_DEAL", "1")
        monkeypatch.setenv("MXGPU_TILE_SPLIT", "1")
        first = None
        for var, nw in ((0, 0), (variant(1, 3), 5), (variant(2, 2), 14), (variant(1, 2, True), 9)):
            got = _run(p, j, x, B, colmajor, var, nw, rows_sorted=False)
            check(got)
            if var == 0:
                first = got
        monkeypatch.delenv("MXGPU_TILE_DEAL"); monkeypatch.delenv("MXGPU_TILE_SPLIT")
        A = D.DeviceCSR.from_host(p, j, x, K)
        Bd = torch.from_numpy(B).cuda()
        a = D.spmm(A, Bd, colmajor=colmajor, algo=TILE).cpu().numpy()
        check(a)
        b = D.spmm(A, Bd, colmajor=colmajor, algo=TILE).cpu().numpy()             # the kept map
        assert np.array_equal(a, b)
        _lib.check(_lib.load().mxd_release_workspaces())
        assert np.array_equal(D.spmm(A, Bd, colmajor=colmajor, algo=TILE).cpu().numpy(), a)
        if m >= 64:
            assert not np.array_equal(a, ref), "no row was cut: the test did not reach the parts"


def test_tile_parts_beyond_the_scratch_stay_whole(gpu, monkeypatch):
    """The parts' partial sums live in a scratch matrix of at most 64 MiB: a matrix that wants more parts than it has rows keeps
    the rest of its long rows whole (tile_deal_rows_kernel hands the rows out with one atomic: a refused row has spent a slot of
    the parents' list, which tile_combine_kernel must skip — round 6's fuzz run with tiny parts found it reading that slot).
    20,000 rows of 100 entries, parts of 8 entries (MXGPU_TILE_PART=1): 140,000 parts wanted, 65,536 rows of scratch at n = 128."""
    rng = np.random.default_rng(5)
    m, K, n = 20000, 2000, 128
    lens = np.full(m, 100, dtype=np.int64)
    p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    j = np.sort(rng.integers(0, K, size=(m, 100)), axis=1).astype(np.int32).ravel()
    x = rng.uniform(-1, 1, size=j.size)
    B = rng.normal(size=(K, n))
    ref = _oracle(p, j, x, B)
    monkeypatch.setenv("MXGPU_TILE_DEAL", "1"); monkeypatch.setenv("MXGPU_TILE_SPLIT", "1"); monkeypatch.setenv("MXGPU_TILE_PART", "1")
    for colmajor in (False, True):
        got = _run(p, j, x, B, colmajor, rows_sorted=True)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
        assert not np.array_equal(got, ref)                          # (parts were cut)
        assert np.array_equal(got, _run(p, j, x, B, colmajor, rows_sorted=True))
