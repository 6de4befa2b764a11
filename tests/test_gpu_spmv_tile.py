"""The flat-stream SpMV kernel (csrc/spmv_flat.hip, MX_SPMV_FLAT, what AUTO runs for large matrices) and the LDS-panel
kernel (csrc/spmv_tile.hip, MX_SPMV_TILE) against the oracle (matmul_csr_dvec<>,
src/matmul.cpp:381-419).  Rows of up to 256 entries are summed in storage order from separately rounded products, i.e.
exactly the reference's loop without FMA contraction: compared BITWISE with the oracle (built -ffp-contract=off); longer
rows are summed by a wavefront (reassociated): 1e-12."""
import ctypes as C

import numpy as np
import pytest

from matrixextra_amd import _lib, synth
from oracle import oracle as O
from conftest import rand_csr
from devmem import spmv_device

pytestmark = pytest.mark.gpu
NA = np.int32(-2147483648)
TILE, GROUP, FLAT = 2, 1, 3


def ragged_csr(lens, K, seed, sort=True, dup=False):
    rng = np.random.default_rng(seed)
    lens = np.asarray(lens, dtype=np.int64)
    p = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    j = rng.integers(0, K, size=int(p[-1]), dtype=np.int32)         # duplicates inside a row are allowed (they add up)
    if sort:
        for r in range(lens.size):
            j[p[r]:p[r + 1]].sort()
    if not dup and K >= lens.max(initial=0):
        for r in range(lens.size):
            if lens[r]:
                j[p[r]:p[r + 1]] = np.sort(rng.choice(K, size=lens[r], replace=False)) if sort else \
                    rng.choice(K, size=lens[r], replace=False)
    x = np.round(rng.normal(size=int(p[-1])), 3)
    return p.astype(np.int32), j, x


def compare(got, ref, lens, rtol):
    """Rows of at most 256 entries: bitwise; longer rows (summed by a wavefront): rtol."""
    short = lens <= 256
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_array_equal(got[short & ok], ref[short & ok])
    if (~short & ok).any():
        np.testing.assert_allclose(got[~short & ok], ref[~short & ok], rtol=rtol, atol=rtol * np.abs(ref[ok]).max())


def check_all_kinds(p, j, x, K, seed):
    for algo in (FLAT, TILE):
        _check_all_kinds(p, j, x, K, seed, algo)


def _check_all_kinds(p, j, x, K, seed, TILE):
    rng = np.random.default_rng(seed)
    lens = np.diff(p)
    v = rng.normal(size=K)
    compare(spmv_device(p, j, x, v, _lib.MX_F64, TILE), O.matmul_csr_dvec_numeric(p, j, x, v), lens, 1e-12)
    vf = v.astype(np.float32)
    got = spmv_device(p, j, x, vf, _lib.MX_F32, TILE)
    assert got.dtype == np.float32
    compare(got, O.matmul_csr_dvec_float32(p, j, x, vf), lens, 1e-5)
    vi = rng.integers(-9, 9, size=K).astype(np.int32)
    vl = rng.integers(0, 2, size=K).astype(np.int32)
    for k in rng.integers(0, K, size=max(1, K // 50)):
        vi[k] = NA
        vl[k] = NA
    for kind, of, vv in ((_lib.MX_I32, O.matmul_csr_dvec_integer, vi), (_lib.MX_LGL, O.matmul_csr_dvec_logical, vl)):
        got, ref = spmv_device(p, j, x, vv, kind, TILE), of(p, j, x, vv)
        compare(got, ref, lens, 1e-12)
        na_rows = np.isnan(ref)
        if na_rows.any():                                             # R's NA_real_ (low word 1954) where an NA entry is touched
            assert (got[na_rows].view(np.uint64) & 0xFFFFFFFF == 1954).all()
        # and the same answer as the lane-group kernel
        np.testing.assert_allclose(spmv_device(p, j, x, vv, kind, GROUP)[~na_rows], ref[~na_rows], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("m,K,dens", [(60, 25, 0.3), (1, 10, 0.9), (500, 700, 0.02), (300, 40, 1.0), (64, 2000, 0.2),
                                      (3000, 16384, 0.002), (2000, 16385, 0.004), (900, 40_000, 0.003)])
def test_tile_small_shapes_all_kinds(gpu, m, K, dens):
    p, j, x = rand_csr(m, K, dens, seed=m + K, sorted_cols=False, empty_rows=(0,) if m > 2 else ())
    if p[-1] < 4:
        pytest.skip("fewer than 4 entries")
    check_all_kinds(p, j, x, K, seed=K)


def test_tile_many_tiles_uniform_and_odd_lengths(gpu):
    # several tiles (cuts every 23552 entries), rows of 32 (the headline shape), of 31 and of 7 entries; 100k columns = 7 panels
    for per_row, m in ((32, 9000), (31, 7000), (7, 40_000)):
        p, j, x = synth.csr_fixed(m, 100_000, per_row, seed=per_row)
        check_all_kinds(p, j, x, 100_000, seed=per_row)


def test_tile_ragged_rows_empty_rows_unsorted_duplicates(gpu):
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 90, size=6000)
    lens[rng.random(6000) < 0.3] = 0                                   # many empty rows, some at both ends
    lens[:40] = 0
    lens[-25:] = 0
    p, j, x = ragged_csr(lens, 50_000, seed=6, sort=False, dup=True)
    check_all_kinds(p, j, x, 50_000, seed=7)
    # a stretch of more empty rows than a workgroup has threads, in the middle of a tile
    lens = np.concatenate([np.full(300, 20), np.zeros(5000, dtype=np.int64), np.full(3000, 25)])
    p, j, x = ragged_csr(lens, 20_000, seed=8)
    check_all_kinds(p, j, x, 20_000, seed=9)


def test_tile_long_rows_cross_passes(gpu):
    # rows longer than a pass (32768 entries), than a tile cut, than the wavefront-sum threshold; short rows in between
    lens = np.array([5, 70_000, 3, 0, 300, 257, 256, 40_000, 12, 33_000, 1, 0, 0, 2000] + [30] * 500)
    p, j, x = ragged_csr(lens, 120_000, seed=10, dup=True)
    check_all_kinds(p, j, x, 120_000, seed=11)


def test_tile_special_values(gpu):
    p, j, x = synth.csr_fixed(3000, 30_000, 24, seed=3)
    x = x.copy()
    x[5] = np.inf; x[700] = -np.inf; x[9000] = np.nan; x[100] = 0.0; x[101] = -0.0
    v = np.random.default_rng(4).normal(size=30_000)
    v[j[100]] = np.inf                                                 # 0 * Inf = NaN, as in the reference
    v[j[20_000]] = np.nan
    ref = O.matmul_csr_dvec_numeric(p, j, x, v)
    for algo in (FLAT, TILE):
        got = spmv_device(p, j, x, v, _lib.MX_F64, algo)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])


def test_tile_refuses_what_it_cannot_run(gpu):
    p, j, x = synth.csr_fixed(100, 500_000, 8, seed=1)                 # 500k columns: more than 24 panels
    v = np.zeros(500_000)
    with pytest.raises(_lib.MxError):
        spmv_device(p, j, x, v, _lib.MX_F64, TILE)
    got = spmv_device(p, j, x, v + 1.0, _lib.MX_F64, 0)               # AUTO falls back to the lane-group kernel
    np.testing.assert_allclose(got, np.add.reduceat(x, p[:-1]), rtol=1e-12)


# ----------------------------------------------------------------------------- planned SpMV (csrc/spmv_plan.hip)
def _check_planned(p, j, x, K, seed):
    from devmem import spmv_plan_device
    rng = np.random.default_rng(seed)
    v = rng.normal(size=K)
    vf = v.astype(np.float32)
    vi = rng.integers(-9, 9, size=K).astype(np.int32)
    vl = rng.integers(0, 2, size=K).astype(np.int32)
    for k in rng.integers(0, K, size=max(1, K // 50)):
        vi[k] = NA
        vl[k] = NA
    got = spmv_plan_device(p, j, x, [(v, _lib.MX_F64), (vf, _lib.MX_F32), (vi, _lib.MX_I32), (vl, _lib.MX_LGL)])
    ref = O.matmul_csr_dvec_numeric(p, j, x, v)
    np.testing.assert_allclose(got[0], ref, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(ref).max()))
    reff = O.matmul_csr_dvec_float32(p, j, x, vf)
    assert got[1].dtype == np.float32
    np.testing.assert_allclose(got[1], reff, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(reff).max()))
    for g, of, vv in ((got[2], O.matmul_csr_dvec_integer, vi), (got[3], O.matmul_csr_dvec_logical, vl)):
        r = of(p, j, x, vv)
        np.testing.assert_array_equal(np.isnan(g), np.isnan(r))
        ok = ~np.isnan(r)
        np.testing.assert_allclose(g[ok], r[ok], rtol=1e-12, atol=1e-12 * max(1.0, np.abs(r[ok]).max(initial=0)))
        if (~ok).any():
            assert (g[~ok].view(np.uint64) & 0xFFFFFFFF == 1954).all()       # NA_real_ where an NA element is touched


@pytest.mark.parametrize("m,K,dens", [(60, 25, 0.3), (1, 10, 0.9), (500, 700, 0.02), (5000, 6144, 0.002), (4097, 6145, 0.003),
                                      (9000, 40_000, 0.0008), (300, 13_000, 0.05)])
def test_planned_spmv_shapes_all_kinds(gpu, m, K, dens):
    p, j, x = rand_csr(m, K, dens, seed=m + K, sorted_cols=False, empty_rows=(0,) if m > 2 else ())
    _check_planned(p, j, x, K, seed=K)


def test_planned_spmv_headline_shape_ragged_and_duplicates(gpu):
    p, j, x = synth.csr_fixed(30_000, 100_000, 32, seed=4)                  # 17 panels, 8 row blocks
    _check_planned(p, j, x, 100_000, seed=1)
    rng = np.random.default_rng(8)
    lens = rng.integers(0, 70, size=10_000)
    lens[rng.random(10_000) < 0.3] = 0
    lens[17] = 30_000                                                       # one row with more entries than a block's share
    p, j, x = ragged_csr(lens, 50_000, seed=9, sort=False, dup=True)
    _check_planned(p, j, x, 50_000, seed=2)


def test_planned_spmv_wide_matrices_go_through_l2_super_panels(gpu):
    """More than 64 x 6,144 columns (round 5 refused them): super-panels of 2^18 columns, v read from global memory.  All four
    kinds with NA elements, columns on both sides of every super-panel boundary, the last (partial) super-panel, empty rows."""
    rng = np.random.default_rng(3)
    for m, K, per_row in ((100, 500_000, 8), (20_000, 1_000_003, 12), (9_000, 2_100_000, 40)):
        p, j, x = synth.csr_fixed(m, K, per_row, seed=m)
        j = j.copy()
        hit = rng.integers(0, j.size, size=200)                              # ids at the panel edges and the last column
        j[hit] = rng.choice([0, (1 << 18) - 1, 1 << 18, (1 << 18) + 1, K - 1, (K >> 18 << 18), max(0, (K >> 18 << 18) - 1)], size=200)
        _check_planned(p, j, x, K, seed=K)
    lens = rng.integers(0, 30, size=30_000)
    lens[rng.random(30_000) < 0.2] = 0
    lens[5] = 100_000
    p, j, x = ragged_csr(lens, 700_000, seed=2, sort=False, dup=True)
    _check_planned(p, j, x, 700_000, seed=7)


@pytest.mark.parametrize("K", [60_000, 600_000])
def test_planned_spmv_row_blocks_follow_the_entries(gpu, K):
    """Rows sorted by length (longest first), a few giant rows, long runs of empty rows: row blocks are cut at every 4,096th
    row AND at every E-th entry — every row is in exactly one block whatever the cuts (the result is complete), and the plan's
    padding stays small."""
    rng = np.random.default_rng(K)
    m = 70_000
    lens = np.sort(np.minimum(rng.lognormal(mean=2.5, sigma=1.2, size=m), 20_000).astype(np.int64))[::-1].copy()
    lens[0], lens[1] = 150_000 if K > 150_000 else K, 40_000
    lens[30_000:45_000] = 0
    p, j, x = ragged_csr(lens, K, seed=1, sort=False, dup=True)
    _check_planned(p, j, x, K, seed=5)
    lens = np.zeros(m, dtype=np.int64); lens[m - 1] = 5000                   # everything in the last row
    p, j, x = ragged_csr(lens, K, seed=1, sort=False, dup=True)
    _check_planned(p, j, x, K, seed=6)


def test_planned_spmv_limits(gpu):
    from devmem import spmv_plan_device
    p = (np.arange(11) * 4).astype(np.int32)                                  # more than 2^28 columns
    j = np.arange(40, dtype=np.int32) * 1000
    x = np.ones(40)
    lib = _lib.load()
    plan = C.c_void_p()
    from devmem import Dev
    dp, dj, dx = Dev(p.astype(np.int32)), Dev(j.astype(np.int32)), Dev(x.astype(np.float64))
    assert lib.mxd_spmv_plan_create(C.c_int(10), C.c_int(1 << 29), dp.ptr, dj.ptr, dx.ptr, None, C.byref(plan)) != 0
    assert b"2^28" in lib.mx_last_error()
    p0 = np.zeros(11, dtype=np.int32)                                        # a matrix without entries
    out = spmv_plan_device(p0, np.zeros(0, dtype=np.int32), np.zeros(0), [(np.ones(30), _lib.MX_F64)])
    np.testing.assert_array_equal(out[0], np.zeros(10))
