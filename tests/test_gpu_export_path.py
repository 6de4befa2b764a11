"""The export-level boundary at full size (what one .Call from R goes through): the transfer engine (csrc/xfer.hip), the
device-side CSR cache and the row-block pipeline of the SpMM exports (csrc/api.hip)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from matrixextra_amd import _lib, synth
from matrixextra_amd import exports as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _roundtrip(nbytes, seed):
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    d = C.c_void_p()
    _lib.check(lib.mx_dev_malloc(C.byref(d), C.c_size_t(nbytes)))
    try:
        _lib.check(lib.mx_upload(d, C.c_void_p(src.ctypes.data), C.c_size_t(nbytes)))
        dst = np.empty(nbytes, dtype=np.uint8)                       # fresh, untouched pages: the case of an R result
        _lib.check(lib.mx_download(C.c_void_p(dst.ctypes.data), d, C.c_size_t(nbytes)))
        assert np.array_equal(src, dst)
        dst2 = np.zeros(nbytes + 4096, dtype=np.uint8)[37:37 + nbytes]  # touched, odd alignment
        _lib.check(lib.mx_download(C.c_void_p(dst2.ctypes.data), d, C.c_size_t(nbytes)))
        assert np.array_equal(src, dst2)
    finally:
        lib.mx_dev_free(d)


def test_transfer_engine_moves_large_buffers_both_ways(gpu):
    # >= 64 MiB each way (above the 16 MiB threshold: register + direct DMA), odd sizes, and just below the threshold
    for nbytes, seed in ((96 << 20, 1), ((64 << 20) + 12345, 2), ((16 << 20) - 1, 3), (300 << 20, 4),
                         ((1 << 30) + 4099, 6)):                      # (from 192 MiB: touched, registered and copied piece by piece)
        _roundtrip(nbytes, seed)


def test_transfer_engine_staged_fallback(gpu):
    # MXGPU_XFER=2 forces the pipeline over pinned slots (what runs when the caller's memory cannot be registered)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_export_path import _roundtrip\n_roundtrip((80 << 20) + 777, 5); print('ok')") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MXGPU_XFER="2"), capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def _check_product(out, p, j, x, B, rows=(0, 333_333, -700)):
    """sampled row blocks bitwise-close to the oracle + the column checksum 1^T C = (A^T 1)^T B"""
    m, n = out.shape
    for r0 in rows:
        r0 = r0 % m
        r1 = min(m, r0 + 512)
        ref = np.zeros((r1 - r0) * n, dtype=B.dtype)
        pp = (p[r0:r1 + 1] - p[r0]).astype(np.int32)
        O.gemm_csr_drm_as_drm(r1 - r0, n, pp, j[p[r0]:p[r1]].copy(), x[p[r0]:p[r1]].copy(), B.reshape(-1), n, ref, n, 4, True)
        tol = 1e-12 if B.dtype == np.float64 else 1e-5
        np.testing.assert_allclose(out[r0:r1], ref.reshape(r1 - r0, n), rtol=tol, atol=tol * np.abs(ref).max())
    w = np.bincount(j, weights=x, minlength=B.shape[0])
    scale = np.abs(x).sum() * np.abs(B).max()
    assert np.max(np.abs(out.sum(axis=0, dtype=np.float64) - w @ B.astype(np.float64))) <= (1e-12 if B.dtype == np.float64 else 1e-8) * scale


def test_export_spmm_cfg2_pipeline_and_cache(gpu):
    lib = _lib.load()
    m, K, n = 1_000_000, 100_000, 128
    p, j, x = synth.csr_fixed(m, K, 32)
    B = synth.dense_normal(K, n)
    Y = np.asfortranarray(B.T)
    lib.mx_cache_invalidate(None)
    st0 = _cache_stats(lib)
    out1 = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)             # cold: upload pipelined with compute and download
    _check_product(out1, p, j, x, B)
    st1 = _cache_stats(lib)
    assert st1["entries"] == st0["entries"] + 1 and st1["bytes"] >= p.nbytes + j.nbytes + x.nbytes
    out2 = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)             # same host vectors again: CSR served from the device
    st2 = _cache_stats(lib)
    assert st2["hits"] == st1["hits"] + 1
    np.testing.assert_array_equal(out1, out2)
    # the same vectors rewritten in place: the fingerprint no longer matches, the stale copy must not be used
    x *= 2.0
    out3 = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
    np.testing.assert_array_equal(out3, 2.0 * out1)
    # ONE element overwritten in place, far from either end (what `X@x[i] <- v` does in R when nothing else refers to the
    # vector): the fingerprint covers every byte, so this is a miss too and the row changes
    k = int(p[m // 3]) + 5
    row, old = m // 3, x[k]
    x[k] = old + 1.0
    out3b = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
    np.testing.assert_allclose(out3b[row] - out3[row], B[j[k]], rtol=1e-9, atol=1e-12)
    assert np.array_equal(np.delete(out3b, row, axis=0)[::997], np.delete(out3, row, axis=0)[::997])
    x[k] = old
    # row-major result (dense %*% CSC): C^T = A B^T with the CSC read as CSR of the transpose
    X = np.asfortranarray(B.T[:64])                                  # 64 x 100k dense, Y = CSC 100k x 1M  ->  64 x 1M
    out4 = G.matmul_dense_csc_numeric(X, p, j, x, 1)
    ref_rows = out3[:, :64]                                          # same product, transposed layout
    np.testing.assert_allclose(out4.T, ref_rows, rtol=1e-12, atol=1e-12 * np.abs(ref_rows).max())
    # explicit invalidation and switching the cache off
    lib.mx_cache_invalidate(C.c_void_p(x.ctypes.data))
    assert _cache_stats(lib)["entries"] == 0
    lib.mx_cache_configure(C.c_int64(0))
    out5 = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
    assert _cache_stats(lib)["entries"] == 0
    np.testing.assert_array_equal(out5, out3)
    lib.mx_cache_configure(C.c_int64(8192 << 20))


def test_export_into_plain_malloc_memory_like_r(gpu):
    """What R hands over: vectors and a result from plain malloc — no huge-page advice, so 4-KiB pages on a THP=madvise
    machine.  First-touching and registering 640 MB of those for the download took ~250 ms here (1 GB: 370-600 ms,
    tools/cold_export_probe.py malloc) until the library advised MADV_HUGEPAGE on the destination before touching it.
    Same bytes as the numpy call; the time bound only has to separate ~20 ms from ~250 ms."""
    import time
    lib = _lib.load()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    m, K, n = 640_000, 50_000, 128
    p, j, x = synth.csr_fixed(m, K, 24)
    B = synth.dense_normal(K, n)
    Y = np.asfortranarray(B.T)
    ref = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)              # numpy result (and the warm-up of pool / scratch)

    def from_malloc(a):
        q = libc.malloc(a.nbytes)
        C.memmove(q, a.ctypes.data, a.nbytes)
        return q
    bufs = [from_malloc(a) for a in (p, j, x, Y)]
    fn = lib.mx_tcrossprod_csr_dense_numeric
    best = None
    try:
        for _ in range(3):
            lib.mx_cache_invalidate(None)
            out = libc.malloc(8 * m * n)                             # untouched: the pages do not exist yet
            try:
                t0 = time.perf_counter()
                _lib.check(fn(C.c_void_p(bufs[0]), C.c_void_p(bufs[1]), C.c_void_p(bufs[2]), C.c_int(m), C.c_void_p(bufs[3]),
                              C.c_int(n), C.c_int(K), C.c_int(1), C.c_void_p(out)))
                t = time.perf_counter() - t0
                got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_double)), shape=(n, m)).T
                assert np.array_equal(got, ref)
            finally:
                libc.free(out)
            best = t if best is None else min(best, t)
    finally:
        for q in bufs:
            libc.free(q)
        lib.mx_cache_invalidate(None)
    try:
        thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read()
    except OSError:
        thp = ""
    if "[madvise]" in thp or "[always]" in thp:       # with THP switched off there is nothing the library could ask for
        assert best < 0.15, f"export into malloc'ed memory took {best * 1e3:.0f} ms"


def test_export_spmm_float32_and_blocks_without_entries(gpu):
    # f32 dense operand through the pipeline; the first half of the rows has no entries at all (a block that is only
    # zero-filled) and the last rows are empty too
    m, K, n = 600_000, 50_000, 128
    lens = np.zeros(m, dtype=np.int64)
    lens[320_000:590_000] = 24
    p = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    p = p.astype(np.int32)
    rng = np.random.default_rng(3)
    j = rng.integers(0, K, size=int(p[-1]), dtype=np.int32)
    x = rng.normal(size=int(p[-1]))
    B = synth.dense_normal(K, n, dtype=np.float32)
    out = G.tcrossprod_csr_dense_float32(p, j, x, np.asfortranarray(B.T), 1)
    assert out.dtype == np.float32 and out.shape == (m, n)
    assert not out[:320_000].any() and not out[590_000:].any()
    _check_product(out, p, j, x, B, rows=(319_900, 450_000, 589_800))


def _cache_stats(lib):
    b, e, h, mi = C.c_int64(0), C.c_int(0), C.c_int64(0), C.c_int64(0)
    lib.mx_cache_stats(C.byref(b), C.byref(e), C.byref(h), C.byref(mi))
    return dict(bytes=b.value, entries=e.value, hits=h.value, misses=mi.value)


def test_sharded_export_on_one_gpu(gpu):
    """mx_set_devices with the same GPU listed three times: the sharded path (row ranges balanced by cost, one host
    thread + three queues per shard, pitched downloads into the caller's column-major result) runs on a one-GPU box
    and must give the single-device answer bit for bit (each row is summed by the same kernel either way)."""
    lib = _lib.load()
    m, K, n = 700_000, 60_000, 128
    p, j, x = synth.csr_skewed(m, K, 12, seed=5, sigma=1.0)
    B = synth.dense_normal(K, n)
    Y = np.asfortranarray(B.T)
    single = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
    devs = (C.c_int * 3)(0, 0, 0)
    _lib.check(lib.mx_set_devices(devs, 3))
    try:
        sharded = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
        np.testing.assert_allclose(sharded, single, rtol=1e-12, atol=1e-12 * np.abs(single).max())
        # B (61 MB) is below the whole-buffer registration size and read by all three workers: they once registered and
        # unregistered it each on their own and one call in five aborted inside the HIP runtime ("Memobj map does not
        # have ptr"); now the coordinating thread registers it once.  A few more calls to give a race its chance.
        for _ in range(4):
            again = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
            assert np.array_equal(again, sharded)
        _check_product(sharded, p, j, x, B, rows=(0, 233_000, 466_000, -600))
        Xd = np.asfortranarray(B.T[:96])                              # row-major result: dense (96 x K) %*% CSC (K x m)
        rm = G.matmul_dense_csc_numeric(Xd, p, j, x, 1)
        np.testing.assert_allclose(rm.T, single[:, :96], rtol=1e-12, atol=1e-12 * np.abs(single).max())
    finally:
        _lib.check(lib.mx_set_devices(None, 0))
    with pytest.raises(_lib.MxError):
        _lib.check(lib.mx_set_devices((C.c_int * 1)(99), 1))


def test_sharded_export_bits_do_not_depend_on_the_device_list(gpu):
    """The row-split family's geometry (segments per row, column panels, piece length of the long-rows path) is chosen for the
    WHOLE product and carried by value into every shard, on whatever worker thread runs it (it used to sit in
    thread-local storage of the calling thread, so the shards re-derived it from their own sizes and the long-rows path was
    off on every shard).  Column-major f32 — where segments, panels and narrow lane groups all reassociate — on skewed rows
    with a few giant ones: the same bits for the GPU listed 2, 3 and 5 times."""
    lib = _lib.load()
    rng = np.random.default_rng(3)
    # (fewer than 32,768 rows: AUTO weighs the row-split kernel against the tile kernel only, and at density 7e-4 the tile
    # kernel has nothing to re-use; n = 528 f32 makes the result 64.4 MiB: the exports shard from 64 MiB on)
    m, K, n = 32_000, 120_000, 528
    lens = np.minimum(np.maximum(rng.lognormal(mean=np.log(80) - 0.72, sigma=1.2, size=m), 1), 4000).astype(np.int64)
    lens[rng.choice(m, size=6, replace=False)] = 30_000
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = np.unique(row * K + rng.integers(0, K, size=row.size))    # sorted, distinct ids per row (a few draws collapse)
    row = key // K
    j = (key - row * K).astype(np.int32)
    p = np.zeros(m + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=m), out=p[1:])
    p = p.astype(np.int32)
    lens = np.diff(p)
    x = rng.uniform(-1, 1, size=j.size)
    B = synth.dense_normal(K, n, dtype=np.float32)
    Y = np.asfortranarray(B.T)
    outs = {}
    fam = [C.c_int(0) for _ in range(4)]
    try:
        for nd in (2, 3, 5):
            _lib.check(lib.mx_set_devices((C.c_int * nd)(*([0] * nd)), nd))
            outs[nd] = G.tcrossprod_csr_dense_float32(p, j, x, Y, 1)
            _lib.check(lib.mx_debug_last_export_family(*[C.byref(f) for f in fam]))
            # the family under test: row-split, with a long-rows piece chosen for the whole product
            assert fam[0].value == 4 and fam[3].value > 0, [f.value for f in fam]
    finally:
        _lib.check(lib.mx_set_devices(None, 0))
    assert np.array_equal(outs[2], outs[3]) and np.array_equal(outs[3], outs[5])
    rows = np.concatenate([np.argsort(lens)[-4:], rng.integers(0, m, size=40)])
    for r in rows:
        s, e = p[r], p[r + 1]
        want = x[s:e] @ B[j[s:e]].astype(np.float64)
        np.testing.assert_allclose(outs[3][r], want, rtol=2e-5, atol=2e-5 * np.abs(want).max())


def test_export_of_a_dense_ish_matrix_with_uneven_rows_deals_its_rows(gpu):
    """Round 6: the tile kernel's rows dealt by length reach the exports — a result below 64 MiB is ONE device-level product with
    AUTO inside and no matrix profile: the cv comes from the caller's row pointers (csrc/api_core.inc host_row_cv); the
    pipelined / sharded exports carry the cv of the whole product's matrix in the family chosen for it —, and the exports never
    cut a long row into parts: bit for bit the reference's storage-order chain (src/matmul.cpp:150-185), as before."""
    lib = _lib.load()
    rng = np.random.default_rng(11)
    m, K, n = 10_000, 10_000, 100
    lens = np.minimum(np.floor(rng.lognormal(np.log(500) - 0.72, 1.2, size=m)), K).astype(np.int64)
    lens[17] = K                                                     # one full row: several windows per tile
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = np.unique(row * K + rng.integers(0, K, size=row.size))
    row = key // K
    j = (key - row * K).astype(np.int32)
    p = np.zeros(m + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=m), out=p[1:])
    p = p.astype(np.int32)
    x = rng.uniform(-1, 1, size=j.size)
    B = synth.dense_normal(K, n, dtype=np.float64)
    Y = np.asfortranarray(B.T)
    got = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
    assert lib.mxd_spmm_last_kernel().decode() == "spmm_tile_kernel"
    assert lib.mxd_debug_spmm_tile_mode() == 1                       # dealt by length, no row cut into parts
    ref = O.tcrossprod_csr_dense(p, j, x, Y, 1, True)
    assert np.array_equal(got, ref)


def _option(lib, name):
    v = C.c_int64(0)
    _lib.check(lib.mx_get_option(name.encode(), C.byref(v)))
    return v.value


def test_export_spmv_is_bitwise_by_default_and_planned_on_request(gpu):
    """matmul_csr_dvec_* by default runs the one-shot flat kernel on EVERY call — the same bits on the first and on later
    calls, bit for bit the reference's loop (the planned kernel used to take over silently on a cache hit).
    With mx_set_option("spmv_planned", 1) a matrix that is still on the device from the previous call goes through the
    planned kernel (v's panels in LDS): same answers to 1e-12, all four kinds, NA rows exact."""
    lib = _lib.load()
    lib.mx_cache_invalidate(None)
    m, K = 300_000, 50_000
    p, j, x = synth.csr_fixed(m, K, 20, seed=6)                      # 6 M entries: above the planning threshold
    rng = np.random.default_rng(2)
    v = rng.normal(size=K)
    ref = O.matmul_csr_dvec_numeric(p, j, x, v)
    assert _option(lib, "spmv_planned") == 0
    n0 = _option(lib, "spmv_planned_calls")
    first = G.matmul_csr_dvec_numeric(p, j, x, v)                    # miss: one-shot flat kernel, bitwise the oracle's loop
    again = G.matmul_csr_dvec_numeric(p, j, x, v)                    # hit: still the flat kernel
    np.testing.assert_array_equal(first, ref)
    np.testing.assert_array_equal(again, ref)
    assert _option(lib, "spmv_planned_calls") == n0 and _cache_stats(lib)["hits"] >= 1
    _lib.check(lib.mx_set_option(b"spmv_planned", C.c_int64(1)))
    try:
        planned = G.matmul_csr_dvec_numeric(p, j, x, v)              # hit: plan built, planned kernel
        assert _option(lib, "spmv_planned_calls") == n0 + 1
        np.testing.assert_allclose(planned, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        v2 = rng.normal(size=K)
        np.testing.assert_allclose(G.matmul_csr_dvec_numeric(p, j, x, v2), O.matmul_csr_dvec_numeric(p, j, x, v2), rtol=1e-12,
                                   atol=1e-12)
        vi = rng.integers(-5, 6, size=K).astype(np.int32)
        vi[::97] = -2147483648
        gi, ri = G.matmul_csr_dvec_integer(p, j, x, vi), O.matmul_csr_dvec_integer(p, j, x, vi)
        # NA rows carry R's NA_real_ payload exactly (low word 1954), not just "a NaN"
        np.testing.assert_array_equal(np.isnan(gi), np.isnan(ri))
        na_bits = np.uint64(0x7FF00000000007A2)
        assert np.all(gi[np.isnan(ri)].view(np.uint64) == na_bits) and np.isnan(ri).sum() > 1000
        np.testing.assert_allclose(gi[~np.isnan(ri)], ri[~np.isnan(ri)], rtol=1e-12, atol=1e-12)
        # a NaN that comes out of the arithmetic (NaN value x finite integer) stays an ordinary NaN, it is not turned into NA
        x2 = x.copy()
        k = int(p[1234]) + 3
        x2[k] = np.nan
        vi2 = np.abs(vi) % 7 + 1
        G.matmul_csr_dvec_integer(p, j, x2, vi2.astype(np.int32))            # miss (new values): flat
        gn = G.matmul_csr_dvec_integer(p, j, x2, vi2.astype(np.int32))       # hit: planned
        assert np.isnan(gn[1234]) and gn[1234:1235].view(np.uint64)[0] != na_bits and np.isnan(gn).sum() == 1
        vf = v.astype(np.float32)
        np.testing.assert_allclose(G.matmul_csr_dvec_float32(p, j, x, vf), O.matmul_csr_dvec_float32(p, j, x, vf), rtol=1e-5, atol=1e-5)
        assert _option(lib, "spmv_planned_calls") >= n0 + 5
    finally:
        _lib.check(lib.mx_set_option(b"spmv_planned", C.c_int64(-1)))
    with pytest.raises(_lib.MxError):
        _lib.check(lib.mx_set_option(b"no_such_option", C.c_int64(1)))


def test_device_blocks_come_back_from_the_pool(gpu):
    """Export calls hand their device blocks (operands, results, the kept plan) to csrc/pool.hip instead of hipFree: the
    second call of a kind is served from the pool, results are the same bits either way (a reused block is not zeroed:
    nothing may depend on fresh memory), and mxd_release_workspaces() gives everything back to the device."""
    lib = _lib.load()
    lib.mx_cache_invalidate(None)
    _lib.check(lib.mxd_release_workspaces())
    assert _option(lib, "pool_idle_bytes") == 0 and _option(lib, "pool_idle_blocks") == 0
    p1, j1, x1 = synth.csr_fixed(60_000, 3_000, 10, seed=21)
    p2, j2, x2 = synth.csr_overlapping(p1, j1, 3_000, 10, share=0.4, seed=22)
    ref = O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)
    h0 = _option(lib, "pool_hits")
    first = G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)
    assert _option(lib, "pool_idle_blocks") > 0                     # the result's blocks (the operands sit in the CSR cache)
    h1 = _option(lib, "pool_hits")
    again = G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)
    assert _option(lib, "pool_hits") > h1 >= h0
    for k in ("indptr", "indices", "values"):
        np.testing.assert_array_equal(first[k], ref[k])
        np.testing.assert_array_equal(again[k], ref[k])
    # dirty blocks under a row gather with empty rows and under a product with rows without entries
    pe = p1.copy()
    pe[30_000:] = pe[30_000]                                        # rows 30000.. are empty
    rows = np.array([5, 5, 0, 59_999, 30_000, 17], dtype=np.int32)
    B = np.asfortranarray(np.random.default_rng(3).normal(size=(16, 3_000)))
    for _ in range(2):
        g, r = G.copy_csr_rows_numeric(pe, j1, x1, rows), O.copy_csr_rows_numeric(pe, j1, x1, rows)
        for k in ("indptr", "indices", "values"):
            np.testing.assert_array_equal(g[k], r[k])
        out = G.tcrossprod_csr_dense_numeric(pe, j1, x1, B)
        np.testing.assert_allclose(out, O.tcrossprod_csr_dense_numeric(pe, j1, x1, B), rtol=1e-12, atol=1e-12)
        assert not out[30_000:].any()
    lib.mx_cache_invalidate(None)
    assert _option(lib, "pool_idle_bytes") > 0
    _lib.check(lib.mxd_release_workspaces())
    assert _option(lib, "pool_idle_bytes") == 0


@pytest.mark.parametrize("f32", [False, True])
def test_cold_export_comes_down_in_tiles(gpu, f32):
    """A cold product into a column-major result comes down as tiles (row block x column group), each group's pages
    touched and registered on their own; a group's last partial page belongs to the next group's registration and is
    written by a copy of its own.  Result at every byte offset the page cuts can fall on (the result starts 8 .. 4088
    bytes into a page), rows not a multiple of the block granularity, both element sizes: every column equal to the cached
    call's (column blocks of the kept plan, contiguous downloads) and to the oracle on a row sample; the bytes before and
    after the result untouched."""
    lib = _lib.load()
    m, K, n = 301_003, 40_000, 160 if f32 else 80                     # ~193 MB of result either way: 3 column groups
    p, j, x = synth.csr_fixed(m, K, 96, seed=31)                      # 29 M entries: the upload outlasts the page work
    dt = np.float32 if f32 else np.float64
    B = synth.dense_normal(K, n).astype(dt)
    Y = np.asfortranarray(B.T)
    fn = lib.mx_tcrossprod_csr_dense_float32 if f32 else lib.mx_tcrossprod_csr_dense_numeric
    item = np.dtype(dt).itemsize
    guard = 8192
    raw = np.empty(m * n * item + 3 * guard, dtype=np.uint8)
    page0 = (-raw.ctypes.data) % 4096
    rows = np.r_[0:300, m // 2:m // 2 + 300, m - 300:m]
    ref = None
    for off in (8, 2048 + 24, 4088):
        raw[:] = 0x5A
        start = page0 + off
        out = raw[start:start + m * n * item].view(dt).reshape(n, m).T      # column-major m x n, `off` bytes into a page
        lib.mx_cache_invalidate(None)
        _lib.check(fn(_lib.ptr(p), _lib.ptr(j), _lib.ptr(x), C.c_int(m), _lib.ptr(Y), C.c_int(n), C.c_int(K), C.c_int(1),
                      C.c_void_p(out.ctypes.data)))
        buf = C.create_string_buffer(512)
        lib.mx_last_call_phases(buf, C.c_size_t(512))
        assert b"tiles=" in buf.value, buf.value                      # the tiled form ran
        assert (raw[:start] == 0x5A).all() and (raw[start + m * n * item:] == 0x5A).all()
        if ref is None:
            ref = out.copy()
            dense = np.zeros((rows.size, n))
            for k, r in enumerate(rows):
                dense[k] = x[p[r]:p[r + 1]] @ B[j[p[r]:p[r + 1]]].astype(np.float64)
            np.testing.assert_allclose(out[rows], dense, rtol=2e-5 if f32 else 1e-12, atol=2e-4 if f32 else 1e-11)
        else:
            assert np.array_equal(out, ref)
    cached = np.empty((m, n), dtype=dt, order="F")
    _lib.check(fn(_lib.ptr(p), _lib.ptr(j), _lib.ptr(x), C.c_int(m), _lib.ptr(Y), C.c_int(n), C.c_int(K), C.c_int(1),
                  C.c_void_p(cached.ctypes.data)))
    lib.mx_last_call_phases(buf, C.c_size_t(512))
    assert b"csr=cached" in buf.value, buf.value
    if not np.array_equal(cached, ref):                               # (the last, partial octet of 64 rows may be laid out differently)
        np.testing.assert_allclose(cached, ref, rtol=1e-5 if f32 else 1e-13, atol=1e-4 if f32 else 1e-11)
        assert np.array_equal(cached[: m - m % 1024], ref[: m - m % 1024])
    lib.mx_cache_invalidate(None)


def test_pool_switched_off_and_poisoned(gpu):
    """MXGPU_POOL_MB=0: every block goes straight back to the device (the behaviour before the pool);
    MXGPU_POOL_POISON=1: every block handed out is filled with 0xA5 first.  Both read once per process: a child."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import ctypes as C, numpy as np\n"
        "from matrixextra_amd import _lib, synth, exports as G\n"
        "from oracle import oracle as O\n"
        "lib = _lib.load()\n"
        "p1, j1, x1 = synth.csr_fixed(50_000, 4_000, 12, seed=3)\n"
        "p2, j2, x2 = synth.csr_overlapping(p1, j1, 4_000, 12, share=0.5, seed=4)\n"
        "for _ in range(3):\n"
        "    g = G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, True); r = O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, True)\n"
        "    assert all(np.array_equal(g[k], r[k]) for k in ('indptr', 'indices', 'values'))\n"
        "    rows = np.array([7, 7, 0, 49_999], dtype=np.int32)\n"
        "    g = G.copy_csr_rows_numeric(p1, j1, x1, rows); r = O.copy_csr_rows_numeric(p1, j1, x1, rows)\n"
        "    assert all(np.array_equal(g[k], r[k]) for k in ('indptr', 'indices', 'values'))\n"
        "    v = np.random.default_rng(1).normal(size=4_000)\n"
        "    assert np.allclose(G.matmul_csr_dvec_numeric(p1, j1, x1, v), O.matmul_csr_dvec_numeric(p1, j1, x1, v), rtol=1e-12, atol=1e-12)\n"
        "n = C.c_int64(-1); _lib.check(lib.mx_get_option(b'pool_idle_bytes', C.byref(n)))\n"
        "print('idle', n.value)\n") % ROOT
    for env, want_zero in ((dict(MXGPU_POOL_MB="0"), True), (dict(MXGPU_POOL_POISON="1"), False)):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        idle = int(r.stdout.strip().split()[-1])
        assert (idle == 0) if want_zero else (idle > 0), (env, idle)
