#!/usr/bin/env python3
"""Randomised stress of the structure-producing exports (merges, row gather, column slices, sort, SpMV, CSR op vector)
against the CPU oracle, bit for bit (run on the GPU box):  python tools/fuzz_structure.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from conftest import rand_csr
from devmem import gather_fused_device, merge_fused_device, rows_sorted_device, spmv_device, spmv_plan_device
from matrixextra_amd import _lib
from matrixextra_amd import exports as G
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
NA = int(O.NA_INTEGER)


import os
TRACE = os.environ.get("FUZZ_TRACE")


def trace(*a):
    if TRACE:
        print(*a, flush=True)


def same(a, b, what):
    trace("  checked", what)
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        assert np.array_equal(np.isnan(a), np.isnan(b)), what
        ok = ~np.isnan(a)
        assert np.array_equal(a[ok].view(np.int64), b[ok].view(np.int64)), what
    else:
        assert np.array_equal(a, b), what


def same_list(g, o, what):
    for k in ("indptr", "indices", "values"):
        if k in o:
            same(g[k], o[k], f"{what}/{k}")


def footprint():
    """(live device blocks of the library's pool, live bytes, idle bytes, device bytes in use on the GPU, host RSS): what a
    leak in the exports' begin / finish / error paths would move"""
    import ctypes, gc, psutil, torch
    lib = _lib.load()
    gc.collect()
    ctypes.CDLL(None).malloc_trim(0)                   # freed heap back to the system: RSS then counts what is HELD
    assert lib.mx_cache_invalidate(None) == 0          # the CSR cache holds device copies by design (cap: MXGPU_CSR_CACHE_MB)
    out = []
    for name in (b"pool_live_blocks", b"pool_live_bytes", b"pool_idle_bytes"):
        v = ctypes.c_int64(0)
        assert lib.mx_get_option(name, ctypes.byref(v)) == 0
        out.append(int(v.value))
    free, total = torch.cuda.mem_get_info()
    return out + [int(total - free), psutil.Process().memory_info().rss]


BASE_AT = 300
TRACK = bool(os.environ.get("FUZZ_TRACK"))
base = None
SKIP = set(filter(None, os.environ.get("FUZZ_SKIP", "").split(",")))     # gatherfused, sortedview, spmvplan, naroute (bisecting)
t_end = time.time() + budget
cases = 0

def one_case(cases):
    """one random case; its operands and results die with the frame, so the footprint below sees the library and not them"""
    m = int(rng.choice([1, 3, 64, 65, 200, 1000, 4000]))
    K = int(rng.choice([1, 2, 9, 70, 400, 3000]))
    d1, d2 = float(rng.choice([0.0, 0.02, 0.2, 0.7])), float(rng.choice([0.0, 0.05, 0.3, 0.9]))
    s1, s2 = int(rng.integers(1 << 30)), int(rng.integers(1 << 30))
    er = tuple(int(r) for r in rng.integers(0, m, size=min(3, m)))
    p1, j1, x1 = rand_csr(m, K, d1, seed=s1, empty_rows=er)
    p2, j2, x2 = rand_csr(m, K, d2, seed=s2)
    if rng.random() < 0.3:                                          # overlapping pattern with cancellation
        p2, j2, x2 = p1.copy(), j1.copy(), -x1.copy()
    what = None
    trace("case", cases, dict(m=m, K=K, d1=d1, d2=d2, s1=s1, s2=s2))
    try:
        for sub in (False, True):
            what = f"add sub={sub}"; trace("  start", what)
            same_list(G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), what)
        if "dropzeros" not in SKIP:                                  # what remove_zeros runs after the subtraction above
            d = O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, True)
            dv = d["values"].copy()
            if dv.size and rng.random() < 0.5:
                dv[rng.random(dv.size) < 0.2] = np.nan
            for rm in (False, True):
                what = f"remove zeros na.rm={rm}"; trace("  start", what)
                same_list(G.remove_zero_valued_csr_numeric(d["indptr"], d["indices"], dv, rm),
                          O.remove_zero_valued_csr_numeric(d["indptr"], d["indices"], dv, rm), what)
            what = "check valid"; trace("  start", what)
            assert G.check_valid_csr_matrix(p1, j1, m, K) == O.check_valid_csr_matrix(p1, j1, m, K) == {}, what
        what = "mul"; trace("  start", what)
        same_list(G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), what)
        l1 = rand_csr(m, K, d1, seed=s1, dtype="l")
        l2 = rand_csr(m, K, d2, seed=s2, dtype="l")
        for xor in (False, True):
            what = f"or xor={xor}"; trace("  start", what)
            same_list(G.logicalor_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2], xor),
                      O.logicalor_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2], xor), what)
        what = "and"; trace("  start", what)
        same_list(G.logicaland_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2]),
                  O.logicaland_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2]), what)
        rows = rng.integers(0, m, size=int(rng.integers(0, 2 * m + 1)), dtype=np.int32)
        what = "gather"; trace("  start", what)
        same_list(G.copy_csr_rows_numeric(p1, j1, x1, rows), O.copy_csr_rows_numeric(p1, j1, x1, rows), what)
        if rows.size:
            c0 = int(rng.integers(0, K)); c1 = int(rng.integers(c0, K))
            cols_seq = np.arange(c0 + 1, c1 + 2, dtype=np.int32)    # 1-based, as R passes them
            what = "col_seq"; trace("  start", what)
            same_list(G.copy_csr_rows_col_seq_numeric(p1, j1, x1, rows, cols_seq, True),
                      O.copy_csr_rows_col_seq_numeric(p1, j1, x1, rows, cols_seq, True), what)
            cols = rng.integers(0, K, size=int(rng.integers(1, K + 3)), dtype=np.int32)
            what = "arbitrary"; trace("  start", what)
            same_list(G.copy_csr_arbitrary_numeric(p1, j1, x1, rows, cols), O.copy_csr_arbitrary_numeric(p1, j1, x1, rows, cols), what)
        if p1[-1] + p2[-1] > 0:                                       # the one-pass merge kernel (device level)
            for op, ref in ((_lib.MX_OP_ADD, O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)),
                            (_lib.MX_OP_MUL, O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2)),
                            (_lib.MX_OP_XOR, O.logicalor_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2], True))):
                what = f"fused merge op={op}"; trace("  start", what)
                lg = op == _lib.MX_OP_XOR
                gp, gj, gx = merge_fused_device(op, l1[0] if lg else p1, l1[1] if lg else j1, l1[2] if lg else x1,
                                                l2[0] if lg else p2, l2[1] if lg else j2, l2[2] if lg else x2)
                same_list(dict(indptr=gp, indices=gj, values=gx), ref, what)
        if rows.size and "gatherfused" not in SKIP:                   # the one-launch gather (device level): any capacity
            total = int((p1[1:] - p1[:-1])[rows].sum())
            cap = int(rng.choice([total, total + 5, max(total // 2, 0), 0, 3 * total + 1]))
            kind = int(rng.integers(3))
            vals, dt = ((x1, _lib.MX_F64), (l1[2] if l1[1].size == j1.size else None, _lib.MX_LGL), (None, _lib.MX_NONE))[kind]
            if vals is None:
                dt = _lib.MX_NONE
            what = f"gather fused cap={cap} total={total} kind={kind}"; trace("  start", what)
            ref = O.copy_csr_rows_numeric(p1, j1, x1, rows)
            gp, gj, gx, nnz = gather_fused_device(p1, j1, vals, rows, cap, dt)
            assert nnz == total, what
            refp = np.concatenate([[0], np.cumsum((p1[1:] - p1[:-1])[rows])]).astype(np.int32)
            same(gp, refp, what + "/indptr")
            if cap >= total and total:
                same(gj, ref["indices"], what + "/indices")
                if dt == _lib.MX_F64:
                    same(gx, ref["values"], what + "/values")
            elif total:                                               # everything that ends within the capacity is there
                for t in np.nonzero(refp[1:] <= cap)[0][:50]:
                    rr = rows[t]
                    same(gj[refp[t]:refp[t + 1]], j1[p1[rr]:p1[rr + 1]], what + "/row")
        what = "sorted check"; trace("  start", what)
        assert G.check_indices_are_sorted(p1, j1) == O.check_indices_are_sorted(p1, j1)
        if m > 2 and "sortedview" not in SKIP:                        # device level: unaligned indices, row-block views (indptr[0] > 0)
            r0 = int(rng.integers(0, m - 1))
            mis = int(rng.integers(0, 4))
            what = f"sorted check view r0={r0} misalign={mis}"; trace("  start", what)
            pu0, ju0, _ = rand_csr(m, K, d2, seed=s2 + 2, sorted_cols=bool(rng.integers(2)))
            lo_, hi_ = int(pu0[r0]), int(pu0[-1])
            ref_sorted = all(np.all(np.diff(ju0[pu0[r]:pu0[r + 1]]) >= 0) for r in range(r0, m)) if hi_ > lo_ else True
            assert rows_sorted_device(pu0[r0:], ju0, misalign=mis) == ref_sorted, what
        what = "sort"; trace("  start", what)
        pu, ju, xu = rand_csr(m, K, d2, seed=s2 + 1, sorted_cols=False)
        assert G.check_indices_are_sorted(pu, ju) == O.check_indices_are_sorted(pu, ju)
        jg, xg = ju.copy(), xu.copy()
        G.sort_sparse_indices_inplace(pu, jg, xg)
        jo, xo = O.sort_sparse_indices(pu, ju, xu)
        same(jg, jo, "sort/j"); same(xg, xo, "sort/x")
        what = "spmv"; trace("  start", what)
        v = rng.normal(size=K).round(3)
        np.testing.assert_allclose(G.matmul_csr_dvec_numeric(pu, ju, xu, v), O.matmul_csr_dvec_numeric(pu, ju, xu, v), rtol=1e-11, atol=1e-12)
        if pu[-1] >= 4:                                               # flat / tile kernels: storage-order sums, bitwise for rows <= 256
            ref = O.matmul_csr_dvec_numeric(pu, ju, xu, v)
            short = np.diff(pu) <= 256
            for algo in (3, 2):
                what = f"spmv algo {algo}"; trace("  start", what)
                got = spmv_device(pu, ju, xu, v, _lib.MX_F64, algo)
                same(got[short], ref[short], what)
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
        if pu[-1] >= 1 and K <= 64 * 6144 and "spmvplan" not in SKIP:  # planned SpMV, all kinds incl. NA elements
            vi = rng.integers(-3, 4, size=K).astype(np.int32)
            vi[rng.random(K) < 0.1] = NA
            what = "spmv plan"; trace("  start", what)
            outs = spmv_plan_device(pu, ju, xu, [(v, _lib.MX_F64), (vi, _lib.MX_I32), (vi, _lib.MX_LGL), (v.astype(np.float32), _lib.MX_F32)])
            refs = [O.matmul_csr_dvec_numeric(pu, ju, xu, v), O.matmul_csr_dvec_integer(pu, ju, xu, vi),
                    O.matmul_csr_dvec_logical(pu, ju, xu, vi), O.matmul_csr_dvec_float32(pu, ju, xu, v.astype(np.float32))]
            # error scale of a row: sum |a| |v| (the float32 kind: the reference rounds after every term, the plan once)
            row_scale = O.matmul_csr_dvec_numeric(pu, ju, np.abs(xu), np.abs(v))
            for got, ref, tol in zip(outs, refs, (1e-12, 1e-12, 1e-12, 1e-5)):
                # R's NA_real_ = a NaN whose low word is 1954 (arithmetic may set the quiet bit: ISNA looks at the low word only)
                na_g = np.isnan(got) & ((got.view(np.uint64) & np.uint64(0xFFFFFFFF)) == 1954) if got.dtype == np.float64 else np.isnan(got)
                na_r = np.isnan(ref) & ((ref.view(np.uint64) & np.uint64(0xFFFFFFFF)) == 1954) if ref.dtype == np.float64 else np.isnan(ref)
                assert np.array_equal(na_g, na_r), what + " NA rows"
                ok = ~np.isnan(ref)
                lim = tol * np.maximum(row_scale if tol > 1e-8 else np.abs(ref), 1.0) * (4.0 if tol > 1e-8 else 1.0)
                assert np.all(np.abs(got[ok].astype(np.float64) - ref[ok].astype(np.float64)) <= lim[ok] + tol), what + f" tol {tol}"
        what = "dvec mul"; trace("  start", what)
        ln = int(rng.choice([1, m, m * K, max(1, m // 2), 7]))
        dv = rng.uniform(0.5, 2.0, size=ln).round(3)
        same(G.multiply_csr_by_dvec_no_NAs_numeric(p1, j1, x1, dv, K, 1, 0, 0, 0, 0, 1),
             O.multiply_csr_by_dvec_no_NAs_numeric(p1, j1, x1, dv, K, 1, 0, 0, 0, 0, 1), what)
        # the structure-changing NA route of CSR (op) vector: any length, any operation, specials sprinkled in
        what = "dvec NA route"; trace("  start", what)
        opn = int(rng.integers(5))
        flags = [0, 0, 0, 0, 0]; flags[opn] = 1                      # multiply, powerto, divide, divrest, intdiv
        ln = int(rng.choice([m, max(1, m // 2) if m % 2 == 0 else m, m * K, 5, m + 1, max(2, (m * K) // 3)]))
        dv = rng.uniform(0.5, 2.0, size=ln).round(2)
        pool = np.array([np.frombuffer(np.uint64(0x7FF00000000007A2).tobytes(), dtype=np.float64)[0], np.nan] +
                        ([np.inf, -np.inf] if opn == 0 else [0.0]) + ([-1.5] if opn == 1 else []))
        hit = rng.random(ln) < float(rng.choice([0.0, 0.05, 0.3]))
        dv[hit] = rng.choice(pool, size=int(hit.sum()))
        if not np.isnan(dv).all() and "naroute" not in SKIP:
            ps, js, xs = p1, j1, x1                                   # (sorted rows: rand_csr default)
            want = O.multiply_csr_by_dvec_with_NAs(ps, js, xs, dv, K, *flags, True)
            got = G.multiply_csr_by_dvec_with_NAs(ps, js, xs, dv, K, *flags, True)
            if not want["alias_structure"]:
                same(got["indptr"], want["indptr"], what + "/indptr"); same(got["indices"], want["indices"], what + "/indices")
            else:
                assert got["indptr"] is ps, what + " alias"
            gv, wv = got["values"], want["values"]
            assert gv.shape == wv.shape and np.array_equal(np.isnan(gv), np.isnan(wv)), what + " NaN pattern"
            na = lambda a: np.isnan(a) & ((a.view(np.uint64) & np.uint64(0xFFFFFFFF)) == 1954)
            assert np.array_equal(na(gv), na(wv)), what + " NA vs NaN"
            ok = ~np.isnan(wv)
            assert np.array_equal(np.isinf(gv[ok]), np.isinf(wv[ok])), what + " inf"
            fin = ok & np.isfinite(wv)
            np.testing.assert_allclose(gv[fin], wv[fin], rtol=1e-13, atol=1e-300)
    except Exception as exc:
        print("FAIL", dict(m=m, K=K, d1=d1, d2=d2, s1=s1, s2=s2, seed=seed, case=cases, what=what), repr(exc)[:600])
        sys.exit(1)


while time.time() < t_end:
    one_case(cases)
    cases += 1
    if cases == BASE_AT:
        base = footprint()
    elif TRACK and cases % 500 == 0:
        print("  footprint at case", cases, footprint(), flush=True)
if base is not None:
    end = footprint()
    print("footprint after %d cases [live blocks, live bytes, idle bytes, device bytes in use, host RSS]: %s" % (BASE_AT, base))
    print("footprint after %d cases: %s" % (cases, end))
    assert end[0] == base[0] and end[1] == base[1], "device blocks still held by finished calls: a leak"
    # (blocks idle in the pool are device memory in use by design: what must not grow is the rest)
    assert (end[3] - end[2]) - (base[3] - base[2]) <= 256 << 20, "device memory in use outside the pool grew by more than 256 MB"
    assert end[4] - base[4] <= 512 << 20, "host RSS grew by more than 512 MB over the run"
print(f"fuzz OK: {cases} cases in {budget:.0f} s (seed {seed})")
