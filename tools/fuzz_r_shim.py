#!/usr/bin/env python3
"""Randomised edge shapes through the `.Call` shim by routine NAME (the mock R runtime of tests/r_mock, gctorture on):
0 / 1-row matrices, matrices without entries, all-empty rows, one column, empty and repeated row selections, NA / NaN
values — results against the CPU oracle (bit for bit for structures and copies, 1e-12 / 1e-5 for products).
    python tools/fuzz_r_shim.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "r_mock"))
import numpy as np
import rmock
from conftest import rand_csr
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
rmock.build()
R = rmock.runtime()
R.gctorture(True)
NA = rmock.NA_INTEGER
# every object a case makes (arguments and results) is released at its end: the collector marks from all held objects on
# every allocation
made = []
_alloc, _call = R.L.rmock_alloc, R.call
class _L:                                                  # the ctypes library with rmock_alloc recorded
    def __getattr__(self, k):
        return getattr(R_L, k)
    def rmock_alloc(self, t, n):
        o = _alloc(t, n); made.append(o); return o
R_L = R.L
R.L = _L()
def _call_rec(name, *a):
    o = _call(name, *a)
    if o is not None:
        made.append(o)
    return o
R.call = _call_rec


def eq(g, w, what):
    g, w = np.asarray(g), np.asarray(w)
    assert g.dtype == w.dtype and g.shape == w.shape and g.tobytes() == w.tobytes(), (what, g.dtype, w.dtype, g.shape, w.shape)


def eq_list(s, want, what):
    g = R.as_py(s)
    assert list(g) == [k for k in ("indptr", "indices", "values") if k in want], (what, list(g), list(want))
    for k in g:
        wv = want[k]
        wv = np.zeros(0, dtype=g[k].dtype) if wv is None else np.asarray(wv)
        eq(g[k], wv.astype(g[k].dtype, copy=False), f"{what}/{k}")


def close(g, w, what, tol):
    """tol 1e-5 = a float32 product: the sums differ from the oracle's by accumulation order, i.e. by ~eps32 * sqrt(terms) *
    the magnitude of the partial sums — the absolute bar scales with the row length"""
    g, w = np.asarray(g), np.asarray(w)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    assert np.array_equal(np.isnan(g), np.isnan(w)), what + " NaN pattern"
    ok = ~np.isnan(w)
    np.testing.assert_allclose(g[ok], w[ok], rtol=tol, atol=tol * (1 if tol < 1e-6 else max(1.0, K / 8)), err_msg=what)


cases = 0
t_end = time.time() + budget
while time.time() < t_end:
    m = int(rng.choice([0, 1, 1, 2, 5, 63, 64, 65, 200]))
    K = int(rng.choice([1, 1, 2, 9, 70, 300]))
    d = float(rng.choice([0.0, 0.0, 0.1, 0.5, 1.0]))
    n = int(rng.choice([1, 2, 17]))
    s = int(rng.integers(1 << 30))
    what = "?"
    try:
        if m == 0:
            p, j, x = np.zeros(1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0)
        else:
            p, j, x = rand_csr(m, K, d, seed=s)
        if x.size and rng.random() < 0.3:
            x = x.copy(); x[rng.integers(x.size)] = rng.choice([np.nan, np.inf, -np.inf, 0.0])
        xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
        sp, sj, sx, sxl = R.integer(p), R.integer(j), R.real(x), R.logical(xl)
        one = R.integer([1])
        # ---- products
        Y = rng.normal(size=(n, K))
        what = "tcrossprod_csr_dense_numeric"
        o = R.call(what, sp, sj, sx, R.matrix(Y), one)
        assert R.view(o).shape == (m, n), what
        if m:
            close(R.view(o), O.tcrossprod_csr_dense_numeric(p, j, x, np.asfortranarray(Y)), what, 1e-12)
        what = "tcrossprod_csr_dense_float32"
        xf = np.nan_to_num(x, nan=0.5, posinf=2.0, neginf=-2.0)
        Y32 = Y.astype(np.float32)
        o = R.call(what, sp, sj, R.real(xf), R.matrix(Y32, "float32"), one)
        assert R.view(o).shape == (m, n) and R.typeof(o) == rmock.INTSXP, what
        if m:
            close(R.view(o).view(np.float32), O.tcrossprod_csr_dense_float32(p, j, xf, np.asfortranarray(Y32)), what, 1e-5)
        if m:
            what = "matmul_dense_csc_numeric"                            # the CSR arrays read as the CSC of a K x m matrix
            X = rng.normal(size=(n, K))
            o = R.call(what, R.matrix(X), sp, sj, sx, one)
            assert R.view(o).shape == (n, m), what
            close(R.view(o), O.matmul_dense_csc_numeric(np.asfortranarray(X), p, j, x), what, 1e-12)
            what = "tcrossprod_dense_csr_numeric"
            o = R.call(what, R.matrix(X), sp, sj, sx, one, R.integer([K]))
            close(R.view(o), O.tcrossprod_dense_csr_numeric(np.asfortranarray(X), p, j, x, 1, K), what, 1e-12)
        # ---- SpMV, four kinds
        v = rng.normal(size=K)
        if m:
            what = "matmul_csr_dvec_numeric"
            close(R.view(R.call(what, sp, sj, sx, R.real(v), one)), O.matmul_csr_dvec_numeric(p, j, x, v), what, 1e-12)
            vi = rng.integers(-3, 4, size=K).astype(np.int32)
            if rng.random() < 0.5:
                vi[rng.integers(K)] = NA
            what = "matmul_csr_dvec_integer"
            close(R.view(R.call(what, sp, sj, sx, R.integer(vi), one)), O.matmul_csr_dvec_integer(p, j, x, vi), what, 1e-12)
            vl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=K)
            what = "matmul_csr_dvec_logical"
            close(R.view(R.call(what, sp, sj, sx, R.logical(vl), one)), O.matmul_csr_dvec_logical(p, j, x, vl), what, 1e-12)
            what = "matmul_csr_dvec_float32"
            v32 = v.astype(np.float32)
            close(R.view(R.call(what, sp, sj, R.real(xf), R.float32(v32), one)).view(np.float32), O.matmul_csr_dvec_float32(p, j, xf, v32), what, 1e-5)
        # ---- merges
        if m:
            p2, j2, x2 = rand_csr(m, K, float(rng.choice([0.0, 0.2, 1.0])), seed=s + 1)
            xl2 = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x2.size)
            sp2, sj2, sx2, sxl2 = R.integer(p2), R.integer(j2), R.real(x2), R.logical(xl2)
            what = "multiply_csr_elemwise"
            eq_list(R.call(what, sp, sp2, sj, sj2, sx, sx2), O.multiply_csr_elemwise(p, p2, j, j2, x, x2), what)
            for sub in (0, 1):
                what = f"add_csr_elemwise sub={sub}"
                eq_list(R.call("add_csr_elemwise", sp, sp2, sj, sj2, sx, sx2, R.logical([sub])), O.add_csr_elemwise(p, p2, j, j2, x, x2, bool(sub)), what)
            what = "logicaland_csr_elemwise"
            eq_list(R.call(what, sp, sp2, sj, sj2, sxl, sxl2), O.logicaland_csr_elemwise(p, p2, j, j2, xl, xl2), what)
            xo = int(rng.integers(2))
            what = f"logicalor_csr_elemwise xor={xo}"
            eq_list(R.call("logicalor_csr_elemwise", sp, sp2, sj, sj2, sxl, sxl2, R.logical([xo])), O.logicalor_csr_elemwise(p, p2, j, j2, xl, xl2, bool(xo)), what)
        # ---- row gather
        if m:
            rows = rng.integers(0, m, size=int(rng.choice([0, 1, 3, 2 * m + 1]))).astype(np.int32)
            sr = R.integer(rows)
            what = "copy_csr_rows_numeric"
            eq_list(R.call(what, sp, sj, sx, sr), O.copy_csr_rows_numeric(p, j, x, rows), what)
            what = "copy_csr_rows_logical"
            eq_list(R.call(what, sp, sj, sxl, sr), O.copy_csr_rows_logical(p, j, xl, rows), what)
            what = "copy_csr_rows_binary"
            eq_list(R.call(what, sp, sj, sr), O.copy_csr_rows_binary(p, j, rows), what)
            what = "check_is_seq"
            assert R.view(R.call(what, sr)).tolist() == [int(O.check_is_seq(rows))], what
            what = "check_is_rev_seq"
            assert R.view(R.call(what, sr)).tolist() == [int(O.check_is_rev_seq(rows))], what
        # ---- the widened rows: column slices, reversals, cbind, sparse vector, dense elemwise, sort, drop zeros
        if m:
            rows = rng.integers(0, m, size=int(rng.choice([1, 3, m + 2]))).astype(np.int32)
            sr = R.integer(rows)
            c0 = int(rng.integers(0, K)); c1 = int(rng.integers(c0, K))
            cols, index1 = (np.arange(c0, c1 + 1, dtype=np.int32), False) if rng.random() < 0.5 else (np.arange(c1, c0 - 1, -1, dtype=np.int32), False)
            sc, si = R.integer(cols), R.logical([int(index1)])
            what = "copy_csr_rows_col_seq_numeric"
            eq_list(R.call(what, sp, sj, sx, sr, sc, si), O.copy_csr_rows_col_seq_numeric(p, j, x, rows, cols, index1), what)
            what = "copy_csr_rows_col_seq_binary"
            eq_list(R.call(what, sp, sj, sr, sc, si), O.copy_csr_rows_col_seq_binary(p, j, rows, cols, index1), what)
            acols = rng.integers(0, K, size=int(rng.choice([1, 2, K + 3]))).astype(np.int32)
            what = "copy_csr_arbitrary_numeric"
            eq_list(R.call(what, sp, sj, sx, sr, R.integer(acols)), O.copy_csr_arbitrary_numeric(p, j, x, rows, acols), what)
            what = "copy_csr_arbitrary_logical"
            eq_list(R.call(what, sp, sj, sxl, sr, R.integer(acols)), O.copy_csr_arbitrary_logical(p, j, xl, rows, acols), what)
            what = "reverse_rows_numeric"
            eq_list(R.call(what, sp, sj, sx), O.reverse_rows_numeric(p, j, x), what)
            what = "reverse_columns_inplace_numeric"
            jj, vv = j.copy(), x.copy()
            O.reverse_columns_inplace(p, jj, vv, K)
            tj, tv = R.integer(j), R.real(x)
            assert R.call(what, sp, tj, tv, R.integer([K])) is None, what
            eq(R.view(tj), jj, what); eq(R.view(tv), vv, what)
            what = "cbind_csr_numeric"
            pb, jb, xb = rand_csr(m, 5, float(rng.choice([0.0, 0.4])), seed=s + 2)
            jbs = (jb + K).astype(np.int32)
            eq_list(R.call(what, sp, sj, sx, R.integer(pb), R.integer(jbs), R.real(xb)), O.cbind_csr_numeric(p, j, x, pb, jbs, xb), what)
            what = "matmul_csr_svec_numeric"
            ny = int(rng.integers(0, K + 1))
            yi = (np.sort(rng.permutation(K)[:ny]) + 1).astype(np.int32)
            yv = rng.normal(size=ny)
            if ny:
                close(R.view(R.call(what, sp, sj, R.real(xf), R.integer(yi), R.real(yv), one)), O.matmul_csr_svec_numeric(p, j, xf, yi, yv), what, 1e-11)
            what = "multiply_csr_by_dense_elemwise_double"
            Dm = rng.normal(size=(m, K))
            eq(R.view(R.call(what, sp, sj, sx, R.matrix(Dm))), O.multiply_csr_by_dense_elemwise_double(p, j, x, Dm), what)
            what = "sort_sparse_indices_numeric"
            pu, ju, xu = rand_csr(m, K, d, seed=s + 3, sorted_cols=False)
            wj, wv = O.sort_sparse_indices(pu, ju, xu)
            tj, tv = R.integer(ju), R.real(xu)
            assert R.call(what, R.integer(pu), tj, tv) is None, what
            eq(R.view(tj), wj, what); eq(R.view(tv), wv, what)
            what = "check_indices_are_unsorted"
            assert R.view(R.call(what, R.integer(pu), R.integer(ju))).tolist() == [int(O.check_indices_are_sorted(pu, ju))], what
            what = "remove_zero_valued_csr_numeric"
            xz = x.copy()
            if xz.size:
                xz[rng.random(xz.size) < 0.3] = 0.0
            na = bool(rng.integers(2))
            eq_list(R.call(what, sp, sj, R.real(xz), R.logical([int(na)])), O.remove_zero_valued_csr_numeric(p, j, xz, na), what)
        R.check_clean()
        for o in made:
            R_L.rmock_release(o)
        made.clear()
        R_L.rmock_sweep_dead()
    except Exception as exc:
        print("FAIL", dict(m=m, K=K, d=d, n=n, s=s, seed=seed, case=cases, what=what), repr(exc)[:800])
        sys.exit(1)
    cases += 1
print(f"shim fuzz OK: {cases} cases in {budget:.0f} s (seed {seed})")
