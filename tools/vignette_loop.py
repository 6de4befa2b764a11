#!/usr/bin/env python3
"""The reference vignette's usage loop, replayed through the EXPORT level (what R would call), on a matrix of real-sim's
shape (vignettes/Introducing_MatrixExtra.Rmd:442-502; the data set itself is not in the image):

    X <- cbind(rep(1, nrow(X)), X)              cbind_csr_numeric           src/cbind.cpp:101-119
    X_train <- X[ix_train, ]                    copy_csr_rows_numeric       src/slice.cpp:276-291
    repeat (optim / L-BFGS-B evaluates these per step):
        pred <- 1 / (1 + exp(-(X_train %*% w))) matmul_csr_dvec_numeric     src/matmul.cpp:381-426
        G    <- X_train * as.numeric(pred - y)  multiply_csr_by_dvec_no_NAs src/operators.cpp:1604-2175
        grad <- colMeans(G) + 2 lambda w        (colMeans: not on the path — host arithmetic here as in R)

The optimiser is replaced by plain gradient steps of a fixed length: the sequence of library calls per evaluation is the
same and the run is deterministic.  Every library call goes through matrixextra_amd.exports (ctypes -> C-ABI -> HIP);
the oracle replays the same loop on the CPU.  Reported: per-call milliseconds of the two hot calls, CSR-cache hits, how many
products the planned SpMV served (opt-in: mx_set_option("spmv_planned", 1)), and the parity of every call of the first
iterations plus of the final coefficients.  The reference prints 4.01 s for its whole optim() run on unstated hardware
(inst/doc/Introducing_MatrixExtra.html:829) — context, not a comparison: iteration counts differ."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def colmeans(p, j, v, ncol):
    return np.bincount(j, weights=v, minlength=ncol) / (p.size - 1)


def run(iters=200, m=72_309, K=20_958, mean_nnz=51, planned=True, check_iters=3, verbose=False):
    from matrixextra_amd import _lib, exports as G, synth
    from oracle import oracle as O
    lib = _lib.load()

    def opt(name):
        v = C.c_int64(0)
        _lib.check(lib.mx_get_option(name.encode(), C.byref(v)))
        return v.value

    def stats():
        b, e, h, mi = C.c_int64(0), C.c_int(0), C.c_int64(0), C.c_int64(0)
        lib.mx_cache_stats(C.byref(b), C.byref(e), C.byref(h), C.byref(mi))
        return dict(bytes=b.value, entries=e.value, hits=h.value, misses=mi.value)
    lib.mx_cache_invalidate(None)
    p, j, x = synth.csr_skewed_fast(m, K, mean_nnz, seed=11, sigma=0.8)
    rng = np.random.default_rng(1)
    y = (rng.random(m) < 0.3).astype(np.float64)
    res = {"shape": {"rows": m, "cols": K, "nnz": int(p[-1])}, "iterations": iters, "spmv_planned_requested": bool(planned)}

    # ---- cbind(1, X): an m x 1 CSR of ones in front; the data's column ids shifted by one (R/cbind.R passes them so)
    ones_p = np.arange(m + 1, dtype=np.int32)
    ones_j = np.zeros(m, dtype=np.int32)
    ones_x = np.ones(m)
    t0 = time.perf_counter()
    Xc = G.cbind_csr_numeric(ones_p, ones_j, ones_x, p, j + 1, x)
    res["cbind_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    Xo = O.cbind_csr_numeric(ones_p, ones_j, ones_x, p, (j + 1).astype(np.int32), x)
    assert all(np.array_equal(Xc[k], Xo[k]) for k in ("indptr", "indices", "values")), "cbind differs from the oracle"
    Kc = K + 1

    # ---- X[ix_train, ]
    ix = rng.permutation(m)[: m // 2].astype(np.int32)
    t0 = time.perf_counter()
    Xt = G.copy_csr_rows_numeric(Xc["indptr"], Xc["indices"], Xc["values"], ix)
    res["slice_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    Xto = O.copy_csr_rows_numeric(Xo["indptr"], Xo["indices"], Xo["values"], ix)
    assert all(np.array_equal(Xt[k], Xto[k]) for k in ("indptr", "indices", "values")), "row slice differs from the oracle"
    tp, tj, tx = Xt["indptr"], Xt["indices"], Xt["values"]
    yt = y[ix]
    mt = tp.size - 1
    short_rows = (tp[1:] - tp[:-1]) <= 256
    res["train"] = {"rows": mt, "cols": Kc, "nnz": int(tp[-1])}

    lam, step = 1e-5, 2.0
    _lib.check(lib.mx_set_option(b"spmv_planned", C.c_int64(1 if planned else 0)))
    try:
        s0, n0 = stats(), opt("spmv_planned_calls")
        w = np.zeros(Kc)
        wo = np.zeros(Kc)
        t_mv, t_mul = [], []
        worst_mv = worst_mul = 0.0
        for it in range(iters):
            t0 = time.perf_counter()
            z = G.matmul_csr_dvec_numeric(tp, tj, tx, w)
            t_mv.append(time.perf_counter() - t0)
            pred = 1.0 / (1.0 + np.exp(-z))
            d = pred - yt
            t0 = time.perf_counter()
            gv = G.multiply_csr_by_dvec_no_NAs_numeric(tp, tj, tx, d, Kc, True, False, False, False, False, True)
            t_mul.append(time.perf_counter() - t0)
            grad = colmeans(tp, tj, gv, Kc) + 2.0 * lam * w
            grad[0] -= 2.0 * lam * w[0]
            if it < check_iters:                                    # every call of the first iterations against the oracle, same inputs
                zo = O.matmul_csr_dvec_numeric(tp, tj, tx, w)
                if planned and it > 0:
                    worst_mv = max(worst_mv, float(np.max(np.abs(z - zo)) / max(np.max(np.abs(zo)), 1e-300)))
                else:
                    # flat kernel: bit for bit the oracle's loop for rows of up to 256 entries (one thread adds them in
                    # storage order); longer rows are summed by a wavefront (reassociated)
                    assert np.array_equal(z[short_rows], zo[short_rows]), "SpMV (flat kernel) differs from the oracle's loop"
                    np.testing.assert_allclose(z, zo, rtol=1e-12, atol=1e-12 * max(float(np.max(np.abs(zo))), 1e-300))
                gvo = O.multiply_csr_by_dvec_no_NAs_numeric(tp, tj, tx, d, Kc, True, False, False, False, False, True)
                assert np.array_equal(gv, gvo), "CSR * vector differs from the oracle"
            w = w - step * grad
        # the oracle's own loop, start to end
        for it in range(iters):
            zo = O.matmul_csr_dvec_numeric(tp, tj, tx, wo)
            do = 1.0 / (1.0 + np.exp(-zo)) - yt
            gvo = O.multiply_csr_by_dvec_no_NAs_numeric(tp, tj, tx, do, Kc, True, False, False, False, False, True)
            go = colmeans(tp, tj, gvo, Kc) + 2.0 * lam * wo
            go[0] -= 2.0 * lam * wo[0]
            wo = wo - step * go
        s1, n1 = stats(), opt("spmv_planned_calls")
    finally:
        _lib.check(lib.mx_set_option(b"spmv_planned", C.c_int64(-1)))
    coef_err = float(np.max(np.abs(w - wo)) / np.max(np.abs(wo)))
    res.update({
        "cache_hits": s1["hits"] - s0["hits"], "cache_misses": s1["misses"] - s0["misses"],
        "spmv_planned_calls": n1 - n0,
        "spmv_ms": {"first": round(t_mv[0] * 1e3, 3), "median": round(float(np.median(t_mv[2:])) * 1e3, 3),
                    "min": round(min(t_mv) * 1e3, 3)},
        "csr_times_vector_ms": {"first": round(t_mul[0] * 1e3, 3), "median": round(float(np.median(t_mul[2:])) * 1e3, 3),
                                "min": round(min(t_mul) * 1e3, 3)},
        "loop_s": round(sum(t_mv) + sum(t_mul), 4),
        "parity": {"spmv_first_call": "bitwise for rows of <= 256 entries, 1e-12 beyond (flat kernel)", "spmv_planned_max_rel_err": worst_mv,
                   "csr_times_vector": "bitwise", "final_coefficients_max_rel_err_vs_oracle_loop": coef_err},
        "reference_published": "4.01 s for the vignette's optim() run, hardware unstated (html:829) — context only"})
    assert coef_err <= 1e-9, f"coefficients after {iters} steps differ from the oracle loop: {coef_err}"
    assert worst_mv <= 1e-12
    if verbose:
        print(json.dumps(res, indent=1), file=sys.stderr)
    return res


if __name__ == "__main__":
    print(json.dumps(run(verbose=False, planned=(len(sys.argv) < 2 or sys.argv[1] != "flat"))))
