"""The two workloads the reference's vignette itself runs (vignettes/Introducing_MatrixExtra.Rmd):
  * :247-251  dense 100 x 1e4 %*% CSC 1e4 x 1e4 (density 0.05) — the one product it publishes a time for (72.74 ms, html:668);
    path matmul_dense_csc_numeric (src/matmul.cpp:188-235);
  * :442-502  cbind(1, X) -> X[ix, ] -> an optimiser loop of `X %*% w` and `X * v` on one X (tools/vignette_loop.py)."""
import os
import sys

import numpy as np
import pytest

from matrixextra_amd import exports as G, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_vignette_dense_times_csc(gpu):
    m, K, n = 10_000, 10_000, 100
    p, i, x = synth.csr_fixed(m, K, 500, seed=7)                       # CSC columns: 500 of 1e4 rows each = density 0.05
    X = np.asfortranarray(synth.dense_normal(n, K, seed=8))            # Y_dense 100 x 1e4
    out = G.matmul_dense_csc_numeric(X, p, i, x, 1)
    ref = O.matmul_dense_csc(X, p, i, x, O.max_threads(), True)
    assert out.shape == (n, m) and out.flags.f_contiguous
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    dense = np.zeros((K, m))                                           # the test the reference itself applies: dense equivalence
    dense[i, np.repeat(np.arange(m), 500)] = x
    np.testing.assert_allclose(out, X @ dense, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    # float32 twin and the unsorted-column case (dense x CSC never needs sorted indices: matmul.cpp:118-142 accumulates)
    perm = np.random.default_rng(0).permutation(500)
    i2, x2 = i.reshape(m, 500)[:, perm].reshape(-1), x.reshape(m, 500)[:, perm].reshape(-1)
    out2 = G.matmul_dense_csc_numeric(X, p, i2, x2, 1)
    np.testing.assert_allclose(out2, ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max())
    outf = G.matmul_dense_csc_float32(X.astype(np.float32), p, i, x, 1)
    np.testing.assert_allclose(outf, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


def test_vignette_optimiser_loop_through_the_exports(gpu):
    import vignette_loop
    r = vignette_loop.run(iters=200)
    print(r)
    # X_train is handed over 400 times by the same three host vectors: everything after the first call of each kind is a hit
    assert r["cache_hits"] >= 2 * 200 - 2 and r["cache_misses"] <= 4
    assert r["spmv_planned_calls"] >= 199                              # requested with mx_set_option("spmv_planned", 1)
    assert r["parity"]["final_coefficients_max_rel_err_vs_oracle_loop"] <= 1e-9
    assert r["parity"]["spmv_planned_max_rel_err"] <= 1e-12
    # the default (bit-exact flat kernel on every call) gives the oracle's coefficients too
    r0 = vignette_loop.run(iters=20, planned=False)
    assert r0["spmv_planned_calls"] == 0 and r0["parity"]["final_coefficients_max_rel_err_vs_oracle_loop"] <= 1e-12
