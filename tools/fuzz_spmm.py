#!/usr/bin/env python3
"""Randomised stress of the SpMM kernels against a dense numpy product (run on the GPU box):
    python tools/fuzz_spmm.py [seconds] [seed]
Shapes, row-length distributions (uniform / heavy-tailed / mostly empty / one giant row), COLUMN distributions (uniform /
power-law / one hot column / every entry inside one column panel / one panel without entries), panel counts, layouts, dtypes
and kernels (the LDS-tile kernel included) are drawn at random; any mismatch prints the case and exits non-zero."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from devmem import spmm_device, spmm_planned_device

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    m = int(rng.choice([1, 7, 63, 64, 65, 127, 500, 1023, 1024, 1025, 3000, 9000]))
    K = int(rng.choice([1, 5, 64, 300, 2000, 20000]))
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    vec = 2 if dtype == np.float64 else 4
    n = int(rng.choice([vec, 2 * vec, 16, 32, 48, 128, 132, 260])) // vec * vec
    kind = rng.choice(["uniform", "lognormal", "sparse", "giant", "empty"])
    if kind == "uniform":
        lens = np.full(m, int(rng.integers(1, 40)))
    elif kind == "lognormal":
        lens = np.floor(rng.lognormal(2.0, float(rng.uniform(0.3, 1.8)), size=m)).astype(np.int64)
    elif kind == "sparse":
        lens = (rng.random(m) < 0.1) * rng.integers(1, 20, size=m)
    elif kind == "giant":
        lens = rng.integers(0, 6, size=m); lens[int(rng.integers(0, m))] = int(rng.integers(500, 5000))
    else:
        lens = np.zeros(m, dtype=np.int64)
    lens = np.minimum(lens, 6000).astype(np.int64)
    p = np.zeros(m + 1, dtype=np.int64); p[1:] = np.cumsum(lens)
    nnz = int(p[-1])
    cols = rng.choice(["uniform", "zipf", "hot", "one_panel", "hole"], p=[0.4, 0.25, 0.1, 0.15, 0.1])
    if cols == "zipf":                                  # power-law popularity, hot columns anywhere
        w = 1.0 / np.arange(1, K + 1) ** float(rng.uniform(0.7, 1.4))
        j = rng.permutation(K)[np.searchsorted(np.cumsum(w) / w.sum(), rng.random(nnz)).clip(0, K - 1)].astype(np.int32)
    elif cols == "hot":                                 # most entries in ONE column
        j = np.where(rng.random(nnz) < 0.7, int(rng.integers(0, K)), rng.integers(0, K, size=nnz)).astype(np.int32)
    elif cols == "one_panel":                           # every entry inside one narrow column range (one panel / one K-tile)
        lo = int(rng.integers(0, max(1, K - 8)))
        j = rng.integers(lo, min(K, lo + max(1, K // 16)), size=nnz, dtype=np.int32)
    elif cols == "hole":                                # a range of columns (a panel, a run of K-tiles) without any entry
        j = rng.integers(0, K, size=nnz, dtype=np.int32)
        a, b = K // 3, 2 * K // 3
        if b > a:
            hit = (j >= a) & (j < b)
            j[hit] = (j[hit] % max(1, a)).astype(np.int32)
    else:
        j = rng.integers(0, K, size=nnz, dtype=np.int32)
    exact = rng.random() < 0.3                          # small integers: every sum is exact in f32 and f64, a lost update is a wrong integer
    x = rng.integers(-2, 3, size=nnz).astype(np.float64) if exact else rng.uniform(-1, 1, size=nnz).round(3)
    B = (rng.integers(-3, 4, size=(K, n)) if exact else rng.normal(size=(K, n)).round(3)).astype(dtype)
    ref = np.zeros((m, n))
    if nnz:
        np.add.at(ref, np.repeat(np.arange(m), lens), x[:, None] * B[j].astype(np.float64))
    colmajor = bool(rng.random() < 0.5)
    tol = 1e-11 if dtype == np.float64 else 5e-4
    which = rng.choice(["planned", "auto", "rowwave", "rowsplit", "tile"], p=[0.35, 0.1, 0.1, 0.3, 0.15])
    if which in ("rowsplit", "tile") and rng.random() < (0.5 if which == "rowsplit" else 0.8) and nnz:   # rows sorted by column (the panel cursors' / the LDS sweep's case)
        for r in range(m):
            j[p[r]:p[r + 1]].sort()
        ref = np.zeros((m, n))
        np.add.at(ref, np.repeat(np.arange(m), lens), x[:, None] * B[j].astype(np.float64))
    npanels = int(rng.choice([0, 1, 2, 3, 8, 13]))
    try:
        if which == "planned":
            got = spmm_planned_device(p.astype(np.int32), j, x, B, colmajor, npanels=npanels,
                                      wg_per_cu=int(rng.choice([0, 1, 2, 4])), sync_mode=int(rng.choice([-1, 0, 1, 2])))
        elif which == "rowsplit":                                     # segments x column panels, any operands (odd n: the scalar path)
            if rng.random() < 0.3:
                n_odd = int(rng.choice([1, 3, 7, 33, 101]))
                B = (rng.integers(-3, 4, size=(K, n_odd)) if exact else rng.normal(size=(K, n_odd)).round(3)).astype(dtype)
                ref = np.zeros((m, n_odd))
                if nnz:
                    np.add.at(ref, np.repeat(np.arange(m), lens), x[:, None] * B[j].astype(np.float64))
            got = spmm_device(p.astype(np.int32), j, x, B, colmajor, 4, 0, npanels=int(rng.choice([0, 1, 2, 3, 7, 32])),
                              wg_per_cu=int(rng.choice([0, 1, 2, 4, 8, -1, -1])))   # -1: the row-group form
        elif which == "tile":                                         # every geometry; unsorted rows go through the flag pass
            got = spmm_device(p.astype(np.int32), j, x, B, colmajor, 5, 0, npanels=int(rng.choice([0, 0, 4, 9, 15])),
                              wg_per_cu=int(rng.choice([0, 0, 1 + 4 * 1, 1 + 4 * 5, 2 + 4 * 3, 1 + 4 * 2 + 32, 2 + 4 * 2 + 32])))
        else:
            got = spmm_device(p.astype(np.int32), j, x, B, colmajor, 0 if which == "auto" else 1, 0)
        if exact:
            assert np.array_equal(got, ref), "exact (small-integer) case differs"
        else:
            np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * 100)
    except Exception as exc:
        print("FAIL", dict(exact=exact, m=m, K=K, n=n, dtype=dtype.__name__, kind=kind, cols=str(cols), nnz=nnz, colmajor=colmajor, which=which,
                           npanels=npanels, seed=seed, case=cases), repr(exc)[:500])
        sys.exit(1)
    cases += 1
print(f"fuzz OK: {cases} cases in {budget:.0f} s (seed {seed})")
