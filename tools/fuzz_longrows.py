#!/usr/bin/env python3
"""Randomised stress of round 5's paths for rows of very uneven length (run on the GPU box):
    python tools/fuzz_longrows.py [seconds] [seed]
  * the row-split kernel's long-rows path (pieces + in-order combine) through DeviceCSR.spmm with the matrix profile, random
    piece length / segments / row groups / panels (both launch forms) / layout / dtype, small-integer operands: EXACT;
  * CSR (+) CSR with row pairs beyond the lane group and beyond the one-workgroup-per-pair threshold (>= 2^21 entries),
    every operation, structure and values bit for bit against the oracle;
  * the one-launch gather on tiles of very uneven rows (flat copy) and on few tiles (size-only launch + per-row copy)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from matrixextra_amd import device as D, exports as G, _lib
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = {"spmm": 0, "merge": 0, "gather": 0}


def csr_from_lengths(lens, K, dtype="d"):
    lens = np.minimum(lens, K)
    m = lens.size
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = np.unique(row * K + rng.integers(0, K, size=row.size))
    row = key // K
    j = (key - row * K).astype(np.int32)
    p = np.zeros(m + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=m), out=p[1:])
    x = rng.integers(-2, 3, size=j.size).astype(np.float64)
    return p.astype(np.int32), j, x


def same(got, want, what):
    for k in ("indptr", "indices", "values"):
        if k in want and not np.array_equal(np.asarray(got[k]), np.asarray(want[k])):
            raise AssertionError(f"{what}: {k} differs")


while time.time() < t_end:
    which = rng.choice(["spmm", "merge", "gather"], p=[0.5, 0.25, 0.25])
    what = None
    try:
        if which == "spmm":
            m = int(rng.choice([300, 1500, 5000])); K = int(rng.choice([2000, 9000]))
            lens = rng.integers(0, 30, size=m)
            for r in rng.choice(m, size=int(rng.integers(1, 12)), replace=False):
                lens[r] = int(rng.choice([128, 129, 255, 256, 257, 600, 1024, 1500, K - 1, K]))
            p, j, x = csr_from_lengths(lens, K)
            dtype = np.float64 if rng.random() < 0.6 else np.float32
            vec = 2 if dtype == np.float64 else 4
            n = int(rng.choice([vec, 16, 64, 100, 132, 260])) // vec * vec
            B = rng.integers(-3, 4, size=(K, n)).astype(dtype)
            colmajor = bool(rng.random() < 0.5)
            S = int(rng.choice([1, 1, 2, 4, 8, -1])); P = int(rng.choice([1, 1, 2, 5]))
            os.environ["MXGPU_LONG_PIECE"] = str(int(rng.choice([128, 256, 512])))
            os.environ["MXGPU_ROWSPLIT_LAUNCHES"] = str(int(rng.integers(0, 2)))
            what = dict(m=m, K=K, n=n, dtype=dtype.__name__, colmajor=colmajor, S=S, P=P, piece=os.environ["MXGPU_LONG_PIECE"],
                        launches=os.environ["MXGPU_ROWSPLIT_LAUNCHES"], nnz=int(j.size))
            A = D.DeviceCSR.from_host(p, j, x, K)
            got = D.spmm(A, torch.from_numpy(B).cuda(), colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S).cpu().numpy()
            ref = np.zeros((m, n))
            np.add.at(ref, np.repeat(np.arange(m), np.diff(p)), x[:, None] * B[j].astype(np.float64))
            assert np.array_equal(got, ref), "exact (small-integer) product differs"
            del A
        elif which == "merge":
            m = int(rng.choice([2048, 2600, 3500])); K = 70_000
            l1 = rng.integers(0, 40, size=m); l2 = rng.integers(0, 40, size=m)
            big = rng.choice(m, size=70, replace=False)
            l1[big] = rng.integers(9_000, 30_000, size=big.size)
            l2[big] = np.where(rng.random(big.size) < 0.6, rng.integers(9_000, 30_000, size=big.size), rng.integers(0, 70, size=big.size))
            for r in rng.choice(m, size=12, replace=False):
                l1[r] = int(rng.choice([64, 65, 1024, 1025, 2048, 5000])); l2[r] = int(rng.choice([0, 1, 63, 1025, 3000]))
            p1, j1, x1 = csr_from_lengths(l1, K); p2, j2, x2 = csr_from_lengths(l2, K)
            if rng.random() < 0.3:                                   # heavy overlap: B's long rows are subsets of A's
                for r in big[:20]:
                    a = j1[p1[r]:p1[r + 1]]; nb = p2[r + 1] - p2[r]
                    if 0 < nb <= a.size:
                        j2[p2[r]:p2[r + 1]] = np.sort(rng.choice(a, size=nb, replace=False))
            what = dict(m=m, nnz=[int(j1.size), int(j2.size)])
            for sub in (False, True):
                same(G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), f"add sub={sub}")
            same(G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), "mul")
            b1, b2 = (x1 > 0).astype(np.int32), (x2 > 0).astype(np.int32)
            same(G.logicalor_csr_elemwise(p1, p2, j1, j2, b1, b2, False), O.logicalor_csr_elemwise(p1, p2, j1, j2, b1, b2, False), "or")
            same(G.logicaland_csr_elemwise(p1, p2, j1, j2, b1, b2), O.logicaland_csr_elemwise(p1, p2, j1, j2, b1, b2), "and")
        else:
            m = int(rng.choice([3000, 40_000])); K = 50_000
            lens = rng.integers(0, 50, size=m) if rng.random() < 0.7 else np.full(m, 300)
            for r in rng.choice(m, size=int(rng.integers(0, 30)), replace=False):
                lens[r] = int(rng.choice([500, 3000, 20_000, 49_999]))
            p, j, x = csr_from_lengths(lens, K)
            r = int(rng.choice([1, 200, 257, 2000, 20_000]))
            rows = rng.integers(0, m, size=r).astype(np.int32)
            what = dict(m=m, r=r, nnz=int(j.size))
            A = D.DeviceCSR.from_host(p, j, x, K)
            R = D.csr_gather_rows(A, torch.from_numpy(rows).cuda())
            want = O.copy_csr_rows_numeric(p, j, x, rows) if hasattr(O, "copy_csr_rows_numeric") else None
            gp, gj, gx = R.indptr.cpu().numpy(), R.indices.cpu().numpy(), R.values.cpu().numpy()
            lens_t = np.diff(p)[rows]
            ep = np.zeros(r + 1, dtype=np.int64); ep[1:] = np.cumsum(lens_t)
            assert np.array_equal(gp, ep.astype(np.int32)), "gather: indptr differs"
            src = np.concatenate([np.arange(p[q], p[q + 1]) for q in rows]) if r else np.zeros(0, dtype=np.int64)
            assert np.array_equal(gj, j[src]) and np.array_equal(gx, x[src]), "gather: entries differ"
            del A, R, want
    except Exception as exc:  # noqa: BLE001
        print("FAIL", which, what, repr(exc)[:400], dict(seed=seed, cases=cases))
        sys.exit(1)
    cases[which] += 1
for k in ("MXGPU_LONG_PIECE", "MXGPU_ROWSPLIT_LAUNCHES"):
    os.environ.pop(k, None)
print(f"long-rows fuzz OK: {cases} in {budget:.0f} s (seed {seed})")
