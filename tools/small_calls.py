"""What one export-level call costs for SMALL operands (SURVEY §5 "min-size threshold for GPU offload"):
p50 microseconds per call of the four hot-path exports at the reference's own test sizes (100 x 50, density .4, %*% 50 x 20:
tests/testthat/test-matmul.R:108-114; 1000 x 500 slice: test-slice.R:6-16) and at 1e4 / 1e5 / 1e6 entries, beside the CPU
restatement of the reference (oracle, one thread and all threads) on the same inputs — the crossover is what
`mxgpu_enable(min_nnz =)` (matrixextra_amd/R/mxgpu_overlay.R) should default to.

  python tools/small_calls.py [--out gpurun_out/small_calls.json]

Host arrays in, host arrays out (what R's .Call pays), through ctypes (a few microseconds per call, like .Call itself)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from matrixextra_amd import _lib, exports as G, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def p50_us(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    t = np.empty(reps)
    for k in range(reps):
        t0 = time.perf_counter()
        fn()
        t[k] = time.perf_counter() - t0
    return {"p50_us": round(float(np.median(t)) * 1e6, 1), "p10_us": round(float(np.quantile(t, 0.1)) * 1e6, 1),
            "p90_us": round(float(np.quantile(t, 0.9)) * 1e6, 1)}


def csr(m, K, per_row, seed):
    return synth.csr_fixed(m, K, per_row, seed=seed)


def point(name, m, K, per_row, n, r):
    p, j, x = csr(m, K, per_row, 3)
    p2, j2, x2 = synth.csr_overlapping(p, j, K, per_row)
    Y = np.asfortranarray(synth.dense_normal(n, K, seed=4))          # tcrossprod_csr_dense: Y is n x K column-major
    v = synth.dense_normal(K, 1, seed=5).reshape(-1)
    rows = synth.rows_with_replacement(r, m)
    nt = O.max_threads()
    reps = 200 if p[-1] <= 200_000 else 60
    out = {"shape": {"m": m, "K": K, "per_row": per_row, "nnz": int(p[-1]), "n": n, "rows_taken": r}}
    legs = {
        "tcrossprod_csr_dense_numeric": (lambda: G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1),
                                         lambda t: (lambda: O.tcrossprod_csr_dense_numeric(p, j, x, Y, t))),
        "matmul_csr_dvec_numeric": (lambda: G.matmul_csr_dvec_numeric(p, j, x, v, 1), lambda t: (lambda: O.matmul_csr_dvec_numeric(p, j, x, v, t))),
        "add_csr_elemwise": (lambda: G.add_csr_elemwise(p, p2, j, j2, x, x2, False), lambda t: (lambda: O.add_csr_elemwise(p, p2, j, j2, x, x2, False))),
        "multiply_csr_elemwise": (lambda: G.multiply_csr_elemwise(p, p2, j, j2, x, x2), lambda t: (lambda: O.multiply_csr_elemwise(p, p2, j, j2, x, x2))),
        "copy_csr_rows_numeric": (lambda: G.copy_csr_rows_numeric(p, j, x, rows), lambda t: (lambda: O.copy_csr_rows_numeric(p, j, x, rows))),
    }
    for leg, (gpu_fn, cpu_fn) in legs.items():
        g = p50_us(gpu_fn, reps)
        c1 = p50_us(cpu_fn(1), max(20, reps // 4), 3)
        rec = {"gpu": g, "cpu_1_thread": c1}
        if leg in ("tcrossprod_csr_dense_numeric", "matmul_csr_dvec_numeric"):      # the reference's OpenMP loops; merges / gather are serial there
            rec["cpu_all_threads"] = dict(p50_us(cpu_fn(nt), max(20, reps // 4), 3), threads=nt)
        best_cpu = min(rec["cpu_1_thread"]["p50_us"], rec.get("cpu_all_threads", {"p50_us": 1e30})["p50_us"])
        rec["gpu_over_cpu"] = round(g["p50_us"] / best_cpu, 2)
        out[leg] = rec
    print(f"[small_calls] {name}: " + ", ".join(f"{k} {v['gpu']['p50_us']:.0f}us ({v['gpu_over_cpu']}x cpu)" for k, v in out.items() if k != "shape"),
          file=sys.stderr, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "small_calls.json"))
    args = ap.parse_args()
    _lib.load()
    doc = {"device": _lib.device_name(), "host_threads": O.max_threads(), "points": {}}
    for name, (m, K, per_row, n, r) in {
        "test_matmul_R_100x50": (100, 50, 20, 20, 30),                 # density .4 (test-matmul.R:108-114)
        "test_slice_R_1000x500": (1000, 500, 50, 20, 300),             # density .1 (test-slice.R:6-16)
        "nnz_1e4": (1000, 2000, 10, 32, 300),
        "nnz_1e5": (5000, 10_000, 20, 32, 1500),
        "nnz_1e6": (31_250, 50_000, 32, 32, 10_000),
        "nnz_1e7": (312_500, 100_000, 32, 32, 60_000),
    }.items():
        doc["points"][name] = point(name, m, K, per_row, n, r)
    # the crossover per export: the smallest measured nnz from which the GPU call is faster than the best CPU figure
    cross = {}
    for leg in ("tcrossprod_csr_dense_numeric", "matmul_csr_dvec_numeric", "add_csr_elemwise", "multiply_csr_elemwise", "copy_csr_rows_numeric"):
        wins = [pt["shape"]["nnz"] for pt in doc["points"].values() if pt[leg]["gpu_over_cpu"] < 1.0]
        cross[leg] = min(wins) if wins else None
    doc["gpu_faster_from_nnz"] = cross
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(cross))


if __name__ == "__main__":
    main()
