"""AUTO on inputs that look like real dgRMatrix data: power-law columns, log-normal row lengths
(matrixextra_amd.synth.device_csr_zipf; the vignette's own application is LibSVM real-sim, Rmd:442-502).  Every SpMM kernel
family on each shape, AUTO both ways (plan kept on the matrix / the C-ABI's one-shot AUTO), the model's estimates, and the
share of the entries that the hottest columns filling one XCD's L2 hold (the quantity AUTO's `hit` term should price).

  python tools/zipf_map.py [--out gpurun_out/zipf_map.json]"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import _lib, device as D, synth  # noqa: E402
from auto_map import timeit  # noqa: E402

NAMES = {0: "auto", 1: "rowwave", 2: "slab", 3: "planned", 4: "rowsplit", 5: "tile"}


def point(m, K, mean, alpha, sigma, n, colmajor, dtype, lib):
    p, j, x = synth.device_csr_zipf(m, K, mean, alpha=alpha, sigma=sigma, seed=21)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
    A.rows_sorted()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    B = torch.randn((K, n), dtype=dtype, device="cuda", generator=g)
    out = torch.empty((n, m) if colmajor else (m, n), dtype=dtype, device="cuda")
    sz = B.element_size()
    # mass of the hottest columns that fit 4 MiB of B rows
    cnt = torch.bincount(j.to(torch.int64), minlength=K)
    top = max(1, min(K, (4 << 20) // (n * sz)))
    mass = float(torch.sort(cnt, descending=True).values[:top].sum().item()) / max(1, int(j.numel()))
    ms, kern = {}, {}

    def run(name, fn):
        try:
            fn()
            kern[name] = lib.mxd_spmm_last_kernel().decode()
            timeit(fn, reps=20)
            ms[name] = round(min(timeit(fn), timeit(fn, warm=0)), 5)
        except _lib.MxError as e:
            ms[name] = None
            kern[name] = "n/a: " + str(e)[:60]

    run("auto_kept_plan", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=True))
    run("auto_one_shot", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=False))
    run("rowwave", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=1))
    run("rowsplit", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4))
    run("rowsplit_one_panel", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=1))
    run("slab", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=2))
    if m >= 32768:
        run("planned_kept", lambda: D.spmm_planned(A, B, out=out, colmajor=colmajor))
        run("planned_rebuilt", lambda: D.spmm_planned(A, B, out=out, colmajor=colmajor, rebuild_plan=True))
    run("tile", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5))
    dt = 0 if dtype == torch.float64 else 1
    a, b, t, P, cp = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_int()
    prof = A.profile()
    _lib.check(lib.mxd_spmm_auto_cost3(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(A.nnz), C.c_int(1), C.c_int(dt), C.c_int(int(colmajor)),
                                       C.c_int(1), prof, C.byref(a), C.byref(b), C.byref(t), C.byref(P), C.byref(cp)))
    top_l2 = max(1.0, (4 << 20) / (n * sz))
    import math
    li = min(30, int(math.log2(top_l2)))
    mass_prof = prof[li] + (prof[li + 1] - prof[li]) * (math.log2(top_l2) - li)
    kept = {k: v for k, v in ms.items() if v is not None and k in ("rowwave", "rowsplit", "rowsplit_one_panel", "slab", "planned_kept", "tile")}
    one = {k: v for k, v in ms.items() if v is not None and k in ("rowwave", "rowsplit", "rowsplit_one_panel", "slab", "planned_rebuilt", "tile")}
    bk, b1 = min(kept, key=kept.get), min(one, key=one.get)
    rec = {"m": m, "K": K, "mean_drawn": mean, "mean": round(A.nnz / m, 2), "nnz": A.nnz, "alpha": alpha, "sigma": sigma, "n": n,
           "layout": "col" if colmajor else "row", "dtype": "f64" if dtype == torch.float64 else "f32", "l2_mass": round(mass, 4), "l2_mass_profile": round(float(mass_prof), 4), "row_cv": round(float(prof[32]), 3),
           "l2_byte_share": round(min(1.0, (4 << 20) / (K * n * sz)), 4), "ms": ms, "kernels": {k: kern[k] for k in ("auto_kept_plan", "auto_one_shot")},
           "model_us": {"rowsplit": round(a.value, 1), "planned": round(b.value, 1), "tile": round(t.value, 1), "panels": P.value},
           "best_kept": bk, "best_one_shot": b1, "auto_kept_over_best": round(ms["auto_kept_plan"] / kept[bk], 3),
           "auto_one_shot_over_best": round(ms["auto_one_shot"] / one[b1], 3)}
    del A, B, out, p, j, x
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "zipf_map.json"))
    args = ap.parse_args()
    lib = _lib.load()
    t0 = time.time()
    doc = {"device": _lib.device_name(), "spmm": []}
    shapes = [  # m, K, mean drawn, alpha, sigma, n
        (72_309, 20_958, 51, 1.0, 1.0, 16), (72_309, 20_958, 51, 1.0, 1.0, 64), (72_309, 20_958, 51, 1.0, 1.0, 128),     # real-sim's shape
        (1_000_000, 100_000, 40, 1.0, 1.0, 128), (1_000_000, 100_000, 40, 0.8, 0.5, 128), (1_000_000, 100_000, 40, 1.3, 1.0, 128),   # cfg2's shape
        (1_000_000, 100_000, 40, 1.0, 1.0, 16),
        (100_000, 10_000, 40, 1.0, 1.0, 128), (100_000, 100_000, 160, 1.0, 1.0, 64), (1_000_000, 10_000, 10, 1.0, 0.5, 16),
        (10_000, 10_000, 700, 1.0, 0.5, 100), (10_000, 100_000, 700, 1.1, 0.5, 100), (200_000, 50_000, 64, 1.0, 1.5, 256),
        (30_000, 5_000, 300, 0.9, 1.0, 64),
    ]
    for (m, K, mean, alpha, sigma, n) in shapes:
        for colmajor in (True, False):
            rec = point(m, K, mean, alpha, sigma, n, colmajor, torch.float64, lib)
            doc["spmm"].append(rec)
            print(f"m={m:7d} K={K:6d} mean={rec['mean']:6.1f} a={alpha} s={sigma} n={n:3d} {rec['layout']} mass={rec['l2_mass']:.2f}~{rec['l2_mass_profile']:.2f}/{rec['l2_byte_share']:.2f} cv={rec['row_cv']}: "
                  + "  ".join(f"{k}={v}" for k, v in rec["ms"].items())
                  + f"  | auto: {rec['kernels']['auto_kept_plan']} kept/best={rec['auto_kept_over_best']} ({rec['best_kept']}) "
                    f"one-shot/best={rec['auto_one_shot_over_best']} ({rec['best_one_shot']}) model={rec['model_us']}", flush=True)
        torch.cuda.empty_cache()
    doc["seconds"] = round(time.time() - t0, 1)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"[zipf_map] {len(doc['spmm'])} points in {doc['seconds']} s -> {args.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
