"""AUTO regime map: every SpMM kernel family timed over a grid of shapes between the benchmarks, so that
MX_SPMM_AUTO's rule (csrc/spmm.hip spmm_auto_algo) is what the measurements say and not a guess.

  python tools/auto_map.py [--quick] [--out profiles/r04_auto_map.json]

Grid: m in {1e4, 1e5, 1e6} x entries/row in {8, 32, 128, 500} x n in {16, 64, 100, 128, 256} x K in {1e4, 1e5}, f64, both
layouts of C (column-major = tcrossprod_csr_dense, what R's `%*%` takes; row-major = dense x CSC / dense x t(CSR)), plus a
few f32 points; shapes with more than 1.3e8 entries or a result above 2 GiB are skipped.  Per shape: the row-wave kernel,
the row-split kernel (AUTO's segment count), the slab kernel, the planned kernel with the plan kept / rebuilt per call, and
AUTO both ways (plan kept on the matrix = what the exports do for a cached operand; C-ABI AUTO = plan rebuilt).  Device
time per product from HIP events around 10 launches after 2 warm-ups, operands resident.

Also: SpMV kernels (lane-group / flat / tile / AUTO) over m x entries/row, and the gather's lane-group width is left to
bench.py's gather leg (one shape).  Writes one JSON document."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matrixextra_amd import _lib, device as D, synth  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def spmm_point(m, K, npr, n, colmajor, dtype, lib):
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=11)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    B = torch.randn((K, n), dtype=dtype, device="cuda", generator=g)
    out = torch.empty((n, m) if colmajor else (m, n), dtype=dtype, device="cuda")
    ms, kern = {}, {}

    def run(name, fn):
        try:
            fn()
            kern[name] = lib.mxd_spmm_last_kernel().decode()
            ms[name] = round(min(timeit(fn), timeit(fn, warm=0)), 5)     # best of two rounds of 10
        except _lib.MxError as e:                                 # operands a kernel does not take (alignment rules)
            ms[name] = None
            kern[name] = "n/a: " + str(e)[:60]

    dt = 0 if dtype == torch.float64 else 1
    pick, pick_kept = C.c_int(0), C.c_int(0)
    for keep, dst in ((0, pick), (1, pick_kept)):
        _lib.check(lib.mxd_spmm_auto_algo2(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(A.nnz), C.c_int(keep), C.c_int(dt),
                                           C.c_void_p(B.data_ptr()), C.c_size_t(n), C.c_void_p(out.data_ptr()),
                                           C.c_size_t(m if colmajor else n), C.c_int(int(colmajor)), C.byref(dst)))
    model = {}
    for keep in (0, 1):
        a, b, P = C.c_double(), C.c_double(), C.c_int()
        _lib.check(lib.mxd_spmm_auto_cost(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(A.nnz), C.c_int(keep), C.c_int(dt), C.byref(a),
                                          C.byref(b), C.byref(P)))
        model["kept" if keep else "one_shot"] = {"rowsplit_ms": round(a.value / 1e3, 4), "planned_ms": round(b.value / 1e3, 4), "panels": P.value}
    # AUTO first: nothing has been freed yet (a plan that goes back to the pool / the driver makes the next launches slower
    # for a while — pool.hip — and showed up as 1.3-2.4x "AUTO over best" artefacts in the first version of this map)
    run("auto_one_shot", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=False))
    run("auto_kept_plan", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=True))
    run("rowwave", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=1))
    run("rowsplit", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4))
    run("rowsplit_one_panel", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=1))
    # the two forms of the row-split kernel forced (one panel): one wavefront per row / several rows per wavefront — the
    # `rowsplit` leg above runs whichever rowsplit_segments() picks, these show whether it picked right
    run("rowsplit_wave_per_row", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=1, wg_per_cu=1))
    if n * B.element_size() <= 512:
        run("rowsplit_row_groups", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=1, wg_per_cu=-1))
    run("slab", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=2))
    if npr * 64 >= K:                      # the LDS-tile kernel (round 5): only where a row meets a 256-row K-tile more than ~4 times
        run("tile", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5))
    run("planned_kept", lambda: D.spmm_planned(A, B, out=out, colmajor=colmajor))
    run("planned_rebuilt", lambda: D.spmm_planned(A, B, out=out, colmajor=colmajor, rebuild_plan=True))
    forms = ("rowsplit", "rowsplit_one_panel", "rowsplit_wave_per_row", "rowsplit_row_groups")
    one_shot = {k: v for k, v in ms.items() if v is not None and k in ("rowwave", "slab", "planned_rebuilt", "tile") + forms}
    kept = {k: v for k, v in ms.items() if v is not None and k in ("rowwave", "slab", "planned_kept", "tile") + forms}
    best1, bestk = min(one_shot, key=one_shot.get), min(kept, key=kept.get)
    rec = {"m": m, "K": K, "per_row": npr, "n": n, "layout": "col" if colmajor else "row", "dtype": "f64" if dtype == torch.float64 else "f32",
           "ms": ms, "auto_family": {0: "auto", 1: "rowwave", 2: "slab", 3: "planned", 4: "rowsplit", 5: "tile"}[pick.value],
           "auto_family_kept_plan": {0: "auto", 1: "rowwave", 2: "slab", 3: "planned", 4: "rowsplit", 5: "tile"}[pick_kept.value],
           "model": model, "kernels": {k: kern[k] for k in ("auto_one_shot", "auto_kept_plan")},
           "best_one_shot": best1, "best_kept": bestk,
           "auto_one_shot_over_best": round(ms["auto_one_shot"] / one_shot[best1], 3),
           "auto_kept_over_best": round(ms["auto_kept_plan"] / kept[bestk], 3)}
    del A, B, out, p, j, x
    return rec


def spmv_point(m, K, npr, lib):
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=13)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
    v = torch.randn(K, dtype=torch.float64, device="cuda")
    y = torch.empty(m, dtype=torch.float64, device="cuda")
    ms = {}
    for name, algo in (("auto", 0), ("group", 1), ("tile", 2), ("flat", 3)):
        try:
            ms[name] = round(timeit(lambda: D.spmv(A, v, out=y, algo=algo), reps=20), 5)
        except _lib.MxError:
            ms[name] = None
    ms["planned_kept"] = round(timeit(lambda: D.spmv_planned(A, v, out=y), reps=20), 5)
    cand = {k: t for k, t in ms.items() if t is not None and k in ("group", "tile", "flat")}
    best = min(cand, key=cand.get)
    return {"m": m, "K": K, "per_row": npr, "ms": ms, "best_one_shot": best, "auto_over_best": round(ms["auto"] / cand[best], 3)}


def gather_point(m, K, npr, r, lib):
    """X[rows, ]: the one-launch gather with every lane-group width (the width follows the row-length hint: 4 .. 64 lanes
    per row) against the width AUTO derives from the matrix's mean row length"""
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=17)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    rows = torch.randint(0, m, (r,), dtype=torch.int32, device="cuda", generator=g)
    cap = r * npr
    new_p = torch.empty(r + 1, dtype=torch.int32, device="cuda")
    new_j = torch.empty(cap, dtype=torch.int32, device="cuda")
    new_x = torch.empty(cap, dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nnz_out = C.c_int64(0)

    def run(hint):
        _lib.check(lib.mxd_csr_gather_fused(C.c_int(r), C.c_void_p(p.data_ptr()), C.c_void_p(j.data_ptr()), C.c_void_p(x.data_ptr()),
                                            C.c_void_p(rows.data_ptr()), C.c_void_p(new_p.data_ptr()), C.c_void_p(new_j.data_ptr()),
                                            C.c_void_p(new_x.data_ptr()), C.c_int(_lib.MX_F64), C.c_int64(cap), C.c_double(hint), None,
                                            C.byref(nnz_out), st))
    ms = {"auto": round(min(timeit(lambda: run(float(npr)), reps=20), timeit(lambda: run(float(npr)), reps=20, warm=0)), 5)}
    for G in (4, 8, 16, 32, 64):
        ms[f"lanes_{G}"] = round(min(timeit(lambda: run(float(G)), reps=20), timeit(lambda: run(float(G)), reps=20, warm=0)), 5)
    best = min((k for k in ms if k != "auto"), key=ms.get)
    return {"m": m, "per_row": npr, "rows_taken": r, "ms": ms, "best": best, "auto_over_best": round(ms["auto"] / ms[best], 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="the reduced grid of tests/test_gpu_auto_map.py")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "auto_map.json"))
    ap.add_argument("--no-spmv", action="store_true", help="SpMM only (no SpMV / gather legs)")
    ap.add_argument("--only-gather", action="store_true")
    args = ap.parse_args()
    lib = _lib.load()
    t0 = time.time()
    doc = {"device": _lib.device_name(), "spmm": [], "spmv": [], "skipped": []}
    ms_, nprs, ns, Ks = (10_000, 100_000, 1_000_000), (8, 32, 128, 500), (16, 32, 64, 100, 128, 256), (10_000, 100_000)
    if args.quick:
        ms_, nprs, ns, Ks = (10_000, 100_000), (32, 500), (16, 100, 128), (10_000, 100_000)
    if args.only_gather:
        ms_ = ()
    for m in ms_:
        for npr in nprs:
            for K in Ks:
                if m * npr > 130_000_000:
                    doc["skipped"].append({"m": m, "per_row": npr, "K": K, "why": "more than 1.3e8 entries"})
                    continue
                for n in ns:
                    if m * n * 8 > 2 << 30:
                        doc["skipped"].append({"m": m, "per_row": npr, "K": K, "n": n, "why": "result above 2 GiB"})
                        continue
                    for colmajor in (True, False):
                        doc["spmm"].append(spmm_point(m, K, npr, n, colmajor, torch.float64, lib))
                torch.cuda.empty_cache()
            print(f"[auto_map] m={m} per_row={npr} done at {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    for (m, K, npr, n) in ((10_000, 10_000, 500, 100), (100_000, 100_000, 32, 256), (1_000_000, 100_000, 32, 256), (10_000, 100_000, 128, 64)):
        if args.quick and m > 100_000:
            continue
        for colmajor in (True, False):
            doc["spmm"].append(spmm_point(m, K, npr, n, colmajor, torch.float32, lib))
    if not args.no_spmv:
        for m in ms_:
            for npr in nprs:
                if m * npr > 130_000_000:
                    continue
                doc["spmv"].append(spmv_point(m, 100_000, npr, lib))
    doc["gather"] = []
    if not args.no_spmv:
        for npr in (4, 8, 32, 128, 500):
            for r in (10_000, 200_000, 1_000_000):
                if r * npr > 130_000_000:
                    continue
                doc["gather"].append(gather_point(1_000_000, 100_000, npr, r, lib))
    w1 = max(doc["spmm"], key=lambda r: r["auto_one_shot_over_best"])
    wk = max(doc["spmm"], key=lambda r: r["auto_kept_over_best"])
    # products of a few microseconds are timed through Python's ~10 us per call (the same kernel measures 0.011 and 0.021 ms
    # in two legs of one point): the ratios that say something about the CHOICE are those of products above 50 us
    long1 = [r for r in doc["spmm"] if min(v for k, v in r["ms"].items() if v is not None) >= 0.05]
    doc["summary"] = {
        "points": len(doc["spmm"]), "seconds": round(time.time() - t0, 1),
        "points_above_50us": len(long1),
        "above_50us_auto_one_shot_worst_over_best": max((r["auto_one_shot_over_best"] for r in long1), default=None),
        "above_50us_auto_kept_worst_over_best": max((r["auto_kept_over_best"] for r in long1), default=None),
        "above_50us_points_above_1.25": [sum(r["auto_one_shot_over_best"] > 1.25 for r in long1), sum(r["auto_kept_over_best"] > 1.25 for r in long1)],
        "auto_one_shot_worst_over_best": {"ratio": w1["auto_one_shot_over_best"], "at": {k: w1[k] for k in ("m", "K", "per_row", "n", "layout", "dtype")}},
        "auto_kept_worst_over_best": {"ratio": wk["auto_kept_over_best"], "at": {k: wk[k] for k in ("m", "K", "per_row", "n", "layout", "dtype")}},
        "points_above_1.25_one_shot": sum(r["auto_one_shot_over_best"] > 1.25 for r in doc["spmm"]),
        "points_above_1.25_kept": sum(r["auto_kept_over_best"] > 1.25 for r in doc["spmm"]),
        "spmv_worst_over_best": max((r["auto_over_best"] for r in doc["spmv"]), default=None),
        "gather_worst_over_best": max((r["auto_over_best"] for r in doc["gather"]), default=None)}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["summary"]))


if __name__ == "__main__":
    main()
