"""Where the LDS-tile kernel (csrc/spmm_tile.hip) pays: dense-ish CSR operands over density x m x K x n, both layouts of C,
against the row-split kernel (AUTO's panels / segments), the planned sweep with its plan kept, and AUTO itself.

  python tools/tile_map.py [--out gpurun_out/tile_map.json] [--quick]

Device time per product from HIP events (best of two rounds of 10 after a warm-up), operands resident, rows sorted by
column (the sortedness is cached on the DeviceCSR, as the exports do)."""
from __future__ import annotations

import argparse
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


def point(m, K, npr, n, colmajor, dtype, lib):
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=11)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
    A.rows_sorted()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    B = torch.randn((K, n), dtype=dtype, device="cuda", generator=g)
    out = torch.empty((n, m) if colmajor else (m, n), dtype=dtype, device="cuda")
    ms, kern = {}, {}

    def run(name, fn):
        try:
            fn()
            kern[name] = lib.mxd_spmm_last_kernel().decode()
            timeit(fn, reps=30)
            ms[name] = round(min(timeit(fn), timeit(fn, warm=0)), 5)
        except _lib.MxError as e:
            ms[name] = None
            kern[name] = "n/a: " + str(e)[:60]

    run("auto_kept_plan", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=True))
    run("auto_one_shot", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=False))
    run("rowsplit", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4))
    if m >= 32768:
        run("planned_kept", lambda: D.spmm_planned(A, B, out=out, colmajor=colmajor))
    run("tile", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5))
    run("tile_cpl1", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5, wg_per_cu=1))
    run("tile_cpl2", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5, wg_per_cu=2))
    run("tile_32k", lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5, wg_per_cu=1 + 32))
    others = {k: v for k, v in ms.items() if v is not None and k in ("rowsplit", "planned_kept")}
    tiles = {k: v for k, v in ms.items() if v is not None and k.startswith("tile")}
    rec = {"m": m, "K": K, "per_row": npr, "density": round(npr / K, 5), "n": n, "layout": "col" if colmajor else "row",
           "dtype": "f64" if dtype == torch.float64 else "f32", "ms": ms, "kernels": {k: kern[k] for k in ("auto_kept_plan", "auto_one_shot")},
           "best_other": min(others, key=others.get), "best_tile": min(tiles, key=tiles.get) if tiles else None}
    if tiles:
        rec["other_over_tile"] = round(others[rec["best_other"]] / ms["tile"], 3)
        rec["auto_over_best"] = round(ms["auto_kept_plan"] / min(min(others.values()), min(tiles.values())), 3)
    del A, B, out, p, j, x
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tile_map.json"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    lib = _lib.load()
    t0 = time.time()
    doc = {"device": _lib.device_name(), "spmm": []}
    shapes = []
    for K in (1_000, 10_000):
        for d in (0.01, 0.05, 0.2, 0.4):
            for m in (1_000, 10_000, 100_000):
                if m * K * d > 1.3e8:
                    continue
                for n in ((16, 100, 256) if not args.quick else (100,)):
                    shapes.append((m, K, max(1, int(K * d)), n))
    # sparser operands: where the row-split kernel / the planned sweep must stay
    shapes += [(10_000, 10_000, 32, 128), (100_000, 10_000, 32, 128), (100_000, 10_000, 128, 128), (10_000, 100_000, 500, 100),
               (100_000, 100_000, 128, 128), (1_000_000, 10_000, 32, 64), (1_000_000, 100_000, 32, 128)]
    for (m, K, npr, n) in shapes:
        for colmajor in (False, True):
            rec = point(m, K, npr, n, colmajor, torch.float64, lib)
            doc["spmm"].append(rec)
            print(f"m={m:7d} K={K:6d} d={rec['density']:.4f} n={n:3d} {rec['layout']}: " +
                  "  ".join(f"{k}={v}" for k, v in rec["ms"].items()) + f"  other/tile={rec.get('other_over_tile')}", flush=True)
        torch.cuda.empty_cache()
    for (m, K, npr, n) in ((10_000, 10_000, 500, 128), (10_000, 10_000, 500, 256), (100_000, 10_000, 500, 64), (1_000, 1_000, 400, 20)):
        for colmajor in (False, True):
            rec = point(m, K, npr, n, colmajor, torch.float32, lib)
            doc["spmm"].append(rec)
            print(f"f32 m={m:7d} K={K:6d} d={rec['density']:.4f} n={n:3d} {rec['layout']}: " +
                  "  ".join(f"{k}={v}" for k, v in rec["ms"].items()) + f"  other/tile={rec.get('other_over_tile')}", flush=True)
    doc["seconds"] = round(time.time() - t0, 1)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"[tile_map] {len(doc['spmm'])} points in {doc['seconds']} s -> {args.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
