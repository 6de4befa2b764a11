#!/usr/bin/env python3
"""Long rows cut into parts inside the tile kernel (MXGPU_TILE_SPLIT=1) against whole rows (=0): 1e4 x 1e4, 500 per row, n = 100,
f64 and f32, both layouts; ms per call and the largest relative difference of the two results."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from matrixextra_amd import _lib, device as D
from auto_map import timeit
from cliff_hunt import build, lens_of

lib = _lib.load()
out = []
shapes = [(10_000, 10_000, 500, 100), (30_000, 5_000, 300, 64), (4_000, 50_000, 2_000, 128), (20_000, 20_000, 300, 128), (30_000, 30_000, 200, 64)]
if os.environ.get("PROBE_PICK"):
    shapes = [shapes[int(i)] for i in os.environ["PROBE_PICK"].split(",")]
if os.environ.get("PROBE_SHAPES"):
    shapes = shapes[:int(os.environ["PROBE_SHAPES"])]
os.environ["MXGPU_TILE_DEAL"] = "1"
for (m, K, mean, n) in shapes:
    for kind in os.environ.get("PROBE_KINDS", "equal,lognormal_1.0,lognormal_1.5,giant,blocks").split(","):
        A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
        for dt in (torch.float64, torch.float32):
            B = torch.randn((K, n), dtype=dt, device="cuda")
            for colmajor in (False, True):
                C = torch.empty((n, m) if colmajor else (m, n), dtype=dt, device="cuda")
                r = {"m": m, "K": K, "mean": mean, "n": n, "rows": kind, "dtype": str(dt)[6:], "colmajor": colmajor, "nnz": A.nnz}
                ref = None
                for name, env in (("whole", "0"), ("split", "1"), ("auto_split", None)):
                    if env is None:
                        os.environ.pop("MXGPU_TILE_SPLIT", None)
                    else:
                        os.environ["MXGPU_TILE_SPLIT"] = env
                    f = lambda: D.spmm(A, B, out=C, colmajor=colmajor, algo=5)
                    f(); f()
                    r[name] = round(min(timeit(f), timeit(f, warm=0)), 4)
                    got = C.clone()
                    if ref is None:
                        ref = got
                    else:
                        r[name + "_maxrel"] = float(((got - ref).abs().max() / ref.abs().max()).item())
                print(r, flush=True)
                out.append(r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tile_split_probe.json"), "w"), indent=1)
