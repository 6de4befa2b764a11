#!/usr/bin/env python3
"""The dense-ish product with rows of uneven length: 1e4 x 1e4, 500 per row, n = 100 — equal rows,
log-normal sigma 1 / 1.5, four giant rows, rows sorted by length — on the tile kernel with consecutive rows (MXGPU_TILE_DEAL=0),
with its rows dealt by length (=1), the row-split kernel, and AUTO (plan / profile kept).  ms per call, f64, both layouts."""
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
shapes = [(10_000, 10_000, 500, 100), (30_000, 5_000, 300, 64)]
for (m, K, mean, n) in shapes:
    for kind in ("equal", "lognormal_1.0", "lognormal_1.5", "giant", "blocks", "half_empty"):
        A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
        B = torch.randn((K, n), dtype=torch.float64, device="cuda")
        for colmajor in (False, True):
            C = torch.empty((n, m) if colmajor else (m, n), dtype=torch.float64, device="cuda")
            r = {"m": m, "K": K, "mean": mean, "n": n, "rows": kind, "colmajor": colmajor, "nnz": A.nnz, "cv": round(float(A.profile()[32]), 3)}
            ref = None
            for name, algo, env in (("tile_consecutive", 5, "0"), ("tile_dealt", 5, "1"), ("row_split", 4, None), ("auto", 0, None)):
                if env is None:
                    os.environ.pop("MXGPU_TILE_DEAL", None)
                else:
                    os.environ["MXGPU_TILE_DEAL"] = env
                f = lambda: D.spmm(A, B, out=C, colmajor=colmajor, algo=algo)
                f(); f()
                r[name] = round(min(timeit(f), timeit(f, warm=0)), 4)
                if name == "auto":
                    r["auto_kernel"] = lib.mxd_spmm_last_kernel().decode()
                got = C.clone()
                if ref is None:
                    ref = got
                elif name == "tile_dealt":
                    r["dealt_bitwise_equal_to_consecutive"] = bool(torch.equal(got, ref))
            os.environ.pop("MXGPU_TILE_DEAL", None)
            print(r, flush=True)
            out.append(r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tile_skew_probe.json"), "w"), indent=1)
