#!/usr/bin/env python3
"""Every SpMM kernel family on the skewed points of tools/cliff_hunt.py that stay above 1.6x of the equal-rows rate: which
kernel SHOULD AUTO take there?  ms per call, plan / profile kept."""
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
for (m, K, mean, n) in ((100_000, 10_000, 64, 64), (200_000, 50_000, 100, 32), (1_000_000, 10_000, 12, 16)):
    for kind in ("equal", "lognormal_1.0", "lognormal_1.5", "blocks", "giant"):
        A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
        for dt in (torch.float64, torch.float32):
            B = torch.randn((K, n), dtype=dt, device="cuda")
            C = torch.empty((m, n), dtype=dt, device="cuda")
            r = {}
            for name, f in (("auto", lambda: D.spmm(A, B, out=C)), ("planned_kept", lambda: D.spmm_planned(A, B, out=C)),
                            ("rowsplit", lambda: D.spmm(A, B, out=C, algo=4)), ("tile", lambda: D.spmm(A, B, out=C, algo=5)),
                            ("rowwave", lambda: D.spmm(A, B, out=C, algo=1))):
                try:
                    f(); f()
                    r[name] = round(min(timeit(f), timeit(f, warm=0)), 4)
                    if name == "auto":
                        r["auto_kernel"] = lib.mxd_spmm_last_kernel().decode()[5:-7]
                except Exception as exc:  # noqa: BLE001
                    r[name] = str(exc)[:40]
            print(f"{m}x{K} {mean}/row n={n} {str(dt)[6:]:8s} {kind:14s} {r}", flush=True)
        del A
        torch.cuda.empty_cache()
