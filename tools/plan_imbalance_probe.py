#!/usr/bin/env python3
"""mxd_spmm_plan_imbalance against what the planned sweep and the row-split kernel really take, on the skewed matrices of
tools/cliff_hunt.py: where should AUTO leave a plan for the row-split kernel?"""
import ctypes as C
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
for (m, K, mean, n) in ((100_000, 10_000, 64, 64), (200_000, 50_000, 100, 32), (1_000_000, 100_000, 32, 128), (1_000_000, 10_000, 12, 16), (300_000, 100_000, 48, 64)):
    for kind in ("lognormal_1.0", "lognormal_1.5", "blocks", "giant"):
        A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
        for dt, code in ((torch.float64, _lib.MX_F64), (torch.float32, _lib.MX_F32)):
            B = torch.randn((K, n), dtype=dt, device="cuda")
            Cc = torch.empty((m, n), dtype=dt, device="cuda")
            plan = A.auto_plan(0)
            imb = C.c_double(-1.0)
            if plan is not None:
                _lib.check(lib.mxd_spmm_plan_imbalance(plan, C.c_int(n), C.c_int(code), C.byref(imb)))
            r = {"imbalance": round(imb.value, 2), "octet_cv": A.plan_info()["octet_length_cv"] if plan is not None else None}
            for name, f in (("planned_kept", lambda: D.spmm_planned(A, B, out=Cc)), ("rowsplit", lambda: D.spmm(A, B, out=Cc, algo=4)), ("auto", lambda: D.spmm(A, B, out=Cc))):
                f(); f()
                r[name] = round(min(timeit(f), timeit(f, warm=0)), 4)
            r["auto_kernel"] = lib.mxd_spmm_last_kernel().decode()[5:-7]
            print(f"{m}x{K} {mean}/row n={n} {str(dt)[6:]:8s} {kind:14s} {r}", flush=True)
        del A
        torch.cuda.empty_cache()
