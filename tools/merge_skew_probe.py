#!/usr/bin/env python3
"""CSR + CSR / CSR * CSR on one row-length distribution at BASELINE configs[3]'s shape (2M x 2M, 50 per row), a few calls —
for per-kernel durations:  bash tools/prof_kernels.sh gpurun_out/merge_skew merge,scan tools/merge_skew_probe.py lognormal_1.0
Prints ms per call (events) as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from matrixextra_amd import _lib, device as D
from auto_map import timeit
from cliff_hunt import build, lens_of

kind = sys.argv[1] if len(sys.argv) > 1 else "lognormal_1.0"
m, K, mean = (int(os.environ.get("PROBE_M", 2_000_000)), int(os.environ.get("PROBE_K", 2_000_000)), int(os.environ.get("PROBE_MEAN", 50)))
A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
A2 = build(m, K, lens_of(kind, m, mean, np.random.default_rng(8)), 8)
for g in ([None] if not os.environ.get("PROBE_G_SWEEP") else [None, "16", "32", "64"]):
    if g is None:
        os.environ.pop("MXGPU_MERGE_G", None)
    else:
        os.environ["MXGPU_MERGE_G"] = g
    for name, op in (("add", _lib.MX_OP_ADD), ("mul", _lib.MX_OP_MUL)):
        f = lambda: D.csr_elemwise(op, A, A2)
        f(); f()
        print(kind, "G", g, name, "ms", round(timeit(f, reps=6), 4), "nnz", A.nnz, A2.nnz, flush=True)
