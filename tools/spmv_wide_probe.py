#!/usr/bin/env python3
"""Planned SpMV on shapes round 5 could not plan or planned badly, beside the one-shot flat kernel:
  * BASELINE configs[3]'s own shape 2M x 2M, 50 / row (v = 16 MB > an XCD's L2): flat 1.49 ms = 0.10 of the roofline in round 5;
  * cfg3 (1M x 100k, 32 / row) with equal rows, log-normal rows, and rows SORTED by length (12.5x cliff in round 5).
python3 tools/spmv_wide_probe.py  -> one JSON line (gpurun_out/spmv_wide_probe.json)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from matrixextra_amd import device as D, synth


def timeit(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def case(name, A, K, lens=None):
    v = torch.randn(K, dtype=torch.float64, device="cuda")
    y_flat = D.spmv(A, v, algo=3)
    y_plan = D.spmv_planned(A, v)
    scale = float(y_flat.abs().max())
    err = float((y_flat - y_plan).abs().max()) / max(scale, 1e-300)
    byts = 4 * (A.m + 1) + 12 * A.nnz + 8 * K + 8 * A.m
    t_flat, t_auto, t_plan = timeit(lambda: D.spmv(A, v, algo=3)), timeit(lambda: D.spmv(A, v)), timeit(lambda: D.spmv_planned(A, v))
    r = {"m": A.m, "K": K, "nnz": A.nnz, "algorithmic_MB": round(byts / 1e6, 1), "flat_ms": round(t_flat, 4), "auto_ms": round(t_auto, 4),
         "planned_ms": round(t_plan, 4), "planned_frac_of_8TBps": round(byts / (t_plan * 1e-3) / 8e12, 4),
         "flat_frac_of_8TBps": round(byts / (t_flat * 1e-3) / 8e12, 4), "planned_vs_flat_max_err": err}
    if lens is not None:
        r["longest_row"] = int(lens.max())
    print(name, r, flush=True)
    return r


out = {}
m, K, k = 2_000_000, 2_000_000, 50
_, j, x = synth.device_csr_fixed(m, K, k, seed=1)
p = (torch.arange(m + 1, dtype=torch.int64, device="cuda") * k).to(torch.int32)
out["cfg4_shape_2Mx2M_50"] = case("2M x 2M, 50/row", D.DeviceCSR(p, j, x, m, K, int(j.numel())), K)
del p, j, x
torch.cuda.empty_cache()
m, K = 1_000_000, 100_000
p, j, x = synth.csr_fixed(m, K, 32)
out["cfg3_equal"] = case("cfg3 equal rows", D.DeviceCSR.from_host(p, j, x, K), K)
p, j, x = synth.csr_skewed_fast(m, K, 32, seed=3, sigma=1.0) if hasattr(synth, "csr_skewed_fast") else synth.csr_skewed(200_000, K, 32, seed=3)
lens = np.diff(p)
out["cfg3_lognormal"] = case("cfg3 log-normal sigma 1", D.DeviceCSR.from_host(p, j, x, K), K, lens)
order = np.argsort(-lens, kind="stable")                       # rows sorted by length, longest first
ps = np.zeros(lens.size + 1, dtype=np.int64)
np.cumsum(lens[order], out=ps[1:])
idx = np.concatenate([np.arange(p[r], p[r + 1]) for r in order[:0]]) if False else None
starts = p[:-1][order].astype(np.int64)
take = np.repeat(starts - ps[:-1], lens[order]) + np.arange(ps[-1])
out["cfg3_sorted_by_length"] = case("cfg3 log-normal, rows sorted by length", D.DeviceCSR.from_host(ps.astype(np.int32), j[take], x[take], K), K, lens)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/spmv_wide_probe.json", "w"), indent=1)
print(json.dumps(out))
