#!/usr/bin/env python3
"""How the SpMM kernels behave on log-normal row lengths (run on the GPU box): python tools/skew_probe.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from matrixextra_amd import _lib, device as D, synth

lib = _lib.load()
m, K, n = 400_000, 100_000, 128
B = torch.from_numpy(synth.dense_normal(K, n)).cuda()
for sigma in (0.0, 0.5, 1.0, 1.5):
    if sigma == 0.0:
        p, j, x = synth.csr_fixed(m, K, 32)
    else:
        rng = np.random.default_rng(1)
        mu = np.log(32) - 0.5 * sigma * sigma
        lens = np.minimum(np.floor(rng.lognormal(mu, sigma, size=m)).astype(np.int64), 4000)
        p = np.zeros(m + 1, dtype=np.int64); np.cumsum(lens, out=p[1:])
        j = rng.integers(0, K, size=p[-1], dtype=np.int32)          # unsorted, duplicates possible: fine for SpMM
        x = rng.uniform(-1, 1, size=p[-1]); p = p.astype(np.int32)
    A = D.DeviceCSR.from_host(p, j, x, K)
    C = torch.empty((n, m), dtype=torch.float64, device="cuda")
    res = {}
    for name, fn in (("auto", lambda: D.spmm(A, B, out=C, colmajor=True, algo=0)),
                     ("rowwave", lambda: D.spmm(A, B, out=C, colmajor=True, algo=1)),
                     ("planned(cached)", lambda: D.spmm_planned(A, B, out=C, colmajor=True)),
                     ("planned(cached, XCD barrier)", lambda: D.spmm_planned(A, B, out=C, colmajor=True, sync_mode=2))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 5 * 1e3
        if name == "auto":
            res["auto_kernel"] = lib.mxd_spmm_last_kernel().decode()
    info = A.plan_info()
    print(f"sigma {sigma}: nnz {A.nnz}, plan slots/nnz {info['padded_entries'] / max(A.nnz, 1):.2f}, "
          + ", ".join(f"{k} {v if isinstance(v, str) else round(v, 3)}" for k, v in res.items()), flush=True)
