#!/usr/bin/env python3
"""Would cutting long rows into interleaved parts pay in the tile kernel?  Emulation without touching the kernel: every row
longer than L entries is replaced by P rows holding its entries p, p + P, p + 2P ... (sorted columns stay sorted), the dealt
tile kernel runs on the longer matrix (the partial rows would still have to be added up: not timed).  1e4 x 1e4, 500 per row,
n = 100, the row-length distributions of tools/cliff_hunt.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from matrixextra_amd import _lib, device as D
from auto_map import timeit
from cliff_hunt import build, lens_of

m, K, mean, n = 10_000, 10_000, 500, 100
for dt in (torch.float64, torch.float32):
    B = torch.randn((K, n), dtype=dt, device="cuda")
    for kind in ("lognormal_1.0", "lognormal_1.5", "giant", "blocks"):
        A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
        p, j, x = A.indptr.cpu().numpy().astype(np.int64), A.indices.cpu().numpy(), A.values.cpu().numpy()
        lens = np.diff(p)
        C = torch.empty((m, n), dtype=dt, device="cuda")
        f = lambda: D.spmm(A, B, out=C, algo=5)
        f(); f()
        r = {"whole_rows": round(min(timeit(f), timeit(f, warm=0)), 4)}
        for L in (2560, 1280, 640):
            newp, order = [0], []
            for row in range(m):
                Ln = int(lens[row])
                P = 1 if Ln <= L else min(16, -(-Ln // L))
                idx = np.arange(p[row], p[row + 1])
                for q in range(P):
                    part = idx[q::P]
                    order.append(part)
                    newp.append(newp[-1] + part.size)
            order = np.concatenate(order)
            A2 = D.DeviceCSR.from_host(np.asarray(newp, dtype=np.int32), j[order], x[order], K)
            C2 = torch.empty((A2.m, n), dtype=dt, device="cuda")
            g = lambda: D.spmm(A2, B, out=C2, algo=5)
            g(); g()
            r[f"parts_of_{L}"] = (round(min(timeit(g), timeit(g, warm=0)), 4), A2.m, int(np.diff(np.asarray(newp)).max()))
            del A2, C2
        print(str(dt)[6:], kind, r, flush=True)
        del A
