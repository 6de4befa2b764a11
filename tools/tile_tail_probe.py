#!/usr/bin/env python3
"""The ragged last slab of the tile kernel (n = 100: three full 256-byte slabs + 4 columns): what do the pieces cost?
tile kernel on all 100 columns, on the first 96 only, and the 4 tail columns on the row-split kernel's row-group form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from matrixextra_amd import _lib, device as D, synth
from auto_map import timeit

lib = _lib.load()
for (m, K, npr, n, nm) in ((10_000, 10_000, 500, 100, 96), (10_000, 10_000, 500, 72, 64), (30_000, 5_000, 200, 40, 32)):
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=7)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel())); A.rows_sorted()
    B = torch.randn((K, n), dtype=torch.float64, device="cuda")
    out = torch.empty((m, n), dtype=torch.float64, device="cuda")
    out_main = torch.empty((m, nm), dtype=torch.float64, device="cuda")
    out_tail = torch.empty((m, n - nm), dtype=torch.float64, device="cuda")
    Bm, Bt = B[:, :nm], B[:, nm:]
    full = lambda: D.spmm(A, B, out=out, algo=5)
    main = lambda: D.spmm(A, Bm, out=out_main, algo=5)
    tail_rg = lambda: D.spmm(A, Bt, out=out_tail, algo=4, wg_per_cu=-1, npanels=1)
    tail_rs = lambda: D.spmm(A, Bt, out=out_tail, algo=4, wg_per_cu=1, npanels=1)
    r = {}
    for name, f in (("tile_all", full), ("tile_main", main), ("tail_rowgroup", tail_rg), ("tail_rowsplit", tail_rs)):
        f(); f()
        r[name] = round(min(timeit(f, reps=30), timeit(f, reps=30, warm=0)), 4)
    both = lambda: (main(), tail_rg())
    both(); r["main_then_tail"] = round(min(timeit(both, reps=30), timeit(both, reps=30, warm=0)), 4)
    ok = torch.equal(out[:, :nm], out_main) and torch.equal(out[:, nm:], out_tail)
    print(f"m={m} K={K} per_row={npr} n={n} (main {nm}): {r}  bitwise_equal_to_tile_all={ok}", flush=True)
