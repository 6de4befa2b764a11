"""A/B of the tile kernel at the vignette's shape: MXGPU_LIB=<other build> python tools/tile_dbg.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D, synth
from auto_map import timeit
shapes = [(10_000, 10_000, 500, 100), (10_000, 10_000, 2000, 256), (100_000, 10_000, 500, 100), (30_000, 5_000, 200, 64)]
for (m, K, npr, n) in shapes:
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=7)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel())); A.rows_sorted()
    B = torch.randn((K, n), dtype=torch.float64, device="cuda")
    out = torch.empty((m, n), dtype=torch.float64, device="cuda")
    f = lambda: D.spmm(A, B, out=out, algo=5)
    timeit(f, reps=300)
    print(f"m={m} K={K} per_row={npr} n={n}: tile {min(timeit(f, reps=30) for _ in range(4)):.4f} ms", flush=True)
    del A, B, out, p, j, x
