"""Geometries of the LDS-tile kernel (csrc/spmm_tile.hip) on one shape (default: the vignette's dense x CSC product),
against the row-split kernel: python tools/tile_sweep.py [m K nnz_per_row n [col]]"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matrixextra_amd import device as D, synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from auto_map import timeit  # noqa: E402
m, K, npr, n = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (10_000, 10_000, 500, 100)
colmajor = len(sys.argv) > 5 and sys.argv[5] == "col"
p, j, x = synth.device_csr_fixed(m, K, npr, seed=7)
A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
A.rows_sorted()
B = torch.randn((K, n), dtype=torch.float64, device="cuda")
out = torch.empty((n, m) if colmajor else (m, n), dtype=torch.float64, device="cuda")
ref = D.spmm(A, B, colmajor=colmajor, algo=4, npanels=1, wg_per_cu=1).clone()
f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=0, keep_plan=False)
print(f"auto: {min(timeit(f), timeit(f, warm=0)):.4f} ms")
f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4)
print(f"row-split: {min(timeit(f), timeit(f, warm=0)):.4f} ms")
f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5)
t = min(timeit(f), timeit(f, warm=0))
err = float((D.spmm(A, B, colmajor=colmajor, algo=5) - ref).abs().max() / ref.abs().max())
print(f"tile (geometry chosen by the library): {t:.4f} ms   max err vs row-split {err:.2e}")
for small in (0, 1):
    for cpl in (1,):
        for rg in (2, 3, 4, 5):
            row = []
            for nw in (8, 10, 12, 13, 14, 15):
                if colmajor and small and cpl == 2 and nw * 4 * rg > 128:
                    row.append(float("nan"))
                    continue
                f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=5, npanels=nw, wg_per_cu=cpl + 4 * rg + 32 * small)
                row.append(min(timeit(f), timeit(f, warm=0)))
            print(f"tile {'32K' if small else '64K'} cpl={cpl} rg={rg}: " + "  ".join(f"nw={w}: {t:.4f}" for w, t in zip((8, 10, 12, 13, 14, 15), row)))
