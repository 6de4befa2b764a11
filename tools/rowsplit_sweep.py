"""Segments x column panels of the row-split kernel on one shape (default: the vignette's dense x CSC product)."""
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
B = torch.randn((K, n), dtype=torch.float64, device="cuda")
out = torch.empty((n, m) if colmajor else (m, n), dtype=torch.float64, device="cuda")
for P in (1, 2, 3, 4, 5, 6, 8, 12):
    row = []
    for S in (1, 2, 4, 8):
        f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S)
        row.append(min(timeit(f), timeit(f, warm=0)))
    print(f"P={P:2d}: " + "  ".join(f"S={S}: {t:.4f}" for S, t in zip((1, 2, 4, 8), row)))
