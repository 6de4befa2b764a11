"""SpMV kernels (mx_spmv_algo: 0 auto, 1 lane groups, 2 LDS-panel tile, 3 flat) on rows of uneven length."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D  # noqa: E402
from auto_map import timeit  # noqa: E402
from cliff_hunt import lens_of, build  # noqa: E402
for (m, K, mean) in [(10_000, 10_000, 500), (100_000, 10_000, 64), (30_000, 100_000, 200), (1_000_000, 100_000, 32)]:
    for kind in ("equal", "lognormal_1.0", "lognormal_1.5", "giant", "blocks"):
        A = build(m, K, lens_of(kind, m, mean, np.random.default_rng(7)), 7)
        v = torch.randn(K, dtype=torch.float64, device="cuda")
        y = torch.empty(m, dtype=torch.float64, device="cuda")
        out = []
        for algo in (0, 1, 2, 3):
            try:
                f = lambda: D.spmv(A, v, out=y, algo=algo)
                f(); f()
                out.append(f"{algo}:{min(timeit(f), timeit(f, warm=0)):.4f}")
            except Exception as exc:  # noqa: BLE001
                out.append(f"{algo}:n/a")
        print(f"{m}x{K} {mean}/row {kind:14s} " + "  ".join(out), flush=True)
        del A
