"""Tile kernel against the row-split kernel (with its long-rows path) on dense-ish operands with rows of uneven length."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D, _lib  # noqa: E402
from auto_map import timeit  # noqa: E402
from cliff_hunt import lens_of, build  # noqa: E402
lib = _lib.load()
for (m, K, mean, n) in [(10_000, 10_000, 500, 100), (10_000, 10_000, 2000, 64), (30_000, 5_000, 250, 256), (3_000, 3_000, 600, 128)]:
    for dt in (torch.float64, torch.float32):
        for kind in ("equal", "lognormal_0.5", "lognormal_1.0", "lognormal_1.5", "half_empty", "giant", "blocks"):
            rng = np.random.default_rng(7)
            A = build(m, K, lens_of(kind, m, mean, rng), 7)
            B = torch.randn((K, n), dtype=dt, device="cuda")
            C = torch.empty((m, n), dtype=dt, device="cuda")
            res = {}
            for name, algo in (("auto", 0), ("rowsplit", 4), ("tile", 5)):
                f = lambda: D.spmm(A, B, out=C, colmajor=False, algo=algo)
                try:
                    f(); f()
                    res[name] = min(timeit(f), timeit(f, warm=0))
                    if algo == 0:
                        res["pick"] = lib.mxd_spmm_last_kernel().decode()
                except Exception as exc:  # noqa: BLE001
                    res[name] = float("nan")
            print(f"{m}x{K} {mean}/row n={n} {str(dt)[6:]} {kind:14s} auto {res['auto']:.4f} ({res['pick']})  rowsplit {res['rowsplit']:.4f}  tile {res['tile']:.4f}", flush=True)
            del A, B, C
