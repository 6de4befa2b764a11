"""Piece length of the row-split kernel's long-rows path (MXGPU_LONG_PIECE; 0 = off) on the shapes where tools/cliff_hunt.py
found a tail: ms per call of the row-split kernel (algo 4, AUTO's segments / panels) by piece length and row distribution."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D  # noqa: E402
from auto_map import timeit  # noqa: E402
from cliff_hunt import lens_of, build  # noqa: E402

SHAPES = [(100_000, 10_000, 64, 64), (10_000, 10_000, 500, 100), (200_000, 50_000, 100, 32), (1_000_000, 10_000, 12, 16)]
PIECES = [0, 128, 256, 512, 1024, 2048, 4096]
for (m, K, mean, n) in SHAPES:
    for dt in (torch.float64, torch.float32):
        for kind in ("equal", "lognormal_1.0", "lognormal_1.5", "giant", "blocks"):
            rng = np.random.default_rng(7)
            A = build(m, K, lens_of(kind, m, mean, rng), 7)
            A.profile()
            B = torch.randn((K, n), dtype=dt, device="cuda")
            C = torch.empty((m, n), dtype=dt, device="cuda")
            row = []
            for piece in PIECES:
                os.environ["MXGPU_LONG_PIECE"] = str(piece)
                f = lambda: D.spmm(A, B, out=C, colmajor=False, algo=4)
                f(); f()
                row.append(min(timeit(f), timeit(f, warm=0)))
            print(f"{m}x{K} {mean}/row n={n} {str(dt)[6:]} {kind:14s} " + "  ".join(f"{p}:{t:.4f}" for p, t in zip(PIECES, row)), flush=True)
            del A, B, C
