"""A/B of the row-split kernel's column panels: ONE launch (panel-major grid, agent-scope hand-over of C between the
panels) against one launch per panel (MXGPU_ROWSPLIT_LAUNCHES=1, rounds 3-4).  Same bits asked of both; ms per call."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matrixextra_amd import device as D, synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from auto_map import timeit  # noqa: E402

SHAPES = [  # m, K, per row, n, dtype, colmajor, segments (wg_per_cu: -1 = row groups), panels
    (10_000, 10_000, 500, 100, torch.float64, False, 1, 3),
    (10_000, 10_000, 500, 100, torch.float64, True, 1, 3),
    (10_000, 100_000, 500, 100, torch.float64, False, 1, 15),
    (100_000, 10_000, 128, 128, torch.float64, False, 1, 4),
    (100_000, 100_000, 128, 128, torch.float64, True, 1, 4),
    (100_000, 100_000, 128, 256, torch.float32, False, 1, 8),
    (1_000_000, 400_000, 32, 16, torch.float64, False, -1, 4),
    (1_000_000, 400_000, 32, 16, torch.float64, True, -1, 4),
    (3_000, 50_000, 4000, 64, torch.float64, False, 4, 6),
]
for m, K, npr, n, dt, colmajor, S, P in SHAPES:
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=11)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
    B = torch.randn((K, n), dtype=dt, device="cuda")
    outs, times = [], []
    for per_panel in ("0", "1"):
        os.environ["MXGPU_ROWSPLIT_LAUNCHES"] = per_panel
        out = torch.full((n, m) if colmajor else (m, n), float("nan"), dtype=dt, device="cuda")
        f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S)
        f(); torch.cuda.synchronize()
        outs.append(out.clone())
        times.append(min(timeit(f), timeit(f, warm=0)))
    del os.environ["MXGPU_ROWSPLIT_LAUNCHES"]
    same = bool(torch.equal(outs[0], outs[1]))
    # repeated runs of the one-launch form give the same bits (the hand-over has no order of its own)
    again = torch.empty_like(outs[0]); stable = True
    os.environ["MXGPU_ROWSPLIT_LAUNCHES"] = "0"
    for _ in range(20):
        D.spmm(A, B, out=again, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S)
        stable &= bool(torch.equal(again, outs[0]))
    del os.environ["MXGPU_ROWSPLIT_LAUNCHES"]
    D.spmm(A, B, out=again, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S)
    default = min(timeit(lambda: D.spmm(A, B, out=again, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S)) for _ in range(2))
    print(f"m={m} K={K} {npr}/row n={n} {str(dt)[6:]} {'col' if colmajor else 'row'} S={S} P={P}: one launch {times[0]:.4f} ms, "
          f"{P} launches {times[1]:.4f} ms ({times[1] / times[0]:.3f}x), default {default:.4f}  same bits: {same}  stable: {stable}", flush=True)
