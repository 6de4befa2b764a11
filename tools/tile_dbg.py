import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from matrixextra_amd import device as D, synth
from auto_map import timeit
m, K, npr, n = 10_000, 10_000, 500, 100
p, j, x = synth.device_csr_fixed(m, K, npr, seed=7)
A = D.DeviceCSR(p, j, x, m, K, int(j.numel())); A.rows_sorted()
B = torch.randn((K, n), dtype=torch.float64, device="cuda")
out = torch.empty((m, n), dtype=torch.float64, device="cuda")
f0 = lambda: D.spmm(A, B, out=out, algo=5)
timeit(f0, reps=400)
for dbg in (0, 1, 0, 1):
    os.environ["MXGPU_TILE_DEBUG"] = str(dbg)
    for (rg, nw, one) in ((4, 10, 0), (4, 10, 1), (3, 14, 0), (3, 14, 1), (3, 13, 0), (5, 8, 0), (5, 8, 1)):
        f = lambda: D.spmm(A, B, out=out, algo=5, npanels=nw, wg_per_cu=1 + 4 * rg + 64 * one)
        print(f"dbg={dbg} rg={rg} nw={nw} loaders={2 - one}: {min(timeit(f, reps=30), timeit(f, warm=0, reps=30)):.4f} ms")
