"""Run-to-run bitwise reproducibility of the planned SpMM kernel on matrices with log-normal row lengths (dealt octets:
rows shared by several lane groups of ONE wavefront, folded with LDS atomics)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matrixextra_amd import device as D, synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tot_diff = 0
for (m, K, mean, sigma, n, dt, seed) in [(300_000, 120_000, 20, 1.4, 128, torch.float64, 11), (1_000_000, 100_000, 32, 1.0, 128, torch.float64, 1),
                                         (200_000, 50_000, 40, 1.8, 64, torch.float32, 5), (400_000, 100_000, 16, 1.2, 256, torch.float32, 6)]:
    p, j, x = synth.csr_skewed_fast(m, K, mean, seed=seed, sigma=sigma)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.randn((K, n), dtype=dt, device="cuda")
    for colmajor in (False, True):
        ref = D.spmm_planned(A, B, colmajor=colmajor).clone()
        info = A.plan_info()
        nd = 0
        for r in range(runs):
            other = torch.randn((4096, 4096), device="cuda") @ torch.randn((4096, 4096), device="cuda") if r % 3 == 0 else None   # perturb timing
            got = D.spmm_planned(A, B, colmajor=colmajor, rebuild_plan=(r % 2 == 1))
            nd += int((got.view(torch.int64 if dt == torch.float64 else torch.int32) != ref.view(torch.int64 if dt == torch.float64 else torch.int32)).sum().item())
        print(f"m={m} K={K} mean={mean} sigma={sigma} n={n} {dt} colmajor={colmajor}: plan {info}; differing elements over {runs} runs: {nd}", flush=True)
        tot_diff += nd
print("TOTAL differing:", tot_diff)
