"""The planned sweep on rows of uneven length: sync mode 1 (the wavefronts of a CU meet at the panel boundaries) against 2
(+ one XCD-wide timing barrier per generation) and the default (-1: chosen from the coefficient of variation of the
plan's octet lengths).  cfg2's shape (f64, n = 128, C column-major) with log-normal row lengths of growing sigma, the
Zipf-column variant, and the cfg5 shard shape (f32, n = 256)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matrixextra_amd import device as D, synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from auto_map import timeit  # noqa: E402


def run(tag, A, K, n, dt):
    B = torch.randn((K, n), dtype=dt, device="cuda")
    out = torch.empty((n, A.m), dtype=dt, device="cuda")
    row = []
    for sm in (-1, 1, 2):
        f = lambda: D.spmm_planned(A, B, out=out, colmajor=True, sync_mode=sm)
        f()
        row.append(min(timeit(f), timeit(f, warm=0)))
    print(f"{tag}: default {row[0]:.4f}  sync1 {row[1]:.4f}  sync2 {row[2]:.4f}  plan {A.plan_info()}", flush=True)


def host(m, K, mean, sigma, seed=1):
    hp, hj, hx = synth.csr_skewed_fast(m, K, mean, sigma=sigma, seed=seed)
    p, j, x = (torch.from_numpy(a).cuda() for a in (hp, hj, hx))
    return D.DeviceCSR(p, j, x, m, K, int(j.numel()))


m, K = 1_000_000, 100_000
ONLY_F32 = len(sys.argv) > 1 and sys.argv[1] == "f32"
if ONLY_F32:
    K5 = 200_000
    for sigma in (0.5, 1.0, 1.3):
        run(f"cfg5 shard log-normal sigma {sigma}", host(m, K5, 64, sigma), K5, 256, torch.float32)
    sys.exit(0)
p, j, x = synth.device_csr_fixed(m, K, 32, seed=1)
run("cfg2 equal rows", D.DeviceCSR(p, j, x, m, K, int(j.numel())), K, 128, torch.float64)
for sigma in (0.3, 0.5, 0.7, 0.85, 1.0, 1.3):
    run(f"cfg2 log-normal sigma {sigma}", host(m, K, 32, sigma), K, 128, torch.float64)
p, j, x = synth.device_csr_zipf(m, K, 40, alpha=1.0, sigma=1.0, seed=1)
run("cfg2 zipf columns, sigma 1", D.DeviceCSR(p, j, x, m, K, int(j.numel())), K, 128, torch.float64)
p, j, x = synth.device_csr_zipf(m, K, 40, alpha=1.0, sigma=0.5, seed=1)
run("cfg2 zipf columns, sigma 0.5", D.DeviceCSR(p, j, x, m, K, int(j.numel())), K, 128, torch.float64)
K5 = 200_000
p, j, x = synth.device_csr_fixed(m, K5, 64, seed=1)
run("cfg5 shard equal rows", D.DeviceCSR(p, j, x, m, K5, int(j.numel())), K5, 256, torch.float32)
for sigma in (0.5, 1.0):
    run(f"cfg5 shard log-normal sigma {sigma}", host(m, K5, 64, sigma), K5, 256, torch.float32)
