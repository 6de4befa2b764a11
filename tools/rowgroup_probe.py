"""The row-group form of the row-split kernel (several rows per wavefront) against the other kernels on MANY SHORT rows x
narrow B; bitwise against the row-wave kernel (the storage-order FMA chain)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D, synth  # noqa: E402
from auto_map import timeit  # noqa: E402
shapes = [(1_000_000, 10_000, 8, 16), (1_000_000, 10_000, 32, 16), (1_000_000, 100_000, 8, 16), (1_000_000, 100_000, 32, 16),
          (1_000_000, 10_000, 8, 64), (1_000_000, 10_000, 32, 64), (1_000_000, 100_000, 32, 64), (100_000, 10_000, 32, 16),
          (100_000, 10_000, 128, 16), (100_000, 10_000, 128, 64), (100_000, 100_000, 32, 64), (1_000_000, 10_000, 128, 16),
          (1_000_000, 10_000, 32, 32), (300_000, 50_000, 64, 32), (100_000, 10_000, 500, 16), (1_000_000, 100_000, 16, 100)]
if len(sys.argv) > 4:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
for dt in (torch.float64, torch.float32):
    for (m, K, npr, n) in shapes:
        if dt == torch.float32 and n == 16 and npr > 32:
            continue
        p, j, x = synth.device_csr_fixed(m, K, npr, seed=7)
        A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
        B = torch.randn((K, n), dtype=dt, device="cuda")
        for colmajor in (False, True):
            out = torch.empty((n, m) if colmajor else (m, n), dtype=dt, device="cuda")
            ref = D.spmm(A, B, colmajor=colmajor, algo=1).clone()
            got = D.spmm(A, B, out=out, colmajor=colmajor, algo=4, npanels=1, wg_per_cu=-1)
            same = torch.equal(got.contiguous().view(torch.int64 if dt == torch.float64 else torch.int32),
                               ref.contiguous().view(torch.int64 if dt == torch.float64 else torch.int32))
            legs = {"rowgroup": dict(algo=4, npanels=1, wg_per_cu=-1), "rowsplit": dict(algo=4, npanels=1, wg_per_cu=1),
                    "rowsplit_auto": dict(algo=4), "slab": dict(algo=2), "auto_kept": dict(algo=0)}
            t = {}
            for name, kw in legs.items():
                f = lambda: D.spmm(A, B, out=out, colmajor=colmajor, **kw)
                t[name] = min(timeit(f), timeit(f, warm=0))
            print(f"{str(dt)[6:]:8s} m={m:8d} K={K:7d} /row={npr:4d} n={n:4d} {'col' if colmajor else 'row'}: bitwise={same}  " +
                  "  ".join(f"{k} {v:.4f}" for k, v in t.items()), flush=True)
        del A, B
