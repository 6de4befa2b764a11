"""Look for performance cliffs of AUTO's SpMM on row-length distributions: for each shape and dtype, ms per call (plan kept)
and ns per (entry x dense column) relative to rows of equal length.  Anything far above 1.5x deserves a look."""
import sys, os, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D, synth, _lib  # noqa: E402
from auto_map import timeit  # noqa: E402

lib = _lib.load()


def lens_of(kind, m, mean, rng):
    if kind == "equal":
        return np.full(m, mean, dtype=np.int64)
    if kind.startswith("lognormal"):
        s = float(kind.split("_")[1])
        return np.floor(rng.lognormal(np.log(mean) - 0.5 * s * s, s, size=m)).astype(np.int64)
    if kind == "half_empty":
        l = np.full(m, 2 * mean, dtype=np.int64); l[rng.random(m) < 0.5] = 0; return l
    if kind == "giant":
        l = np.full(m, mean, dtype=np.int64); l[rng.integers(0, m, size=4)] = 50_000; return l
    if kind == "blocks":                                   # long rows first, short rows after (sorted-by-length data)
        l = np.sort(np.floor(rng.lognormal(np.log(mean) - 0.5, 1.0, size=m)).astype(np.int64))[::-1].copy(); return l
    raise ValueError(kind)


def build(m, K, lens, seed):
    lens = np.minimum(lens, K)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    l = torch.from_numpy(lens).cuda()
    total = int(l.sum().item())
    row = torch.repeat_interleave(torch.arange(m, dtype=torch.int64, device="cuda"), l)
    key = torch.sort(row * K + torch.randint(0, K, (total,), dtype=torch.int64, device="cuda", generator=g)).values
    keep = torch.ones(total, dtype=torch.bool, device="cuda"); keep[1:] = key[1:] != key[:-1]
    key = key[keep]; row = key // K
    j = (key - row * K).to(torch.int32).contiguous()
    p = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); torch.cumsum(torch.bincount(row, minlength=m), 0, out=p[1:])
    x = torch.rand(j.numel(), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    return D.DeviceCSR(p.to(torch.int32), j, x, m, K, int(j.numel()))


def main():
    SHAPES = [(1_000_000, 100_000, 32, 128), (1_000_000, 200_000, 64, 256), (100_000, 10_000, 64, 64), (1_000_000, 10_000, 12, 16),
              (10_000, 10_000, 500, 100), (200_000, 50_000, 100, 32)]
    KINDS = ["equal", "lognormal_0.5", "lognormal_1.0", "lognormal_1.5", "half_empty", "giant", "blocks"]
    out = []
    for (m, K, mean, n) in SHAPES:
        for dt in (torch.float64, torch.float32):
            for colmajor in (True, False):
                base = None
                for kind in KINDS:
                    rng = np.random.default_rng(7)
                    A = build(m, K, lens_of(kind, m, mean, rng), 7)
                    B = torch.randn((K, n), dtype=dt, device="cuda")
                    C = torch.empty((n, m) if colmajor else (m, n), dtype=dt, device="cuda")
                    f = lambda: D.spmm(A, B, out=C, colmajor=colmajor)
                    f(); f()
                    t = min(timeit(f), timeit(f, warm=0))
                    per = t * 1e6 / (A.nnz * n)
                    if kind == "equal":
                        base = per
                    rec = dict(m=m, K=K, mean=mean, n=n, dtype=str(dt)[6:], colmajor=colmajor, rows=kind, nnz=A.nnz, ms=round(t, 4),
                               kernel=lib.mxd_spmm_last_kernel().decode(), rel=round(per / base, 2))
                    out.append(rec)
                    flag = "  <<<<" if rec["rel"] > 1.6 else ""
                    print(f"{m}x{K} {mean}/row n={n} {rec['dtype']} {'col' if colmajor else 'row'} {kind:14s} {rec['kernel']:22s} {t:8.4f} ms  x{rec['rel']}{flag}", flush=True)
                    del A, B, C
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cliff_hunt.json"), "w"), indent=1)



if __name__ == "__main__":
    main()
