"""tools/cliff_hunt.py for the other rows of the hot path: SpMV (AUTO and the kept plan), CSR + CSR, CSR * CSR, row gather —
ms per call on row-length distributions, relative to rows of equal length (per entry)."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from matrixextra_amd import device as D, _lib  # noqa: E402
from auto_map import timeit  # noqa: E402
from cliff_hunt import lens_of, build  # noqa: E402

KINDS = ["equal", "lognormal_1.0", "lognormal_1.5", "half_empty", "giant", "blocks"]
for (m, K, mean) in [(1_000_000, 100_000, 32), (2_000_000, 2_000_000, 50), (100_000, 10_000, 64), (10_000, 10_000, 500)]:
    base = {}
    for kind in KINDS:
        rng = np.random.default_rng(7)
        lens = lens_of(kind, m, mean, rng)
        A = build(m, K, lens, 7)
        A2 = build(m, K, lens_of(kind, m, mean, np.random.default_rng(8)) if kind != "giant" else lens, 8)
        v = torch.randn(K, dtype=torch.float64, device="cuda")
        y = torch.empty(m, dtype=torch.float64, device="cuda")
        rows = torch.randint(0, m, (m // 5,), dtype=torch.int32, device="cuda")
        ops = {
            "spmv": lambda: D.spmv(A, v, out=y),
            "spmv_planned": lambda: D.spmv_planned(A, v, out=y),
            "add": lambda: D.csr_elemwise(_lib.MX_OP_ADD, A, A2),
            "mul": lambda: D.csr_elemwise(_lib.MX_OP_MUL, A, A2),
            "gather": lambda: D.csr_gather_rows(A, rows),
        }
        line = []
        for name, f in ops.items():
            try:
                f(); f()
                t = min(timeit(f, reps=5), timeit(f, reps=5, warm=0))
            except Exception as exc:  # noqa: BLE001
                line.append(f"{name} n/a({str(exc)[:40]})"); continue
            work = A.nnz + (A2.nnz if name in ("add", "mul") else 0)
            per = t / work
            if kind == "equal":
                base[name] = per
            rel = per / base[name]
            line.append(f"{name} {t:.4f} x{rel:.2f}{'<<<<' if rel > 1.6 else ''}")
        print(f"{m}x{K} {mean}/row {kind:14s} " + "  ".join(line), flush=True)
        del A, A2
        torch.cuda.empty_cache()
