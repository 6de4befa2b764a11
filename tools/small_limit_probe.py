"""Where does the small-call path (one pinned block up, one down: csrc/api.hip SMALL_LIMIT) stop paying against the regular
path?  p50 us per export call at growing operand sizes; run once per MXGPU_SMALL_LIMIT_KB setting:
    MXGPU_SMALL_LIMIT_KB=16 python tools/small_limit_probe.py ; MXGPU_SMALL_LIMIT_KB=65536 python tools/small_limit_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from matrixextra_amd import _lib, exports as G, synth
from small_calls import p50_us
lib = _lib.load()
def small_count():
    v = ctypes.c_int64(0); lib.mx_get_option(b"small_calls", ctypes.byref(v)); return v.value
print("MXGPU_SMALL_LIMIT_KB =", os.environ.get("MXGPU_SMALL_LIMIT_KB", "(default 512)"))
for nnz_target in (20_000, 50_000, 100_000, 200_000, 400_000, 800_000, 1_600_000):
    per_row, K, n = 20, 10_000, 16
    m = nnz_target // per_row
    p, j, x = synth.csr_fixed(m, K, per_row, seed=3)
    p2, j2, x2 = synth.csr_overlapping(p, j, K, per_row)
    Y = np.asfortranarray(synth.dense_normal(n, K, seed=4))
    v = synth.dense_normal(K, 1, seed=5).reshape(-1)
    rows = synth.rows_with_replacement(m // 3, m)
    legs = {"spmm": lambda: G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1), "spmv": lambda: G.matmul_csr_dvec_numeric(p, j, x, v, 1),
            "add": lambda: G.add_csr_elemwise(p, p2, j, j2, x, x2, False), "mul": lambda: G.multiply_csr_elemwise(p, p2, j, j2, x, x2),
            "rows": lambda: G.copy_csr_rows_numeric(p, j, x, rows)}
    row = []
    for name, fn in legs.items():
        c0 = small_count()
        t = p50_us(fn, 100, 10)["p50_us"]
        row.append(f"{name} {t:7.1f}{'s' if small_count() > c0 else ' '}")
    print(f"nnz {nnz_target:8d} (CSR {nnz_target * 12 / 1e6:5.2f} MB): " + "  ".join(row), flush=True)
