"""Where the tile kernel's wavefronts spend their cycles (diagnostic build path: mxd_debug_spmm_tile_stamps): per compute
wavefront the shader-clock cycles at the per-tile barriers against the whole sweep, at the vignette's shape."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matrixextra_amd import _lib, device as D, synth
lib = _lib.load()
m, K, npr, n = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (10_000, 10_000, 500, 100)
kind = sys.argv[5] if len(sys.argv) > 5 else "equal"          # a row-length distribution of tools/cliff_hunt.py (lognormal_1.5, giant, blocks ...)
if kind == "equal":
    p, j, x = synth.device_csr_fixed(m, K, npr, seed=7)
    A = D.DeviceCSR(p, j, x, m, K, int(j.numel()))
else:
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from cliff_hunt import build, lens_of
    A = build(m, K, lens_of(kind, m, npr, np.random.default_rng(7)), 7)
A.rows_sorted()
print(kind, "nnz", A.nnz, "cv", round(float(A.profile()[32]), 3), "MXGPU_TILE_DEAL", os.environ.get("MXGPU_TILE_DEAL"))
B = torch.randn((K, n), dtype=torch.float64, device="cuda")
out = torch.empty((m, n), dtype=torch.float64, device="cuda")
for _ in range(200):
    D.spmm(A, B, out=out, algo=5)
st = torch.zeros(2 * 16 * 16384, dtype=torch.int64, device="cuda")
lib.mxd_debug_spmm_tile_stamps(C.c_void_p(st.data_ptr()))
D.spmm(A, B, out=out, algo=5)
torch.cuda.synchronize()
lib.mxd_debug_spmm_tile_stamps(None)
s = st.cpu().numpy().reshape(-1, 16, 2)
used = s[:, :, 1] > 0
wait, total = s[:, :, 0][used].astype(float), s[:, :, 1][used].astype(float)
print(f"wavefronts {used.sum()}: sweep cycles mean {total.mean():.0f} (min {total.min():.0f}, max {total.max():.0f}); at the barriers mean {wait.mean():.0f} "
      f"= {100 * wait.mean() / total.mean():.1f} % (min {100 * (wait / total).min():.1f} %, max {100 * (wait / total).max():.1f} %)")
wg_tot = s[:, :, 1].max(axis=1)
wg_tot = wg_tot[wg_tot > 0]
print(f"workgroups {wg_tot.size}: slowest / mean sweep {wg_tot.max() / wg_tot.mean():.3f}")

import numpy as np
t = s[:, :, 1].max(axis=1)
t = t[t > 0]
print("workgroup sweep cycles: percentiles 10/50/90/99/max", [int(np.percentile(t, q)) for q in (10, 50, 90, 99, 100)])
