#!/usr/bin/env python3
"""Where a tile of the LDS-panel SpMV kernel spends its cycles (stamped diagnostic build; shares, not lengths)."""
import sys
sys.path.insert(0, ".")
import ctypes as C
import numpy as np, torch
from matrixextra_amd import _lib, device as D, synth
lib = _lib.load()
for (m, K, k) in ((1_000_000, 100_000, 32), (1_000_000, 16_000, 32)):
    p, j, x = synth.csr_fixed(m, K, k)
    A = D.DeviceCSR.from_host(p, j, x, K)
    v = torch.randn(K, dtype=torch.float64, device="cuda")
    nt = (A.nnz + 23551) // 23552
    st = torch.zeros(nt * 8, dtype=torch.int64, device="cuda")
    lib.mxd_debug_spmv_tile_stamps(C.c_void_p(st.data_ptr()))
    for _ in range(3):
        D.spmv(A, v, algo=2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); D.spmv(A, v, algo=2); b.record(); torch.cuda.synchronize()
    lib.mxd_debug_spmv_tile_stamps(None)
    s = st.cpu().numpy().reshape(nt, 8)
    d = np.diff(s[:, :5], axis=1).astype(np.float64)
    print(f"{m}x{K}: stamped launch {a.elapsed_time(b)*1e3:.0f} us; cycles per tile (median): search {np.median(d[:,0]):.0f}  entries "
          f"{np.median(d[:,1]):.0f}  panels {np.median(d[:,2]):.0f}  reduce {np.median(d[:,3]):.0f}; total {np.median(s[:,4]-s[:,0]):.0f}; "
          f"span of all tiles {(s[:,4].max()-s[:,0].min())} ticks")
