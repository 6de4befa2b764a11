import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from matrixextra_amd import _lib, device as D, synth
p, j, x = synth.csr_fixed(1_000_000, 100_000, 32)
A = D.DeviceCSR.from_host(p, j, x, 100_000); B = torch.from_numpy(synth.dense_normal(100_000, 128)).cuda()
C = torch.empty((128, 1_000_000), dtype=torch.float64, device="cuda")
A.rows_sorted()
torch.cuda.synchronize(); time.sleep(0.5)
ts = []
for k in range(14):
    t0 = time.perf_counter(); D.spmm(A, B, out=C, colmajor=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("sync each:", [round(t, 2) for t in ts])
time.sleep(1.0)
t0 = time.perf_counter()
for k in range(10): D.spmm(A, B, out=C, colmajor=True)
torch.cuda.synchronize(); print("10 back-to-back after 1 s idle:", round((time.perf_counter() - t0) / 10 * 1e3, 3))
t0 = time.perf_counter()
for k in range(10): D.spmm(A, B, out=C, colmajor=True)
torch.cuda.synchronize(); print("next 10:", round((time.perf_counter() - t0) / 10 * 1e3, 3))
