#!/usr/bin/env python3
"""Kernels at nnz = 2.1e9 (just under R's int32 limit): 32-bit offset bugs in SpMV (all kernels), the sortedness check,
the row gather, SpMM:  python tools/maxnnz_probe.py"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from matrixextra_amd import _lib, device as D
m, per, K = 21_000_000, 100, 128
nnz = m * per
assert nnz < 2 ** 31
indptr = (torch.arange(m + 1, dtype=torch.int64, device="cuda") * per).to(torch.int32)
idx = (torch.arange(nnz, dtype=torch.int32, device="cuda") % per)
val = torch.ones(nnz, dtype=torch.float64, device="cuda")
val[-per:] = 2.0                                      # the last row differs: catches a wrapped tail
val[(2 ** 31 // 8):(2 ** 31 // 8) + per] = 3.0        # entries around byte offset 2^31 of `values`
A = D.DeviceCSR(indptr, idx, val, m, K, nnz)
v = torch.arange(1, K + 1, dtype=torch.float64, device="cuda")
ref = torch.zeros(m, dtype=torch.float64, device="cuda")
seg = torch.cumsum(val.view(-1), 0)                   # exact: small integers
# y[r] = sum_k val[r*per+k] * v[k]
ref = (val.view(m, per) * v[:per]).sum(dim=1)
print("sorted:", A.rows_sorted(), flush=True)
for algo, name in ((0, "auto"), (1, "group"), (3, "flat"), (2, "tile")):
    y = D.spmv(A, v, algo=algo)
    torch.cuda.synchronize()
    bad = int((y != ref).sum())
    print(f"spmv {name}: mismatches {bad}", flush=True)
    assert bad == 0
A.spmv_plan(); y = D.spmv_planned(A, v); torch.cuda.synchronize()
print("spmv planned: mismatches", int((y != ref).sum()), flush=True)
assert bool((y == ref).all())
A.drop_spmv_plan()
rows = torch.tensor([0, m - 1, 2 ** 31 // 8 // per, 5, m - 1], dtype=torch.int32, device="cuda")
g = D.csr_gather_rows(A, rows)
gp, gj, gx = g.to_host()
assert gp.tolist() == [0, 100, 200, 300, 400, 500] and gx[100:200].tolist() == [2.0] * 100 and gx[400:].tolist() == [2.0] * 100
print("gather ok", flush=True)
n = 16
B = torch.randn(K, n, dtype=torch.float64, device="cuda")
for algo in (0, 1):
    C = torch.full((m, n), float("nan"), dtype=torch.float64, device="cuda")
    D.spmm(A, B, out=C, colmajor=False, algo=algo)
    torch.cuda.synchronize()
    r0 = (B[:per]).sum(dim=0)
    for i in (0, 12345, m - 2):
        assert torch.allclose(C[i], r0, rtol=1e-12, atol=1e-12), (algo, i)
    assert torch.allclose(C[m - 1], 2 * r0, rtol=1e-12, atol=1e-12)
    print(f"spmm algo {algo} ({_lib.load().mxd_spmm_last_kernel().decode()}) ok", flush=True)
C = torch.full((m, n), float("nan"), dtype=torch.float64, device="cuda")
D.spmm_planned(A, B, out=C, colmajor=False)
torch.cuda.synchronize()
for i in (0, 12345, 2 ** 31 // 8 // per - 1, 2 ** 31 // 8 // per, 2 ** 31 // 8 // per + 1, m - 2, m - 1):
    want = (val[i * per:(i + 1) * per, None] * B[:per]).sum(dim=0)
    assert torch.allclose(C[i], want, rtol=1e-12, atol=1e-12), i
print(f"spmm planned ({_lib.load().mxd_spmm_last_kernel().decode()}) ok", flush=True)
del C, A, val, idx, indptr
torch.cuda.empty_cache()
# merges with 1.05e9 entries per operand and 1.575e9 in the union (offsets beyond 2^30 entries = 2^33 bytes of values)
m2 = 10_500_000
ip = (torch.arange(m2 + 1, dtype=torch.int64, device="cuda") * per).to(torch.int32)
ja = (torch.arange(m2 * per, dtype=torch.int32, device="cuda") % per)
xa = torch.ones(m2 * per, dtype=torch.float64, device="cuda")
A1 = D.DeviceCSR(ip, ja, xa, m2, 2 * per, m2 * per)
A2 = D.DeviceCSR(ip, ja + 50, xa * 2, m2, 2 * per, m2 * per)
R = D.csr_elemwise(_lib.MX_OP_ADD, A1, A2)
assert R.nnz == m2 * 150
row = torch.cat([torch.ones(50), torch.full((50,), 3.0), torch.full((50,), 2.0)]).to(torch.float64).cuda()
cols = torch.arange(150, dtype=torch.int32, device="cuda")
for i in (0, 7_158_279, m2 - 1):                       # 7,158,279 * 150 * 8 B = just above 2^33 bytes
    s = i * 150
    assert bool((R.values[s:s + 150] == row).all()) and bool((R.indices[s:s + 150] == cols).all()), i
assert int(R.indptr[-1]) == m2 * 150
Rm = D.csr_elemwise(_lib.MX_OP_MUL, A1, A2)
assert Rm.nnz == m2 * 50 and bool((Rm.values[-50:] == 2.0).all()) and bool((Rm.indices[-50:] == torch.arange(50, 100, dtype=torch.int32, device="cuda")).all())
print("merge ok", flush=True)
print("max nnz ok")
