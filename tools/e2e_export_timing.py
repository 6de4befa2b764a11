import sys, time; sys.path.insert(0, '.')
import numpy as np
from matrixextra_amd import exports as G, synth
m, K, n = 1_000_000, 100_000, 128
p, j, x = synth.csr_fixed(m, K, 32)
B = synth.dense_normal(K, n)
Y = np.asfortranarray(B.T)
out = None
for i in range(4):
    del out                                   # (freeing the previous 1 GB result is not part of the next call)
    t0 = time.perf_counter(); out = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1); t = time.perf_counter() - t0
    print(f"export call {i}: {t*1e3:.1f} ms", flush=True)
from oracle import oracle as O
ref = np.zeros(2048 * n); O.gemm_csr_drm_as_drm(2048, n, p[:2049], j, x, B.reshape(-1), n, ref, n, O.max_threads(), True)
print("max err", np.abs(out[:2048] - ref.reshape(2048, n)).max())
rows = synth.rows_with_replacement(200_000, m)
for i in range(2):
    t0 = time.perf_counter(); r = G.copy_csr_rows_numeric(p, j, x, rows); t = time.perf_counter() - t0
    print(f"gather export {i}: {t*1e3:.1f} ms")
rr = O.copy_csr_rows_numeric(p, j, x, rows); print("gather equal", all(np.array_equal(r[k], rr[k]) for k in r))
