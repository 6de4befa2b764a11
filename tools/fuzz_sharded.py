#!/usr/bin/env python3
"""Randomised geometry of the SHARDED export (csrc/api.hip spmm_host_multi: row ranges balanced by cost, the result
registered piece by piece, pitched downloads split at piece boundaries, B shared per device, persistent workers): random
row / column counts, element size, byte offset of the result inside its page, empty row ranges, device lists of 1 .. 8
entries (this GPU listed several times) changing from case to case, both result layouts — every case bit for bit against
the unsharded call, guard bytes either side of the result.
    python tools/fuzz_sharded.py [seconds] [seed]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matrixextra_amd import _lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
lib = _lib.load()
t_end = time.time() + budget
cases = 0
GUARD = 8192
try:
    while time.time() < t_end:
        f32 = bool(rng.random() < 0.5)
        dt = np.float32 if f32 else np.float64
        item = 4 if f32 else 8
        n = int(rng.choice([1, 3, 16, 33, 64, 129, 256]))
        m = int(rng.choice([1, 7, 64, 1000, 4097, 50_000, 300_001, 1_200_000]))
        if m * n * item > (1 << 30):
            m = (1 << 30) // (n * item)
        K = int(rng.choice([1, 50, 3000, 60_000]))
        nnz_row = min(int(rng.choice([1, 8, 40])), K)
        p, j, x = synth.csr_fixed(m, K, nnz_row, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.3 and m > 2:                                 # a tail (or nearly everything) without entries
            cut = int(rng.integers(0, m))
            p = p.copy()
            p[cut:] = p[cut]
        B = rng.normal(size=(K, n)).astype(dt)
        nd = int(rng.choice([1, 2, 3, 5, 8]))
        rowmajor = bool(rng.random() < 0.35)                             # dense %*% CSC: the result is row-major for the library
        case = dict(m=m, n=n, K=K, nnz_row=nnz_row, f32=f32, nd=nd, rowmajor=rowmajor)
        if rowmajor:
            X = np.asfortranarray(B.T)                                   # n x K
            fn = lib.mx_matmul_dense_csc_float32 if f32 else lib.mx_matmul_dense_csc_numeric
            shape = (n, m)                                               # R result n x m column-major
        else:
            X = np.asfortranarray(B.T)
            fn = lib.mx_tcrossprod_csr_dense_float32 if f32 else lib.mx_tcrossprod_csr_dense_numeric
            shape = (m, n)
        def run(devs, cold):
            raw = np.empty(m * n * item + 3 * GUARD, dtype=np.uint8)
            raw[:] = 0x5A
            off = int(rng.choice([0, 8, 16, 40, 2048, 4088, 4096 - item])) // item * item
            start = (-raw.ctypes.data) % 4096 + off
            out = raw[start:start + m * n * item].view(dt)
            _lib.check(lib.mx_set_devices((C.c_int * max(devs, 1))(*([0] * devs)), devs))
            if cold:                                                      # (a cached operand takes another pipeline: other kernels per block)
                lib.mx_cache_invalidate(None)
            if rowmajor:
                _lib.check(fn(_lib.ptr(X), C.c_int(n), C.c_int(K), _lib.ptr(p), _lib.ptr(j), _lib.ptr(x), C.c_int(m), C.c_int(1),
                              C.c_void_p(out.ctypes.data)))
            else:
                _lib.check(fn(_lib.ptr(p), _lib.ptr(j), _lib.ptr(x), C.c_int(m), _lib.ptr(X), C.c_int(n), C.c_int(K), C.c_int(1),
                              C.c_void_p(out.ctypes.data)))
            assert (raw[:start] == 0x5A).all() and (raw[start + m * n * item:] == 0x5A).all(), ("guard bytes", devs, case)
            return out.copy()
        cold = bool(rng.random() < 0.5)
        case["cold"] = cold
        single = run(0, True)
        if not cold:
            run(nd, True)                                                 # leaves the operand in the cache
        sharded = run(nd, cold)
        # a shard is a smaller product than the whole: AUTO may give it another kernel (another order of the same terms), so
        # sharded == unsharded holds to the product's tolerance, and bit for bit between two sharded calls
        scale = max(1.0, float(np.abs(single).max()))
        err = float(np.max(np.abs(sharded.astype(np.float64) - single.astype(np.float64)))) / scale
        assert err <= (3e-6 if f32 else 1e-12), ("sharded vs unsharded", err, case)
        again = run(nd, cold)
        assert np.array_equal(again.view(np.int32 if f32 else np.int64), sharded.view(np.int32 if f32 else np.int64)), ("sharded call not reproducible", case)
        # sampled rows against a dense product
        res = single.reshape(shape, order="F")
        res = res.T if rowmajor else res
        rows = np.unique(np.r_[0:min(m, 16), rng.integers(0, m, 50), max(0, m - 16):m])
        dense = np.stack([x[p[r]:p[r + 1]] @ B[j[p[r]:p[r + 1]]].astype(np.float64) for r in rows])
        err = np.max(np.abs(res[rows] - dense)) / max(1.0, np.abs(dense).max())
        assert err <= (3e-5 if f32 else 1e-12), ("rows vs dense", err, case)
        cases += 1
finally:
    lib.mx_set_devices(None, 0)
print(f"sharded fuzz OK: {cases} cases in {budget:.0f} s (seed {seed})")
