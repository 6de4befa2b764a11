#!/usr/bin/env python3
"""Randomised geometry of the tiled cold export (csrc/api.hip: row blocks x column groups, page cuts, the groups' last
partial pages): random row counts around the block granularity, column counts, element size, byte offset of the result
inside its first page; every case against the same call with the CSR cached (column blocks of the kept plan, contiguous
downloads — another code path end to end) and against a dense product on sampled rows; guard bytes either side.
    python tools/fuzz_export_tiles.py [seconds] [seed]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matrixextra_amd import _lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
lib = _lib.load()
t_end = time.time() + budget
cases = tiled = 0
GUARD = 8192
while time.time() < t_end:
    f32 = bool(rng.random() < 0.5)
    dt = np.float32 if f32 else np.float64
    item = 4 if f32 else 8
    n = int(rng.choice([33, 64, 80, 129, 160, 256]))
    target = int(rng.integers(130, 420)) << 20                       # bytes of result: 2 .. 6 column groups
    m = max(4096, target // (n * item))
    m += int(rng.choice([0, 1, -1, 511, 1023, 1024, 1025, -1023, 37]))
    K = int(rng.choice([1000, 20_000, 60_000]))
    nnz_row = int(rng.choice([24, 48, 96]))                           # (the upload must outlast half the page work for row blocks)
    p, j, x = synth.csr_fixed(m, K, min(nnz_row, K), seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.3:                                            # rows without entries, a whole block of them sometimes
        cut = int(rng.integers(0, m))
        p = p.copy()
        p[cut:] = p[cut]
    B = rng.normal(size=(K, n)).astype(dt)
    Y = np.asfortranarray(B.T)
    fn = lib.mx_tcrossprod_csr_dense_float32 if f32 else lib.mx_tcrossprod_csr_dense_numeric
    raw = np.empty(m * n * item + 3 * GUARD, dtype=np.uint8)
    raw[:] = 0x5A
    off = int(rng.choice([0, 8, 16, 40, 2048, 4088, 4096 - item])) // item * item
    start = (-raw.ctypes.data) % 4096 + off
    out = raw[start:start + m * n * item].view(dt).reshape(n, m).T
    lib.mx_cache_invalidate(None)
    if p[-1] == p[0]:
        continue
    _lib.check(fn(_lib.ptr(p), _lib.ptr(j), _lib.ptr(x), C.c_int(m), _lib.ptr(Y), C.c_int(n), C.c_int(K), C.c_int(1),
                  C.c_void_p(out.ctypes.data)))
    buf = C.create_string_buffer(512)
    lib.mx_last_call_phases(buf, C.c_size_t(512))
    what = buf.value.decode()
    tiled += "tiles=" in what
    case = dict(m=m, n=n, K=K, nnz_row=nnz_row, f32=f32, off=off, phases=what)
    assert (raw[:start] == 0x5A).all() and (raw[start + m * n * item:] == 0x5A).all(), ("guard bytes", case)
    rows = np.unique(np.r_[0:64, rng.integers(0, m, 200), m - 64:m])
    dense = np.zeros((rows.size, n))
    for k, r in enumerate(rows):
        dense[k] = x[p[r]:p[r + 1]] @ B[j[p[r]:p[r + 1]]].astype(np.float64)
    tol = 3e-5 if f32 else 1e-12
    err = np.max(np.abs(out[rows] - dense)) / max(1.0, np.abs(dense).max())
    assert err <= tol, ("rows vs dense", err, case)
    cached = np.empty((m, n), dtype=dt, order="F")
    _lib.check(fn(_lib.ptr(p), _lib.ptr(j), _lib.ptr(x), C.c_int(m), _lib.ptr(Y), C.c_int(n), C.c_int(K), C.c_int(1),
                  C.c_void_p(cached.ctypes.data)))
    whole = m - m % 1024                                               # (the last, partial octet may be laid out differently)
    if not np.array_equal(cached[:whole], out[:whole]):
        bad = np.argwhere(cached[:whole] != out[:whole])
        d = np.max(np.abs(cached[:whole].astype(np.float64) - out[:whole]))
        assert d <= tol * max(1.0, np.abs(dense).max()), ("cold vs cached", d, bad[:5].tolist(), case)
    assert np.allclose(cached[whole:], out[whole:], rtol=tol, atol=tol * max(1.0, float(np.abs(dense).max()))), ("tail rows", case)
    cases += 1
    del raw, out, cached
lib.mx_cache_invalidate(None)
print(f"export tiles fuzz OK: {cases} cases ({tiled} tiled) in {budget:.0f} s (seed {seed})")
