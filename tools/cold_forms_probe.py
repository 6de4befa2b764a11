#!/usr/bin/env python3
"""A/B of the two cold forms of the export-level product (column blocks after a whole upload vs row blocks), alternating
inside one process, cfg2, malloc'ed result.  `python tools/cold_forms_probe.py [calls] [torch]` — with `torch`, the
process first does what bench.py has done by the time it reaches the export leg (torch's allocator holding device memory)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matrixextra_amd import _lib, synth
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SLEEP = float(os.environ.get("PROBE_SLEEP_MS", "0")) / 1e3          # pause between the invalidate and the call
REUSE = os.environ.get("PROBE_REUSE_RESULT") == "1"                 # one result buffer for all calls (pages stay mapped)
ORDER = os.environ.get("PROBE_ORDER", "10")                        # the forms, cycled: "10" alternates, "1110" etc.
if "torch" in sys.argv:
    import torch
    a = torch.empty(24 << 30, dtype=torch.uint8, device="cuda"); a.zero_(); torch.cuda.synchronize(); del a
    torch.cuda.empty_cache()
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
m, K, n = int(os.environ.get("PROBE_M", 1_000_000)), int(os.environ.get("PROBE_K", 100_000)), int(os.environ.get("PROBE_N", 128))
F32 = os.environ.get("PROBE_F32") == "1"                           # cfg5's kind: float32 dense, f64 CSR values
ITEM = 4 if F32 else 8
p, j, x = synth.csr_fixed(m, K, int(os.environ.get("PROBE_NNZ_ROW", 32)))
Y = np.asfortranarray(synth.dense_normal(K, n).T.astype(np.float32 if F32 else np.float64))
lib = _lib.load()
fn = lib.mx_tcrossprod_csr_dense_float32 if F32 else lib.mx_tcrossprod_csr_dense_numeric
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]


kept = libc.malloc(ITEM * m * n) if REUSE else None


def call():
    out = kept if REUSE else libc.malloc(ITEM * m * n)
    t0 = time.perf_counter()
    _lib.check(fn(p.ctypes.data, j.ctypes.data, x.ctypes.data, m, Y.ctypes.data, n, K, 1, out))
    t = time.perf_counter() - t0
    if not REUSE:
        libc.free(out)
    buf = C.create_string_buffer(512)
    lib.mx_last_call_phases(buf, C.c_size_t(512))
    return t * 1e3, buf.value.decode()


for _ in range(3):
    lib.mx_cache_invalidate(None)
    call()
ts = {"1": [], "0": []}
for i in range(2 * calls):
    form = ORDER[i % len(ORDER)]
    os.environ["MXGPU_EXPORT_COLD_COLS"] = "2" if form == "1" else "0"   # 2 = column blocks forced, 0 = row blocks
    t0 = time.perf_counter()
    lib.mx_cache_invalidate(None)
    t_inv = (time.perf_counter() - t0) * 1e3
    time.sleep(SLEEP)
    t, ph = call()
    ph += f";invalidate={t_inv:.2f}"
    ts[form].append(t)
    print(f"cols={form} {t:6.1f} ms  {ph}", flush=True)
for form, v in ts.items():
    if v:
        print("cols=%s: min %.1f median %.1f max %.1f" % (form, min(v), sorted(v)[len(v) // 2], max(v)))
