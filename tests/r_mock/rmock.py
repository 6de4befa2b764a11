"""Python driver of the mock R runtime (rmock.c) + the .Call shim (matrixextra_amd/csrc/r_shim.cpp), built into one shared
object by tests/r_mock/Makefile.  TEST INFRASTRUCTURE ONLY — it exists so that the shim's marshalling can be executed
without R: objects are made here as R would hand them to `.Call`, routines are looked up BY NAME in the table the shim
registered, and results come back as numpy arrays / dicts.  See the header of rmock.c for what the runtime models."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "_build", "libmxgpu_r_mock.so")

NILSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, S4SXP = 0, 10, 13, 14, 16, 19, 25
NA_INTEGER = -2147483648


class RError(RuntimeError):
    """The routine called Rf_error (R would raise a condition with this message)."""


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return SO


class SEXP(C.c_void_p):
    pass


_rt = None


def runtime() -> "Runtime":
    global _rt
    if _rt is None:
        _rt = Runtime()
    return _rt


class Runtime:
    def __init__(self):
        if not os.path.exists(SO):
            build()
        from matrixextra_amd import _lib
        _lib.load()                                    # libmxgpu.so first (RTLD_GLOBAL): the shim binds to that instance
        L = self.L = C.CDLL(SO)
        for name, res, args in [
            ("rmock_dllinfo", C.c_void_p, []), ("rmock_n_routines", C.c_int, []),
            ("rmock_routine_name", C.c_char_p, [C.c_int]), ("rmock_routine_arity", C.c_int, [C.c_int]),
            ("rmock_dynamic_symbols", C.c_int, []), ("rmock_nil", C.c_void_p, []),
            ("rmock_alloc", C.c_void_p, [C.c_int, C.c_int64]), ("rmock_dataptr", C.c_void_p, [C.c_void_p]),
            ("rmock_length", C.c_int64, [C.c_void_p]), ("rmock_typeof", C.c_int, [C.c_void_p]),
            ("rmock_is_live", C.c_int, [C.c_void_p]), ("rmock_id", C.c_uint64, [C.c_void_p]),
            ("rmock_hold", None, [C.c_void_p]), ("rmock_release", None, [C.c_void_p]),
            ("rmock_set_dim", None, [C.c_void_p, C.c_int, C.c_int]),
            ("rmock_get_dim", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            ("rmock_set_class", None, [C.c_void_p, C.c_char_p]),
            ("rmock_set_slot", None, [C.c_void_p, C.c_char_p, C.c_void_p]),
            ("rmock_get_slot", C.c_void_p, [C.c_void_p, C.c_char_p]),
            ("rmock_mkstring", C.c_void_p, [C.c_char_p]), ("rmock_list_elt", C.c_void_p, [C.c_void_p, C.c_int64]),
            ("rmock_set_list_elt", None, [C.c_void_p, C.c_int64, C.c_void_p]),
            ("rmock_string_elt", C.c_char_p, [C.c_void_p, C.c_int64]), ("rmock_names", C.c_void_p, [C.c_void_p]),
            ("rmock_set_gctorture", None, [C.c_int]), ("rmock_fail_alloc_at", None, [C.c_long]), ("rmock_gc", None, []),
            ("rmock_protect_depth", C.c_int, []), ("rmock_preserved_count", C.c_int, []),
            ("rmock_violations", C.c_long, []), ("rmock_violation_msg", C.c_char_p, []),
            ("rmock_clear_violations", None, []), ("rmock_last_error", C.c_char_p, []),
            ("rmock_alloc_count", C.c_long, []), ("rmock_collected_count", C.c_long, []), ("rmock_live_count", C.c_long, []),
            ("rmock_sweep_dead", None, []),
            ("rmock_set_host_routines", None, [C.c_int]), ("rmock_host_calls", C.c_long, []), ("rmock_host_last", C.c_char_p, []),
            ("rmock_dotcall", C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
            ("R_init_mxgpu_r", None, [C.c_void_p]),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.R_init_mxgpu_r(L.rmock_dllinfo())            # what dyn.load("mxgpu_r.so") does
        self.nil = L.rmock_nil()
        self.calls = 0

    # ---------------------------------------------------------------------------------- the registered table
    def routines(self) -> dict[str, int]:
        L = self.L
        return {L.rmock_routine_name(k).decode(): L.rmock_routine_arity(k) for k in range(L.rmock_n_routines())}

    # ---------------------------------------------------------------------------------- making R objects
    def _vec(self, a, dtype, sxp):
        a = np.ascontiguousarray(a, dtype=dtype).reshape(-1)
        s = self.L.rmock_alloc(sxp, a.size)
        if a.size:
            C.memmove(self.L.rmock_dataptr(s), a.ctypes.data, a.nbytes)
        return s

    def integer(self, a):
        return self._vec(a, np.int32, INTSXP)

    def real(self, a):
        return self._vec(a, np.float64, REALSXP)

    def logical(self, a):
        return self._vec(a, np.int32, LGLSXP)

    def matrix(self, M, kind="real"):
        """Column-major R matrix; kind 'float32': the float32@Data form (INTSXP holding binary32 bits, R/matmul.R:260)."""
        M = np.asarray(M)
        assert M.ndim == 2
        flat = np.asfortranarray(M).reshape(-1, order="F")
        if kind == "real":
            s = self.real(flat)
        elif kind == "integer":
            s = self.integer(flat)
        elif kind == "logical":
            s = self.logical(flat)
        elif kind == "float32":
            s = self.integer(np.ascontiguousarray(flat, dtype=np.float32).view(np.int32))
        else:
            raise ValueError(kind)
        self.L.rmock_set_dim(s, M.shape[0], M.shape[1])
        return s

    def float32(self, v):
        return self.integer(np.ascontiguousarray(v, dtype=np.float32).view(np.int32))

    def string(self, s):
        return self.L.rmock_mkstring(s.encode())

    def list(self, items):
        s = self.L.rmock_alloc(VECSXP, len(items))
        for k, it in enumerate(items):
            self.L.rmock_set_list_elt(s, k, it)
        return s

    def s4(self, cls, **slots):
        o = self.L.rmock_alloc(S4SXP, 0)
        self.L.rmock_set_class(o, cls.encode())
        for k, v in slots.items():
            self.L.rmock_set_slot(o, k.encode(), v)
        return o

    def slot(self, o, name):
        return self.L.rmock_get_slot(o, name.encode())

    def release(self, *sexps):
        for s in sexps:
            self.L.rmock_release(s)

    # ---------------------------------------------------------------------------------- reading R objects
    def typeof(self, s):
        return self.L.rmock_typeof(s)

    def view(self, s):
        """numpy COPY of an atomic vector (with its dim, column-major, when it has one)."""
        t, n = self.L.rmock_typeof(s), self.L.rmock_length(s)
        assert t in (LGLSXP, INTSXP, REALSXP), t
        dt = np.float64 if t == REALSXP else np.int32
        if n:
            a = np.ctypeslib.as_array(C.cast(self.L.rmock_dataptr(s), C.POINTER(C.c_double if t == REALSXP else C.c_int32)),
                                      shape=(n,)).astype(dt, copy=True)
        else:
            a = np.empty(0, dtype=dt)
        nr, nc = C.c_int(), C.c_int()
        if self.L.rmock_get_dim(s, C.byref(nr), C.byref(nc)):
            a = a.reshape((nr.value, nc.value), order="F")
        return a

    def names(self, s):
        n = self.L.rmock_names(s)
        if not n:
            return None
        return [self.L.rmock_string_elt(n, k).decode() for k in range(self.L.rmock_length(n))]

    def elts(self, s):
        return [self.L.rmock_list_elt(s, k) for k in range(self.L.rmock_length(s))]

    def as_py(self, s):
        """R value -> numpy array / dict (named list) / list / str list / None."""
        if s is None or s == self.nil:
            return None
        t = self.L.rmock_typeof(s)
        if t in (LGLSXP, INTSXP, REALSXP):
            return self.view(s)
        if t == STRSXP:
            return [self.L.rmock_string_elt(s, k).decode() for k in range(self.L.rmock_length(s))]
        if t == VECSXP:
            vals = [self.as_py(e) for e in self.elts(s)]
            nm = self.names(s)
            return dict(zip(nm, vals)) if nm is not None else vals
        raise TypeError(f"SEXP type {t}")

    # ---------------------------------------------------------------------------------- .Call
    def call(self, name, *args):
        """.Call("_MatrixExtra_<name>", ...) through the registered table.  Returns the result SEXP (held until
        release()).  Raises RError on Rf_error; asserts the protect stack is balanced and nothing stays preserved."""
        L = self.L
        full = name if name.startswith("_") else "_MatrixExtra_" + name
        arr = (C.c_void_p * max(len(args), 1))(*args)
        out = C.c_void_p()
        depth0, kept0 = L.rmock_protect_depth(), L.rmock_preserved_count()
        rc = L.rmock_dotcall(full.encode(), len(args), arr, C.byref(out))
        self.calls += 1
        assert L.rmock_protect_depth() == depth0, f"{full}: protect stack {depth0} -> {L.rmock_protect_depth()}"
        assert L.rmock_preserved_count() == kept0, f"{full}: R_PreserveObject without R_ReleaseObject"
        if rc == 1:
            raise RError(L.rmock_last_error().decode())
        if rc != 0:
            raise LookupError(L.rmock_last_error().decode())
        assert L.rmock_violations() == 0, L.rmock_violation_msg().decode()
        return None if (not out.value or out.value == self.nil) else out.value     # R_NilValue -> None

    def gctorture(self, on=True):
        self.L.rmock_set_gctorture(1 if on else 0)

    def check_clean(self):
        assert self.L.rmock_violations() == 0, self.L.rmock_violation_msg().decode()
