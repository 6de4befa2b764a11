"""ctypes binding of libmxgpu.so — the only way Python reaches the HIP kernels.

There is no fallback: if the shared library is missing or fails to load, or a
call reports an error, an exception is raised.  `torch` is imported first when
it is installed so that the library binds to the same HIP runtime instance
(same `libamdhip64.so.7` SONAME) as torch's allocator and RCCL — device
pointers of torch tensors can then be handed to the mxd_* entry points.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MXGPU_LIB") or os.path.join(_HERE, "libmxgpu.so")      # MXGPU_LIB: A/B another build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mxgpu.h")

# mx_dtype / mx_merge_op (include/mxgpu.h)
MX_F64, MX_F32, MX_I32, MX_LGL, MX_NONE = 0, 1, 2, 3, 4
MX_OP_ADD, MX_OP_SUB, MX_OP_MUL, MX_OP_OR, MX_OP_XOR, MX_OP_AND = range(6)


class MxError(RuntimeError):
    """An mx_* / mxd_* call returned non-zero (the .Call shim would Rf_error here)."""


class ResultInfo(C.Structure):
    _fields_ = [("indptr_len", C.c_int64), ("nnz", C.c_int64), ("values_len", C.c_int64),
                ("values_dtype", C.c_int), ("alias_structure", C.c_int)]


_lib = None


def declared_symbols() -> list[str]:
    """Every function name include/mxgpu.h declares (used by the export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mxd?_[A-Za-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MxError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C matrixextra_amd/csrc`. There is no CPU fallback.")
    try:  # share torch's HIP runtime when torch is present (see module docstring)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the plain C-ABI
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.mx_last_error.restype = C.c_char_p
    lib.mxd_spmm_last_kernel.restype = C.c_char_p
    lib.mxd_spmm_plan_imbalance_limit.restype = C.c_double
    lib.mxd_merge_workspace_bytes.restype = C.c_size_t
    lib.mxd_gather_workspace_bytes.restype = C.c_size_t
    lib.mxd_scan_workspace_bytes.restype = C.c_size_t
    lib.mxd_merge_workspace_bytes.argtypes = [C.c_int]
    lib.mxd_merge_fused_workspace_bytes.restype = C.c_size_t
    lib.mxd_merge_fused_workspace_bytes.argtypes = [C.c_int]
    lib.mxd_gather_workspace_bytes.argtypes = [C.c_int]
    lib.mxd_scan_workspace_bytes.argtypes = [C.c_int64]
    lib.mxd_csr_profile_workspace_bytes.restype = C.c_size_t
    lib.mxd_csr_profile_workspace_bytes.argtypes = [C.c_int]
    lib.mxd_colmap_workspace_bytes.restype = C.c_size_t
    lib.mxd_colmap_workspace_bytes.argtypes = [C.c_int]
    if lib.mx_abi_version() != 1:
        raise MxError("libmxgpu.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise MxError(load().mx_last_error().decode("utf-8", "replace"))


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    return None if a is None else C.c_void_p(a.ctypes.data)


def device_count() -> int:
    n = C.c_int(0)
    check(load().mx_device_count(C.byref(n)))
    return n.value


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(load().mx_device_name(buf, C.c_size_t(256)))
    return buf.value.decode()
