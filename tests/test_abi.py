"""C-ABI surface checks that need no GPU: the library loads, exports every symbol include/mxgpu.h
declares, and compute calls fail loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from matrixextra_amd import _lib


def test_library_present_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.mx_abi_version() == 1


def test_every_declared_symbol_is_exported():
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in include/mxgpu.h but not exported: {missing}"


def test_hot_path_export_names_follow_the_reference():
    # one mx_* twin per _MatrixExtra_* routine of the hot path (src/RcppExports.cpp:2233-2242,2290-2298,2333-2343)
    names = set(_lib.declared_symbols())
    for n in ["matmul_dense_csc_numeric", "matmul_dense_csc_float32", "tcrossprod_dense_csr_numeric",
              "tcrossprod_dense_csr_float32", "tcrossprod_csr_dense_numeric", "tcrossprod_csr_dense_float32",
              "matmul_csr_dvec_numeric", "matmul_csr_dvec_integer", "matmul_csr_dvec_logical",
              "matmul_csr_dvec_float32", "check_is_seq", "check_is_rev_seq"]:
        assert "mx_" + n in names
    assert {"mx_csr_elemwise_begin", "mx_copy_csr_rows_begin", "mx_result_finish"} <= names


def test_no_undefined_non_runtime_symbols():
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    undefined = [l.split()[-1] for l in out.splitlines() if l.strip()]
    bad = [u for u in undefined if u.startswith(("mx", "_ZN2mx"))]
    assert not bad, bad


def test_product_does_not_reference_the_oracle():
    root = os.path.dirname(os.path.dirname(_lib.LIB_PATH))
    pkg = os.path.join(root, "matrixextra_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".R")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "mx_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


@pytest.mark.skipif(_lib.load() is not None and __import__("conftest")._have_gpu(), reason="GPU present")
def test_compute_fails_loudly_without_gpu():
    from matrixextra_amd import exports
    p = np.array([0, 1], dtype=np.int32)
    j = np.array([0], dtype=np.int32)
    x = np.array([1.0])
    with pytest.raises(_lib.MxError):
        exports.tcrossprod_csr_dense_numeric(p, j, x, np.ones((2, 1), order="F"))
    with pytest.raises(_lib.MxError):
        exports.add_csr_elemwise(p, p.copy(), j, j.copy(), x, x.copy(), False)
