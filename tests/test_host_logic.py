"""Host-side mirror of the R glue: argument checks, messages, index canonicalisation — no GPU needed
(every case here stops before any compute call, as the reference's R code does)."""
import numpy as np
import pytest

import matrixextra_amd as mx
from matrixextra_amd import matmul, operators, slice as mslice
from matrixextra_amd.matrices import MatrixExtraError, check_valid_matrix


def _csr(m=4, K=5):
    return mx.dgRMatrix([0, 1, 4, 5, 6][: m + 1], [4, 1, 2, 4, 1, 0], [-0.91, 0.14, -0.12, -0.12, 1.1, 0.66], (m, K))


def test_dimension_mismatch_messages():
    X = _csr()
    with pytest.raises(MatrixExtraError, match="Matrix dimensions do not match."):       # R/matmul.R:145
        X @ np.ones((4, 3))
    with pytest.raises(MatrixExtraError, match="Matrix dimensions do not match."):
        mx.tcrossprod(X, np.ones((3, 4)))
    with pytest.raises(MatrixExtraError, match="Matrix-vector dimensions do not match."):  # R/matmul.R:547
        X @ np.ones(4)
    Y = mx.dgRMatrix([0, 0, 0, 0, 0], [], [], (4, 6))
    with pytest.raises(MatrixExtraError, match="in order to add/substract them"):         # R/operators.R:716
        X + Y
    with pytest.raises(MatrixExtraError, match="in order to multiply them"):              # R/operators.R:45
        X * Y


def test_check_valid_matrix_negatives():
    # tests/testthat/test-utilities.R:63-70 style structural negatives that R/utils.R:383-392 catches
    X = _csr()
    check_valid_matrix(X)
    bad = _csr(); bad.p = np.array([0, 1, 4, 5, 100], dtype=np.int32)
    with pytest.raises(MatrixExtraError, match="bad start/end"):
        check_valid_matrix(bad)
    bad = _csr(); bad.p = np.array([0, 1, 4, 5], dtype=np.int32)
    with pytest.raises(MatrixExtraError, match="doesn't match with dimension"):
        check_valid_matrix(bad)
    bad = _csr(); bad.x = bad.x[:-1]
    with pytest.raises(MatrixExtraError, match="lengths of indices and values differ"):
        check_valid_matrix(bad)


def test_get_indices_integer():
    g = mslice.get_indices_integer
    assert g([1, 3, 3], 5, None).tolist() == [1, 3, 3]
    assert g([-1, -5], 5, None).tolist() == [2, 3, 4]                 # negative = exclusion (R/slice.R:48-49)
    assert g(np.array([True, False, True, False, True]), 5, None).tolist() == [1, 3, 5]
    assert g(np.array([True, False]), 5, None).tolist() == [1, 3, 5]  # recycled mask
    assert g(["b", "a"], 3, ["a", "b", "c"]).tolist() == [2, 1]
    with pytest.raises(MatrixExtraError, match="not present in matrix"):   # R/slice.R:50-51
        g([1, 6], 5, None)


def test_slice_paths_that_never_reach_native_code():
    X = _csr()
    X.Dimnames = [list("abcd"), None]
    assert mx.subset_csr(X, None) is X
    e = mx.subset_csr(X, np.zeros(0, dtype=np.int32))
    assert e.Dim == (0, 5) and e.p.tolist() == [0] and isinstance(e, mx.dgRMatrix)
    Z = mx.dgRMatrix([0, 0, 0], [], [], (2, 3))
    e = mx.subset_csr(Z, [2, 1, 2])                                    # no entries at all: R/slice.R:404-421
    assert e.Dim == (3, 3) and e.p.tolist() == [0, 0, 0, 0]


def test_ngRMatrix_identity_shortcuts():
    # R/operators.R:47-50, :720-740 — decided on the host from pointer identity, no kernel involved
    p = np.array([0, 1, 2], dtype=np.int32); j = np.array([0, 1], dtype=np.int32)
    A = mx.ngRMatrix(p, j, None, (2, 2)); B = mx.ngRMatrix(p, j, None, (2, 2))
    assert operators.multiply_csr_by_csr(A, B) is A
    assert operators.add_csr_matrices(A, B) is A
    x = operators.xor_csr_matrices(A, B)
    assert isinstance(x, mx.lgRMatrix) and x.p.tolist() == [0, 0, 0] and x.j.size == 0
    s = operators.add_csr_matrices(A, B, True)
    assert isinstance(s, mx.dgRMatrix) and s.x.tolist() == [2.0, 2.0]


def test_partition_rows_balances_cost_and_covers_all_rows():
    """mx_partition_rows (what the sharded exports cut the sparse operand with): host arithmetic only."""
    import ctypes as C
    from matrixextra_amd import _lib, synth
    lib = _lib.load()
    for (m, K, mean, sigma) in ((50_000, 4000, 20, 1.3), (1000, 50, 3, 0.5), (7, 100, 5, 0.1)):
        p, j, x = synth.csr_skewed(m, K, mean, seed=m, sigma=sigma)
        for nparts in (1, 2, 3, 8):
            cuts = (C.c_int * (nparts + 1))()
            _lib.check(lib.mx_partition_rows(p.ctypes.data_as(C.c_void_p), C.c_int(m), C.c_int(nparts), C.c_int(128),
                                             C.c_int(8), cuts))
            cuts = list(cuts)
            assert cuts[0] == 0 and cuts[-1] == m and all(a <= b for a, b in zip(cuts, cuts[1:]))
            cost = [12.0 * (p[b] - p[a]) + 128 * 8.0 * (b - a) for a, b in zip(cuts, cuts[1:])]
            longest = 12.0 * np.diff(p).max() + 128 * 8.0
            assert max(cost) <= sum(cost) / nparts + longest          # no part exceeds its share by more than one row
    # an all-empty matrix is cut by rows
    p0 = np.zeros(101, dtype=np.int32)
    cuts = (C.c_int * 5)()
    _lib.check(lib.mx_partition_rows(p0.ctypes.data_as(C.c_void_p), C.c_int(100), C.c_int(4), C.c_int(16), C.c_int(4), cuts))
    assert list(cuts) == [0, 25, 50, 75, 100]


def test_profile_guard_hashes_the_file_a_kernel_is_built_from():
    """tools/prof_common.sources_of: every kernel name that appears in the committed rocprofv3 summaries maps to the
    source file that defines it (bench.py quotes a summary's counters only while those files are unchanged) — in
    particular the planned SpMV kernel, whose name contains the generic `plan_` of the SpMM plan's build kernels."""
    import csv
    import glob
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import prof_common as PC
    assert PC.sources_of("mx::spmv_plan_kernel<64>(int, ...)")[0] == "spmv_plan.hip"
    assert PC.sources_of("mx::spmm_plan_kernel<double,true,16,false>")[0] == "spmm_plan.hip"
    assert PC.sources_of("void mx::plan_count_kernel(...)")[0] == "spmm_plan.hip"
    assert PC.sources_of("mx::spmm_tile_kernel<double, 1, 3, 2, 65536, false>")[0] == "spmm_tile.hip"
    csrc = os.path.join(root, "matrixextra_amd", "csrc")
    seen = 0
    for f in glob.glob(os.path.join(root, "profiles", "r0[45]_*kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            name = row.get("Name") or row.get("KernelName") or ""
            m = re.search(r"mx::(\w+)", name)
            if not m:
                continue
            files = PC.sources_of(name)
            assert files is not None, name
            text = open(os.path.join(csrc, files[0])).read()
            assert re.search(r"\b%s\b" % re.escape(m.group(1)), text), (name, files[0])
            seen += 1
    assert seen > 10
