"""tools/fit_auto_model.py restates AUTO's cost model (csrc/spmm.hip spmm_auto_cost, csrc/spmm_rowsplit.hip rowsplit_est_us /
rowsplit_panels / rowsplit_segments) in Python to replay it over the measured map; this pins the restatement to the shipped
code: mxd_spmm_auto_cost is host arithmetic only (no GPU needed), and both must give the same estimates and panel counts."""
import ctypes as C
import os
import sys

import pytest

from matrixextra_amd import _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_python_restatement_matches_the_library():
    import fit_auto_model as F
    lib = _lib.load()
    n_checked = 0
    for m in (40_000, 100_000, 1_000_000):
        for K in (10_000, 100_000):
            for per_row in (8, 32, 128, 500):
                for n in (16, 32, 64, 100, 256):
                    for sz, dt in ((8, _lib.MX_F64), (4, _lib.MX_F32)):
                        nnz = m * per_row
                        for keep in (0, 1):
                            a, b, P = C.c_double(), C.c_double(), C.c_int()
                            _lib.check(lib.mxd_spmm_auto_cost(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(nnz), C.c_int(keep), C.c_int(dt),
                                                              C.byref(a), C.byref(b), C.byref(P)))
                            rs, pl, pn, _ = F.cost2(m, n, K, nnz, sz, bool(keep), F.SHIPPED)
                            assert pn == P.value, (m, K, per_row, n, sz, keep, pn, P.value)
                            assert rs == pytest.approx(a.value, rel=1e-9), (m, K, per_row, n, sz, keep, rs, a.value)
                            assert pl == pytest.approx(b.value, rel=1e-9), (m, K, per_row, n, sz, keep, pl, b.value)
                            n_checked += 1
    assert n_checked == 3 * 2 * 4 * 5 * 2 * 2
