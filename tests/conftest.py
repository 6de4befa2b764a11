import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from matrixextra_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path or fail — never skip silently on a GPU box."""
    from matrixextra_amd import _lib
    n = _lib.device_count()          # raises MxError when the library or the device is missing
    assert n > 0
    return _lib


# ---------------------------------------------------------------------------------------------
# small random CSR builders shared by the tests (numpy RNG; R's rsparsematrix is not available)
def rand_csr(m, K, density, seed, sorted_cols=True, dtype="d", empty_rows=()):
    rng = np.random.default_rng(seed)
    mask = rng.random((m, K)) < density
    for r in empty_rows:
        mask[r, :] = False
    indptr = np.zeros(m + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(mask.sum(axis=1))
    rows, cols = np.nonzero(mask)
    indices = cols.astype(np.int32)
    if not sorted_cols:
        for r in range(m):
            s, e = indptr[r], indptr[r + 1]
            indices[s:e] = rng.permutation(indices[s:e])
    nnz = indices.size
    if dtype == "d":
        values = np.round(rng.normal(size=nnz), 2)
        values[values == 0] = 0.5
    elif dtype == "l":
        values = rng.choice(np.array([0, 1, -2147483648], dtype=np.int32), size=nnz, p=[0.2, 0.6, 0.2])
    else:
        values = None
    return indptr, indices, values


def csr_to_dense(indptr, indices, values, K):
    m = indptr.size - 1
    out = np.zeros((m, K))
    for r in range(m):
        for k in range(indptr[r], indptr[r + 1]):
            out[r, indices[k]] += 1.0 if values is None else values[k]
    return out
