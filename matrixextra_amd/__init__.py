"""matrixextra_amd — MI355X (gfx950) backend for MatrixExtra's CSR hot path.

Layers (SURVEY.md §1 / DESIGN.md):
  csrc/*.hip + include/mxgpu.h   hand-written HIP kernels behind a C-ABI (libmxgpu.so)
  exports.py                     twins of R/RcppExports.R wrappers (ctypes -> C-ABI)
  matrices.py                    dgRMatrix / lgRMatrix / ngRMatrix / dgCMatrix / float32 stand-ins
  matmul.py operators.py slice.py   mirrors of the R glue (checks, messages, dimnames, classes)
  device.py                      device-resident CSR + mxd_* launches on torch tensors (bench, multi-GPU)
  distributed.py                 row-block sharding + RCCL all-gather of C

No CPU fallback anywhere: without libmxgpu.so and a GPU every compute call raises.
"""
from . import _lib  # noqa: F401
from .matrices import (DenseMatrix, MatrixExtraError, NA_INTEGER, NA_LOGICAL, NA_REAL, RsparseMatrix,  # noqa: F401
                       as_csr_matrix, check_sparse_matrix, check_valid_matrix, dgCMatrix, dgRMatrix, float32, from_scipy,
                       lgRMatrix, ngRMatrix, options, remove_sparse_zeros, sort_sparse_indices)
from .matmul import RLogical, crossprod, tcrossprod  # noqa: F401  (`%*%` is the @ operator)
from .operators import (add_csr_matrices, logicalor_csr_matrices, multiply_csr_by_csr,  # noqa: F401
                        xor_csr_matrices)
from .slice import subset_csr  # noqa: F401

__version__ = "0.1.0"
