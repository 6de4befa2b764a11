### mxgpu_overlay.R — route MatrixExtra's CSR hot path to the MI355X backend without rebuilding MatrixExtra.
###
### The S4 method registrations of the reference (R/matmul.R, R/operators.R, R/slice.R) and its R glue stay
### exactly as they are: `%*%`, `tcrossprod`, `+`, `-`, `*`, `&`, `|`, `[` on dgRMatrix objects dispatch
### unchanged.  The glue reaches native code only through the one-line wrappers of R/RcppExports.R
### (e.g. :148-150 `tcrossprod_csr_dense_numeric <- function(...) .Call(`_MatrixExtra_tcrossprod_csr_dense_numeric`, ...)`).
### This overlay rebinds the wrappers of the hot path and of its neighbours (SURVEY section 8f) inside the MatrixExtra namespace so that they `.Call` the routines of
### the same names registered by mxgpu_r.so (matrixextra_amd/csrc/r_shim.cpp), which forward to libmxgpu.so.
### Every other native routine (~140 of them) keeps pointing at MatrixExtra's own CPU code.
###
### Usage:
###   library(MatrixExtra)
###   source("mxgpu_overlay.R")
###   mxgpu_enable("/path/to/mxgpu_r.so")     # needs libmxgpu.so on the loader path and a visible MI355X
###   ... X %*% Y, X + Z, X[rows, ] ...       # now run on the GPU
###   mxgpu_disable()                         # restore the CPU wrappers
###
### NOT TESTED in the development image (no R there); see INTEGRATION.md.

.mxgpu_state <- new.env()

.mxgpu_hot_routines <- c(
    "matmul_dense_csc_numeric", "matmul_dense_csc_float32",
    "tcrossprod_dense_csr_numeric", "tcrossprod_dense_csr_float32",
    "tcrossprod_csr_dense_numeric", "tcrossprod_csr_dense_float32",
    "matmul_csr_dvec_numeric", "matmul_csr_dvec_integer", "matmul_csr_dvec_logical", "matmul_csr_dvec_float32",
    "multiply_csr_elemwise", "logicaland_csr_elemwise", "add_csr_elemwise", "logicalor_csr_elemwise",
    "copy_csr_rows_numeric", "copy_csr_rows_logical", "copy_csr_rows_binary",
    "check_is_seq", "check_is_rev_seq",
    ## SURVEY section 8(f) rank 2: column-filtering slices and reversals
    "copy_csr_rows_col_seq_numeric", "copy_csr_rows_col_seq_logical", "copy_csr_rows_col_seq_binary",
    "copy_csr_arbitrary_numeric", "copy_csr_arbitrary_logical", "copy_csr_arbitrary_binary",
    "reverse_rows_numeric", "reverse_rows_logical", "reverse_rows_binary",
    "reverse_columns_inplace_numeric", "reverse_columns_inplace_logical", "reverse_columns_inplace_binary",
    ## rank 4: values-only CSR (op) vector, CSR x sparse vector, CSR (.) dense
    "multiply_csr_by_dvec_no_NAs_numeric", "logicaland_csr_by_dvec_internal", "multiply_csr_by_dvec_with_NAs",
    "matmul_csr_svec_numeric", "matmul_csr_svec_integer", "matmul_csr_svec_logical", "matmul_csr_svec_binary",
    "matmul_csr_svec_float32",
    "multiply_csr_by_dense_elemwise_double", "multiply_csr_by_dense_elemwise_float32",
    "multiply_csr_by_dense_elemwise_int", "multiply_csr_by_dense_elemwise_bool", "logicaland_csr_by_dense_cpp",
    ## rank 3: cbind / rbind
    "cbind_csr_numeric", "cbind_csr_logical", "cbind_csr_binary", "concat_csr_batch",
    ## rank 1: sortedness check and in-place sort (run before every CSR (+) CSR, R/operators.R:58,64,748,754)
    "check_indices_are_unsorted",
    "sort_sparse_indices_numeric", "sort_sparse_indices_logical", "sort_sparse_indices_numeric_known_ncol",
    "sort_sparse_indices_logical_known_ncol", "sort_sparse_indices_binary",
    ## what follows a merge (`remove_zeros`, R/utils.R:286-312) and what `check_sparse_matrix` asks (R/utils.R:448)
    "remove_zero_valued_csr_numeric", "remove_zero_valued_csr_logical", "check_valid_csr_matrix",
    ## float32 row vector %*% CsparseMatrix (R/matmul.R:243-259, 350-365, 411-425)
    "matmul_rowvec_by_csc", "matmul_rowvec_by_cscbin"
)

## From which operand size on a call goes to the GPU (the length of the longest vector passed: the entry count, or the
## dense operand).  The same numbers as the C-ABI's mx_should_offload (include/mxgpu.h; the shim applies them once more
## to direct .Call users and falls back to MatrixExtra's own native routine).  Measured on an MI355X box against MatrixExtra's algorithm on the host's cores (tools/small_calls.py,
## profiles/r04_small_calls.json: host arrays in, host arrays out): one call costs 29-50 us at the reference's own test
## sizes (100 x 50) whatever it does, so products and CSR (+) CSR win from ~5e4 entries on, `X %*% v` (8 bytes of result per
## row against 12 bytes per entry over PCIe) from ~1e6, and `X[rows, ]` — a memcpy on the host — only from ~1e7.
.mxgpu_default_min_len <- function(fn) {
    if (startsWith(fn, "matmul_csr_dvec_") || startsWith(fn, "matmul_csr_svec_") || startsWith(fn, "matmul_rowvec_by_")) return(1000000L)
    if (startsWith(fn, "copy_csr_") || startsWith(fn, "reverse_") || startsWith(fn, "cbind_") || fn == "concat_csr_batch") return(10000000L)
    50000L
}

## min_nnz: NULL = the measured defaults above; 0 = every call goes to the GPU; a number = that threshold for every routine
mxgpu_enable <- function(shim_path, min_nnz = NULL) {
    dll <- dyn.load(shim_path)
    ## options of the backend (include/mxgpu.h, mx_set_option / mx_set_devices), taken from R options at enable time:
    ##   options(MatrixExtra.mxgpu_spmv_planned = TRUE)   `X %*% v` on a matrix still on the device may use the planned
    ##                                                    kernel (faster; equal to 1e-12, not bit for bit: see mxgpu.h)
    ##   options(MatrixExtra.mxgpu_devices = 0:7)         large CSR x dense products sharded by rows over these GPUs
    if (isTRUE(getOption("MatrixExtra.mxgpu_spmv_planned", FALSE)))
        .Call(getNativeSymbolInfo("_mxgpu_set_option", dll), "spmv_planned", 1L)
    devs <- getOption("MatrixExtra.mxgpu_devices", NULL)
    if (!is.null(devs))
        .Call(getNativeSymbolInfo("_mxgpu_set_devices", dll), as.integer(devs))
    ns <- asNamespace("MatrixExtra")
    .mxgpu_state$saved <- list()
    for (fn in .mxgpu_hot_routines) {
        cpu_fun <- get(fn, envir = ns)
        .mxgpu_state$saved[[fn]] <- cpu_fun
        native <- getNativeSymbolInfo(paste0("_MatrixExtra_", fn), dll)
        gpu_fun <- local({
            native <- native; cpu_fun <- cpu_fun
            min_nnz <- if (is.null(min_nnz)) .mxgpu_default_min_len(fn) else as.integer(min_nnz)
            function(...) {
                ## small operands are cheaper on the host than a PCIe round trip: size gate on the length of the
                ## longest vector passed (0 = always use the GPU)
                args <- list(...)
                if (min_nnz > 0L) {
                    lens <- vapply(args, length, integer(1L))
                    if (max(lens) < min_nnz) return(do.call(cpu_fun, args))
                }
                do.call(.Call, c(list(native), args))
            }
        })
        formals_cpu <- formals(cpu_fun)
        unlockBinding(fn, ns)
        assign(fn, gpu_fun, envir = ns)
        lockBinding(fn, ns)
    }
    invisible(TRUE)
}

mxgpu_disable <- function() {
    ns <- asNamespace("MatrixExtra")
    for (fn in names(.mxgpu_state$saved)) {
        unlockBinding(fn, ns)
        assign(fn, .mxgpu_state$saved[[fn]], envir = ns)
        lockBinding(fn, ns)
    }
    .mxgpu_state$saved <- list()
    invisible(TRUE)
}
