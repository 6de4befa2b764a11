// bind.hip — cbind / rbind of CSR matrices for gfx950.
//
// Replaces:
//   cbind_csr<>          src/cbind.cpp:4-99     per-row interleave of two CSR operands
//   concat_indptr2       src/rbind.cpp:9-21
//   concat_csr_batch     src/rbind.cpp:24-173   bulk copy + indptr offset (+ value-type conversion)
// Pure bandwidth copies: 2*12 bytes per entry + 8 bytes per row.  The output row offsets of cbind are
// indptrX[r] + indptrY[r] — no scan is needed.
#include "mx_common.h"

namespace mx {

constexpr int BIND_BLOCK = 256;

template <int G, typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(BIND_BLOCK)
void cbind_kernel(int nX, int nY, const int32_t *__restrict__ Xp, const int32_t *__restrict__ Xj, const VT *__restrict__ Xx,
                  const int32_t *__restrict__ Yp, const int32_t *__restrict__ Yj, const VT *__restrict__ Yx,
                  int32_t *__restrict__ indptr, int32_t *__restrict__ indices, VT *__restrict__ values)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (BIND_BLOCK / G) + threadIdx.x / G;
    const int nrows = max(nX, nY);
    if (row >= nrows) return;
    // rows past the end of an operand contribute nothing and sit after all of its entries
    const int xs = Xp[min((long long)nX, row)], xe = Xp[min((long long)nX, row + 1)];
    const int ys = Yp[min((long long)nY, row)], ye = Yp[min((long long)nY, row + 1)];
    const int o = xs + ys, lx = xe - xs, ly = ye - ys;
    if (lg == 0) {
        indptr[row + 1] = xe + ye;
        if (row == 0) indptr[0] = 0;
    }
    for (int k = lg; k < lx; k += G) {
        indices[o + k] = Xj[xs + k];
        if constexpr (HAS_VALUES) values[o + k] = Xx[xs + k];
    }
    for (int k = lg; k < ly; k += G) {
        indices[o + lx + k] = Yj[ys + k];
        if constexpr (HAS_VALUES) values[o + lx + k] = Yx[ys + k];
    }
}

__global__ __launch_bounds__(256)
void indptr_offset_kernel(const int32_t *__restrict__ src, int n, int offset, int32_t *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = offset + src[i];
}

// value conversion of concat_csr_batch (rbind.cpp:72-166).  in_kind: 0 f64, 1 R logical, 2 none, 4 R integer;
// out_kind: 0 f64, 1 R logical.  add = index shift (-1 for the 1-based sparse-vector inputs).
template <int IN, int OUT>
__global__ __launch_bounds__(256)
void concat_entries_kernel(int64_t n, const int32_t *__restrict__ idx_in, const void *__restrict__ val_in, int add,
                           int32_t *__restrict__ idx_out, void *__restrict__ val_out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        idx_out[i] = idx_in[i] + add;
        if constexpr (OUT == 0) {
            double v;
            if constexpr (IN == 0) v = ((const double *)val_in)[i];
            else if constexpr (IN == 1) { const int l = ((const int32_t *)val_in)[i]; v = l == MX_NA_INT ? na_real() : (add ? (double)(l != 0) : (double)l); }
            else if constexpr (IN == 4) { const int l = ((const int32_t *)val_in)[i]; v = l == MX_NA_INT ? na_real() : (double)l; }
            else v = 1.0;
            ((double *)val_out)[i] = v;
        } else if constexpr (OUT == 1) {
            int v;
            if constexpr (IN == 0) { const double d = ((const double *)val_in)[i]; v = d != d ? MX_NA_INT : (d != 0.0); }
            else if constexpr (IN == 1) v = ((const int32_t *)val_in)[i];
            else if constexpr (IN == 4) { const int l = ((const int32_t *)val_in)[i]; v = l == MX_NA_INT ? MX_NA_INT : (l != 0); }
            else v = 1;
            ((int32_t *)val_out)[i] = v;
        }
    }
}

}  // namespace mx

extern "C" int mxd_csr_cbind(int nX, int nY, const int32_t *Xp, const int32_t *Xj, const void *Xx, const int32_t *Yp,
                             const int32_t *Yj_plus_ncol, const void *Yx, int value_dtype, int64_t nnz_total,
                             int32_t *indptr, int32_t *indices, void *values, void *stream)
{
    MX_REQUIRE(nX >= 0 && nY >= 0, "mxd_csr_cbind: negative size");
    const int nrows = nX > nY ? nX : nY;
    hipStream_t st = mx::as_stream(stream);
    if (nrows == 0) { MX_HIP(hipMemsetAsync(indptr, 0, sizeof(int32_t), st)); return 0; }
    const int G = mx::pick_group(0.5 * (double)nnz_total / (double)nrows);
#define MX_CB(GG, VT, HV)                                                                                       \
    hipLaunchKernelGGL((mx::cbind_kernel<GG, VT, HV>), dim3((unsigned)mx::ceil_div(nrows, mx::BIND_BLOCK / GG)), \
                       dim3(mx::BIND_BLOCK), 0, st, nX, nY, Xp, Xj, (const VT *)Xx, Yp, Yj_plus_ncol,           \
                       (const VT *)Yx, indptr, indices, (VT *)values)
#define MX_CB_G(VT, HV)                                                                                         \
    switch (G) { case 4: MX_CB(4, VT, HV); break; case 8: MX_CB(8, VT, HV); break; case 16: MX_CB(16, VT, HV); break; \
                 case 32: MX_CB(32, VT, HV); break; default: MX_CB(64, VT, HV); break; }
    switch (value_dtype) {
        case MX_F64: MX_CB_G(double, true); break;
        case MX_LGL: case MX_I32: MX_CB_G(int32_t, true); break;
        case MX_NONE: MX_CB_G(int32_t, false); break;
        default: return mx::set_error("mxd_csr_cbind: unsupported value dtype %d", value_dtype);
    }
#undef MX_CB_G
#undef MX_CB
    MX_LAUNCH_CHECK();
    return 0;
}

// Appends one operand of an rbind at (row_offset, entry_offset) of the output arrays.
// in_kind: 0 dgRMatrix, 1 lgRMatrix, 2 ngRMatrix, 3 dsparseVector, 4 isparseVector, 5 lsparseVector,
// 6 nsparseVector (vectors: 1-based indices, one row, indptr_in ignored); out_kind 0 dgR, 1 lgR, 2 ngR.
// out_indptr[row_offset] must already hold entry_offset (out_indptr[0] = 0 is written when row_offset == 0).
extern "C" int mxd_csr_rbind_append(int in_kind, const int32_t *indptr_in, const int32_t *indices_in, const void *values_in,
                                    int nrows_in, int64_t nnz_in, int out_kind, int row_offset, int64_t entry_offset,
                                    int32_t *out_indptr, int32_t *out_indices, void *out_values, void *stream)
{
    MX_REQUIRE(in_kind >= 0 && in_kind <= 6 && out_kind >= 0 && out_kind <= 2, "mxd_csr_rbind_append: bad kind");
    MX_REQUIRE(entry_offset + nnz_in <= (int64_t)INT_MAX, "rbind result exceeds R's int32 index range");
    hipStream_t st = mx::as_stream(stream);
    const bool vec = in_kind >= 3;
    if (row_offset == 0) MX_HIP(hipMemsetAsync(out_indptr, 0, sizeof(int32_t), st));
    if (vec) {
        const int32_t end = (int32_t)(entry_offset + nnz_in);
        MX_HIP(hipMemcpyAsync(out_indptr + row_offset + 1, &end, sizeof(int32_t), hipMemcpyHostToDevice, st));
        MX_HIP(hipStreamSynchronize(st));          // `end` is a stack temporary
    } else if (nrows_in > 0) {
        hipLaunchKernelGGL(mx::indptr_offset_kernel, dim3((unsigned)mx::ceil_div(nrows_in, 256)), dim3(256), 0, st,
                           indptr_in + 1, nrows_in, (int)entry_offset, out_indptr + row_offset + 1);
        MX_LAUNCH_CHECK();
    }
    if (nnz_in == 0) return 0;
    const unsigned grid = (unsigned)(mx::ceil_div(nnz_in, 256) < 4096 ? mx::ceil_div(nnz_in, 256) : 4096);
    int32_t *jo = out_indices + entry_offset;
    const int add = vec ? -1 : 0;
    // value kind of the input: 0 f64, 1 logical, 2 none, 4 integer
    const int vin = (in_kind == 0 || in_kind == 3) ? 0 : (in_kind == 1 || in_kind == 5) ? 1 : (in_kind == 4) ? 4 : 2;
    void *vo = out_kind == 0 ? (void *)((double *)out_values + entry_offset)
             : out_kind == 1 ? (void *)((int32_t *)out_values + entry_offset) : nullptr;
#define MX_CE(IN, OUT) hipLaunchKernelGGL((mx::concat_entries_kernel<IN, OUT>), dim3(grid), dim3(256), 0, st, nnz_in, \
                                          indices_in, values_in, add, jo, vo)
#define MX_CE_IN(OUT) switch (vin) { case 0: MX_CE(0, OUT); break; case 1: MX_CE(1, OUT); break; case 4: MX_CE(4, OUT); break; default: MX_CE(2, OUT); break; }
    if (out_kind == 0) { MX_CE_IN(0) } else if (out_kind == 1) { MX_CE_IN(1) } else { MX_CE_IN(2) }
#undef MX_CE_IN
#undef MX_CE
    MX_LAUNCH_CHECK();
    return 0;
}
