// spmv.hip — CSR x dense-vector SpMV for gfx950.
//
// Replaces matmul_csr_dvec<> (src/matmul.cpp:381-419; exports :421-483):
//   numeric / integer / logical right-hand sides -> f64 result,
//   float32 right-hand side -> f32 result with float accumulation (:403).
// NA_INTEGER / NA_LOGICAL entries contribute NA_REAL (:406-411); a logical
// entry counts as (bool)y (:411).
//
// Design: G lanes of a wavefront own one row (G = power of two picked from the
// mean row length, 64 = one wavefront per row); lanes stride the row's entries
// (coalesced (j, a) reads, gathered v[j] reads served by L2 — v is 0.8 MB for
// the headline config), then a butterfly __shfl_xor reduction inside the group.
// HBM-bound: algorithmic bytes = 4(m+1) + 12 nnz + s*K + s*m.
#include "mx_common.h"

namespace mx {

constexpr int SPMV_BLOCK = 256;

template <int G, int KIND>
__global__ __launch_bounds__(SPMV_BLOCK)
void spmv_group_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                       const double *__restrict__ values, const void *__restrict__ v_, void *__restrict__ y_)
{
    const int lg = threadIdx.x % G;
    const long long row_ll = (long long)blockIdx.x * (SPMV_BLOCK / G) + threadIdx.x / G;
    const bool valid = row_ll < m;
    const int row = valid ? (int)row_ll : 0;
    int s = 0, e = 0;
    if (valid) { s = indptr[row]; e = indptr[row + 1]; }

    double acc = 0.0;
    float accf = 0.0f;
    int na = 0;
    for (int k = s + lg; k < e; k += G) {
        const int j = indices[k];
        const double a = values[k];
        if constexpr (KIND == MX_F64) {
            acc = __builtin_fma(a, ((const double *)v_)[j], acc);
        } else if constexpr (KIND == MX_I32) {
            const int yv = ((const int32_t *)v_)[j];
            if (yv == MX_NA_INT) na = 1; else acc = __builtin_fma(a, (double)yv, acc);
        } else if constexpr (KIND == MX_LGL) {
            const int yv = ((const int32_t *)v_)[j];
            if (yv == MX_NA_INT) na = 1; else acc += a * (double)(yv != 0);
        } else {
            // float accumulator, double product: val += x * y with float val (matmul.cpp:403,413)
            accf = (float)((double)accf + a * (double)((const float *)v_)[j]);
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        if constexpr (KIND == MX_F32) accf += __shfl_xor(accf, off, G);
        else {
            acc += __shfl_xor(acc, off, G);
            if constexpr (KIND == MX_I32 || KIND == MX_LGL) na |= __shfl_xor(na, off, G);
        }
    }
    if (valid && lg == 0) {
        if constexpr (KIND == MX_F32) ((float *)y_)[row] = accf;
        else ((double *)y_)[row] = na ? na_real() : acc;
    }
}

template <int KIND>
static int launch_spmv(int G, int m, const int32_t *indptr, const int32_t *indices, const double *values,
                       const void *v, void *y, hipStream_t st)
{
#define MX_SPMV_CASE(GG)                                                                        \
    case GG: {                                                                                  \
        const unsigned grid = (unsigned)ceil_div(m, SPMV_BLOCK / GG);                           \
        hipLaunchKernelGGL((spmv_group_kernel<GG, KIND>), dim3(grid), dim3(SPMV_BLOCK), 0, st,  \
                           m, indptr, indices, values, v, y);                                   \
        break;                                                                                  \
    }
    switch (G) {
        MX_SPMV_CASE(4) MX_SPMV_CASE(8) MX_SPMV_CASE(16) MX_SPMV_CASE(32) MX_SPMV_CASE(64)
        default: return set_error("spmv: bad group size %d", G);
    }
#undef MX_SPMV_CASE
    MX_LAUNCH_CHECK();
    return 0;
}

// nnz is only used to pick the group width; pass <0 when unknown (=> 32 lanes per row)
int spmv_launch(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                const void *v, int v_dtype, void *y, hipStream_t st)
{
    const int G = nnz < 0 ? 32 : pick_group((double)nnz / (double)(m > 0 ? m : 1));
    switch (v_dtype) {
        case MX_F64: return launch_spmv<MX_F64>(G, m, indptr, indices, values, v, y, st);
        case MX_I32: return launch_spmv<MX_I32>(G, m, indptr, indices, values, v, y, st);
        case MX_LGL: return launch_spmv<MX_LGL>(G, m, indptr, indices, values, v, y, st);
        case MX_F32: return launch_spmv<MX_F32>(G, m, indptr, indices, values, v, y, st);
        default: return set_error("spmv: unsupported vector dtype %d", v_dtype);
    }
}

}  // namespace mx

extern "C" int mxd_spmv_csr_dvec(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                                 const void *v, int v_dtype, void *y, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_spmv_csr_dvec: negative m");
    if (m == 0) return 0;
    MX_REQUIRE(indptr && y, "mxd_spmv_csr_dvec: null pointer");
    return mx::spmv_launch(m, nnz, indptr, indices, values, v, v_dtype, y, mx::as_stream(stream));
}
