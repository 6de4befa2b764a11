// svec.hip — CSR x sparse vector and CSR (.) dense elementwise for gfx950.
//
// Replaces:
//   matmul_csr_svec<>                  src/matmul.cpp:486-641   per-row sorted intersection dot product
//   multiply_csr_by_dense_elemwise<>   src/operators.cpp:239-334
// csr x svec: G lanes per row; every entry of the row looks its column up in the (small, L2-resident) index
// vector of y by binary search, the products are reduced with a butterfly.  NA rules as in SpMV.
// csr (.) dense: one dense read per entry at [row + nrows*col] (column-major, a gather by construction).
#include "mx_common.h"

namespace mx {

constexpr int SV_BLOCK = 256;

// KIND: 0 numeric, 1 integer, 2 logical, 3 binary, 4 float32
template <int G, int KIND>
__global__ __launch_bounds__(SV_BLOCK)
void csr_svec_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                     const double *__restrict__ values, const int32_t *__restrict__ yi, int ny,
                     const void *__restrict__ yv, double *__restrict__ out)
{
    const int lg = threadIdx.x % G;
    const long long row_ll = (long long)blockIdx.x * (SV_BLOCK / G) + threadIdx.x / G;
    const bool valid = row_ll < m;
    const int row = valid ? (int)row_ll : 0;
    int s = 0, e = 0;
    if (valid) { s = indptr[row]; e = indptr[row + 1]; }
    double acc = 0.0;
    int na = 0;
    for (int k = s + lg; k < e; k += G) {
        const int key = indices[k] + 1;                 // y's indices are 1-based (matmul.cpp:523)
        const int lb = lower_bound_dev(yi, ny, key);
        if (lb < ny && yi[lb] == key) {
            const double a = values[k];
            if constexpr (KIND == 0) acc = __builtin_fma(a, ((const double *)yv)[lb], acc);
            else if constexpr (KIND == 1) { const int y = ((const int32_t *)yv)[lb]; if (y == MX_NA_INT) na = 1; else acc = __builtin_fma(a, (double)y, acc); }
            else if constexpr (KIND == 2) { const int y = ((const int32_t *)yv)[lb]; if (y == MX_NA_INT) na = 1; else acc += a * (double)(y != 0); }
            else if constexpr (KIND == 3) acc += a;
            else acc = __builtin_fma(a, (double)((const float *)yv)[lb], acc);
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off, G);
        if constexpr (KIND == 1 || KIND == 2) na |= __shfl_xor(na, off, G);
    }
    if (valid && lg == 0) out[row] = na ? na_real() : acc;
}

// KIND: 0 double, 1 float32, 2 integer, 3 logical (f64 values -> f64), 4 logical AND (int32 -> int32)
template <int G, int KIND>
__global__ __launch_bounds__(SV_BLOCK)
void csr_by_dense_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                         const void *__restrict__ values, const void *__restrict__ dense, void *__restrict__ out)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (SV_BLOCK / G) + threadIdx.x / G;
    if (row >= m) return;
    const int s = indptr[row], e = indptr[row + 1];
    const size_t nr = (size_t)m;
    for (int k = s + lg; k < e; k += G) {
        const size_t at = (size_t)row + nr * (size_t)indices[k];
        if constexpr (KIND == 4) {
            ((int32_t *)out)[k] = r_logical_and(((const int32_t *)values)[k], ((const int32_t *)dense)[at]);
        } else {
            const double v = ((const double *)values)[k];
            double o;
            if constexpr (KIND == 0) o = v * ((const double *)dense)[at];
            else if constexpr (KIND == 1) o = v * (double)((const float *)dense)[at];
            else if constexpr (KIND == 2) { const int d = ((const int32_t *)dense)[at]; o = d == MX_NA_INT ? na_real() : v * (double)d; }
            else { const int d = ((const int32_t *)dense)[at]; o = d == MX_NA_INT ? na_real() : v * (double)(d != 0); }
            ((double *)out)[k] = o;
        }
    }
}

}  // namespace mx

#define MX_SV_G(KERNEL, KIND, ...)                                                                             \
    switch (G) {                                                                                               \
        case 4:  hipLaunchKernelGGL((mx::KERNEL<4, KIND>),  dim3((unsigned)mx::ceil_div(m, mx::SV_BLOCK / 4)),  dim3(mx::SV_BLOCK), 0, st, __VA_ARGS__); break; \
        case 8:  hipLaunchKernelGGL((mx::KERNEL<8, KIND>),  dim3((unsigned)mx::ceil_div(m, mx::SV_BLOCK / 8)),  dim3(mx::SV_BLOCK), 0, st, __VA_ARGS__); break; \
        case 16: hipLaunchKernelGGL((mx::KERNEL<16, KIND>), dim3((unsigned)mx::ceil_div(m, mx::SV_BLOCK / 16)), dim3(mx::SV_BLOCK), 0, st, __VA_ARGS__); break; \
        case 32: hipLaunchKernelGGL((mx::KERNEL<32, KIND>), dim3((unsigned)mx::ceil_div(m, mx::SV_BLOCK / 32)), dim3(mx::SV_BLOCK), 0, st, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((mx::KERNEL<64, KIND>), dim3((unsigned)mx::ceil_div(m, mx::SV_BLOCK / 64)), dim3(mx::SV_BLOCK), 0, st, __VA_ARGS__); break; \
    }

extern "C" int mxd_spmv_csr_svec(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                                 const int32_t *y_indices_base1, int ny, const void *y_values, int kind, double *out,
                                 void *stream)
{
    MX_REQUIRE(m >= 0 && ny >= 0 && kind >= 0 && kind <= 4, "mxd_spmv_csr_svec: bad arguments");
    if (m == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    if (ny == 0) { MX_HIP(hipMemsetAsync(out, 0, sizeof(double) * (size_t)m, st)); return 0; }     // matmul.cpp:495-496
    const int G = nnz < 0 ? 32 : mx::pick_group((double)nnz / (double)m);
    switch (kind) {
        case 0: MX_SV_G(csr_svec_kernel, 0, m, indptr, indices, values, y_indices_base1, ny, y_values, out); break;
        case 1: MX_SV_G(csr_svec_kernel, 1, m, indptr, indices, values, y_indices_base1, ny, y_values, out); break;
        case 2: MX_SV_G(csr_svec_kernel, 2, m, indptr, indices, values, y_indices_base1, ny, y_values, out); break;
        case 3: MX_SV_G(csr_svec_kernel, 3, m, indptr, indices, values, y_indices_base1, ny, y_values, out); break;
        default: MX_SV_G(csr_svec_kernel, 4, m, indptr, indices, values, y_indices_base1, ny, y_values, out); break;
    }
    MX_LAUNCH_CHECK();
    return 0;
}

extern "C" int mxd_csr_by_dense_elemwise(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                                         const void *values, const void *dense_colmajor, int kind, void *values_out,
                                         void *stream)
{
    MX_REQUIRE(m >= 0 && kind >= 0 && kind <= 4, "mxd_csr_by_dense_elemwise: bad arguments");
    if (m == 0 || nnz == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const int G = nnz < 0 ? 32 : mx::pick_group((double)nnz / (double)m);
    switch (kind) {
        case 0: MX_SV_G(csr_by_dense_kernel, 0, m, indptr, indices, values, dense_colmajor, values_out); break;
        case 1: MX_SV_G(csr_by_dense_kernel, 1, m, indptr, indices, values, dense_colmajor, values_out); break;
        case 2: MX_SV_G(csr_by_dense_kernel, 2, m, indptr, indices, values, dense_colmajor, values_out); break;
        case 3: MX_SV_G(csr_by_dense_kernel, 3, m, indptr, indices, values, dense_colmajor, values_out); break;
        default: MX_SV_G(csr_by_dense_kernel, 4, m, indptr, indices, values, dense_colmajor, values_out); break;
    }
    MX_LAUNCH_CHECK();
    return 0;
}
