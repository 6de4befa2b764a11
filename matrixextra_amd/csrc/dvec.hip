// CSR (op) dense vector with R's recycling rules — values-only transforms (second half).
//
// Replaces:
//   multiply_csr_by_dvec_no_NAs<>            src/operators.cpp:1604-2140   (* ^ / %% %/% with X on either side)
//   multiply_csr_by_dvec_no_NAs_numeric      src/operators.cpp:2142-2175
//   logicaland_csr_by_dvec_internal          src/operators.cpp:2177-2200
// The reference has four branches for the length of the vector (== nrows :1640, >= nrows*ncols :1773,
// divides nrows :1870, anything else :2033); all four read dvec[(row + col*nrows) mod length]
// (`recyle_pos`, :1478), which is what the kernel computes — per row when the position does not depend on the
// column, without the modulo when the vector covers the whole matrix, with a 64-bit modulo otherwise.
// R's arithmetic (R_pow, R_modulus = %%, R_intdiv = %/%, :1482-1590) is restated for the device; where the
// reference carries an intermediate in `long double` (x87, 64-bit mantissa) the device uses one fused
// multiply-add (exact product, one rounding): results agree to the last bit or two, not always bit for bit.
// Structure-changing variants (multiply_csr_by_dvec_with_NAs, :2258-) are not built: they stay on the CPU.
#include "mx_common.h"

namespace mx {

constexpr int DV_BLOCK = 256;
constexpr double DV_LD_EPS = 1.0842021724855044e-19;       // LDBL_EPSILON of the x87 format the reference compiles with

__device__ __forceinline__ double dv_nan() { return __builtin_nan(""); }

// R_modulus, src/operators.cpp:1526-1540 (R's myfmod)
__device__ __forceinline__ double r_modulus(double x1, double x2)
{
    if (x2 == 0.0) return dv_nan();
    if (fabs(x2) * DV_LD_EPS > 1 && isfinite(x1) && fabs(x1) <= fabs(x2))
        return (fabs(x1) == fabs(x2)) ? 0 : (((x1 < 0 && x2 > 0) || (x2 < 0 && x1 > 0)) ? x1 + x2 : x1);
    const double q = x1 / x2;
    const double tmp = __builtin_fma(-floor(q), x2, x1);
    return __builtin_fma(-floor(tmp / x2), x2, tmp);
}

// R_intdiv, src/operators.cpp:1500-1513 (R's myfloor)
__device__ __forceinline__ double r_intdiv(double x1, double x2)
{
    const double q = x1 / x2;
    if (x2 == 0.0 || fabs(q) * DV_LD_EPS > 1 || !isfinite(q)) return q;
    if (fabs(q) < 1) return (q < 0) ? -1 : (((x1 < 0 && x2 > 0) || (x1 > 0 && x2 < 0)) ? -1 : 0);
    const double fq = floor(q);
    const double tmp = __builtin_fma(-fq, x2, x1);
    return fq + floor(tmp / x2);
}

// R_pow of R's C API (arithmetic.c; the semantics are quoted at src/operators.cpp:1555-1601)
__device__ __forceinline__ double r_pow(double x, double y)
{
    if (y == 2.0) return x * x;
    if (x == 1. || y == 0.) return 1.;
    if (x == 0.) {
        if (y > 0.) return 0.;
        else if (y < 0) return __builtin_inf();
        else return y;                                       // NA or NaN
    }
    if (isfinite(x) && isfinite(y)) return pow(x, y);
    if (isnan(x) || isnan(y)) return x + y;
    if (!isfinite(x)) {
        if (x > 0) return (y < 0.) ? 0. : __builtin_inf();   // Inf ^ y
        else if (isfinite(y) && y == floor(y))               // (-Inf) ^ n
            return (y < 0.) ? 0. : (r_modulus(y, 2.) != 0 ? x : -x);
    }
    if (!isfinite(y)) {
        if (x >= 0) {
            if (y > 0) return (x >= 1) ? __builtin_inf() : 0.;
            else return (x < 1) ? __builtin_inf() : 0.;
        }
    }
    return dv_nan();
}

__device__ __forceinline__ double dv_apply(int op, bool lhs, double x, double d)
{
    switch (op) {
        case MX_DV_MULTIPLY: return x * d;
        case MX_DV_DIVIDE:   return lhs ? x / d : d / x;
        case MX_DV_DIVREST:  return lhs ? r_modulus(x, d) : r_modulus(d, x);
        case MX_DV_INTDIV:   return lhs ? r_intdiv(x, d) : r_intdiv(d, x);
        default:             return lhs ? r_pow(x, d) : r_pow(d, x);
    }
}

// MODE 0: position depends on the row only (length == nrows, or length divides nrows); 1: the vector covers the
// matrix (row + col*nrows, no wrap); 2: general recycling, 64-bit modulo per entry.
template <int G, bool LOGICAL>
__global__ __launch_bounds__(DV_BLOCK)
void csr_by_dvec_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                        const void *__restrict__ values, const void *__restrict__ dvec, unsigned long long len,
                        int mode, int op, int lhs, void *__restrict__ out)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (DV_BLOCK / G) + threadIdx.x / G;
    if (row >= m) return;
    const int s = indptr[row], e = indptr[row + 1];
    const unsigned long long nr = (unsigned long long)m;
    const unsigned long long rowpos = mode == 0 ? (unsigned long long)row % len : 0ULL;
    for (int k = s + lg; k < e; k += G) {
        unsigned long long at = rowpos;
        if (mode == 1) at = (unsigned long long)row + nr * (unsigned long long)indices[k];
        else if (mode == 2) at = ((unsigned long long)row + nr * (unsigned long long)indices[k]) % len;
        if constexpr (LOGICAL)
            ((int32_t *)out)[k] = r_logical_and(((const int32_t *)values)[k], ((const int32_t *)dvec)[at]);
        else
            ((double *)out)[k] = dv_apply(op, lhs != 0, ((const double *)values)[k], ((const double *)dvec)[at]);
    }
}

}  // namespace mx

#define MX_DV_G(LOGICAL)                                                                                        \
    switch (G) {                                                                                                \
        case 4:  hipLaunchKernelGGL((mx::csr_by_dvec_kernel<4, LOGICAL>),  dim3((unsigned)mx::ceil_div(m, mx::DV_BLOCK / 4)),  dim3(mx::DV_BLOCK), 0, st, m, indptr, indices, values, dvec, len, mode, op, x_is_lhs, values_out); break; \
        case 8:  hipLaunchKernelGGL((mx::csr_by_dvec_kernel<8, LOGICAL>),  dim3((unsigned)mx::ceil_div(m, mx::DV_BLOCK / 8)),  dim3(mx::DV_BLOCK), 0, st, m, indptr, indices, values, dvec, len, mode, op, x_is_lhs, values_out); break; \
        case 16: hipLaunchKernelGGL((mx::csr_by_dvec_kernel<16, LOGICAL>), dim3((unsigned)mx::ceil_div(m, mx::DV_BLOCK / 16)), dim3(mx::DV_BLOCK), 0, st, m, indptr, indices, values, dvec, len, mode, op, x_is_lhs, values_out); break; \
        case 32: hipLaunchKernelGGL((mx::csr_by_dvec_kernel<32, LOGICAL>), dim3((unsigned)mx::ceil_div(m, mx::DV_BLOCK / 32)), dim3(mx::DV_BLOCK), 0, st, m, indptr, indices, values, dvec, len, mode, op, x_is_lhs, values_out); break; \
        default: hipLaunchKernelGGL((mx::csr_by_dvec_kernel<64, LOGICAL>), dim3((unsigned)mx::ceil_div(m, mx::DV_BLOCK / 64)), dim3(mx::DV_BLOCK), 0, st, m, indptr, indices, values, dvec, len, mode, op, x_is_lhs, values_out); break; \
    }

extern "C" int mxd_csr_by_dvec(int m, int ncols, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                               const void *values, const void *dvec, int64_t dvec_len, int op, int x_is_lhs,
                               void *values_out, void *stream)
{
    MX_REQUIRE(m >= 0 && ncols >= 0 && dvec_len >= 0, "mxd_csr_by_dvec: negative size");
    MX_REQUIRE(op >= MX_DV_MULTIPLY && op <= MX_DV_LOGICAL_AND, "mxd_csr_by_dvec: unknown operation %d", op);
    if (m == 0 || nnz == 0) return 0;
    MX_REQUIRE(dvec_len > 0, "mxd_csr_by_dvec: empty vector");       // the R caller returns early (R/operators.R:961-966)
    MX_REQUIRE(indptr && indices && values && dvec && values_out, "mxd_csr_by_dvec: null pointer");
    hipStream_t st = mx::as_stream(stream);
    const unsigned long long len = (unsigned long long)dvec_len;
    int mode = 2;
    if (len == (unsigned long long)m || (len < (unsigned long long)m && (unsigned long long)m % len == 0)) mode = 0;
    else if (len >= (unsigned long long)m * (unsigned long long)ncols) mode = 1;
    const int G = nnz < 0 ? 32 : mx::pick_group((double)nnz / (double)m);
    if (op == MX_DV_LOGICAL_AND) { MX_DV_G(true) } else { MX_DV_G(false) }
    MX_LAUNCH_CHECK();
    return 0;
}
