// r_arith.h — R's arithmetic restated for the device (shared by dvec.hip and dvec_na.hip): R_pow, R_modulus (%%),
// R_intdiv (%/%) as the reference quotes them at src/operators.cpp:1482-1601.  Where the reference carries an intermediate
// in `long double` (x87, 64-bit mantissa) the device uses one fused multiply-add (exact product, one rounding): results
// agree to the last bit or two, not always bit for bit.
#pragma once
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

}  // namespace mx
