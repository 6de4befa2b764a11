// spmv_rows.h — pieces shared by the two "products first, rows afterwards" SpMV kernels (spmv_flat.hip: v gathered from
// L2; spmv_tile.hip: v swept through LDS): the row accumulator with the reference's semantics, the conversion of a
// vector element to the f64 factor, and the reduction of staged products one thread per row in storage order.
#pragma once
#include "mx_common.h"

namespace mx {

constexpr int SPMV_LONG_ROW = 256;      // rows with more staged entries than this are summed by a whole wavefront

__device__ __forceinline__ double na_bits_as_double() { return __longlong_as_double((long long)MX_NA_REAL_BITS); }
__device__ __forceinline__ bool is_na_bits(double d) { return __double_as_longlong(d) == (long long)MX_NA_REAL_BITS; }

// v[j] as the f64 factor of the product:  numeric: as is;  integer: (double)v, NA_INTEGER -> NA_real bits;
// logical: (double)(v != 0), NA_LOGICAL -> NA_real bits (matmul.cpp:406-411);  float32: (double)v
template <int KIND>
__device__ __forceinline__ double vec_factor(const void *__restrict__ v_, int j)
{
    if constexpr (KIND == MX_F64) return ((const double *)v_)[j];
    else if constexpr (KIND == MX_F32) return (double)((const float *)v_)[j];
    else {
        const int yv = ((const int32_t *)v_)[j];
        if constexpr (KIND == MX_LGL) return yv == MX_NA_INT ? na_bits_as_double() : (double)(yv != 0);
        else return yv == MX_NA_INT ? na_bits_as_double() : (double)yv;
    }
}
// the term of one entry: a * factor, or NA_REAL itself when the vector element is NA
template <int KIND>
__device__ __forceinline__ double term(double a, double factor)
{
    const double prod = a * factor;                                   // separately rounded (the sums below do not fuse)
    if constexpr (KIND == MX_I32 || KIND == MX_LGL) return is_na_bits(factor) ? factor : prod;
    else return prod;
}

// running sum of one row, in storage order; float32 kind: float accumulator, double product (matmul.cpp:403,413)
template <int KIND>
struct RowAcc {
    double d = 0.0;
    float f = 0.0f;
    int na = 0;
    __device__ __forceinline__ void add(double prod)
    {
        if constexpr (KIND == MX_F32) f = (float)((double)f + prod);
        else {
            if constexpr (KIND == MX_I32 || KIND == MX_LGL) na |= is_na_bits(prod) ? 1 : 0;
            d += prod;
        }
    }
};

// RA / RB = the first row r in [0, m] with indptr[r] >= cutA / cutB (m when every row starts below the cut): an
// (NT * PR)-ary search for both cuts at once, every thread of the workgroup issues PR independent probes per cut and level
// (1024-ary: two levels — two exposed load latencies — for a million rows).  All NT threads must call it (barriers inside);
// cnt_lds: two ints of LDS.
template <int NT, int PR>
__device__ __forceinline__ void first_rows_at_or_after(const int32_t *__restrict__ indptr, int m, long long cutA,
                                                       long long cutB, int &RA, int &RB, int *cnt_lds)
{
    const int tid = threadIdx.x;
    long long loA = 0, hiA = m, loB = 0, hiB = m;     // rows < lo start before the cut, rows >= hi do not
    while (loA < hiA || loB < hiB) {                  // uniform
        const long long stA = (hiA - loA + NT * PR - 1) / (NT * PR), stB = (hiB - loB + NT * PR - 1) / (NT * PR);
        if (tid < 2) cnt_lds[tid] = 0;
        __syncthreads();
        int a[PR], b[PR];
        bool okA[PR], okB[PR];
#pragma unroll
        for (int p = 0; p < PR; p++) {                // all 2 * PR loads in flight together
            const long long rA = loA + (long long)(tid * PR + p) * stA, rB = loB + (long long)(tid * PR + p) * stB;
            okA[p] = rA < hiA; okB[p] = rB < hiB;
            a[p] = indptr[okA[p] ? rA : loA];
            b[p] = indptr[okB[p] ? rB : loB];
        }
        int cA = 0, cB = 0;
#pragma unroll
        for (int p = 0; p < PR; p++) { cA += okA[p] && (long long)a[p] < cutA; cB += okB[p] && (long long)b[p] < cutB; }
        if (cA) atomicAdd(&cnt_lds[0], cA);
        if (cB) atomicAdd(&cnt_lds[1], cB);
        __syncthreads();
        const int nA = cnt_lds[0], nB = cnt_lds[1];   // monotone: exactly the first n probes start before the cut
        __syncthreads();
        const long long loA2 = nA ? loA + (long long)(nA - 1) * stA + 1 : loA, loB2 = nB ? loB + (long long)(nB - 1) * stB + 1 : loB;
        hiA = min(hiA, loA + (long long)nA * stA); hiB = min(hiB, loB + (long long)nB * stB);
        loA = loA2; loB = loB2;
    }
    RA = (int)loA;
    RB = (int)loB;
}

// Products of the entries [hb, he) of the arrays are staged at stage[pos + (pos >> 5)], pos = entry - hb.  Rows [R0, R1)
// belong to this workgroup; row r is summed by thread (r - R0) % THREADS over its entries that lie in [max(hb, E0), he),
// continuing from `carry` when the row began in an earlier pass; a row whose last entry lies in this pass is written,
// otherwise its running sum is kept in `carry`.  Empty rows are written as 0 when `zero_empty`.
template <int KIND, int THREADS>
__device__ __forceinline__ void reduce_rows(const double *stage, long long hb, long long he, long long E0, int R0, int R1,
                                            const int32_t *__restrict__ indptr, int rs_first, int re_first,
                                            RowAcc<KIND> &carry, int &carry_row, bool zero_empty, void *__restrict__ y_)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const long long lo_lim = max(hb, E0);
    const int nrow_it = (R1 - R0 + THREADS - 1) / THREADS;            // uniform trip count
    for (int it = 0; it < nrow_it; it++) {
        const int row = R0 + it * THREADS + tid;
        const bool have = row < R1;
        int rs = 0, re = 0;
        if (it == 0) { rs = rs_first; re = re_first; }
        else if (have) { rs = indptr[row]; re = indptr[row + 1]; }
        const int s = (int)(max((long long)rs, lo_lim) - hb), e = (int)(min((long long)re, he) - hb);
        const bool empty_row = have && rs == re;
        const bool work = have && s < e;
        RowAcc<KIND> acc;
        if (work && row == carry_row) acc = carry;
        const bool lng = work && (e - s) > SPMV_LONG_ROW;
        if (work && !lng)
            for (int pos = s; pos < e; pos++) acc.add(stage[pos + (pos >> 5)]);
        // long rows: the whole wavefront sums the row's part of this pass (strided + butterfly), the owner adds it
        unsigned long long lm = __ballot(lng);
        while (lm) {
            const int L = __builtin_ctzll(lm);
            lm &= lm - 1;
            const int s_ = __shfl(s, L, 64), e_ = __shfl(e, L, 64);
            double part = 0.0;
            int pna = 0;
            for (int pos = s_ + lane; pos < e_; pos += 64) {
                const double pv = stage[pos + (pos >> 5)];
                if constexpr (KIND == MX_I32 || KIND == MX_LGL) pna |= is_na_bits(pv) ? 1 : 0;
                part += pv;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                part += __shfl_xor(part, off, 64);
                pna |= __shfl_xor(pna, off, 64);
            }
            if (lane == L) {
                if constexpr (KIND == MX_F32) acc.f = (float)((double)acc.f + part);
                else { acc.d += part; acc.na |= pna; }
            }
        }
        if (work) {
            if ((long long)re <= he) {                                // the row ends in this pass: its sum is final
                if constexpr (KIND == MX_F32) ((float *)y_)[row] = acc.f;
                else ((double *)y_)[row] = acc.na ? na_real() : acc.d;
            } else {
                carry = acc;
                carry_row = row;
            }
        } else if (empty_row && zero_empty) {
            if constexpr (KIND == MX_F32) ((float *)y_)[row] = 0.0f; else ((double *)y_)[row] = 0.0;
        }
    }
}

}  // namespace mx
