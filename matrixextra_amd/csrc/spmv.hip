// spmv.hip — CSR x dense-vector SpMV for gfx950.
//
// Replaces matmul_csr_dvec<> (src/matmul.cpp:381-419; exports :421-483):
//   numeric / integer / logical right-hand sides -> f64 result,
//   float32 right-hand side -> f32 result with float accumulation (:403).
// NA_INTEGER / NA_LOGICAL entries contribute NA_REAL (:406-411); a logical
// entry counts as (bool)y (:411).
//
// Design: G lanes of a wavefront own SPMV_ROWS consecutive rows (G = power of two picked from the
// mean row length, 64 = one wavefront per row); lanes stride the row's entries
// (coalesced (j, a) reads, gathered v[j] reads served by L2 — v is 0.8 MB for
// the headline config), then a butterfly __shfl_xor reduction inside the group.
// HBM-bound: algorithmic bytes = 4(m+1) + 12 nnz + s*K + s*m.
#include "spmm_common.h"     // (ProfileScope, profile_longest_over_mean: the matrix profile AUTO reads)
#include <cstdlib>

namespace mx {

constexpr int SPMV_BLOCK = 256;

constexpr int SPMV_ROWS = 1;     // rows per lane group (measured on MI355X: 1 -> 0.197 ms, 4 -> 0.221 ms for cfg3:
                                 // the kernel is bound by the L2->L1 line traffic of the v[j] gather, not by latency)

template <int KIND>
__device__ __forceinline__ void spmv_term(double a, const void *__restrict__ v_, int j, double &acc, float &accf, int &na)
{
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

// G lanes own SPMV_ROWS consecutive rows; every stage (indptr, (j, a), v[j]) issues its loads unconditionally
// (clamped addresses, select afterwards) so that they are in flight together.
template <int G, int KIND>
__global__ __launch_bounds__(SPMV_BLOCK)
void spmv_group_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                       const double *__restrict__ values, const void *__restrict__ v_, void *__restrict__ y_)
{
    constexpr int R = SPMV_ROWS;
    const int lg = threadIdx.x % G;
    const long long row0 = ((long long)blockIdx.x * (SPMV_BLOCK / G) + threadIdx.x / G) * R;
    int s[R], e[R];
    const int nnz = indptr[m];                    // uniform; entry 0 is a safe address for masked lanes iff nnz > 0
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long long row = row0 + r;
        const long long rs = row < m ? row : 0;   // clamped, unconditional loads: all 2R indptr reads in flight at once
        const int ps = indptr[rs], pe = indptr[rs + 1];
        s[r] = row < m ? ps : 0;
        e[r] = row < m ? pe : 0;
    }
    if (nnz == 0) {
#pragma unroll
        for (int r = 0; r < R; r++)
            if (lg == r % G && row0 + r < m) {
                if constexpr (KIND == MX_F32) ((float *)y_)[row0 + r] = 0.0f; else ((double *)y_)[row0 + r] = 0.0;
            }
        return;
    }
    double acc[R];
    float accf[R];
    int na[R];
    int j0[R];
    double a0[R];
#pragma unroll
    for (int r = 0; r < R; r++) {                 // stage 1: first chunk of every row
        acc[r] = 0.0; accf[r] = 0.0f; na[r] = 0;
        const int k = s[r] + lg;
        const bool ok = k < e[r];
        const int ks = ok ? k : 0;                // unconditional (clamped) loads, selected afterwards
        const int jl = indices[ks];
        const double al = values[ks];
        j0[r] = ok ? jl : -1;
        a0[r] = al;
    }
    {                                             // stage 2: gathers of v, again unconditional + select
        double accn[R];
        float accfn[R];
        int nan_[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            accn[r] = 0.0; accfn[r] = 0.0f; nan_[r] = 0;
            spmv_term<KIND>(a0[r], v_, j0[r] >= 0 ? j0[r] : 0, accn[r], accfn[r], nan_[r]);
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (j0[r] >= 0) { acc[r] = accn[r]; accf[r] = accfn[r]; na[r] = nan_[r]; }
    }
#pragma unroll
    for (int r = 0; r < R; r++)                   // rows longer than one group width
        for (int k = s[r] + G + lg; k < e[r]; k += G)
            spmv_term<KIND>(values[k], v_, indices[k], acc[r], accf[r], na[r]);
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (KIND == MX_F32) accf[r] += __shfl_xor(accf[r], off, G);
            else {
                acc[r] += __shfl_xor(acc[r], off, G);
                if constexpr (KIND == MX_I32 || KIND == MX_LGL) na[r] |= __shfl_xor(na[r], off, G);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (lg == r % G && row0 + r < m) {
            if constexpr (KIND == MX_F32) ((float *)y_)[row0 + r] = accf[r];
            else ((double *)y_)[row0 + r] = na[r] ? na_real() : acc[r];
        }
    }
}

template <int KIND>
static int launch_spmv(int G, int m, const int32_t *indptr, const int32_t *indices, const double *values,
                       const void *v, void *y, hipStream_t st)
{
#define MX_SPMV_CASE(GG)                                                                        \
    case GG: {                                                                                  \
        const unsigned grid = (unsigned)ceil_div(m, (SPMV_BLOCK / GG) * SPMV_ROWS);             \
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

// spmv_tile.hip: the LDS-panel kernel (v staged through LDS, rows summed in storage order)
bool spmv_tile_ok(int m, int64_t nnz, int K, const int32_t *indices, const double *values, const void *v, bool forced);
int spmv_tile_launch(int m, int64_t nnz, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                     const void *v, int v_dtype, void *y, hipStream_t st);

// spmv_flat.hip: the flat-stream kernel (slices of entries, v gathered, rows summed in storage order)
bool spmv_flat_ok(int m, int64_t nnz, const int32_t *indices, const double *values);
int spmv_flat_launch(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                     const void *v, int v_dtype, void *y, hipStream_t st);

// algo: MX_SPMV_AUTO picks the flat kernel when it applies (exact nnz known and >= 2^20, aligned arrays), else the
// lane-group kernel; the LDS-panel tile kernel runs on request only.  nnz < 0 = unknown (=> lane-group kernel, 32
// lanes per row).  MXGPU_SPMV_ALGO=1|2|3 overrides AUTO (A/B runs).
int spmv_launch(int m, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                const void *v, int v_dtype, void *y, int algo, hipStream_t st)
{
    if (algo == MX_SPMV_AUTO) {
        static const int forced = [] { const char *e = getenv("MXGPU_SPMV_ALGO"); return e ? atoi(e) : 0; }();
        algo = forced;
    }
    const bool tile_ok = K > 0 && nnz >= 0 && spmv_tile_ok(m, nnz, K, indices, values, v, algo == MX_SPMV_TILE);
    if (algo == MX_SPMV_TILE && !tile_ok)
        return set_error("spmv: the tile kernel needs K, nnz >= 4, 16-byte aligned arrays and K <= %d columns", 24 * 16384);
    if (algo == MX_SPMV_TILE) return spmv_tile_launch(m, nnz, K, indptr, indices, values, v, v_dtype, y, st);
    const bool flat_ok = nnz >= 0 && spmv_flat_ok(m, nnz, indices, values);
    if (algo == MX_SPMV_FLAT && !flat_ok)
        return set_error("spmv: the flat kernel needs the exact nnz (>= 4) and 16-byte aligned index / value arrays");
    // AUTO: the flat kernel once there are enough entries to fill the chip (measured on MI355X: 1M x 100k, 32 / row:
    // flat vs lane-group in DESIGN.md §4.3); small products stay on the lane-group kernel.  Rounds 1-3 switched at 2^20
    // entries whatever the shape; round 4's map (tools/auto_map.py, K = 1e5) shows the flat kernel 1.6-1.9x behind for FEW
    // rows (1e4 x 128 / row: lane-group 0.009 ms, flat 0.016; 1e4 x 500: 0.029 / 0.047) and within 1.2x either way from 1e5
    // rows on (1e5 x 32: 0.020 / 0.024; 1e5 x 500: 0.328 / 0.276; 1e6 x 32: 0.190 / 0.164) — where it stays, because its
    // sums are the reference's loop bit for bit
    // (and from 2^22 entries: at 1e5 x 32 — 3.2 M entries — the lane-group kernel took 0.020 ms, the flat one 0.024-0.029)
    // With the matrix profile in scope (mxd_spmv_csr_dvec_ex2): a row of 16k entries or more is a tail for the lane-group
    // kernel — one group walks it — and nothing special for the flat kernel's equal slices (tools/spmv_skew_probe.py, 3e4 x
    // 1e5, 200 per row: four rows of 50,000 entries 0.237 ms against 0.087, log-normal sigma 1.5 0.152 against 0.065)
    // (Round 6 tried the flat kernel for rows of very uneven length — cv > 0.5 — below 32,768 rows as well: 1e4 x 1e4, 500 per row,
    // log-normal 0.047 -> 0.054 ms, half of the rows empty 0.016 -> 0.044: the flat kernel's two launches cost ~0.045 ms whatever it
    // is given.  Not kept.)
    const bool long_rows = algo == MX_SPMV_AUTO && flat_ok && nnz >= ((int64_t)1 << 20) && m > 0 &&
                           profile_longest_over_mean() * ((double)nnz / m) >= 16384.0;
    if (algo == MX_SPMV_FLAT || long_rows || (algo == MX_SPMV_AUTO && flat_ok && nnz >= ((int64_t)1 << 22) && m >= 32768))
        return spmv_flat_launch(m, nnz, indptr, indices, values, v, v_dtype, y, st);
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
    return mx::spmv_launch(m, 0, nnz, indptr, indices, values, v, v_dtype, y, MX_SPMV_GROUP, mx::as_stream(stream));
}

extern "C" int mxd_spmv_csr_dvec_ex(int m, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                                    const double *values, const void *v, int v_dtype, void *y, int algo, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_spmv_csr_dvec_ex: negative m");
    MX_REQUIRE(algo >= MX_SPMV_AUTO && algo <= MX_SPMV_FLAT, "mxd_spmv_csr_dvec_ex: unknown algo %d", algo);
    if (m == 0) return 0;
    MX_REQUIRE(indptr && y, "mxd_spmv_csr_dvec_ex: null pointer");
    return mx::spmv_launch(m, K, nnz, indptr, indices, values, v, v_dtype, y, algo, mx::as_stream(stream));
}

// the same with the matrix profile (mxd_csr_profile) in scope: AUTO then knows about very long rows
extern "C" int mxd_spmv_csr_dvec_ex2(int m, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                                     const double *values, const void *v, int v_dtype, void *y, int algo, const float *profile,
                                     void *stream)
{
    mx::ProfileScope scope(profile);
    return mxd_spmv_csr_dvec_ex(m, K, nnz, indptr, indices, values, v, v_dtype, y, algo, stream);
}
