// spmv_plan.hip — planned SpMV for repeated products with the same matrix (X %*% v inside an iterative solver: the
// L-BFGS loop of the reference's vignette calls it hundreds of times with one X).
//
// A one-shot SpMV is bound by the v[j] gather, not by the (j, a) stream: 32 M scattered 8-byte reads of a 0.8 MB vector
// move 32 M 128-byte lines from L2 into the CUs, 131 us at the L2's request rate, against 64 us for the stream alone
// (profiles/r02_spmv_ceiling.json).  Taking the gather off the L2 path needs the entries grouped by COLUMN PANEL so that
// the panel of v they touch can sit in LDS — a regrouping that costs more than one product (a pass over A), hence a plan:
//   * rows in blocks of 4096, columns in panels of 6144 (48 KiB of f64); the plan holds, for every (row block, panel),
//     the block's entries of that panel as (row-in-block << 13 | column-in-panel, value), 12 bytes per entry like the CSR
//     arrays, segments padded to 4 entries;
//   * one 1024-thread workgroup per row block keeps the block's 4096 sums in LDS (32 KiB) and two panel buffers
//     (2 x 48 KiB): while it streams panel p's segment — 16 B per lane, fully coalesced, the only HBM traffic — and adds
//     a * v_lds[c] into the row's sum (fire-and-forget ds_add_f64), the next panel of v comes in through registers.
//     v is read once per row block (0.8 MB per 1.5 MB of entries, from L2, coalesced).
// Round 6 — two more things a kept plan has to live with:
//   * WIDE matrices.  64 LDS panels cover 393,216 columns; beyond that — BASELINE configs[3]'s own shape is 2M x 2M, v =
//     16 MB, more than an XCD's 4 MiB L2, and the one-shot gather then runs at the Infinity Cache's line rate: 1.49 ms for
//     1.24 GB, 0.10 of the roofline — the plan regroups the entries by SUPER-PANEL of 2^18 columns (2 MB of v) instead and
//     the product runs ONE LAUNCH PER SUPER-PANEL (spmv_wide_panel_kernel): every workgroup of a launch reads v[j] straight
//     from global memory, all of them from the same 2 MB, which every XCD's L2 then holds; y is added to from launch to launch.
//     Up to 1,024 super-panels (2^28 columns).
//   * rows of UNEVEN length.  A row block used to be 4,096 rows whatever they held: with the rows sorted by length the first
//     blocks carried ten times the entries of the last and their workgroups were the tail (0.95 ms against 0.077).  A block
//     of 4,096 rows that holds more than E entries (1.06 mean blocks) is now cut into sub-blocks of equal entry counts
//     (spmv_plan_cuts_kernel, on the device): rows of equal length keep exactly the blocks they had.
// Measured (cfg3, MI355X): DESIGN.md §4.3.  Summation order: per row, panels in ascending order; inside a panel the
// entries of a row are added with LDS atomics in whatever order their wavefronts arrive: equal to the reference to
// 1e-12 (f64), not bitwise and not run-to-run reproducible in the last bit — the one-shot flat kernel (spmv_flat.hip) is
// the bit-exact path.  float32 kind: sums kept in f64, rounded once (the reference rounds after every term: 1e-5).
#include "mx_common.h"
#include <algorithm>
#include <new>

namespace mx {

int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace, hipStream_t st);
size_t scan_workspace_bytes(int64_t n);

constexpr int SP_RB = 4096;                  // rows per block (one workgroup)
constexpr int SP_PANEL = 6144;               // columns per panel: 48 KiB of f64
constexpr int SP_THREADS = 1024;
constexpr int SP_MAX_PANELS = 64;            // LDS mode: K <= 64 x 6144
constexpr int SP_COL_BITS = 13;              // entry = row_in_block << 13 | col_in_panel  (row 4096 = the padding's dummy row)
static_assert(SP_PANEL <= (1 << SP_COL_BITS), "column field");
constexpr int SP_WIDE_BITS = 18;             // wide mode: super-panels of 2^18 columns (2 MB of f64), read from L2
constexpr int SP_WIDE_MAX_PANELS = 1024;     // 2^28 columns
static_assert(13 + SP_WIDE_BITS <= 31, "row-in-block (0 .. 4096) and column-in-panel share 31 bits");

// Row-block boundaries.  Every 4,096 rows is a cut; a block of 4,096 rows that holds more than E entries is cut further into
// s = ceil(entries / E) sub-blocks of equal entry counts (binary searches inside the block), so that no workgroup carries much
// more than E entries + one row.  One workgroup: the blocks' sub-block counts, their prefix sums (a scan over <= 2^19 values
// in rounds of 1,024), then every sub-block boundary.  cuts[0 .. nrb], nrb = the total — read back by the host (the build
// synchronises anyway).  Rows of equal length: s = 1 everywhere, exactly the blocks of rounds 2-5.
__global__ __launch_bounds__(1024)
void spmv_plan_cuts_kernel(int m, int RB, const int32_t *__restrict__ indptr, long long E, long long E_piece, int32_t *__restrict__ cuts, int max_cuts,
                           long long *__restrict__ nrb_out)
{
    __shared__ int scan[1024];
    __shared__ int carry;
    const int nA = (int)((m + RB - 1) / RB), tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < nA; k0 += 1024) {
        const int k = k0 + tid;
        int s = 0, r0 = 0, r1 = 0;
        long long e0 = 0, nn = 0;
        if (k < nA) {
            r0 = k * RB; r1 = min(r0 + RB, m);
            e0 = indptr[r0]; nn = (long long)indptr[r1] - e0;
            s = nn > E ? (int)((nn + E_piece - 1) / E_piece) : 1;
            s = max(1, min(s, r1 - r0));                               // (never more sub-blocks than rows)
        }
        scan[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                           // inclusive scan
            const int v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int base = carry + scan[tid] - s;                        // index of this block's first cut
        if (k < nA) {
            for (int q = 0; q < s; q++) {
                int at = r0;
                if (q > 0) {                                           // first row of the block that starts at or after the q-th share
                    const long long target = e0 + nn * q / s;
                    int lo = r0, hi = r1;
                    while (lo < hi) { const int mid = lo + (hi - lo) / 2; if ((long long)indptr[mid] < target) lo = mid + 1; else hi = mid; }
                    at = lo;
                }
                if (base + q < max_cuts) cuts[base + q] = at;
            }
        }
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
    if (tid == 0) { if (carry < max_cuts + 1) cuts[min(carry, max_cuts)] = m; *nrb_out = carry; }
}

// panel of a column id: LDS mode divides by 6144, wide mode shifts by 18 (ids outside [0, K) are the caller's bug: kept from
// corrupting memory)
template <bool WIDE> __device__ __forceinline__ int sp_panel_of(int col, int npanels)
{
    const int c = max(col, 0);
    return min(WIDE ? c >> SP_WIDE_BITS : c / SP_PANEL, npanels - 1);
}
// entries of row block `rb` per panel, rounded up to 4 (segments are read 4 entries per lane)
template <bool WIDE>
__global__ __launch_bounds__(SP_THREADS)
void spmv_plan_count_kernel(int npanels, const int32_t *__restrict__ rb_row, const int32_t *__restrict__ indptr,
                            const int32_t *__restrict__ indices, int32_t *__restrict__ counts)
{
    __shared__ int hist[WIDE ? SP_WIDE_MAX_PANELS : SP_MAX_PANELS];
    const int rb = blockIdx.x;
    for (int i = threadIdx.x; i < npanels; i += SP_THREADS) hist[i] = 0;
    __syncthreads();
    const int r0 = rb_row[rb], r1 = rb_row[rb + 1];
    const int s = indptr[r0], e = indptr[r1];
    for (int k = s + threadIdx.x; k < e; k += SP_THREADS) atomicAdd(&hist[sp_panel_of<WIDE>(indices[k], npanels)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < npanels; i += SP_THREADS) counts[(size_t)rb * npanels + i] = (hist[i] + 3) & ~3;
}

// scatter: the block's entries are read flat (coalesced); an entry's row comes from a binary search of the block's row
// pointers held in LDS, its position from the segment start + an LDS cursor per panel (so the order of a segment's
// entries varies from build to build — the sums' last bits with it)
template <bool WIDE>
__global__ __launch_bounds__(SP_THREADS)
void spmv_plan_fill_kernel(int npanels, const int32_t *__restrict__ rb_row, const int32_t *__restrict__ indptr,
                           const int32_t *__restrict__ indices, const double *__restrict__ values, const int32_t *__restrict__ seg_off,
                           int32_t *__restrict__ ent, double *__restrict__ val)
{
    constexpr int CB = WIDE ? SP_WIDE_BITS : SP_COL_BITS;
    __shared__ int cursor[WIDE ? SP_WIDE_MAX_PANELS : SP_MAX_PANELS];
    __shared__ int ip[SP_RB + 1];
    const int rb = blockIdx.x;
    for (int i = threadIdx.x; i < npanels; i += SP_THREADS) cursor[i] = 0;
    const int r0 = rb_row[rb], r1 = rb_row[rb + 1], nr = r1 - r0;
    for (int i = threadIdx.x; i <= nr; i += SP_THREADS) ip[i] = indptr[r0 + i];
    __syncthreads();
    const int32_t *__restrict__ so = seg_off + (size_t)rb * npanels;
    const int s = ip[0], e = ip[nr];
    for (int k = s + threadIdx.x; k < e; k += SP_THREADS) {
        int lo = 0, hi = nr;                                     // the row r with ip[r] <= k < ip[r + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ip[mid] <= k) lo = mid; else hi = mid; }
        const int col = max(indices[k], 0), pan = sp_panel_of<WIDE>(col, npanels);
        const int pos = so[pan] + atomicAdd(&cursor[pan], 1);
        ent[pos] = (lo << CB) | (WIDE ? min(col - (pan << SP_WIDE_BITS), (1 << SP_WIDE_BITS) - 1) : min(col - pan * SP_PANEL, SP_PANEL - 1));
        val[pos] = values[k];
    }
    __syncthreads();
    // padding up to the rounded segment length: the dummy row SP_RB, column 0, value 0
    for (int pan = threadIdx.x >> 2; pan < npanels; pan += SP_THREADS >> 2) {
        const int pos = so[pan] + cursor[pan] + (threadIdx.x & 3);
        if (pos < so[pan + 1]) { ent[pos] = SP_RB << CB; val[pos] = 0.0; }
    }
}

// v[cbase + c] as f64 (NA_INTEGER / NA_LOGICAL -> NA_real bits, logical -> 0 / 1: matmul.cpp:406-411)
template <int KIND>
__device__ __forceinline__ double sp_factor(const void *__restrict__ v_, int j)
{
    if constexpr (KIND == MX_F64) return ((const double *)v_)[j];
    else if constexpr (KIND == MX_F32) return (double)((const float *)v_)[j];
    else {
        const int yv = ((const int32_t *)v_)[j];
        if (yv == MX_NA_INT) return __longlong_as_double((long long)MX_NA_REAL_BITS);
        if constexpr (KIND == MX_LGL) return (double)(yv != 0); else return (double)yv;
    }
}

template <int KIND, bool WIDE>
__global__ __launch_bounds__(SP_THREADS)
void spmv_plan_kernel(int m, int K, int npanels, const int32_t *__restrict__ rb_row, const int32_t *__restrict__ seg_off,
                      const int32_t *__restrict__ ent, const double *__restrict__ val, const void *__restrict__ v_, void *__restrict__ y_)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int CB = WIDE ? SP_WIDE_BITS : SP_COL_BITS;
    constexpr int PER = SP_PANEL / SP_THREADS;                        // panel elements staged per thread (6)
    __shared__ double acc[SP_RB + 8];                                 // + the padding's dummy row
    __shared__ double vpan[WIDE ? 1 : 2][WIDE ? 1 : SP_PANEL];        // (wide mode reads v from global memory: no panel buffers)
    // integer / logical vectors: one bit per row "an NA element took part" (the reference adds NA_REAL itself for such a
    // term, matmul.cpp:406-411: the row's result is NA_real_, whereas a NaN that comes out of the arithmetic stays a NaN —
    // R tells the two apart; the flag, not `sum != sum`, decides, as RowAcc does in spmv_rows.h)
    __shared__ unsigned na_rows[(SP_RB + 8 + 31) / 32];
    const int tid = threadIdx.x, rb = blockIdx.x;
    const int r0 = rb_row[rb], nr = rb_row[rb + 1] - r0;
    if (nr <= 0) return;                                              // (an empty block: two cuts at the same row)
    for (int i = tid; i < SP_RB + 8; i += SP_THREADS) acc[i] = 0.0;
    if constexpr (KIND == MX_I32 || KIND == MX_LGL)
        for (int i = tid; i < (SP_RB + 8 + 31) / 32; i += SP_THREADS) na_rows[i] = 0u;
    const int32_t *__restrict__ so = seg_off + (size_t)rb * npanels;
    // A LIGHT block (the tail of a matrix whose rows are sorted by length: 4,096 rows of one or two entries) would spend its
    // time staging all of v — K x 8 bytes through 2 x 48 KB of LDS, a barrier per panel — for a few thousand products: it reads
    // v[j] from global memory instead, panel after panel without a barrier (rows sorted by length, cfg3's shape: 0.163 -> see
    // docs/experiments/round-6.md §10.5).  Uniform per workgroup.
    const bool direct = !WIDE && (long long)(so[npanels] - so[0]) * 8 < (long long)K;      // (a gathered element moves a 64-byte sector: entries x 64 < K x 8)
    if (direct) {
        __syncthreads();
        for (int p = 0; p < npanels; p++) {
            const int s = so[p], e = so[p + 1], cb = p * SP_PANEL;
            for (int k0 = s + tid * 4; k0 < e; k0 += SP_THREADS * 4) {
                const i4 c0 = *reinterpret_cast<const i4 *>(ent + k0);
                const d2 a0 = *reinterpret_cast<const d2 *>(val + k0), a1 = *reinterpret_cast<const d2 *>(val + k0 + 2);
                const double av[4] = {a0[0], a0[1], a1[0], a1[1]};
                double f[4];
#pragma unroll
                for (int q = 0; q < 4; q++) f[q] = sp_factor<KIND>(v_, min(cb + (c0[q] & ((1 << CB) - 1)), K - 1));
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = (int)((unsigned)c0[q] >> CB);
                    double t = av[q] * f[q];
                    if constexpr (KIND == MX_I32 || KIND == MX_LGL) {
                        if (__double_as_longlong(f[q]) == (long long)MX_NA_REAL_BITS) { t = 0.0; if (r < SP_RB) atomicOr(&na_rows[r >> 5], 1u << (r & 31)); }
                    }
                    __hip_atomic_fetch_add(&acc[r], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < nr; i += SP_THREADS) {
            const double sum = acc[i];
            if constexpr (KIND == MX_F32) ((float *)y_)[r0 + i] = (float)sum;
            else if constexpr (KIND == MX_I32 || KIND == MX_LGL) ((double *)y_)[r0 + i] = (na_rows[i >> 5] >> (i & 31)) & 1u ? na_real() : sum;
            else ((double *)y_)[r0 + i] = sum;
        }
        return;
    }
    double stage[PER];
    if constexpr (!WIDE) {                                            // panel 0 -> buffer 0
#pragma unroll
        for (int i = 0; i < PER; i++) { const int c = i * SP_THREADS + tid; stage[i] = c < K ? sp_factor<KIND>(v_, c) : 0.0; }
#pragma unroll
        for (int i = 0; i < PER; i++) vpan[0][i * SP_THREADS + tid] = stage[i];
    }
    __syncthreads();
    // one entry: its factor (LDS panel / global memory), the product, the row's sum
    auto term = [&](int code, double a, const double *__restrict__ vp, int cbase) {
        const int c = code & ((1 << CB) - 1), r = (int)((unsigned)code >> CB);
        double f;
        if constexpr (WIDE) f = sp_factor<KIND>(v_, min(cbase + c, K - 1));          // (the padding's column 0 of the last panel stays in range)
        else f = vp[c];
        double t = a * f;
        if constexpr (KIND == MX_I32 || KIND == MX_LGL) {
            if (__double_as_longlong(f) == (long long)MX_NA_REAL_BITS) {
                t = 0.0;
                if (r < SP_RB) atomicOr(&na_rows[r >> 5], 1u << (r & 31));
            }
        }
        __hip_atomic_fetch_add(&acc[r], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int p = 0; p < npanels; p++) {
        const double *__restrict__ vp = vpan[WIDE ? 0 : (p & 1)];
        const int cbase = WIDE ? p << SP_WIDE_BITS : 0;
        // the next panel's elements are requested now and written to the other buffer after this panel's stream
        if constexpr (!WIDE) {
            const int cb_next = (p + 1) * SP_PANEL;
            if (p + 1 < npanels) {
#pragma unroll
                for (int i = 0; i < PER; i++) { const int c = cb_next + i * SP_THREADS + tid; stage[i] = c < K ? sp_factor<KIND>(v_, c) : 0.0; }
            }
        }
        const int s = so[p], e = so[p + 1];                           // multiples of 4
        for (int k0 = s + tid * 4; k0 < e; k0 += SP_THREADS * 4 * 2) {
            // two quads per thread and trip: 6 loads in flight (wide mode: then 8 independent reads of v)
            const int k1 = k0 + SP_THREADS * 4;
            const bool two = k1 < e;
            const i4 c0 = *reinterpret_cast<const i4 *>(ent + k0);
            const d2 a0 = *reinterpret_cast<const d2 *>(val + k0), a1 = *reinterpret_cast<const d2 *>(val + k0 + 2);
            const int kb = two ? k1 : k0;
            const i4 c1 = *reinterpret_cast<const i4 *>(ent + kb);
            const d2 b0 = *reinterpret_cast<const d2 *>(val + kb), b1 = *reinterpret_cast<const d2 *>(val + kb + 2);
            const double av[4] = {a0[0], a0[1], a1[0], a1[1]}, bv[4] = {b0[0], b0[1], b1[0], b1[1]};
            if constexpr (WIDE) {
                // all eight factors first (independent loads), then the sums
                double f0[4], f1[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f0[q] = sp_factor<KIND>(v_, min(cbase + (c0[q] & ((1 << CB) - 1)), K - 1));
                    f1[q] = sp_factor<KIND>(v_, min(cbase + (c1[q] & ((1 << CB) - 1)), K - 1));
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = (int)((unsigned)c0[q] >> CB);
                    double t = av[q] * f0[q];
                    if constexpr (KIND == MX_I32 || KIND == MX_LGL) {
                        if (__double_as_longlong(f0[q]) == (long long)MX_NA_REAL_BITS) { t = 0.0; if (r < SP_RB) atomicOr(&na_rows[r >> 5], 1u << (r & 31)); }
                    }
                    __hip_atomic_fetch_add(&acc[r], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (two) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int r = (int)((unsigned)c1[q] >> CB);
                        double t = bv[q] * f1[q];
                        if constexpr (KIND == MX_I32 || KIND == MX_LGL) {
                            if (__double_as_longlong(f1[q]) == (long long)MX_NA_REAL_BITS) { t = 0.0; if (r < SP_RB) atomicOr(&na_rows[r >> 5], 1u << (r & 31)); }
                        }
                        __hip_atomic_fetch_add(&acc[r], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) term(c0[q], av[q], vp, cbase);
                if (two) {
#pragma unroll
                    for (int q = 0; q < 4; q++) term(c1[q], bv[q], vp, cbase);
                }
            }
        }
        if constexpr (!WIDE) {
            if (p + 1 < npanels) {
#pragma unroll
                for (int i = 0; i < PER; i++) vpan[(p + 1) & 1][i * SP_THREADS + tid] = stage[i];
            }
            __syncthreads();           // the next panel is in LDS; every wavefront is done with this one's buffer and sums
        }
    }
    if constexpr (WIDE) __syncthreads();
    for (int i = tid; i < nr; i += SP_THREADS) {
        const double sum = acc[i];
        if constexpr (KIND == MX_F32) ((float *)y_)[r0 + i] = (float)sum;
        else if constexpr (KIND == MX_I32 || KIND == MX_LGL) ((double *)y_)[r0 + i] = (na_rows[i >> 5] >> (i & 31)) & 1u ? na_real() : sum;
        else ((double *)y_)[r0 + i] = sum;
    }
}

// WIDE matrices (more than 64 x 6,144 columns): ONE LAUNCH PER SUPER-PANEL of 2^18 columns.  Launch p adds, for every row block,
// the products of the block's entries in that super-panel to y (p = 0 writes).  All workgroups of a launch gather from the
// same 2 MB of v — which every XCD's L2 then holds — and the launches follow one another on the stream, so y needs no atomics.
// (First version, one launch whose workgroups walked the super-panels on their own: they drift apart within a few panels, two
// rounds of workgroups overlap, and the gather runs at the Infinity Cache's rate again — 2M x 2M, 50 per row: 1.46 ms, the
// one-shot flat kernel's 1.44.)
template <int KIND>
__global__ __launch_bounds__(SP_THREADS)
void spmv_wide_panel_kernel(int K, int npanels, int p, const int32_t *__restrict__ rb_row, const int32_t *__restrict__ seg_off,
                            const int32_t *__restrict__ ent, const double *__restrict__ val, const void *__restrict__ v_, void *__restrict__ y_)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int CB = SP_WIDE_BITS;
    __shared__ double acc[SP_RB + 8];
    __shared__ unsigned na_rows[(SP_RB + 8 + 31) / 32];
    const int tid = threadIdx.x, rb = blockIdx.x;
    const int r0 = rb_row[rb], nr = rb_row[rb + 1] - r0;
    if (nr <= 0) return;
    const int32_t *__restrict__ so = seg_off + (size_t)rb * npanels;
    const int s = so[p], e = so[p + 1];                               // multiples of 4
    if (s == e && p > 0) return;                                      // nothing to add (launch 0 still writes the zeros)
    for (int i = tid; i < SP_RB + 8; i += SP_THREADS) acc[i] = 0.0;
    if constexpr (KIND == MX_I32 || KIND == MX_LGL)
        for (int i = tid; i < (SP_RB + 8 + 31) / 32; i += SP_THREADS) na_rows[i] = 0u;
    __syncthreads();
    const int cbase = p << SP_WIDE_BITS;
    auto add = [&](int code, double a, double f) {
        const int r = (int)((unsigned)code >> CB);
        double t = a * f;
        if constexpr (KIND == MX_I32 || KIND == MX_LGL) {
            if (__double_as_longlong(f) == (long long)MX_NA_REAL_BITS) { t = 0.0; if (r < SP_RB) atomicOr(&na_rows[r >> 5], 1u << (r & 31)); }
        }
        __hip_atomic_fetch_add(&acc[r], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int k0 = s + tid * 4; k0 < e; k0 += SP_THREADS * 4 * 2) {
        // two quads per thread and trip: the six stream loads, then eight independent reads of v, then the sums
        const int k1 = k0 + SP_THREADS * 4;
        const bool two = k1 < e;
        const i4 c0 = *reinterpret_cast<const i4 *>(ent + k0);
        const d2 a0 = *reinterpret_cast<const d2 *>(val + k0), a1 = *reinterpret_cast<const d2 *>(val + k0 + 2);
        const int kb = two ? k1 : k0;
        const i4 c1 = *reinterpret_cast<const i4 *>(ent + kb);
        const d2 b0 = *reinterpret_cast<const d2 *>(val + kb), b1 = *reinterpret_cast<const d2 *>(val + kb + 2);
        const double av[4] = {a0[0], a0[1], a1[0], a1[1]}, bv[4] = {b0[0], b0[1], b1[0], b1[1]};
        double f0[4], f1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {                                 // (the padding's column 0 of the last super-panel stays in range)
            f0[q] = sp_factor<KIND>(v_, min(cbase + (c0[q] & ((1 << CB) - 1)), K - 1));
            f1[q] = sp_factor<KIND>(v_, min(cbase + (c1[q] & ((1 << CB) - 1)), K - 1));
        }
#pragma unroll
        for (int q = 0; q < 4; q++) add(c0[q], av[q], f0[q]);
        if (two) {
#pragma unroll
            for (int q = 0; q < 4; q++) add(c1[q], bv[q], f1[q]);
        }
    }
    __syncthreads();
    for (int i = tid; i < nr; i += SP_THREADS) {
        const double sum = acc[i];
        if constexpr (KIND == MX_F32) {
            float *y = (float *)y_ + r0 + i;
            *y = p == 0 ? (float)sum : (float)((double)*y + sum);
        } else if constexpr (KIND == MX_I32 || KIND == MX_LGL) {
            double *y = (double *)y_ + r0 + i;
            const double prev = p == 0 ? 0.0 : *y;
            const bool was_na = p > 0 && __double_as_longlong(prev) == (long long)MX_NA_REAL_BITS;
            *y = ((na_rows[i >> 5] >> (i & 31)) & 1u) || was_na ? na_real() : prev + sum;
        } else {
            double *y = (double *)y_ + r0 + i;
            *y = p == 0 ? sum : *y + sum;
        }
    }
}

}  // namespace mx

struct mx_spmv_plan {
    int m = 0, K = 0, nrb = 0, npanels = 0;
    bool wide = false;
    long long nnz = 0, slots = 0;
    int32_t *rb_row = nullptr;               // [nrb + 1] first row of every block
    int32_t *seg_off = nullptr;
    int32_t *ent = nullptr;
    double *val = nullptr;
};

extern "C" int mxd_spmv_plan_destroy(mx_spmv_plan *pl)
{
    if (!pl) return 0;
    if (pl->rb_row) (void)hipFree(pl->rb_row);
    if (pl->seg_off) (void)hipFree(pl->seg_off);
    if (pl->ent) (void)hipFree(pl->ent);
    if (pl->val) (void)hipFree(pl->val);
    delete pl;
    return 0;
}

extern "C" int mxd_spmv_plan_create(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                                    void *stream, mx_spmv_plan **plan_out)
{
    using namespace mx;
    MX_REQUIRE(plan_out && m >= 0 && K >= 0, "mxd_spmv_plan_create: bad arguments");
    // up to 64 LDS panels of 6,144 columns; wider matrices: super-panels of 2^18 columns read through L2
    const bool wide = ceil_div(K > 0 ? K : 1, SP_PANEL) > SP_MAX_PANELS;
    const int npanels = wide ? (int)ceil_div(K, 1 << SP_WIDE_BITS) : (int)ceil_div(K > 0 ? K : 1, SP_PANEL);
    MX_REQUIRE(!wide || npanels <= SP_WIDE_MAX_PANELS, "mxd_spmv_plan_create: more than 2^28 columns (use the one-shot kernels)");
    mx_spmv_plan *pl = new (std::nothrow) mx_spmv_plan();
    MX_REQUIRE(pl, "out of host memory");
    pl->m = m; pl->K = K; pl->npanels = npanels; pl->wide = wide;
    *plan_out = pl;
    if (m == 0) return 0;
    hipStream_t st = as_stream(stream);
    int32_t *counts = nullptr;
    void *scan_ws = nullptr;
    int rc = 1;
    do {
        // the entry count sizes the cuts (one 8-byte read-back; the build synchronises further down anyway)
        int32_t ends[2] = {0, 0};
        if (read_back_small(&ends[0], indptr, sizeof(int32_t), st) || read_back_small(&ends[1], indptr + m, sizeof(int32_t), st)) break;
        pl->nnz = (long long)ends[1] - ends[0];
        // row blocks: 4,096 rows, cut further where they hold more than E entries (1.06 mean blocks — 2 in a one-round launch —, at least 32k: a block must
        // be worth a workgroup) — spmv_plan_cuts_kernel
        // rows per block: 4,096 (what the LDS sums hold) for a matrix whose blocks then fill the machine; fewer for fewer rows — ~256
        // blocks, at least 64 rows each: 1e5 rows were 25 workgroups on 256 CUs (1e5 x 1e4, 64 per row: 0.170 -> 0.025 ms, the flat
        // kernel 0.038; 1e4 x 1e4, 500 per row: 2.35 -> 0.154 with blocks of 256 rows)
        const int RB = (int)std::min<long long>(SP_RB, std::max<long long>(64, (ceil_div(m, 256) + 63) / 64 * 64));
        const long long mean_block = (long long)((double)pl->nnz / (double)m * RB);
        // (pieces of half a mean block: with heavy-first dispatch the launch then ends within half a block of the ideal — rows
        // sorted by length, cfg3's shape: pieces of one mean block 0.142 ms, half a block 0.117, equal rows 0.079.  When the
        // blocks fill the machine in ONE round of workgroups — cfg3: 245 on 256 CUs — the first extra block costs a whole second
        // round: log-normal rows, sigma 1.5, a dozen blocks a few percent above the mean, 0.114 -> 0.170 ms; there only a block of
        // twice the mean is cut)
        const bool one_round = ceil_div(m, RB) <= 256;
        const long long E = std::max<long long>(8192, one_round ? 2 * mean_block : mean_block + mean_block / 16);
        const long long E_piece = std::max<long long>(4096, mean_block / 2);
        const int nA = (int)ceil_div(m, RB);
        const int max_cuts = (int)std::min<long long>((long long)nA + pl->nnz / E_piece + 1, (long long)m);
        long long *nrb_dev = nullptr;
        if (hipMalloc((void **)&pl->rb_row, ((size_t)max_cuts + 2) * 4) != hipSuccess) { set_error("spmv plan: allocation failed"); break; }
        if (hipMalloc((void **)&nrb_dev, 16) != hipSuccess) { set_error("spmv plan: allocation failed"); break; }
        hipLaunchKernelGGL(spmv_plan_cuts_kernel, dim3(1), dim3(1024), 0, st, m, RB, indptr, E, E_piece, pl->rb_row, max_cuts, nrb_dev);
        long long nrb_ll = 0;
        const int rb_rc = read_back_small(&nrb_ll, nrb_dev, sizeof(nrb_ll), st);
        (void)hipFree(nrb_dev);
        if (rb_rc) break;
        if (nrb_ll < 1 || nrb_ll > max_cuts) { set_error("spmv plan: %lld row blocks for a bound of %d", nrb_ll, max_cuts); break; }
        pl->nrb = (int)nrb_ll;
        const int64_t nseg = (int64_t)pl->nrb * npanels;
        if (hipMalloc((void **)&counts, (size_t)nseg * 4 + 16) != hipSuccess) { set_error("spmv plan: allocation failed"); break; }
        if (hipMalloc(&scan_ws, scan_workspace_bytes(nseg) + 16) != hipSuccess) { set_error("spmv plan: allocation failed"); break; }
        if (hipMalloc((void **)&pl->seg_off, ((size_t)nseg + 1) * 4) != hipSuccess) { set_error("spmv plan: allocation failed"); break; }
        if (wide) hipLaunchKernelGGL(spmv_plan_count_kernel<true>, dim3((unsigned)pl->nrb), dim3(SP_THREADS), 0, st, npanels, pl->rb_row, indptr, indices, counts);
        else hipLaunchKernelGGL(spmv_plan_count_kernel<false>, dim3((unsigned)pl->nrb), dim3(SP_THREADS), 0, st, npanels, pl->rb_row, indptr, indices, counts);
        if (exclusive_scan_i32(counts, nseg, pl->seg_off, (int64_t *)scan_ws, scan_ws, st)) break;
        long long total = 0;
        if (read_back_small(&total, scan_ws, sizeof(total), st)) break;
        if (total > (long long)INT_MAX) { set_error("spmv plan: too many entries"); break; }
        pl->slots = total;
        if (hipMalloc((void **)&pl->ent, (size_t)total * 4 + 64) != hipSuccess || hipMalloc((void **)&pl->val, (size_t)total * 8 + 64) != hipSuccess) {
            set_error("spmv plan: allocation failed");
            break;
        }
        if (wide) hipLaunchKernelGGL(spmv_plan_fill_kernel<true>, dim3((unsigned)pl->nrb), dim3(SP_THREADS), 0, st, npanels, pl->rb_row, indptr, indices, values,
                                     pl->seg_off, pl->ent, pl->val);
        else hipLaunchKernelGGL(spmv_plan_fill_kernel<false>, dim3((unsigned)pl->nrb), dim3(SP_THREADS), 0, st, npanels, pl->rb_row, indptr, indices, values,
                                pl->seg_off, pl->ent, pl->val);
        if (hipGetLastError() != hipSuccess) { set_error("spmv plan: launch failed"); break; }
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("spmv plan: build failed"); break; }     // scratch is freed below
        rc = 0;
    } while (0);
    if (counts) (void)hipFree(counts);
    if (scan_ws) (void)hipFree(scan_ws);
    if (rc) { mxd_spmv_plan_destroy(pl); *plan_out = nullptr; }
    return rc;
}

extern "C" int mxd_spmv_plan_info(const mx_spmv_plan *pl, int *npanels, int64_t *padded_entries)
{
    MX_REQUIRE(pl, "mxd_spmv_plan_info: null plan");
    if (npanels) *npanels = pl->npanels;
    if (padded_entries) *padded_entries = pl->slots;
    return 0;
}

extern "C" int mxd_spmv_plan_run(const mx_spmv_plan *pl, const void *v, int v_dtype, void *y, void *stream)
{
    using namespace mx;
    MX_REQUIRE(pl && y, "mxd_spmv_plan_run: bad arguments");
    if (pl->m == 0) return 0;
    MX_REQUIRE(v || pl->K == 0, "mxd_spmv_plan_run: null vector");
    hipStream_t st = as_stream(stream);
#define MX_SP(KIND)                                                                                                          \
    do {                                                                                                                     \
        if (pl->wide) {                                                                                                      \
            for (int p = 0; p < pl->npanels; p++)                                                                            \
                hipLaunchKernelGGL((spmv_wide_panel_kernel<KIND>), dim3((unsigned)pl->nrb), dim3(SP_THREADS), 0, st, pl->K,  \
                                   pl->npanels, p, pl->rb_row, pl->seg_off, pl->ent, pl->val, v, y);                         \
        } else hipLaunchKernelGGL((spmv_plan_kernel<KIND, false>), dim3((unsigned)pl->nrb), dim3(SP_THREADS), 0, st, pl->m, pl->K,       \
                                  pl->npanels, pl->rb_row, pl->seg_off, pl->ent, pl->val, v, y);                             \
    } while (0)
    switch (v_dtype) {
        case MX_F64: MX_SP(MX_F64); break;
        case MX_I32: MX_SP(MX_I32); break;
        case MX_LGL: MX_SP(MX_LGL); break;
        case MX_F32: MX_SP(MX_F32); break;
        default: return set_error("mxd_spmv_plan_run: unsupported vector dtype %d", v_dtype);
    }
#undef MX_SP
    MX_LAUNCH_CHECK();
    return 0;
}
