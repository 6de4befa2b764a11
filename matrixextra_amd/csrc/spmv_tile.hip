// spmv_tile.hip — CSR x dense-vector SpMV with the vector staged through LDS in column panels ("tile" kernel).
//
// Why: the lane-group kernel of spmv.hip reads v[j] with one 8-byte gather per nonzero.  For the headline CSR
// (1M x 100k, 32 / row) that is 32 M random reads of a 0.8 MB vector: no 32 KiB L1 holds it, so every gather moves a
// whole 128-B line from L2 into a CU (3.9 GB per product against 0.4 GB of algorithmic bytes) and the kernel runs at
// the L2 -> L1 line rate (~21 TB/s, 0.19 ms) whatever is done to the streaming side.  Here the gather never leaves the
// CU:
//   * ONE 1024-thread workgroup per CU takes a tile of ~24 k consecutive entries (whole rows; found by a two-level
//     1024-ary search of indptr) and reads them once, 16 B per lane per load (the (j, a) stream at full width: 24
//     entries per thread in registers, 288 KB in flight per CU);
//   * v is then swept through LDS in panels of 16 k columns (128 KiB, converted to f64 on the way in, NA -> NA_real):
//     for every panel each thread turns the entries whose column lies in the panel into products a * v[j], in place
//     (LDS reads, no global gather).  v is read once per tile, coalesced: 0.8 MB per 384 KB of entries;
//   * the products go to LDS (the panel buffer is free by then) and the rows are summed one thread per row in
//     STORAGE ORDER: y[row] = ((0 + p0) + p1) + ... with separately rounded products — bit for bit what the reference's
//     loop gives without FMA contraction (matmul.cpp:401-416), float accumulate for the float32 kind (:403).  Rows longer
//     than 256 entries are summed by a whole wavefront instead (reassociated, still within 1e-12).
// Entries may come in any column order; duplicates add up.  Used by spmv_launch (spmv.hip) when the vector spans at most
// TV_MAX_PANELS panels and there are enough entries to fill the chip; the lane-group kernel takes everything else.
#include "spmv_rows.h"

namespace mx {

constexpr int TV_THREADS = 1024;
constexpr int TV_EPT = 24;                              // entries per thread (24 x 3 registers: leaves room to batch the LDS reads)
constexpr int TV_ROUNDS = TV_EPT / 4;                   // rounds of 4 consecutive entries per thread
constexpr int TV_ROUND_ENTRIES = TV_THREADS * 4;        // 4096
constexpr int TV_CAP = TV_THREADS * TV_EPT;             // 24576 entries per pass
constexpr int TV_TARGET = TV_CAP - 1024;                // tiles are cut every TV_TARGET entries (slack for whole rows)
constexpr int TV_PANEL = 16384;                         // columns per LDS panel (128 KiB of f64)
constexpr int TV_HALF = TV_CAP / 2;                     // products staged per reduction pass
constexpr int TV_STAGE = TV_HALF + TV_HALF / 32;        // + one pad per 32 (row-stride-32 reads stay conflict-free)
constexpr int TV_ZERO = (TV_STAGE > TV_PANEL ? TV_STAGE : TV_PANEL);   // slot that always reads 0.0 (lanes outside the panel)
constexpr int TV_VBUF = TV_ZERO + 8;
constexpr int TV_MAX_PANELS = 24;

// R(e) = the largest r in [0, m] with indptr[r] <= e, for two targets at once: 1024-ary search, every thread probes
// once per level (two levels for a million rows).
__device__ __forceinline__ void tile_row_bounds(const int32_t *__restrict__ indptr, int m, long long e0, long long e1,
                                                int &R0, int &R1)
{
    const int tid = threadIdx.x;
    long long lo0 = 0, hi0 = m, lo1 = 0, hi1 = m;
    while (max(hi0 - lo0, hi1 - lo1) + 1 > TV_THREADS) {                // uniform
        const long long st0 = (hi0 - lo0 + TV_THREADS) / TV_THREADS, st1 = (hi1 - lo1 + TV_THREADS) / TV_THREADS;
        const long long i0 = lo0 + tid * st0, i1 = lo1 + tid * st1;
        const bool ok0 = i0 <= hi0, ok1 = i1 <= hi1;
        const int a0 = indptr[ok0 ? i0 : lo0], a1 = indptr[ok1 ? i1 : lo1];
        const int c0 = __syncthreads_count(ok0 && a0 <= e0), c1 = __syncthreads_count(ok1 && a1 <= e1);
        lo0 += (long long)(c0 - 1) * st0; hi0 = min(lo0 + st0 - 1, hi0);
        lo1 += (long long)(c1 - 1) * st1; hi1 = min(lo1 + st1 - 1, hi1);
    }
    const long long i0 = lo0 + tid, i1 = lo1 + tid;
    const bool ok0 = i0 <= hi0, ok1 = i1 <= hi1;
    const int a0 = indptr[ok0 ? i0 : lo0], a1 = indptr[ok1 ? i1 : lo1];
    const int c0 = __syncthreads_count(ok0 && a0 <= e0), c1 = __syncthreads_count(ok1 && a1 <= e1);
    R0 = (int)(lo0 + c0 - 1);
    R1 = (int)(lo1 + c1 - 1);
}

// columns [cbase, cbase + ncols) of v -> vbuf[0 .. ncols) as f64 (what the products are formed with):
//   numeric: as is;  integer: (double)v, NA_INTEGER -> NA_real bits;  logical: (double)(v != 0), NA -> NA_real bits
//   (matmul.cpp:406-411);  float32: (double)v
template <int KIND>
__device__ __forceinline__ void load_panel(double *__restrict__ vbuf, const void *__restrict__ v_, int cbase, int ncols)
{
    const int tid = threadIdx.x;
    if constexpr (KIND == MX_F64) {
        // LDS-DMA (global_load_lds_dwordx4): no VGPRs (the 32 entries per thread leave none for a staged panel), one
        // wave-instruction = 1 KiB of v landing at a wave-uniform LDS base + 16 B x lane.  Wave w takes the KiB chunks
        // w, w + 16, ...; lanes whose pair of columns is not whole are masked off (EXEC), the odd last column is
        // written by thread 0 once the DMA has landed (caller's barrier).
        const double *__restrict__ v = (const double *)v_ + cbase;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        constexpr int CHUNK = 128;                                    // doubles per wave-instruction
#pragma unroll
        for (int i = 0; i < TV_PANEL / (CHUNK * (TV_THREADS / 64)); i++) {
            const int c0 = (i * (TV_THREADS / 64) + wave) * CHUNK;   // wave-uniform
            if (c0 < ncols) {
                const int c = c0 + lane * 2;
                if (c + 2 <= ncols)
                    __builtin_amdgcn_global_load_lds(v + c, (__attribute__((address_space(3))) void *)(vbuf + c0), 16, 0, 0);
            }
        }
    } else if constexpr (KIND == MX_F32) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const float *__restrict__ v = (const float *)v_ + cbase;
        for (int c = tid * 4; c < ncols; c += TV_THREADS * 4) {
            if (c + 4 <= ncols) {
                const f4 q = *reinterpret_cast<const f4 *>(v + c);
#pragma unroll
                for (int i = 0; i < 4; i++) vbuf[c + i] = (double)q[i];
            } else {
                for (int i = 0; c + i < ncols; i++) vbuf[c + i] = (double)v[c + i];
            }
        }
    } else {
        typedef int i4 __attribute__((ext_vector_type(4)));
        const int32_t *__restrict__ v = (const int32_t *)v_ + cbase;
        auto conv = [](int yv) -> double {
            if (yv == MX_NA_INT) return na_bits_as_double();
            if constexpr (KIND == MX_LGL) return (double)(yv != 0); else return (double)yv;
        };
        for (int c = tid * 4; c < ncols; c += TV_THREADS * 4) {
            if (c + 4 <= ncols) {
                const i4 q = *reinterpret_cast<const i4 *>(v + c);
#pragma unroll
                for (int i = 0; i < 4; i++) vbuf[c + i] = conv(q[i]);
            } else {
                for (int i = 0; c + i < ncols; i++) vbuf[c + i] = conv(v[c + i]);
            }
        }
    }
    if (tid == 0) vbuf[TV_ZERO] = 0.0;
}

// STAMP: diagnostic build (tools/spmv_stamps.py): thread 0 of every workgroup records s_memtime at the phase boundaries
// (with a full wait in front of each stamp, so the phases do not overlap as they do in the real kernel: read shares)
template <int KIND, bool STAMP>
__global__ __launch_bounds__(TV_THREADS)
void spmv_tile_kernel(int m, long long nnz, int K, const int32_t *__restrict__ indptr,
                      const int32_t *__restrict__ indices, const double *__restrict__ values,
                      const void *__restrict__ v_, void *__restrict__ y_, int npanels,
                      unsigned long long *__restrict__ stamps)
{
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + i] = __builtin_readcyclecounter();
        }
    };
    stamp(0);
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ double vbuf[TV_VBUF];
    const int tid = threadIdx.x;
    const long long cut0 = (long long)blockIdx.x * TV_TARGET, cut1 = cut0 + TV_TARGET;
    int R0, R1;
    tile_row_bounds(indptr, m, cut0, cut1, R0, R1);
    if (blockIdx.x == 0) R0 = 0;                                      // leading empty rows belong to the first tile
    if (cut1 >= nnz) R1 = m;                                          // the last tile takes the trailing rows
    if (R0 >= R1) return;                                             // swallowed by a long row of an earlier tile
    const long long E0 = indptr[R0], E1 = indptr[R1];                 // uniform
    stamp(1);
    // the first row this thread owns: its bounds are needed in every reduction pass
    const int row_first = R0 + tid;
    int rs_first = 0, re_first = 0;
    if (row_first < R1) { rs_first = indptr[row_first]; re_first = indptr[row_first + 1]; }

    RowAcc<KIND> carry;                                               // the (one) owned row that crosses a pass boundary
    int carry_row = -1;
    bool first_pass = true;

    for (long long sb = E0 & ~3LL; first_pass || sb < E1; sb += TV_CAP) {
        // ---- the tile's entries, 4 consecutive ones per thread and round.  All 24 loads of a pass are issued back to
        // back (clamped offsets, no branch in between: a branch per round made the compiler wait for each round's
        // loads before issuing the next, i.e. eight exposed HBM latencies per tile), masks are applied afterwards.
        // Addresses are a wave-uniform base + a 32-bit offset per lane.
        int col[TV_EPT];
        double x[TV_EPT];
        const long long lim4 = (nnz - 4) & ~3LL;                      // the arrays' last whole aligned quad (nnz >= 4)
        const bool any_whole = sb <= lim4;                            // uniform
        const int32_t *__restrict__ ip = indices + (any_whole ? sb : 0);
        const double *__restrict__ xp = values + (any_whole ? sb : 0);
        const int n_rel = (int)min(E1 - sb, (long long)TV_CAP);       // our entries of this pass are [head_rel, n_rel)
        const int head_rel = first_pass ? (int)(E0 - sb) : 0;
        const int lim_rel = any_whole ? (int)min(lim4 - sb, (long long)TV_CAP) : -1;
#pragma unroll
        for (int r = 0; r < TV_ROUNDS; r++) {
            const int er = r * TV_ROUND_ENTRIES + tid * 4;
            const int off = er < n_rel ? max(min(er, lim_rel), 0) : 0;   // aligned, inside the arrays
            const i4 c = *reinterpret_cast<const i4 *>(ip + off);
            const d2 a01 = *reinterpret_cast<const d2 *>(xp + off);
            const d2 a23 = *reinterpret_cast<const d2 *>(xp + off + 2);
            col[r * 4 + 0] = c[0]; col[r * 4 + 1] = c[1]; col[r * 4 + 2] = c[2]; col[r * 4 + 3] = c[3];
            x[r * 4 + 0] = a01[0]; x[r * 4 + 1] = a01[1]; x[r * 4 + 2] = a23[0]; x[r * 4 + 3] = a23[1];
        }
#pragma unroll
        for (int r = 0; r < TV_ROUNDS; r++) {
            const int er = r * TV_ROUND_ENTRIES + tid * 4;
            const bool whole = er >= head_rel && er + 4 <= n_rel && er <= lim_rel;
            if (!whole) {                                             // the tile's first / last quad, quads past lim_rel (fixed below)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool mine = er + q >= head_rel && er + q < n_rel && er <= lim_rel;
                    col[r * 4 + q] = mine ? col[r * 4 + q] : -1;
                }
            }
        }
        if (sb + TV_CAP > lim4) {                                     // uniform: only the pass that holds the arrays' last quad
#pragma unroll
            for (int r = 0; r < TV_ROUNDS; r++) {
                const int er = r * TV_ROUND_ENTRIES + tid * 4;
                if (er > lim_rel && er < n_rel) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (er + q >= head_rel && er + q < n_rel) {
                            col[r * 4 + q] = indices[sb + er + q];
                            x[r * 4 + q] = values[sb + er + q];
                        }
                    }
                }
            }
        }
        stamp(2);
        // ---- sweep v through LDS: x[s] <- a * v[col] for the entries whose column is in the panel
        for (int pb = 0; pb < npanels; pb++) {
            const int cbase = pb * TV_PANEL;
            const int ncols = min(TV_PANEL, K - cbase);
            __syncthreads();                                          // the previous panel / staging pass is done with vbuf
            load_panel<KIND>(vbuf, v_, cbase, ncols);
            if constexpr (KIND == MX_F64) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed
                if ((ncols & 1) && tid == 0) vbuf[ncols - 1] = ((const double *)v_)[cbase + ncols - 1];
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < TV_EPT; s++) {
                const unsigned t = (unsigned)(col[s] - cbase);        // col = -1 (not ours) is never inside
                const bool in = t < (unsigned)ncols;
                const double val = vbuf[in ? t : (unsigned)TV_ZERO];
                const double prod = term<KIND>(x[s], val);            // NA element: the term is NA_REAL itself
                x[s] = in ? prod : x[s];
                if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // 8 LDS reads in flight at a time: bounds the temporaries
            }
        }
        stamp(3);
        // ---- products -> LDS, half a pass at a time; rows summed one thread per row in storage order
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const long long hb = sb + (long long)half * TV_HALF;      // staging origin
            if (hb >= E1 && !(first_pass && half == 0)) break;        // uniform
            const long long he = min(hb + TV_HALF, E1);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < TV_ROUNDS / 2; r++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int pos = r * TV_ROUND_ENTRIES + tid * 4 + q;
                    vbuf[pos + (pos >> 5)] = x[(half * (TV_ROUNDS / 2) + r) * 4 + q];
                }
            }
            __syncthreads();
            reduce_rows<KIND, TV_THREADS>(vbuf, hb, he, E0, R0, R1, indptr, rs_first, re_first, carry, carry_row,
                                          first_pass && half == 0, y_);
        }
        first_pass = false;
        stamp(4);
    }
}

static unsigned long long *g_tile_stamps = nullptr;                   // set by mxd_debug_spmv_tile_stamps

// can the tile kernel take this product?  (16-B aligned arrays for the wide loads, a vector of at most TV_MAX_PANELS
// panels, enough entries to give every CU a tile)
// `forced`: an explicit MX_SPMV_TILE request only has to be runnable (the tests drive small shapes through it)
bool spmv_tile_ok(int m, int64_t nnz, int K, const int32_t *indices, const double *values, const void *v, bool forced)
{
    if (m <= 0 || K <= 0 || nnz < (forced ? 4 : (int64_t)64 * TV_TARGET)) return false;
    if (((uintptr_t)indices | (uintptr_t)values | (uintptr_t)v) & 15) return false;
    return (K + TV_PANEL - 1) / TV_PANEL <= TV_MAX_PANELS;
}

int spmv_tile_launch(int m, int64_t nnz, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                     const void *v, int v_dtype, void *y, hipStream_t st)
{
    const unsigned grid = (unsigned)ceil_div(nnz, TV_TARGET);
    const int npanels = (K + TV_PANEL - 1) / TV_PANEL;
#define MX_TV(KIND)                                                                                          \
    do { if (g_tile_stamps)                                                                                   \
        hipLaunchKernelGGL((spmv_tile_kernel<KIND, true>), dim3(grid), dim3(TV_THREADS), 0, st, m,            \
                           (long long)nnz, K, indptr, indices, values, v, y, npanels, g_tile_stamps);         \
    else                                                                                                      \
        hipLaunchKernelGGL((spmv_tile_kernel<KIND, false>), dim3(grid), dim3(TV_THREADS), 0, st, m,           \
                           (long long)nnz, K, indptr, indices, values, v, y, npanels,                         \
                           (unsigned long long *)nullptr); } while (0)
    switch (v_dtype) {
        case MX_F64: MX_TV(MX_F64); break;
        case MX_I32: MX_TV(MX_I32); break;
        case MX_LGL: MX_TV(MX_LGL); break;
        case MX_F32: MX_TV(MX_F32); break;
        default: return set_error("spmv: unsupported vector dtype %d", v_dtype);
    }
#undef MX_TV
    MX_LAUNCH_CHECK();
    return 0;
}

}  // namespace mx

// diagnostic: device buffer of 8 x (number of tiles) uint64 that the tile kernel's STAMP build fills; NULL switches back
extern "C" int mxd_debug_spmv_tile_stamps(void *buf)
{
    mx::g_tile_stamps = (unsigned long long *)buf;
    return 0;
}
