// spmm_tile.hip — CSR x dense SpMM for gfx950 (MI355X): the kernel for DENSE-ISH sparse operands, where a row block of A
// comes back to every row of B several times (density x rows-per-block >~ 2).  That is the operand the reference itself
// publishes and tests: dense 100 x 1e4 %*% CSC 1e4 x 1e4 at density .05 (vignettes/Introducing_MatrixExtra.Rmd:247-251 ->
// matmul_dense_csc_numeric -> gemm_csr_drm_as_drm, src/matmul.cpp:188-235, :118-142), test matrices at density .4
// (tests/testthat/test-matmul.R:108-114).
//
// Why another kernel: every other SpMM kernel here gathers one row of B per entry from L2 into registers — nnz * n * s
// bytes of L2 -> L1 line reads (4 GB for the vignette product against an 8 MB B), the ceiling the row-split kernel sits at.
// Here B goes through LDS:
//
//   * a workgroup owns a ROW BLOCK of R rows of A (up to ~220) and one COLUMN SLAB of W = 16 lanes x 16 bytes x CPL
//     columns of B / C (256 or 512 bytes of a row); workgroups that share an XCD share a slab;
//   * it walks [0, K) in K-TILES: TK rows of the slab of B (64 KB) are brought into LDS by LOADER WAVEFRONTS (two; their
//     instruction stream is on the critical path: a pointer add and the DMA per 1-KB piece) with LDS-DMA
//     (global_load_lds_dwordx4: coalesced 256 / 512-byte row pieces, no registers), tile t + 1 while the compute wavefronts
//     work on tile t (two buffers, one barrier per tile) — B is read from L2 once per row block instead of once per entry:
//     m * K * n * s / R bytes;
//   * a compute wavefront is 4 lane groups of 16 (one DPP row each); a group owns 2 .. 5 rows of A — dealt by the host so that
//     every SIMD makes the same number of (row, tile) visits, tile_deal below — and keeps their sums in registers through
//     the whole sweep.  Rows are sorted by column, so the entries of a row that fall into a tile are the
//     next ones in storage order: the group keeps a window of the row's next 16 / 32 entries in registers (lane l = entry
//     pos + l), counts those below the tile's end with a ballot, and streams them: the entry's LDS row offset is handed to
//     the group's 16 lanes by DPP row_newbcast folded INTO the address add (v_add_u32_dpp), its value into the FMA
//     (v_fmac_f64_dpp) — three VALU instructions and one ds_read_b128 per (4 rows x 1 entry x 256 bytes);
//   * a group that has fewer entries in the tile than its neighbours multiplies the window's -0.0 padding with a row of
//     zeros kept in LDS: x + (-0.0 * 0.0) = x for every x (signed zeros, Inf and NaN included), so there is no select in
//     the loop and the sum of a row is the reference's storage-order FMA chain BIT FOR BIT, in both layouts of C;
//   * a row that is NOT sorted by column (flagged by the caller's sortedness pass) is summed whole from global memory
//     before the sweep, by the same group in storage order — correct, slow, rare;
//   * row-major C: 16-byte stores straight from the registers; column-major C: through LDS, whole column segments of R rows.
//
// Roofline: HBM-bound by the contract's algorithmic bytes.  In practice (rocprofv3 at the vignette's shape,
// profiles/r05_vignette_extras_pmc.json): LDS 40 % busy (no bank conflicts), VALU 54 %, waves a quarter of their cycles at the
// tile barrier — the steps executed are 1.5x the useful ones (four groups in lockstep, batches of four, the 4-column last
// slab of n = 100).  `lds_read` in bench.py: nnz x slabs x 256 B per launch.
//
// Rows of uneven length (round 6; the kernels behind tile_spmm's map, built once per matrix and geometry and kept):
//   tile_cuts_kernel / tile_cuts_greedy_kernel  row blocks of about equal entries (every R rows, heavy ones cut further) / of
//                                               about equal weight in one pass over the rows (units = rows + parts <= the slots);
//   tile_deal_rows_kernel                       a block's rows ranked by length and dealt to the lane groups (four rows of nearly
//                                               one length per visit, visits to the least-loaded SIMD); with a matrix profile
//                                               that shows rows several windows per tile long: those rows cut into 2 / 4 / 8
//                                               interleaved parts on slots of their own (count pass, tile_part_scan_kernel, cut pass);
//   tile_combine_kernel                         adds a cut row's parts in order.
// Dealt rows keep every bit; a cut row is the chain regrouped (device-level callers with a profile only, never the exports).
#include "spmm_common.h"
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace mx {

constexpr int TL_G = 16;                  // lanes per row of the slab: one DPP row
constexpr int TL_NG = MX_WAVE / TL_G;     // rows of A a wavefront walks at once
constexpr int TL_MAX_WAVES = 15;          // compute wavefronts per workgroup (+ 1 or 2 loaders <= 1,024 threads)

template <int U> __device__ __forceinline__ unsigned tl_addr(unsigned off, unsigned lane16)
{
    unsigned r;
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(off), "v"(lane16), "i"(U));
    return r;
}
template <int U> __device__ __forceinline__ void tl_fmac(double &acc, double a, double b)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "i"(U));
}
template <int U> __device__ __forceinline__ void tl_fmac(float &acc, float a, float b)
{
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "i"(U));
}

// One visit of (a row of every group, the tile): entries 0 .. maxc - 1 of the window in batches of NB = 4 / CPL entries (four
// ds_read_b128 either way).  The reads of batch q + 1 are issued BEFORE the FMAs of batch q (two register sets): without
// that the compiler waits for every read right behind its issue and a wavefront spends an LDS round trip per entry.  Whole
// batches only: a padded step (-0.0 x the row of zeros) is a no-op on the sums.
template <typename real_t, int CPL, int NB> struct TlBatch { typename VecT<real_t, 16 / (int)sizeof(real_t)>::type b[NB][CPL]; };

template <typename real_t, int CPL, int NB, int U0>
__device__ __forceinline__ void tl_read(const char *smem, unsigned off, unsigned lane16, TlBatch<real_t, CPL, NB> &t)
{
    using V = typename VecT<real_t, 16 / (int)sizeof(real_t)>::type;
    unsigned a[NB];
    a[0] = tl_addr<(U0 + 0) & 15>(off, lane16);
    if constexpr (NB > 1) a[1] = tl_addr<(U0 + 1) & 15>(off, lane16);
    if constexpr (NB > 2) { a[2] = tl_addr<(U0 + 2) & 15>(off, lane16); a[3] = tl_addr<(U0 + 3) & 15>(off, lane16); }
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int e = 0; e < NB; e++) t.b[e][c] = *reinterpret_cast<const V *>(smem + a[e] + c * 256);
}
template <typename real_t, int CPL, int NB, int U0>
__device__ __forceinline__ void tl_fma(real_t aa, const TlBatch<real_t, CPL, NB> &t, real_t (&acc)[CPL][16 / sizeof(real_t)])
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int v = 0; v < VEC; v++) tl_fmac<(U0 + 0) & 15>(acc[c][v], aa, t.b[0][c][v]);
    if constexpr (NB > 1) {
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int v = 0; v < VEC; v++) tl_fmac<(U0 + 1) & 15>(acc[c][v], aa, t.b[1][c][v]);
    }
    if constexpr (NB > 2) {
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int v = 0; v < VEC; v++) tl_fmac<(U0 + 2) & 15>(acc[c][v], aa, t.b[2][c][v]);
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int v = 0; v < VEC; v++) tl_fmac<(U0 + 3) & 15>(acc[c][v], aa, t.b[3][c][v]);
    }
}
// NBATCH batches of one window, straight-line: the reads of batch b + 1 go out before the FMAs of batch b, no branch in between
// (rocprofv3 on the first version, one exit test + one read-ahead test per batch: 12 branches and 62 scalar instructions per
// visit beside 75 vector ones, waves waiting half of their cycles)
template <typename real_t, int CPL, int NB, int NBATCH>
__device__ __forceinline__ void tl_fixed(const char *smem, unsigned off, real_t aa, unsigned lane16, real_t (&acc)[CPL][16 / sizeof(real_t)])
{
    TlBatch<real_t, CPL, NB> p, q;
    tl_read<real_t, CPL, NB, 0>(smem, off, lane16, p);
    if constexpr (NBATCH > 1) tl_read<real_t, CPL, NB, NB>(smem, off, lane16, q);
    tl_fma<real_t, CPL, NB, 0>(aa, p, acc);
    if constexpr (NBATCH > 2) tl_read<real_t, CPL, NB, 2 * NB>(smem, off, lane16, p);
    if constexpr (NBATCH > 1) tl_fma<real_t, CPL, NB, NB>(aa, q, acc);
    if constexpr (NBATCH > 3) tl_read<real_t, CPL, NB, 3 * NB>(smem, off, lane16, q);
    if constexpr (NBATCH > 2) tl_fma<real_t, CPL, NB, 2 * NB>(aa, p, acc);
    if constexpr (NBATCH > 4) tl_read<real_t, CPL, NB, 4 * NB>(smem, off, lane16, p);
    if constexpr (NBATCH > 3) tl_fma<real_t, CPL, NB, 3 * NB>(aa, q, acc);
    if constexpr (NBATCH > 5) tl_read<real_t, CPL, NB, 5 * NB>(smem, off, lane16, q);
    if constexpr (NBATCH > 4) tl_fma<real_t, CPL, NB, 4 * NB>(aa, p, acc);
    if constexpr (NBATCH > 6) tl_read<real_t, CPL, NB, 6 * NB>(smem, off, lane16, p);
    if constexpr (NBATCH > 5) tl_fma<real_t, CPL, NB, 5 * NB>(aa, q, acc);
    if constexpr (NBATCH > 7) tl_read<real_t, CPL, NB, 7 * NB>(smem, off, lane16, q);
    if constexpr (NBATCH > 6) tl_fma<real_t, CPL, NB, 6 * NB>(aa, p, acc);
    if constexpr (NBATCH > 7) tl_fma<real_t, CPL, NB, 7 * NB>(aa, q, acc);
}
// entries [16 w, 16 w + 16) of the window: `left` = how many of them any group still has (wave-uniform, >= 1); whole batches:
// a padded step (-0.0 x the row of zeros) is a no-op on the sums
template <typename real_t, int CPL>
__device__ __forceinline__ void tl_window(const char *smem, int left, unsigned off, real_t aa, unsigned lane16,
                                          real_t (&acc)[CPL][16 / sizeof(real_t)])
{
    constexpr int NB = 4 / CPL;                         // (batches of 2 with 256-byte slabs: measured, no gain — 0.111 against 0.107 ms)
    const int nb = (left + NB - 1) / NB;
    if constexpr (NB == 4) {
        if (nb <= 2) { if (nb <= 1) tl_fixed<real_t, CPL, NB, 1>(smem, off, aa, lane16, acc); else tl_fixed<real_t, CPL, NB, 2>(smem, off, aa, lane16, acc); }
        else { if (nb == 3) tl_fixed<real_t, CPL, NB, 3>(smem, off, aa, lane16, acc); else tl_fixed<real_t, CPL, NB, 4>(smem, off, aa, lane16, acc); }
    } else {
        if (nb <= 4) {
            if (nb <= 2) { if (nb <= 1) tl_fixed<real_t, CPL, NB, 1>(smem, off, aa, lane16, acc); else tl_fixed<real_t, CPL, NB, 2>(smem, off, aa, lane16, acc); }
            else { if (nb == 3) tl_fixed<real_t, CPL, NB, 3>(smem, off, aa, lane16, acc); else tl_fixed<real_t, CPL, NB, 4>(smem, off, aa, lane16, acc); }
        } else {
            if (nb <= 6) { if (nb == 5) tl_fixed<real_t, CPL, NB, 5>(smem, off, aa, lane16, acc); else tl_fixed<real_t, CPL, NB, 6>(smem, off, aa, lane16, acc); }
            else { if (nb == 7) tl_fixed<real_t, CPL, NB, 7>(smem, off, aa, lane16, acc); else tl_fixed<real_t, CPL, NB, 8>(smem, off, aa, lane16, acc); }
        }
    }
}

// TILE: bytes of one K-tile of the slab in LDS (two of them + one row of zeros); WIN: 16-entry windows per row (1 or 2)
// SPLIT: slots may hold PARTS of long rows (tile_deal_rows_kernel) — a stride and a row of Cx per slot
template <typename real_t, int CPL, int RG, int WIN, int TILE, bool COLMAJOR, bool SPLIT>
__global__ __launch_bounds__((TL_MAX_WAVES + 1) * MX_WAVE)
void spmm_tile_kernel(int m, int n, int K, int nslabs, int nrb,
                      const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const double *__restrict__ values,
                      const unsigned char *__restrict__ unsorted, const int32_t *__restrict__ perm, const int32_t *__restrict__ cuts,
                      const int32_t *__restrict__ split, real_t *__restrict__ Cx,
                      const real_t *__restrict__ B, size_t ldb, real_t *__restrict__ C, size_t ldc, int c_vec, int nl, unsigned long long rgw, unsigned long long *__restrict__ stamps)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = TL_G * VEC * CPL;              // columns of the slab
    constexpr int ROWB = 256 * CPL;                  // bytes of one row of the tile
    constexpr int TK = TILE / ROWB;                  // rows of B per tile
    constexpr unsigned ZOFF = 2 * TILE;              // the row of zeros
    using V = typename VecT<real_t, VEC>::type;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE + ROWB];

    // workgroups that share an XCD (blockIdx % 8) share a slab whenever the slabs divide 8: an XCD's L2 then holds one
    // slab of B (K x 256 bytes), not all of it
    int slab, rb;
    {
        const int id = blockIdx.x;
        if (8 % nslabs == 0) { const int x = id & 7, per = 8 / nslabs; slab = x % nslabs; rb = (id >> 3) * per + x / nslabs; }
        else { slab = id % nslabs; rb = id / nslabs; }
    }
    if (rb >= nrb) return;
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x / MX_WAVE);
    const int nw = (int)(blockDim.x / MX_WAVE) - nl; // compute wavefronts; the last nl wavefronts are the loaders
    // rows per lane group of every compute wavefront, 4 bits each (<= RG): consecutive wavefronts sit on different SIMDs (dealt
    // 0 -> 2 -> 1 -> 3 round robin), so the host deals the rows such that every SIMD makes the same number of visits per tile
    int R = 0, my_rg = 0, my_base = 0;
    for (int w = 0; w < nw; w++) {
        const int r = (int)((rgw >> (4 * w)) & 15);
        if (w == wave) { my_rg = r; my_base = R; }
        R += r;
    }
    R *= TL_NG;
    // rows of this block: R consecutive ones, or — rows of uneven length — the range the cuts give it (<= R rows, about equal
    // entry counts; blocks past the last cut are empty: the grid is sized by a bound, tile_cuts_kernel)
    const int row0 = cuts ? uniform(cuts[rb]) : rb * R;
    const int nrows_blk = cuts ? uniform(cuts[rb + 1]) - row0 : min(R, m - row0);
    if (nrows_blk <= 0) return;
    const int c0 = slab * W;
    const int T = (K + TK - 1) / TK;

    if (wave >= nw) {
        // ---- the loaders (nl wavefronts, 1-KB pieces dealt round robin): tile t + 1 travels while tile t is being read.
        // Their instruction stream is on the critical path (a tile is 64 instructions; the wavefront shares its SIMD with
        // compute wavefronts): per piece one 64-bit pointer add and the DMA
        constexpr int LPR = ROWB / 16;               // lanes per row piece
        constexpr int RPP = MX_WAVE / LPR;           // rows per 1-KB instruction
        constexpr int NP = TILE / 1024;
        const int li = wave - nw;
        const int rp = lane / LPR;
        int col = c0 + (lane % LPR) * VEC;
        if (col > n - VEC) col = n - VEC;            // past the last column: any valid address (those sums are never stored)
        const real_t *src0 = B + col;
        const size_t stride = (size_t)RPP * ldb * (size_t)nl;
        auto fill = [&](int t) {
            const unsigned dst = (unsigned)(t & 1) * TILE;
            const int k0 = t * TK;
            if (k0 + TK <= K) {
                const real_t *src = src0 + (size_t)(k0 + li * RPP + rp) * ldb;
#pragma unroll 4
                for (int p = li; p < NP; p += nl) {
                    __builtin_amdgcn_global_load_lds((const void *)src, (__attribute__((address_space(3))) void *)(smem + dst + p * 1024), 16, 0, 0);
                    src += stride;
                }
            } else {
                for (int p = li; p < NP; p += nl) {
                    int k = k0 + p * RPP + rp;
                    if (k > K - 1) k = K - 1;        // past the last row: never referenced by an entry
                    __builtin_amdgcn_global_load_lds((const void *)(src0 + (size_t)k * ldb),
                                                     (__attribute__((address_space(3))) void *)(smem + dst + p * 1024), 16, 0, 0);
                }
            }
        };
        fill(0);
        for (int t = 0; t < T; t++) {
            __syncthreads();                         // (waits for the DMA of tile t first: vmcnt(0))
            if (t + 1 < T) fill(t + 1);
        }
        if constexpr (!COLMAJOR) return;
    }

    const int g = lane / TL_G, lg = lane % TL_G;
    real_t acc[RG][CPL][VEC];
    // SPLIT costs no register: log2 of a part's stride (0: a whole row) rides in bits 29 / 30 of rows[] (m < 2^29, the host
    // checks), and the part's row of Cx (-1: the sum goes to C) is read again from split[] by the epilogue
    int rows[RG];
    auto ROW = [&](int i) -> int { if constexpr (SPLIT) return rows[i] & 0x1FFFFFFF; else return rows[i]; };
    auto SH = [&](int i) -> int { if constexpr (SPLIT) return (int)((unsigned)rows[i] >> 29); else return 0; };
    auto XR = [&](int i) -> int {
        if constexpr (SPLIT) {
            const int sp = split[(size_t)rb * R + (my_base + i) * TL_NG + lane / TL_G];
            return sp >= 0 ? (sp >> 12) - 1 : -1;
        } else return -1;
    };
    if (wave < nw) {
        const unsigned lane16 = (unsigned)lg * 16;
        int pos[RG], end[RG];
        int jv[RG][WIN];
        double av[RG][WIN];
        if (wave == 0) {                             // the row of zeros
            if (lane < ROWB / 16) *reinterpret_cast<V *>(smem + ZOFF + lane * 16) = V{};
        }
#pragma unroll
        for (int i = 0; i < RG; i++) {
            // slot -> row: consecutive rows, or — rows of uneven length — the block's rows dealt by length (tile_deal_rows_kernel)
            const int slot = (my_base + i) * TL_NG + g;
            const int row = i < my_rg ? (perm ? perm[(size_t)rb * R + slot] : row0 + slot) : m;      // (m: no such row)
            rows[i] = row;
            pos[i] = end[i] = 0;
            if (row < m) { pos[i] = indptr[row]; end[i] = indptr[row + 1]; }
            // a LONG row cut into P = 2 / 4 / 8 interleaved parts (tile_deal_rows_kernel): this slot sums entries q, q + P, q + 2P ...
            // of the row — its column ids still ascend — into a row of the scratch matrix Cx (part 0: into C), added up afterwards
            if constexpr (SPLIT) if (split && row < m) {
                const int sp = split[(size_t)rb * R + slot];
                if (sp >= 0) {                                        // q | log2(P) << 8 | (row of Cx + 1) << 12 (0: part 0)
                    const int q = sp & 255, len = end[i] - pos[i];
                    // (a kept map may have cut a row that is NOT sorted by column in today's matrix: part 0 then sums all of it —
                    // whole, from global memory, below — and the other parts nothing: their rows of Cx become zeros)
                    const bool whole = unsorted && unsorted[row] != 0;
                    if (whole) {
                        if (q > 0) end[i] = pos[i];
                    } else {
                        const int lg2 = (sp >> 8) & 3;
                        rows[i] |= lg2 << 29;
                        pos[i] += q;
                        end[i] = pos[i] + (len > q ? (len - q + (1 << lg2) - 1) >> lg2 : 0);  // (pos + the part's entry COUNT)
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CPL; c++)
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[i][c][v] = 0;
        }
        // ---- rows that are not sorted by column: summed whole from global memory, in storage order
        if (unsorted) {
#pragma unroll
            for (int i = 0; i < RG; i++) {
                const bool slow = ROW(i) < m && unsorted[ROW(i)] != 0;
                if (__ballot(slow) == 0) continue;
                const int gbase = g * TL_G;
                int longest = slow ? end[i] - pos[i] : 0;
#pragma unroll
                for (int o = TL_G; o < MX_WAVE; o <<= 1) longest = max(longest, __shfl_xor(longest, o, MX_WAVE));
                longest = uniform(longest);
                for (int k0 = 0; k0 < longest; k0 += TL_G) {
                    const int k = pos[i] + k0 + lg;
                    int jw = 0;
                    real_t aw = 0;
                    if (slow && k < end[i]) { jw = indices[k]; aw = (real_t)values[k]; }
                    for (int u = 0; u < TL_G; u++) {
                        const int jj = __shfl(jw, gbase + u, MX_WAVE);
                        const real_t a = __shfl(aw, gbase + u, MX_WAVE);
                        const bool valid = slow && pos[i] + k0 + u < end[i];
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            int col = c0 + c * (TL_G * VEC) + lg * VEC;
                            if (col > n - VEC) col = n - VEC;
                            real_t b[VEC];
                            vload<real_t, VEC>(b, B + (size_t)(valid ? jj : 0) * ldb + col);
#pragma unroll
                            for (int v = 0; v < VEC; v++) acc[i][c][v] = valid ? mx_fma(a, b[v], acc[i][c][v]) : acc[i][c][v];
                        }
                    }
                }
                if (slow) pos[i] = end[i];
            }
        }
        // ---- the sweep.  The window of row i: entries pos .. pos + 16 WIN - 1, lane l of the group holds entry pos + l (+ 16).
        // Entries are read through buffer descriptors that start at the row block's first entry and end with the matrix:
        // a lane past the end reads zero instead of faulting, so EVERY lane loads on every visit — the number of loads in
        // flight is the same on every path and the compiler waits for THIS row's window with a counted vmcnt, the other
        // rows' windows still in flight — and the second window is the first one's address + an immediate.
        const int first = uniform(indptr[min(row0, m)]);
        const long long left_entries = (long long)uniform(indptr[m]) - first;
        const unsigned bytes_j = (unsigned)min(left_entries * 4, 0xFFFFFFFFll), bytes_x = (unsigned)min(left_entries * 8, 0xFFFFFFFFll);
        const __amdgpu_buffer_rsrc_t rs_j = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(indices + first), 0, bytes_j, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(values + first), 0, bytes_x, 0x00020000);
        int posl[RG], rem[RG];                       // pos - first + lane-in-group; entries the row has left
#pragma unroll
        for (int i = 0; i < RG; i++) { posl[i] = pos[i] - first + (lg << SH(i)); rem[i] = end[i] - pos[i]; }
        auto load_window = [&](int i) {
            const int o4 = posl[i] << 2, o8 = posl[i] << 3;
#pragma unroll
            for (int w = 0; w < WIN; w++) {
                jv[i][w] = __builtin_amdgcn_raw_buffer_load_b32(rs_j, o4 + ((w * TL_G * 4) << SH(i)), 0, 0);     // (used as loaded: nothing waits here)
                av[i][w] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_x, o8 + ((w * TL_G * 8) << SH(i)), 0, 0));
            }
        };
#pragma unroll
        for (int i = 0; i < RG; i++) {
            load_window(i);
            asm volatile("" : : : "memory");         // (the loop's order of loads, so that its counted waits hold from the first round on)
        }
        const int g8 = g * 8;

        unsigned long long st_wait = 0, st_t0 = 0;
        if (stamps) st_t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < T; t++) {
            unsigned long long st_a = 0;
            if (stamps) st_a = __builtin_amdgcn_s_memtime();
            __syncthreads();                         // tile t has landed; everybody is done with tile t - 1
            if (stamps) st_wait += __builtin_amdgcn_s_memtime() - st_a;
            const int kend = (t + 1) * TK;
            const unsigned basek = (unsigned)(t & 1) * TILE - (unsigned)(t * TK) * ROWB;     // LDS offset of row j: basek + j * ROWB
            // one visit of (row i of every group, tile t); returns the largest number of entries a group had in the tile
            auto visit = [&](int i) -> int {
                unsigned off[WIN];
                real_t aa[WIN];
                unsigned lo[WIN], hi[WIN];           // the ballots: 16 bits per group
#pragma unroll
                for (int w = 0; w < WIN; w++) {
                    // (the ballot of each comparison is the comparison's own result register; the ballot of their AND would be
                    // re-materialised lane by lane)
                    const bool valid = lg + w * TL_G < rem[i], below = jv[i][w] < kend;
                    const bool in = valid & below;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(valid) & __builtin_amdgcn_ballot_w64(below);
                    lo[w] = (unsigned)mask; hi[w] = (unsigned)(mask >> 32);
                    off[w] = in ? basek + (unsigned)jv[i][w] * ROWB : ZOFF;
                    aa[w] = in ? (real_t)av[i][w] : (real_t)-0.0;      // narrowed per entry for f32 (matmul.cpp:53-57)
                }
                // entries in the tile per group (scalar): both windows of a group in one word, one popcount
                int t0, t1, t2, t3;
                if constexpr (WIN == 1) {
                    t0 = __popc(lo[0] & 0xFFFFu); t1 = __popc(lo[0] >> 16); t2 = __popc(hi[0] & 0xFFFFu); t3 = __popc(hi[0] >> 16);
                } else {
                    t0 = __popc((lo[0] & 0xFFFFu) | (lo[1] << 16)); t1 = __popc((lo[0] >> 16) | (lo[1] & 0xFFFF0000u));
                    t2 = __popc((hi[0] & 0xFFFFu) | (hi[1] << 16)); t3 = __popc((hi[0] >> 16) | (hi[1] & 0xFFFF0000u));
                }
                const unsigned packed = (unsigned)t0 | (unsigned)t1 << 8 | (unsigned)t2 << 16 | (unsigned)t3 << 24;
                unsigned m01, m23, mx;
                asm("s_max_u32 %0, %1, %2" : "=s"(m01) : "s"(t0), "s"(t1) : "scc");
                asm("s_max_u32 %0, %1, %2" : "=s"(m23) : "s"(t2), "s"(t3) : "scc");
                asm("s_max_u32 %0, %1, %2" : "=s"(mx) : "s"(m01), "s"(m23) : "scc");
                const int maxc = (int)mx;
                const int cnt = (int)__builtin_amdgcn_ubfe(packed, (unsigned)g8, 8u);
                // the next window is on its way while this one is summed.  (The empty statement pins the order: the window's
                // last readers above, the loads below — so that a load may land in the register it replaces instead of a
                // fresh one that the compiler then has to copy, i.e. wait for, at the end of the round.  Its s_nop: a DPP
                // read of a VGPR needs two wait states after the VALU write.)
                if constexpr (WIN == 1) asm volatile("s_nop 1" : "+v"(off[0]), "+v"(aa[0]) : : "memory");
                else asm volatile("s_nop 1" : "+v"(off[0]), "+v"(aa[0]), "+v"(off[WIN - 1]), "+v"(aa[WIN - 1]) : : "memory");
                posl[i] += cnt << SH(i);
                rem[i] -= cnt;
                load_window(i);
                if (maxc > 0) {                      // (wave-uniform)
                    tl_window<real_t, CPL>(smem, maxc, off[0], aa[0], lane16, acc[i]);
                    if constexpr (WIN > 1) {
                        if (maxc > TL_G) tl_window<real_t, CPL>(smem, maxc - TL_G, off[1], aa[1], lane16, acc[i]);
                    }
                }
                return maxc;
            };
            // One round over the wavefront's rows: every row's window is loaded exactly once per round, in the same order, so that a
            // row's window is awaited RG - 1 visits after it was asked for (counted vmcnt).  A group that FILLED its window may have
            // more entries in this tile: those rows — and only those — are visited again until none is left (round 5 went round
            // ALL the wavefront's rows again: a row of 10,000 entries in K = 10,000 has 256 per tile, eight windows, and made its
            // wavefront pay seven extra visits of every other row it holds, per tile — tools/tile_stamps.py, `giant`: the
            // workgroups with such a row took 2.2x the others).  The extra visits are a separate, rarely taken path: the
            // compiler's waits there are conservative, which is fine.
            // (spelled out: a `#pragma unroll` over bodies this large is declined, and a rolled loop indexes the sums in scratch;
            // a wavefront skips the visits of the rows it does not have — my_rg is wave-uniform)
            constexpr int FULL = WIN * TL_G;
            unsigned more = visit(0) == FULL ? 1u : 0u;
            if constexpr (RG > 1) { if (my_rg > 1 && visit(1) == FULL) more |= 2u; }
            if constexpr (RG > 2) { if (my_rg > 2 && visit(2) == FULL) more |= 4u; }
            if constexpr (RG > 3) { if (my_rg > 3 && visit(3) == FULL) more |= 8u; }
            if constexpr (RG > 4) { if (my_rg > 4 && visit(4) == FULL) more |= 16u; }
            while (more) {                           // (wave-uniform)
                if (more & 1u) { if (visit(0) != FULL) more &= ~1u; }
                if constexpr (RG > 1) { if (more & 2u) { if (visit(1) != FULL) more &= ~2u; } }
                if constexpr (RG > 2) { if (more & 4u) { if (visit(2) != FULL) more &= ~4u; } }
                if constexpr (RG > 3) { if (more & 8u) { if (visit(3) != FULL) more &= ~8u; } }
                if constexpr (RG > 4) { if (more & 16u) { if (visit(4) != FULL) more &= ~16u; } }
            }
        }
        if (stamps && lane == 0) {                   // diagnostic build only (mxd_debug_spmm_tile_stamps): cycles at the barriers / in all
            stamps[((size_t)blockIdx.x * 16 + wave) * 2] = st_wait;
            stamps[((size_t)blockIdx.x * 16 + wave) * 2 + 1] = __builtin_amdgcn_s_memtime() - st_t0;
        }
        if constexpr (!COLMAJOR) {
#pragma unroll
            for (int i = 0; i < RG; i++) {
                if (ROW(i) >= m) continue;
                const int xr = XR(i);
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int col = c0 + c * (TL_G * VEC) + lg * VEC;
                    if (col >= n) continue;
                    real_t *dst = xr < 0 ? C + (size_t)ROW(i) * ldc + col : Cx + (size_t)xr * n + col;     // (Cx: row-major, ld = n)
                    if (c_vec || xr >= 0) vstore<real_t, VEC>(dst, acc[i][c]);
                    else {
#pragma unroll
                        for (int v = 0; v < VEC; v++) dst[v] = acc[i][c][v];
                    }
                }
            }
            return;
        }
    }
    if constexpr (COLMAJOR) {
        // the block's R x W sums through LDS (the tiles are done with): column segments of R consecutive rows
        real_t *tr = reinterpret_cast<real_t *>(smem);
        const int RP = R | 1;
        __syncthreads();
        if (wave < nw) {
#pragma unroll
            for (int i = 0; i < RG; i++) {
                if (i >= my_rg || ROW(i) >= m) continue;
                const int xr = XR(i);
                if (xr >= 0) {                                       // a part of a long row: straight into its row of Cx
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        const int col = c0 + c * (TL_G * VEC) + lg * VEC;
                        if (col < n) vstore<real_t, VEC>(Cx + (size_t)xr * n + col, acc[i][c]);
                    }
                    continue;
                }
                const int r = ROW(i) - row0;                         // (the block's rows, whatever slots they were dealt to)
#pragma unroll
                for (int c = 0; c < CPL; c++)
#pragma unroll
                    for (int v = 0; v < VEC; v++) tr[(c * (TL_G * VEC) + lg * VEC + v) * RP + r] = acc[i][c][v];
            }
        }
        __syncthreads();
        const int ncols = min(W, n - c0), nrows = nrows_blk;
        for (int idx = threadIdx.x; idx < ncols * R; idx += blockDim.x) {
            const int c = idx / R, r = idx % R;
            if (r < nrows) C[(size_t)(c0 + c) * ldc + row0 + r] = tr[c * RP + r];
        }
    }
}

// per-row flags for the tile kernel: 1 = the row is NOT sorted by column (src/misc.cpp:118-128's test); one wavefront per row
__global__ __launch_bounds__(512)
void tile_unsorted_rows_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                               unsigned char *__restrict__ flags)
{
    const int lane = lane_id();
    const int row = blockIdx.x * 8 + uniform(threadIdx.x / MX_WAVE);
    if (row >= m) return;
    const int s = uniform(indptr[row]), e = uniform(indptr[row + 1]);
    bool bad = false;
    for (int k = s + lane; k + 1 < e; k += MX_WAVE) bad |= indices[k] > indices[k + 1];
    const bool any = __ballot(bad) != 0;
    if (lane == 0) flags[row] = any ? 1 : 0;
}

// Row blocks for rows of uneven length: every R rows is a cut, and a block of R rows that holds more than E entries is cut
// further into sub-blocks of equal entry counts — a block of the 10 % longest rows of a sorted matrix is otherwise the tail of
// the launch (1e4 x 1e4, 500 per row, rows sorted by length: 0.38 ms where equal rows take 0.12).  One workgroup (block counts,
// their prefix sums in rounds of 1,024, the boundaries by binary search); cuts[0 .. max_blocks]: everything past the last real
// block is m (empty blocks: the launch is sized by the bound max_blocks, nothing is read back — the product stays capturable).
__global__ __launch_bounds__(1024)
void tile_cuts_kernel(int m, int R, const int32_t *__restrict__ indptr, long long E, long long E_piece, int32_t *__restrict__ cuts, int max_blocks)
{
    __shared__ int scan[1024];
    __shared__ int carry;
    const int nA = (m + R - 1) / R, tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < nA; k0 += 1024) {
        const int k = k0 + tid;
        int s = 0, r0 = 0, r1 = 0;
        long long e0 = 0, nn = 0;
        if (k < nA) {
            r0 = k * R; r1 = min(r0 + R, m);
            e0 = indptr[r0]; nn = (long long)indptr[r1] - e0;
            s = nn > E ? (int)min((long long)(r1 - r0), max(1LL, (nn + E_piece - 1) / E_piece)) : 1;
        }
        scan[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int base = carry + scan[tid] - s;
        for (int q = 0; q < s; q++) {
            int at = r0;
            if (q > 0) {
                const long long target = e0 + nn * q / s;
                int lo = r0, hi = r1;
                while (lo < hi) { const int mid = lo + (hi - lo) / 2; if ((long long)indptr[mid] < target) lo = mid + 1; else hi = mid; }
                at = lo;
            }
            if (base + q < max_blocks) cuts[base + q] = at;
        }
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
    for (int i = min(carry, max_blocks) + tid; i <= max_blocks; i += 1024) cuts[i] = m;
}

// Rows of UNEVEN length (round 6).  The four lane groups of a wavefront walk one row each in lockstep: a visit
// costs the LONGEST of the four, and the wavefronts of a SIMD add up.  With consecutive rows in consecutive slots a log-normal
// matrix (sigma 1.5) paid E[max of 4] / mean ~ 2x in every visit and its busiest SIMD half as much again (1e4 x 1e4, 500 per row,
// n = 100: 0.31 ms where equal rows take 0.108).  Here the rows of a block are RANKED by length (R <= 300: every thread counts
// the rows ahead of its own) and dealt: ranks 4q .. 4q + 3 share visit q — four rows of nearly the same length —, and the
// visits, longest first, go to the SIMD with the least work so far (wavefront w sits on SIMD w % 4).
// A row is still summed by one group in storage order: the same bits.  perm[rb * R + slot] = row, m = no row.
// split: nullptr = no row is cut.  Otherwise a row longer than 1.5 * part_len entries (and sorted by column) is cut into
// P = 2 / 4 / 8 interleaved parts (round 6, tools/tile_split_emulation.py: the storage-order chain of a 10,000-entry row is 0.16 ms
// whatever else happens — cut into parts of <= 1,280 entries the product takes 0.13 ms in f64 and 0.09 in f32 where whole rows
// take 0.18): each part is a slot of its own — the block has spare slots for that, Rslots > the rows of a base block —, part 0
// sums into C, the others into rows of the scratch matrix Cx (capacity xcap), handed out in block order (pass 0 / 1, blk[]:
// below); the parent rows are listed (parents[]: row, first row of Cx, P) for tile_combine_kernel, which adds the parts in
// order: the same bits on every run, but no longer the storage-order chain for THOSE rows.
struct TileParent { int row, x0, parts; };
// log2 of the number of parts a row of L entries is cut into when the block has the slots (part_len = 0: never)
__device__ __forceinline__ int tile_parts_lg2(int L, int part_len)
{
    int lg2 = 0;
    if (part_len > 0 && 2LL * L > 3LL * part_len)
        while (lg2 < 3 && (L >> lg2) > part_len) lg2++;
    return lg2;
}

// Row blocks in ONE pass over the rows, for matrices of <= 32,768 rows (round 6): a block takes rows while its UNITS — rows, and the
// parts the long ones will be cut into — fit the workgroup's slots and its WEIGHT stays below E: a row weighs its entries + w_row,
// what its slot costs whatever it holds (8 entries' worth per tile, measured: tools/tile_split_probe.py with MXGPU_TILE_WROW =
// 0 / 8 / 16 / 40 — rows sorted by length, 1e4 x 1e4: 0.214 / 0.182 / 0.199 / 0.237 ms; 4,000 x 50,000: 0.59 / 0.43 / 0.50 / 0.48).  tile_cuts_kernel cuts every R
// rows whatever they hold: a matrix with its rows sorted by length then needs ceil(m / R) blocks for its rows and again as many
// pieces for its entries — 1e4 x 1e4, 500 per row: 155 blocks x 2 slabs, a second round of workgroups for the last 54 —, and the
// spare slots of a block of short rows stay empty.  Here: every row's farthest block end (two monotone conditions: a binary
// search in the prefix sums of the units, U[], and in indptr), kept as a 16-bit distance in LDS, then one thread walks from row
// 0 (<= max_blocks hops through LDS).  When the launch could be ONE round of workgroups (target_blocks > 0) and E1 gives more
// blocks than that, E2 is tried, and E1 again if that is still too many.  max_blocks bounds what the walk can produce (two
// consecutive blocks always overflow one of the two limits); a walk that would pass it traps: loud, never a lost row.
__global__ __launch_bounds__(1024)
void tile_cuts_greedy_kernel(int m, int Rslots, const int32_t *__restrict__ indptr, const unsigned char *__restrict__ unsorted, int part_len,
                             int w_row, long long E1, long long E2, int target_blocks, int32_t *__restrict__ U, int32_t *__restrict__ cuts, int max_blocks)
{
    __shared__ unsigned short delta[32768];
    __shared__ int scan[1024];
    __shared__ int carry, nblocks;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < m; k0 += 1024) {
        const int r = k0 + tid;
        int u = 0;
        if (r < m) u = 1 << ((unsorted && unsorted[r]) ? 0 : tile_parts_lg2(indptr[r + 1] - indptr[r], part_len));
        scan[tid] = u;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        if (r < m) U[r] = carry + scan[tid] - u;
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
    if (tid == 0) U[m] = carry;
    __syncthreads();
    for (int attempt = 0; attempt < 3; attempt++) {
        const long long E = attempt == 1 ? E2 : E1;
        for (int r = tid; r < m; r += 1024) {
            const int u0 = U[r];
            const long long e0 = indptr[r];
            int lo = r + 1, hi = min(m, r + Rslots);                 // the block [r, j): the largest j that keeps both limits; one row at least
            while (lo < hi) {
                const int mid = lo + (hi - lo + 1) / 2;
                if (U[mid] - u0 <= Rslots && (long long)indptr[mid] - e0 + (long long)(mid - r) * w_row <= E) lo = mid; else hi = mid - 1;
            }
            delta[r] = (unsigned short)(lo - r);
        }
        __syncthreads();
        if (tid == 0) {
            int i = 0, b = 0;
            while (i < m) {
                if (b >= max_blocks) __builtin_trap();
                cuts[b++] = i;
                i += delta[i];
            }
            nblocks = b;
        }
        __syncthreads();
        if (target_blocks <= 0 || nblocks <= target_blocks || attempt == 2) break;
        __syncthreads();
    }
    for (int i = nblocks + tid; i <= max_blocks; i += 1024) cuts[i] = m;
}
__global__ __launch_bounds__(256)
void tile_deal_rows_kernel(int m, int R, int nw, unsigned long long rgw, const int32_t *__restrict__ indptr, const int32_t *__restrict__ cuts,
                           int32_t *__restrict__ perm, int32_t *__restrict__ split, int part_len, const unsigned char *__restrict__ unsorted,
                           int pass, int32_t *__restrict__ blk, int nblk, unsigned xcap, TileParent *__restrict__ parents, unsigned pcap)
{
    constexpr int MAXR = TL_MAX_WAVES * TL_NG * 5;
    __shared__ int len[MAXR + 4], sorted_len[MAXR + 4];
    __shared__ short row_of_rank[MAXR + 4], visit_of[TL_MAX_WAVES * 5 + 1];
    // units: what a slot holds — a whole row or one part of a cut row — in dealing order
    __shared__ short unit_rank[MAXR + 4];
    __shared__ int unit_code[MAXR + 4], unit_len[MAXR + 4];
    __shared__ int n_units;
    const int rb = blockIdx.x, row0 = cuts[rb], nrows = cuts[rb + 1] - row0;
    if (nrows <= 0) return;
    for (int r = threadIdx.x; r < R; r += blockDim.x) len[r] = r < nrows ? indptr[row0 + r + 1] - indptr[row0 + r] : -1;
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int mine = len[r];
        int rank = 0;
        for (int o = 0; o < R; o++) rank += (len[o] > mine || (len[o] == mine && o < r)) ? 1 : 0;
        sorted_len[rank] = mine;
        row_of_rank[rank] = (short)r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // which rows are cut: longest first, while the block has spare slots (a block never holds more rows than slots).
        // The parts' rows of Cx and the places in the parents' list are handed out in BLOCK order: pass 0 only counts what this
        // block wants (blk[rb] parts, blk[nblk + rb] parents), tile_part_scan_kernel turns the counts into offsets, pass 1 cuts — a
        // row whose range does not fit the scratch stays whole (its range and its spare slots stay spent, so that every later
        // decision is the one pass 0 counted): which rows those are does not depend on the order the blocks ran in.
        int spare = R - nrows, nu = 0, my_parts = 0, my_parents = 0;
        const unsigned x_base = split && pass ? (unsigned)blk[rb] : 0u, p_base = split && pass ? (unsigned)blk[nblk + rb] : 0u;
        for (int rank = 0; rank < nrows; rank++) {
            const int L = sorted_len[rank], row = row0 + row_of_rank[rank];
            int lg2 = 0;
            if (split && spare > 0 && !(unsorted && unsorted[row])) {
                lg2 = tile_parts_lg2(L, part_len);
                while (lg2 > 0 && (1 << lg2) - 1 > spare) lg2--;
            }
            int x0 = 0;
            if (lg2 > 0) {
                const unsigned at = x_base + (unsigned)my_parts, pa = p_base + (unsigned)my_parents;
                my_parts += (1 << lg2) - 1; my_parents++;
                spare -= (1 << lg2) - 1;
                if (!pass) lg2 = 0;
                else if (at + (1u << lg2) - 1u <= xcap && pa < pcap) {
                    x0 = (int)at;
                    parents[pa].row = row; parents[pa].x0 = x0; parents[pa].parts = 1 << lg2;
                } else {
                    if (pa < pcap) parents[pa].parts = 0;             // (tile_combine_kernel skips it)
                    lg2 = 0;
                }
            }
            if (!pass) continue;
            const int P = 1 << lg2;
            for (int q = 0; q < P; q++) {
                unit_rank[nu] = (short)rank;
                unit_code[nu] = q | (lg2 << 8) | ((q > 0 ? x0 + q : 0) << 12);       // (row of Cx + 1: part q > 0 sums into row x0 + q - 1)
                unit_len[nu] = (L - q + P - 1) >> lg2;
                nu++;
            }
        }
        if (!pass) { blk[rb] = my_parts; blk[nblk + rb] = my_parents; }
        n_units = nu;
        // visits (units 4v .. 4v + 3, cost = the longest of them), longest first, each to the SIMD with the least work so far that
        // still has a free visit slot, there to the wavefront with the most free slots
        // (per SIMD, not per wavefront: dealing to the least-loaded WAVEFRONT was measured too — log-normal sigma 1 / 1.5, 1e4 x 1e4:
        // 0.263 / 0.397 ms against 0.218 / 0.339; the wavefronts of a SIMD share its issue slots, so what counts is their sum)
        int base[TL_MAX_WAVES + 1], free_[TL_MAX_WAVES];
        long long load[4] = {0, 0, 0, 0};
        base[0] = 0;
        for (int w = 0; w < nw; w++) { free_[w] = (int)((rgw >> (4 * w)) & 15); base[w + 1] = base[w] + free_[w]; }
        const int nq = R / TL_NG;
        for (int v = 0; v < nq; v++) {
            int best_c = -1;
            for (int c = 0; c < 4; c++) {
                bool has = false;
                for (int w = c; w < nw; w += 4) has = has || free_[w] > 0;
                if (has && (best_c < 0 || load[c] < load[best_c])) best_c = c;
            }
            int best_w = -1;
            for (int w = best_c; w < nw; w += 4)
                if (free_[w] > 0 && (best_w < 0 || free_[w] > free_[best_w])) best_w = w;
            const int rg_w = (int)((rgw >> (4 * best_w)) & 15);
            visit_of[v] = (short)(base[best_w] + (rg_w - free_[best_w]));
            free_[best_w]--;
            // (a visit costs its window bookkeeping whatever its rows hold: ~40 entries' worth; a visit without rows nothing)
            int longest = 0;
            for (int u = v * TL_NG; u < min(v * TL_NG + TL_NG, nu); u++) longest = max(longest, unit_len[u]);
            load[best_c] += v * TL_NG < nu ? 40 + longest : 0;
        }
    }
    __syncthreads();
    if (!pass) return;
    const int nu = n_units;
    for (int u = threadIdx.x; u < R; u += blockDim.x) {
        const size_t slot = (size_t)rb * R + visit_of[u / TL_NG] * TL_NG + u % TL_NG;
        perm[slot] = u < nu ? row0 + row_of_rank[unit_rank[u]] : m;
        if (split) split[slot] = u < nu ? unit_code[u] : 0;
    }
}

// the blocks' part / parent counts (tile_deal_rows_kernel, pass 0) -> their offsets; the totals into xstate[0 .. 1]
__global__ __launch_bounds__(1024)
void tile_part_scan_kernel(int nblk, int32_t *__restrict__ blk, unsigned *__restrict__ xstate)
{
    __shared__ int scan[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    for (int which = 0; which < 2; which++) {
        int32_t *a = blk + (size_t)which * nblk;
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int k0 = 0; k0 < nblk; k0 += 1024) {
            const int k = k0 + tid, v = k < nblk ? a[k] : 0;
            scan[tid] = v;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {
                const int t = tid >= o ? scan[tid - o] : 0;
                __syncthreads();
                scan[tid] += t;
                __syncthreads();
            }
            if (k < nblk) a[k] = carry + scan[tid] - v;
            __syncthreads();
            if (tid == 1023) carry += scan[1023];
            __syncthreads();
        }
        if (tid == 0) xstate[which] = (unsigned)carry;
        __syncthreads();
    }
}

// adds the parts of the cut rows: C[row, :] += Cx[x0, :] + Cx[x0 + 1, :] + ... in order (one wavefront per cut row)
template <typename real_t>
__global__ __launch_bounds__(256)
void tile_combine_kernel(int n, const unsigned *__restrict__ xstate, unsigned pcap, const TileParent *__restrict__ parents,
                         const real_t *__restrict__ Cx, real_t *__restrict__ C, size_t ldc, int colmajor)
{
    const unsigned np = min(xstate[1], pcap);
    const int lane = lane_id();
    for (unsigned i = blockIdx.x * 4 + uniform(threadIdx.x / MX_WAVE); i < np; i += gridDim.x * 4) {
        const int row = uniform(parents[i].row), x0 = uniform(parents[i].x0), P = uniform(parents[i].parts);
        if (P < 2) continue;
        for (int c = lane; c < n; c += MX_WAVE) {
            real_t *dst = colmajor ? C + (size_t)c * ldc + row : C + (size_t)row * ldc + c;
            real_t sum = *dst;
            for (int q = 1; q < P; q++) sum += Cx[(size_t)(x0 + q - 1) * n + c];
            *dst = sum;
        }
    }
}

static thread_local float g_tile_deal_hint = -1.0f;    // TileDealScope: >= 0 = the cv to deal by, no parts
static std::atomic<int> g_tile_last_mode{0};                     // diagnostic (mxd_debug_spmm_tile_mode; the last product of ANY thread): 1 dealt | 2 parts | 4 one-pass blocks
static unsigned long long *g_tile_stamps = nullptr;   // diagnostic (tools/tile_stamps.py): per wavefront, cycles at the tile barriers / in all

// ---------------------------------------------------------------------------------------------------------------
// Geometry: rows per workgroup so that ONE round of workgroups fills the 256 CUs when the product is small (every
// workgroup keeps two 64 KB tiles: one per CU), the largest block otherwise (the tile fills shrink with 1 / R).
struct TileGeom {
    int cpl, rg, nw, nl, nslabs, nrb;
    int R, serial;                    // rows per workgroup; visits per tile on the busiest SIMD
    unsigned long long rgw;           // rows per lane group of every compute wavefront, 4 bits each
};

// Deal `gr` group-rows (4 rows of A each) to nw compute wavefronts so that every SIMD makes the same number of visits per tile:
// a workgroup's wavefronts go to the SIMDs round robin, wavefront w to position w % 4 of the cycle, the nl loaders behind the
// compute wavefronts — so a SIMD holds the compute wavefronts of one residue class, and a class that also holds a loader has
// one compute wavefront fewer when nw + nl = 16.  (The first version gave every wavefront the same 3 rows: the two SIMDs with
// four compute wavefronts made 12 visits per tile, the two with three + a loader 9 and waited — 28 % of all wavefront cycles
// were spent at the tile barrier, tools/tile_stamps.py.)  Returns false when a wavefront would need more than rgmax rows.
static bool tile_deal(int gr, int nw, int nl, int rgmax, TileGeom &gm)
{
    int nc[4] = {0, 0, 0, 0}, loaders[4] = {0, 0, 0, 0};
    for (int w = 0; w < nw; w++) nc[w & 3]++;
    for (int k = 0; k < nl; k++) loaders[(nw + k) & 3]++;
    // quotas: gr / 4 each, the remainder to the classes without a loader first
    int quota[4], order[4] = {0, 1, 2, 3};
    std::sort(order, order + 4, [&](int a, int b) { return loaders[a] != loaders[b] ? loaders[a] < loaders[b] : a < b; });
    for (int c = 0; c < 4; c++) quota[c] = gr / 4;
    for (int k = 0; k < gr % 4; k++) quota[order[k]]++;
    gm.rgw = 0; gm.serial = 0; gm.rg = 0;
    int given = 0;
    for (int c = 0; c < 4; c++) {
        if (nc[c] == 0) { if (quota[c]) return false; continue; }
        const int base = quota[c] / nc[c], extra = quota[c] % nc[c];
        int idx = 0;
        for (int w = c; w < nw; w += 4, idx++) {
            const int r = base + (idx < extra ? 1 : 0);
            if (r > rgmax) return false;
            gm.rgw |= (unsigned long long)r << (4 * w);
            gm.rg = std::max(gm.rg, r);
            given += r;
        }
        gm.serial = std::max(gm.serial, quota[c]);
    }
    if (gm.rg < 1) gm.rg = 1;
    gm.nw = nw; gm.nl = nl; gm.R = given * TL_NG;
    return given == gr;
}

static TileGeom tile_geometry(int m, int n, int dense_bytes, int cpl, int rg, int nw, int colmajor, int tile_bytes)
{
    const int vec = 16 / dense_bytes, w1 = TL_G * vec;
    TileGeom gm;
    if (cpl != 2) cpl = 1;
    gm.cpl = cpl;
    gm.nslabs = (n + w1 * cpl - 1) / (w1 * cpl);
    const int rgmax = cpl == 1 ? 5 : 4;
    if (rg >= 1 && rg <= rgmax && nw >= 1 && nw <= TL_MAX_WAVES) {
        // a geometry asked for (tools/tile_sweep.py, the tests): the same rg rows for every lane group
        gm.rgw = 0;
        for (int w = 0; w < nw; w++) gm.rgw |= (unsigned long long)rg << (4 * w);
        gm.rg = rg; gm.nw = nw; gm.nl = nw <= TL_MAX_WAVES - 1 ? 2 : 1;
        gm.R = nw * TL_NG * rg;
        gm.serial = (int)ceil_div(nw + gm.nl, 4) * rg;
    } else {
        const int per_slab = 256 / gm.nslabs > 0 ? 256 / gm.nslabs : 1;
        int want = (int)ceil_div(m, per_slab);                            // rows per workgroup for one full round
        // more rows than one round holds: workgroups follow one another on a CU, and 144 rows (3 rows x 12 wavefronts + 2
        // loaders) run faster than the largest block (m = 1e5, K = 1e4, 500 per row, n = 100: 1.14 ms against 1.29 with 240 rows
        // — tools/tile_sweep.py; the windows of A then stream from HBM and more, smaller workgroups hide that better)
        if (want > 14 * TL_NG * 4) want = 12 * TL_NG * 3;
        // column-major C leaves through the tiles' LDS: (R | 1) x W sums must fit
        const int fit = (2 * tile_bytes + 256 * cpl) / (256 * cpl) - 1;
        if (colmajor && want > fit) want = fit / TL_NG * TL_NG;
        if (want < 4 * TL_NG) want = 4 * TL_NG;
        const int gr = (int)ceil_div(want, TL_NG);
        // the fewest visits per tile on the busiest SIMD, then the most wavefronts (latency hiding)
        bool found = false;
        TileGeom best = gm;
        for (int w = 14; w >= 4; w--) {
            TileGeom t = gm;
            if (!tile_deal(gr, w, 2, rgmax, t)) continue;
            if (!found || t.serial < best.serial) { best = t; found = true; }
        }
        if (!found) {                                                 // (cannot happen for want <= 14 x 4 x 4; belt and braces)
            tile_deal(12 * 3, 12, 2, rgmax, best);
        }
        gm = best;
    }
    gm.nrb = (int)ceil_div(m, gm.R);
    return gm;
}

// What the kernel costs, microseconds (the model AUTO compares with the row-split kernel's and the planned sweep's,
// csrc/spmm.hip; fitted to tools/tile_map.py's map, profiles/r05_tile_map.json: within 25 % of the measured time at 120 of 154
// points and every point above 0.1 ms).  A workgroup walks T K-tiles; per tile its busiest SIMD makes `serial` visits of
// (a row of every lane group, the tile), one after the other: ~180 cycles of window bookkeeping per 32 entries and 4.5
// cycles per issue slot of a step (address add, CPL reads, 2 CPL FMAs), steps = the largest of four Poisson(mu) counts
// rounded up to a batch, mu = entries per row and tile; never below what the fill and the window loads take (~1.3 us + 0.1
// us per visit in a row: 2,000 rows of the vignette's matrix take 0.064 ms = 1.6 us per tile, 10,000 rows 2.7).  Workgroups beyond one per CU come in rounds (x 1.05: their windows stream A from HBM, not from
// the Infinity Cache).
static double tile_est_us_cpl(int m, int n, int K, int dense_bytes, double avg_len, int cpl, int colmajor)
{
    const TileGeom gm = tile_geometry(m, n, dense_bytes, cpl, 0, 0, colmajor, 65536);
    const double wgs = (double)gm.nrb * gm.nslabs;
    const double rounds = wgs <= 256.0 ? 1.0 : (wgs <= 768.0 ? std::ceil(wgs / 256.0) : wgs / 256.0);
    const int TK = 65536 / (256 * cpl);
    const double T = (double)ceil_div(K, TK);
    const double mu = avg_len * (double)(TK < K ? TK : K) / (double)(K > 0 ? K : 1);
    // (entries per row and tile: Poisson around the ROW's own mean — with skewed row lengths a mixture: variance mu + (cv mu)^2)
    const double cv = profile_cv(), sd = std::sqrt(mu + cv * cv * mu * mu);
    const double steps = mu + 1.03 * sd + (4 / cpl) / 2.0;
    const double passes = std::max(1.0, std::ceil((mu + 2.0 * sd) / 32.0));
    const double visit = 180.0 * passes + 4.5 * (1 + 3 * cpl) * steps;
    const double serial = (double)gm.serial;
    const double tile_us = std::max(1.3 + 0.1 * serial, serial * visit / 2400.0);
    return 6.0 + rounds * (T * tile_us + 4.0) * (wgs > 256.0 ? 1.05 : 1.0);
}
// the cheaper of 256- and 512-byte slabs (512: half the workgroups — it wins where 256-byte slabs need a second round — but
// half the rows of B per tile and seven issue slots per step instead of four)
double tile_est_us(int m, int n, int K, int dense_bytes, double avg_len, int colmajor, int *cpl)
{
    const double c1 = tile_est_us_cpl(m, n, K, dense_bytes, avg_len, 1, colmajor);
    const double c2 = n > TL_G * (16 / dense_bytes) ? tile_est_us_cpl(m, n, K, dense_bytes, avg_len, 2, colmajor) : 2.0 * c1;
    if (cpl) *cpl = c2 < 0.9 * c1 ? 2 : 1;
    return c2 < 0.9 * c1 ? c2 : c1;
}

template <typename real_t, int CPL, int RG, int WIN, int TILE>
static void launch_tile(const TileGeom &gm, int m, int n, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                        const unsigned char *unsorted, const int32_t *perm, const int32_t *cuts, const int32_t *split, real_t *Cx, const real_t *B, size_t ldb, real_t *C, size_t ldc,
                        int colmajor, int c_vec, hipStream_t st)
{
    const int per = 8 % gm.nslabs == 0 ? 8 / gm.nslabs : 0;
    const unsigned grid = per ? (unsigned)(8 * ceil_div(gm.nrb, per)) : (unsigned)(gm.nrb * gm.nslabs);
    const dim3 block((unsigned)(gm.nw + gm.nl) * MX_WAVE);
    constexpr bool CAN_SPLIT = true;
    if constexpr (CAN_SPLIT) {
        if (split) {
            if (colmajor)
                hipLaunchKernelGGL((spmm_tile_kernel<real_t, CPL, RG, WIN, TILE, true, true>), dim3(grid), block, 0, st, m, n, K, gm.nslabs,
                                   gm.nrb, indptr, indices, values, unsorted, perm, cuts, split, Cx, B, ldb, C, ldc, c_vec, gm.nl, gm.rgw, g_tile_stamps);
            else
                hipLaunchKernelGGL((spmm_tile_kernel<real_t, CPL, RG, WIN, TILE, false, true>), dim3(grid), block, 0, st, m, n, K, gm.nslabs,
                                   gm.nrb, indptr, indices, values, unsorted, perm, cuts, split, Cx, B, ldb, C, ldc, c_vec, gm.nl, gm.rgw, g_tile_stamps);
            return;
        }
    }
    if (colmajor)
        hipLaunchKernelGGL((spmm_tile_kernel<real_t, CPL, RG, WIN, TILE, true, false>), dim3(grid), block, 0, st, m, n, K, gm.nslabs,
                           gm.nrb, indptr, indices, values, unsorted, perm, cuts, nullptr, nullptr, B, ldb, C, ldc, c_vec, gm.nl, gm.rgw, g_tile_stamps);
    else
        hipLaunchKernelGGL((spmm_tile_kernel<real_t, CPL, RG, WIN, TILE, false, false>), dim3(grid), block, 0, st, m, n, K, gm.nslabs,
                           gm.nrb, indptr, indices, values, unsorted, perm, cuts, nullptr, nullptr, B, ldb, C, ldc, c_vec, gm.nl, gm.rgw, g_tile_stamps);
}

template <typename real_t, int CPL, int WIN, int TILE>
static void launch_tile_rg(const TileGeom &gm, int m, int n, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                           const unsigned char *unsorted, const int32_t *perm, const int32_t *cuts, const int32_t *split, real_t *Cx, const real_t *B, size_t ldb, real_t *C, size_t ldc,
                           int colmajor, int c_vec, hipStream_t st)
{
#define MX_TL_RG(RG) launch_tile<real_t, CPL, RG, WIN, TILE>(gm, m, n, K, indptr, indices, values, unsorted, perm, cuts, split, Cx, B, ldb, C, ldc, colmajor, c_vec, st)
    switch (gm.rg) {
    case 1: MX_TL_RG(1); break;
    case 2: MX_TL_RG(2); break;
    case 3: MX_TL_RG(3); break;
    case 4: MX_TL_RG(4); break;
    default:
        if constexpr (CPL == 1) MX_TL_RG(5);         // (five rows per group still fit 128 registers with 256-byte slabs)
        else MX_TL_RG(4);
        break;
    }
#undef MX_TL_RG
}

// can the tile kernel take these operands?  16-byte aligned rows of B (the LDS-DMA moves 16 bytes per lane), at least one
// whole vector per row
template <typename real_t>
bool tile_ok(int n, const real_t *B, size_t ldb)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    return n >= VEC && n % VEC == 0 && ldb % VEC == 0 && (uintptr_t)B % 16 == 0;
}
template bool tile_ok<double>(int, const double *, size_t);
template bool tile_ok<float>(int, const float *, size_t);

// variant: 0 = chosen here; otherwise cpl + 4 * rg + 32 * (32 KB tiles, 16-entry windows)
// (tools/tile_sweep.py); nw: compute wavefronts per workgroup (0 = chosen here).  rows_sorted != 0: the caller vouches for column-sorted rows (no flag pass).
template <typename real_t>
int tile_spmm(int m, int n, int K, int64_t nnz, int variant, int nw, int rows_sorted, const int32_t *indptr, const int32_t *indices,
              const double *values, const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    if (!tile_ok<real_t>(n, B, ldb)) return set_error("tile_spmm: rows of B must be 16-byte aligned whole vectors (n = %d, ldb = %zu)", n, ldb);
    // (a row block's windows are addressed by 32-bit byte offsets from the block's first entry)
    if (nnz > (1LL << 29)) return set_error("tile_spmm: more than 2^29 entries (%lld)", (long long)nnz);
    int cpl = variant & 3;
    // 256- or 512-byte slabs: the model's choice when the caller knows how many entries there are
    if (cpl != 1 && cpl != 2) { cpl = 1; if (nnz >= 0 && m > 0) tile_est_us(m, n, K, (int)sizeof(real_t), (double)nnz / m, colmajor, &cpl); }
    const int rg = (variant >> 2) & 7, small_tile = (variant >> 5) & 1, one_loader = (variant >> 6) & 1;
    TileGeom gm = tile_geometry(m, n, (int)sizeof(real_t), cpl, rg, nw, colmajor, small_tile ? 32768 : 65536);
    if (one_loader) gm.nl = 1;
    const int c_vec = colmajor || ((ldc % VEC == 0) && ((uintptr_t)C % 16 == 0));
    {
        const int R = gm.R, rowb = 256 * gm.cpl, lds = 2 * (small_tile ? 32768 : 65536) + rowb;
        if (colmajor && (R | 1) * rowb > lds)
            return set_error("tile_spmm: %d rows per workgroup do not fit the column-major epilogue's LDS (%d bytes)", R, lds);
    }
    unsigned char *flags = nullptr;
    if (!rows_sorted) {
        flags = (unsigned char *)scratch_buffer(MX_SCRATCH_TILE_FLAGS, (size_t)m);
        scratch_acquire(MX_SCRATCH_TILE_FLAGS, stream);
        if (!flags) return set_error("tile_spmm: cannot allocate %d bytes of row flags", m);
    }
    // rows of uneven length (the matrix profile in scope says so; MXGPU_TILE_DEAL=0 / 1 forces): row blocks of about equal entry
    // counts (tile_cuts_kernel) whose rows are dealt to the lane groups by length (tile_deal_rows_kernel) — two small launches,
    // worth it from cv ~ 0.15 on (Poisson row lengths of a uniform-density matrix: 0.045 at the vignette's shape, which stays on
    // consecutive rows).  Kept per thread and device for the next product with the same row pointers and geometry: whatever the
    // matrix holds by then, the cuts and the map stay a partition of its rows into blocks of <= R — stale ones can cost
    // balance, never a row.
    // LONG rows (round 6; with the map, when the profile's longest row is at least 2.5 parts long; MXGPU_TILE_SPLIT=0 / 1 forces):
    // a row of more than 1.5 parts is cut into 2 / 4 / 8 interleaved parts of ~part_len entries (1.25 windows per tile) that
    // take slots of their own — the geometry is given a third more slots than a base block has rows — and are added up by
    // tile_combine_kernel.  Those rows' sums are regrouped (1e-12), all others stay the storage-order chain bit for bit.
    int32_t *perm = nullptr, *cuts = nullptr, *split = nullptr;
    real_t *Cx = nullptr;
    unsigned *xstate = nullptr;
    TileParent *parents = nullptr;
    unsigned xcap = 0, pcap = 0;
    bool build_map = false;
    {
        const char *de = getenv("MXGPU_TILE_DEAL");                      // (read per call: the tests and the A/B runs switch it)
        const int deal_env = de ? atoi(de) : -1;
        const bool hinted = g_tile_deal_hint >= 0.0f;
        const bool deal = deal_env >= 0 ? deal_env != 0 : (hinted ? g_tile_deal_hint : profile_cv()) > 0.15;
        g_tile_last_mode = 0;
        if (deal && nnz > 0 && gm.R <= TL_MAX_WAVES * TL_NG * 5) {
            const int Rrows = gm.R;                                      // rows of a base block (the slots may be more, below)
            const int tile_rows = (small_tile ? 32768 : 65536) / (256 * gm.cpl), ntiles = (int)ceil_div(K, tile_rows);
            const char *pe = getenv("MXGPU_TILE_PART");                    // (entries per tile and part; experiments)
            const int part_len = (pe && atoi(pe) > 0 ? atoi(pe) : 40) * ntiles;
            const char *se = getenv("MXGPU_TILE_SPLIT");
            // (the longest row at least 2.5 parts long; 2 parts for row-major C when a row has few entries per tile — tools/tile_split_probe.py,
            // 20,000 x 20,000, 300 per row, f64, longest row 2 parts: row-major 0.50 -> 0.41 / 0.50 -> 0.34, column-major, whose cut-row
            // instances are the ones short of registers, 0.50 -> 0.54 / 0.51 -> 0.50; with 7 entries per tile — 30,000 x 5,000, rows
            // sorted by length — cutting at 2 parts cost 0.157 -> 0.195)
            const double per_tile = (double)nnz / (double)m / (double)ntiles;
            bool want_split = se ? atoi(se) != 0
                                 : profile_longest_over_mean() * ((double)nnz / (double)m) >= (!colmajor && per_tile < 3.0 ? 2.0 : 2.5) * part_len;
            if (m >= (1 << 29) || (hinted && !se)) want_split = false;                     // (the kernel keeps a part's stride in the row number's top bits)
            if (want_split) {
                // a third more slots than rows: one more row per lane group where the registers allow it
                const int gr2 = Rrows / TL_NG + (Rrows / TL_NG + 2) / 3, rgmax = gm.cpl == 1 ? 5 : 4;
                const int lds = 2 * (small_tile ? 32768 : 65536) + 256 * gm.cpl;
                bool found = false;
                TileGeom best = gm;
                for (int w = 14; w >= 4; w--) {
                    TileGeom t = gm;
                    if (!tile_deal(gr2, w, 2, rgmax, t)) continue;
                    if (colmajor && (t.R | 1) * 256 * gm.cpl > lds) continue;
                    if (!found || t.serial < best.serial) { best = t; found = true; }
                }
                // (no larger geometry — the block already has all the rows a lane group can hold: with <= 32,768 rows the one-pass
                // blocks below still make room, a block takes fewer rows where parts want slots; otherwise every row stays whole)
                if (found) { const int nrb0 = gm.nrb; gm = best; gm.nrb = nrb0; } else if (m > 32768) want_split = false;
            }
            // a block above E entries is cut into pieces of E_piece.  A product whose row blocks fill the machine in ONE round of
            // workgroups pays a whole second round for the first extra block (1e4 x 1e4, log-normal sigma 1.5: 60 -> 68 blocks x 4
            // slabs = 272 workgroups, 0.235 -> 0.338 ms): there only a block of twice the mean is cut (rows sorted by length:
            // 0.378 -> 0.230); otherwise from 1.25 mean blocks on.
            const long long mean_block = (long long)((double)nnz / (double)m * Rrows);
            const bool one_round = (long long)ceil_div(m, Rrows) * gm.nslabs <= 256;
            const long long E = std::max<long long>(1024, one_round ? 2 * mean_block : mean_block + mean_block / 4);
            const long long E_piece = std::max<long long>(1024, mean_block + mean_block / 4);
            // with spare slots (the cut rows' geometry) and <= 32,768 rows: blocks by units and entries in one pass (tile_cuts_greedy_kernel)
            // — also without cut rows when a row has few entries per tile (< 3 on average): a slot then costs its visit more than its
            // entries, and tile_cuts_kernel's blocks of equal ENTRIES are the wrong balance (rows sorted by length, 30,000 x 30,000,
            // 200 per row, n = 64: 0.72 -> 0.41 ms f64, 0.37 -> 0.25 f32; 20,000 x 20,000, 300 per row, f64: 0.57 -> 0.49).  With more
            // entries per tile the entry-balanced blocks stay (1e4 x 1e4, 500 per row — 6 per tile: 0.195 -> 0.226 with the one-pass
            // blocks; 30,000 x 5,000, 300 per row: 0.156 -> 0.215).  MXGPU_TILE_GREEDY: 0 never, 1 only with cut rows, 2 whenever the
            // rows are dealt.
            const char *ge = getenv("MXGPU_TILE_GREEDY");
            const int gmode = ge ? atoi(ge) : -1;
            const bool greedy = m <= 32768 && (gmode < 0 ? (want_split || per_tile < 3.0) : (gmode >= 2 || (gmode == 1 && want_split)));
            long long bound = (long long)ceil_div(m, Rrows) + nnz / E_piece + 1;
            // (two consecutive blocks of the walk overflow the slots or the weight limit, which is >= 1.1 mean weights of Rrows rows)
            if (greedy) bound = 2 * ((long long)ceil_div((long long)m + (want_split ? 2 * (long long)nnz / part_len : 0), gm.R) + std::max<long long>(ceil_div(m, Rrows), one_round ? 256 / gm.nslabs : 0)) + 2;
            if (bound * gm.nslabs < (1LL << 30)) {
                const int max_blocks = (int)std::min<long long>(bound, m);
                const size_t cuts_b = (((size_t)max_blocks + 2) * sizeof(int32_t) + 255) & ~(size_t)255;
                const size_t perm_b = ((size_t)max_blocks * gm.R * sizeof(int32_t) + 255) & ~(size_t)255;
                pcap = want_split ? (unsigned)std::min<long long>((long long)max_blocks * gm.R, 1 << 18) : 0;
                xcap = want_split ? (unsigned)std::min<long long>(std::min<long long>((long long)max_blocks * gm.R, (1 << 19) - 2),
                                                                  ((long long)64 << 20) / ((long long)n * (long long)sizeof(real_t))) : 0;
                const size_t par_b = ((size_t)pcap * sizeof(TileParent) + 255) & ~(size_t)255;
                const size_t u_b = greedy ? (((size_t)m + 1) * sizeof(int32_t) + 255) & ~(size_t)255 : 0;
                const size_t blk_b = want_split ? ((size_t)2 * max_blocks * sizeof(int32_t) + 255) & ~(size_t)255 : 0;
                const size_t split_b = want_split ? perm_b + 256 + par_b + blk_b : 0;      // split | xstate | parents | blk, then U
                char *buf = (char *)scratch_buffer(MX_SCRATCH_TILE_PERM, cuts_b + perm_b + split_b + u_b);
                if (buf && want_split) {
                    Cx = (real_t *)scratch_buffer(MX_SCRATCH_TILE_X, (size_t)xcap * n * sizeof(real_t) + 256);
                    if (!Cx) { (void)hipGetLastError(); buf = nullptr; }   // (no memory for the parts' rows: consecutive rows, whole)
                }
                if (buf) {
                    struct Key { const void *indptr, *buf, *cx; long long nnz; unsigned long long rgw; int m, R, Rrows, nw, nl, max_blocks, n, part_len; unsigned gen, genx; };   // (no padding bytes: compared with memcmp; gen: the buffers' allocations — a freed and re-allocated buffer may come back at the same address)
                    static thread_local Key kept[16] = {};
                    int dev = 0;
                    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
                    Key now;
                    memset(&now, 0, sizeof(now));
                    now.indptr = indptr; now.buf = buf; now.cx = Cx; now.nnz = (long long)nnz; now.rgw = gm.rgw; now.m = m; now.R = gm.R; now.Rrows = Rrows;
                    now.nw = gm.nw; now.nl = gm.nl | (greedy ? 256 : 0); now.max_blocks = max_blocks; now.n = want_split ? n : 0; now.part_len = want_split ? part_len : 0;
                    now.gen = scratch_generation(MX_SCRATCH_TILE_PERM); now.genx = want_split ? scratch_generation(MX_SCRATCH_TILE_X) : 0;
                    build_map = memcmp(&kept[dev], &now, sizeof(Key)) != 0;
                    kept[dev] = now;
                    scratch_acquire(MX_SCRATCH_TILE_PERM, stream);
                    if (want_split) scratch_acquire(MX_SCRATCH_TILE_X, stream);
                    cuts = (int32_t *)buf; perm = (int32_t *)(buf + cuts_b);
                    if (want_split) {
                        split = (int32_t *)(buf + cuts_b + perm_b);
                        xstate = (unsigned *)(buf + cuts_b + 2 * perm_b);
                        parents = (TileParent *)(buf + cuts_b + 2 * perm_b + 256);
                    }
                    gm.nrb = max_blocks;
                    g_tile_last_mode = 1 | (want_split ? 2 : 0) | (greedy ? 4 : 0);
                    if (build_map) {
                        // (the cut decisions need this call's sortedness flags: a row that is not sorted by column stays whole)
                        if (flags && want_split)
                            hipLaunchKernelGGL(tile_unsorted_rows_kernel, dim3((unsigned)ceil_div(m, 8)), dim3(512), 0, stream, m, indptr, indices, flags);
                        if (greedy) {
                            // weights: entries + w_row per row; one round of workgroups when the rows allow it (1.1, then 1.3 mean weights)
                            const char *we = getenv("MXGPU_TILE_WROW"), *e2 = getenv("MXGPU_TILE_E2");   // (experiments)
                            const int w_row = (we ? atoi(we) : 8) * ntiles;
                            const double f2 = e2 ? atoi(e2) / 100.0 : 1.3;
                            const double w_total = (double)nnz + (double)m * w_row;
                            const int nb1 = 256 / gm.nslabs;
                            const double w_block = one_round ? w_total / nb1 : ((double)mean_block + (double)Rrows * w_row) * 1.14;
                            hipLaunchKernelGGL(tile_cuts_greedy_kernel, dim3(1), dim3(1024), 0, stream, m, gm.R, indptr, (const unsigned char *)(want_split ? flags : nullptr),
                                               want_split ? part_len : 0,
                                               w_row, (long long)(1.1 * w_block) + 1024, (long long)(f2 * w_block) + 1024, one_round ? nb1 : 0,
                                               (int32_t *)(buf + cuts_b + perm_b + split_b), cuts, max_blocks);
                        }
                        else
                            hipLaunchKernelGGL(tile_cuts_kernel, dim3(1), dim3(1024), 0, stream, m, Rrows, indptr, E, E_piece, cuts, max_blocks);
                        int32_t *blk = want_split ? (int32_t *)(buf + cuts_b + 2 * perm_b + 256 + par_b) : nullptr;
                        if (want_split) {                                // the parts' places in block order: count, scan, then cut
                            MX_HIP(hipMemsetAsync(blk, 0, (size_t)2 * max_blocks * sizeof(int32_t), stream));
                            hipLaunchKernelGGL(tile_deal_rows_kernel, dim3((unsigned)max_blocks), dim3(256), 0, stream, m, gm.R, gm.nw, gm.rgw, indptr, cuts, perm,
                                               split, part_len, (const unsigned char *)flags, 0, blk, max_blocks, xcap, parents, pcap);
                            hipLaunchKernelGGL(tile_part_scan_kernel, dim3(1), dim3(1024), 0, stream, max_blocks, blk, xstate);
                        }
                        hipLaunchKernelGGL(tile_deal_rows_kernel, dim3((unsigned)max_blocks), dim3(256), 0, stream, m, gm.R, gm.nw, gm.rgw, indptr, cuts, perm,
                                           split, part_len, (const unsigned char *)flags, 1, blk, max_blocks, xcap, parents, pcap);
                    }
                } else { (void)hipGetLastError(); Cx = nullptr; }        // (no memory for the map: consecutive rows)
            }
        }
    }
    kt_begin(stream);
    if (flags)
        hipLaunchKernelGGL(tile_unsorted_rows_kernel, dim3((unsigned)ceil_div(m, 8)), dim3(512), 0, stream, m, indptr, indices, flags);
#define MX_TL_GO(CPL, WIN, TILE)                                                                                                  \
    launch_tile_rg<real_t, CPL, WIN, TILE>(gm, m, n, K, indptr, indices, values, flags, perm, cuts, split, Cx, B, ldb, C, ldc, colmajor, c_vec, stream)
    if (gm.cpl == 2) { if (small_tile) MX_TL_GO(2, 1, 32768); else MX_TL_GO(2, 2, 65536); }
    else { if (small_tile) MX_TL_GO(1, 1, 32768); else MX_TL_GO(1, 2, 65536); }
#undef MX_TL_GO
    if (split)
        hipLaunchKernelGGL((tile_combine_kernel<real_t>), dim3(256), dim3(256), 0, stream, n, (const unsigned *)xstate, pcap, (const TileParent *)parents,
                           (const real_t *)Cx, C, ldc, colmajor);
    if (flags) scratch_done(MX_SCRATCH_TILE_FLAGS, stream);
    if (perm) scratch_done(MX_SCRATCH_TILE_PERM, stream);
    if (split) scratch_done(MX_SCRATCH_TILE_X, stream);
    kt_end(stream);
    MX_LAUNCH_CHECK();
    return 0;
}
template int tile_spmm<double>(int, int, int, int64_t, int, int, int, const int32_t *, const int32_t *, const double *, const double *, size_t,
                               double *, size_t, int, hipStream_t);
template int tile_spmm<float>(int, int, int, int64_t, int, int, int, const int32_t *, const int32_t *, const double *, const float *, size_t,
                              float *, size_t, int, hipStream_t);

TileDealScope::TileDealScope(float cv) : saved(g_tile_deal_hint) { g_tile_deal_hint = cv; }
TileDealScope::~TileDealScope() { g_tile_deal_hint = saved; }

}  // namespace mx

// diagnostic: how the process's last tile product laid out its rows — 0 consecutive rows, 1 dealt by length, | 2 long rows cut
// into parts, | 4 row blocks made in one pass
extern "C" int mxd_debug_spmm_tile_mode(void) { return mx::g_tile_last_mode.load(std::memory_order_relaxed); }

// diagnostic: a device buffer of 2 x 16 x (workgroups) uint64 makes the tile kernel record, per compute wavefront, the shader
// clock cycles it spent at the tile barriers and in its sweep; NULL switches back
extern "C" int mxd_debug_spmm_tile_stamps(void *stamps_dev)
{
    mx::g_tile_stamps = (unsigned long long *)stamps_dev;
    return 0;
}