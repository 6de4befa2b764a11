// spmm.hip — CSR x dense SpMM for gfx950 (MI355X), hand-written HIP.
//
// Replaces the two OpenMP loops of the reference:
//   gemm_csr_drm_as_drm  src/matmul.cpp:118-142  (C row-major)
//   gemm_csr_drm_as_dcm  src/matmul.cpp:150-185  (C column-major, what R needs)
//
// Design (v1, "row-wave"): a workgroup of 4 wavefronts owns a tile of TR
// consecutive rows; each wavefront walks TR/4 rows.  For one row the 64 lanes
// span 64*VEC consecutive columns of the output, so every nonzero a_ij turns
// into ONE fully coalesced read of B[j, slab] (1 KiB for f64 n=128 / f32
// n=256: 16 B per lane) with the row base address in SGPRs (v_readlane of the
// column id), followed by VEC FMAs per lane.  (j, a) of the row are loaded
// coalesced by the wave, kept in one VGPR pair and broadcast lane by lane; the
// next row's first chunk is prefetched while the current row streams B.
// Accumulation runs in CSR storage order, one FMA per nonzero per column —
// the same order as the reference's axpy loop — so results differ from an
// FMA-enabled BLAS in nothing and from a non-FMA one by one rounding per term.
//
// Column-major epilogue: the tile's rows are parked in LDS (row stride odd ->
// conflict-free transposed reads) and written out as TR-row-contiguous
// segments per output column (256 B for f64, TR=32), instead of the
// stride-m scatter the CPU code does with dcopy.
//
// Roofline: HBM-bound.  Algorithmic bytes per launch =
//   4(m+1) + 12 nnz + s*K*n + s*m*n   (SURVEY §8d).  The gather of B rows is
// served by L2 / Infinity Cache (B = 102 MB for the headline config).
#include "mx_common.h"
#include <cstdlib>

namespace mx {

template <typename T, int N> struct VecT;
template <> struct VecT<double, 1> { using type = double; };
template <> struct VecT<double, 2> { using type = double __attribute__((ext_vector_type(2))); };
template <> struct VecT<float, 1>  { using type = float; };
template <> struct VecT<float, 2>  { using type = float __attribute__((ext_vector_type(2))); };
template <> struct VecT<float, 4>  { using type = float __attribute__((ext_vector_type(4))); };

template <typename real_t, int VEC>
__device__ __forceinline__ void vload(real_t (&dst)[VEC], const real_t *__restrict__ p)
{
    using V = typename VecT<real_t, VEC>::type;
    if constexpr (VEC == 1) {
        dst[0] = *p;
    } else {
        const V v = *reinterpret_cast<const V *>(p);
#pragma unroll
        for (int i = 0; i < VEC; i++) dst[i] = v[i];
    }
}

template <typename real_t, int VEC>
__device__ __forceinline__ void vstore(real_t *__restrict__ p, const real_t (&src)[VEC])
{
    using V = typename VecT<real_t, VEC>::type;
    if constexpr (VEC == 1) {
        *p = src[0];
    } else {
        V v;
#pragma unroll
        for (int i = 0; i < VEC; i++) v[i] = src[i];
        *reinterpret_cast<V *>(p) = v;
    }
}

__device__ __forceinline__ double mx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float mx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

constexpr int SPMM_WAVES = 4;     // wavefronts per workgroup
constexpr int SPMM_UNROLL = 8;    // B-row reads in flight per wavefront

// one chunk of <=64 nonzeros of the current row: lane k holds (jv, av) of entry k.
// B is wave-uniform and `col` a per-lane element offset, so each read is
// "SGPR row base + VGPR lane offset" (global_load ... s[base], no 64-bit VALU
// address arithmetic per nonzero).  Lanes past the last column read a clamped,
// valid column instead of branching; their results are never stored.
template <typename real_t, int VEC>
__device__ __forceinline__ void spmm_chunk(int cnt, int jv, double av,
                                           const real_t *__restrict__ B, size_t ldb, unsigned col,
                                           real_t (&acc)[VEC])
{
    int k = 0;
    for (; k + SPMM_UNROLL <= cnt; k += SPMM_UNROLL) {
        real_t b[SPMM_UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < SPMM_UNROLL; u++) {
            const int j = __builtin_amdgcn_readlane(jv, k + u);
            const real_t *__restrict__ rowp = B + (size_t)j * ldb;
            vload<real_t, VEC>(b[u], rowp + col);
        }
#pragma unroll
        for (int u = 0; u < SPMM_UNROLL; u++) {
            const real_t a = (real_t)readlane_f64(av, k + u);   // narrowed per nonzero for f32 (matmul.cpp:53-57)
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[u][v], acc[v]);
        }
    }
    for (; k < cnt; k++) {
        const int j = __builtin_amdgcn_readlane(jv, k);
        const real_t a = (real_t)readlane_f64(av, k);
        const real_t *__restrict__ rowp = B + (size_t)j * ldb;
        real_t b[VEC];
        vload<real_t, VEC>(b, rowp + col);
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[v], acc[v]);
    }
}

// TR rows per workgroup.  COLMAJOR: stage the tile in LDS and write transposed.
// VSTORE (COLMAJOR only): two consecutive rows per lane -> wider stores; needs
// even ldc and 2*sizeof(real_t)-aligned C.
template <typename real_t, int VEC, bool COLMAJOR, bool VSTORE, int TR>
__global__ __launch_bounds__(SPMM_WAVES * MX_WAVE)
void spmm_rowwave_kernel(int m, int n,
                         const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                         const double *__restrict__ values,
                         const real_t *__restrict__ B, size_t ldb,
                         real_t *__restrict__ C, size_t ldc)
{
    constexpr int W = MX_WAVE * VEC;          // output columns per workgroup pass
    constexpr int S = W + 1;                  // odd LDS row stride (elements)
    constexpr int ROWS_PER_WAVE = TR / SPMM_WAVES;
    __shared__ real_t tile[COLMAJOR ? TR * S : 1];

    const int lane = lane_id();
    const int wave = uniform(threadIdx.x / MX_WAVE);
    const int row0 = blockIdx.x * TR;
    const int c0 = blockIdx.y * W;
    const int col = c0 + lane * VEC;
    const bool active = col < n;
    // clamped column for the reads of inactive lanes (n >= VEC always holds here)
    const unsigned lcol = active ? (unsigned)col : (unsigned)(n - VEC);

    const int r_begin = row0 + wave * ROWS_PER_WAVE;
    const int r_end = min(r_begin + ROWS_PER_WAVE, m);

    int s = 0, e = 0, jv = 0;
    double av = 0.0;
    if (r_begin < r_end) {
        s = uniform(indptr[r_begin]);
        e = uniform(indptr[r_begin + 1]);
        if (s + lane < e) { jv = indices[s + lane]; av = values[s + lane]; }
    }
    for (int row = r_begin; row < r_end; row++) {
        // prefetch the first chunk of the next row
        int e2 = e, jv2 = 0;
        double av2 = 0.0;
        if (row + 1 < r_end) {
            e2 = uniform(indptr[row + 2]);
            if (e + lane < e2) { jv2 = indices[e + lane]; av2 = values[e + lane]; }
        }
        real_t acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = 0;

        spmm_chunk<real_t, VEC>(min(MX_WAVE, e - s), jv, av, B, ldb, lcol, acc);
        for (int k0 = s + MX_WAVE; k0 < e; k0 += MX_WAVE) {   // rows longer than one wavefront
            int jc = 0;
            double ac = 0.0;
            if (k0 + lane < e) { jc = indices[k0 + lane]; ac = values[k0 + lane]; }
            spmm_chunk<real_t, VEC>(min(MX_WAVE, e - k0), jc, ac, B, ldb, lcol, acc);
        }

        if constexpr (COLMAJOR) {
            real_t *t = tile + (row - row0) * S + lane * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v++) t[v] = acc[v];
        } else {
            if (active) vstore<real_t, VEC>(C + (size_t)row * ldc + col, acc);
        }
        s = e; e = e2; jv = jv2; av = av2;
    }

    if constexpr (COLMAJOR) {
        __syncthreads();
        const int ncols = min(W, n - c0);
        constexpr int RPL = VSTORE ? 2 : 1;        // rows per lane in the write-out
        constexpr int LPC = TR / RPL;              // lanes per output column
        constexpr int CPW = MX_WAVE / LPC;         // columns per wave-instruction
        const int q = lane % LPC;
        const int r = q * RPL;
        const int grow = row0 + r;
        for (int cb = wave * CPW; cb < ncols; cb += SPMM_WAVES * CPW) {
            const int c = cb + lane / LPC;
            if (c < ncols && grow < m) {
                real_t *dst = C + (size_t)(c0 + c) * ldc + grow;
                if constexpr (VSTORE) {
                    real_t two[2] = { tile[r * S + c], tile[(r + 1) * S + c] };
                    vstore<real_t, 2>(dst, two);   // m even & grow even => grow+1 < m
                } else {
                    *dst = tile[r * S + c];
                }
            }
        }
    }
}

template <typename real_t, int VEC, bool COLMAJOR, bool VSTORE>
static int launch_spmm(int m, int n, const int32_t *indptr, const int32_t *indices, const double *values,
                       const real_t *B, size_t ldb, real_t *C, size_t ldc, hipStream_t stream)
{
    constexpr int TR = 32;
    constexpr int W = MX_WAVE * VEC;
    dim3 grid((unsigned)ceil_div(m, TR), (unsigned)ceil_div(n, W));
    hipLaunchKernelGGL((spmm_rowwave_kernel<real_t, VEC, COLMAJOR, VSTORE, TR>), grid,
                       dim3(SPMM_WAVES * MX_WAVE), 0, stream, m, n, indptr, indices, values, B, ldb, C, ldc);
    MX_LAUNCH_CHECK();
    return 0;
}

template <typename real_t, int VECMAX>
static int dispatch_spmm(int m, int n, const int32_t *indptr, const int32_t *indices, const double *values,
                         const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream)
{
    // widest per-lane access the operands allow (16 B when rows of B are 16-B aligned)
    const bool b_vec = (n % VECMAX == 0) && (ldb % VECMAX == 0) && ((uintptr_t)B % (VECMAX * sizeof(real_t)) == 0);
    if (colmajor) {
        const bool vs = (ldc % 2 == 0) && ((uintptr_t)C % (2 * sizeof(real_t)) == 0);
        if (b_vec) return vs ? launch_spmm<real_t, VECMAX, true, true>(m, n, indptr, indices, values, B, ldb, C, ldc, stream)
                             : launch_spmm<real_t, VECMAX, true, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
        return vs ? launch_spmm<real_t, 1, true, true>(m, n, indptr, indices, values, B, ldb, C, ldc, stream)
                  : launch_spmm<real_t, 1, true, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
    }
    const bool c_vec = b_vec && (ldc % VECMAX == 0) && ((uintptr_t)C % (VECMAX * sizeof(real_t)) == 0);
    if (c_vec) return launch_spmm<real_t, VECMAX, false, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
    return launch_spmm<real_t, 1, false, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
}


// =====================================================================================================
// v2 "slab / panel sweep" kernel.
//
// Why: with B = K x n row-major far larger than one XCD's 4 MiB L2 (102 MB for the headline config) the
// row-wave kernel above gets a 7 % L2 hit rate and runs at the Infinity-Cache gather rate (~7 TB/s of
// fabric reads for nnz*n*8 = 32.8 GB -> 4.5 ms; profiles/r01_v1_*).  The same kernel with a 4 MB B runs in
// 1.5 ms.  This kernel restructures the iteration space so that each XCD's *working set* of B fits its L2:
//
//   * column slabs: the n output columns are cut into 128-byte slabs (16 f64 / 32 f32 columns = one cache
//     line of a B row).  Work items (slab, row block) are dealt slab-major to the 8 XCDs (blockIdx % 8 is
//     the XCD a workgroup lands on — a locality heuristic only, never a correctness assumption), so one
//     XCD touches K x 128 B of B (12.8 MB) instead of all of it;
//   * column panels: [0, K) is cut into `npanels` ranges so that one slab-panel (K/npanels x 128 B) fits
//     L2.  A workgroup keeps the accumulators of its RB = 32*RPG rows in registers and sweeps the panels
//     in order, visiting for each of its rows only the entries whose column lies in the current panel
//     (rows are sorted, so a cursor per row suffices).  All workgroups of an XCD start together and do
//     statistically equal work per panel, so they stay on the same panel (soft synchronisation).
//     npanels > 1 requires rows sorted by column; npanels == 1 works for any order.
//   * 8 lanes own one row (x 16 B per lane = the 128-B slab line), so a wave-instruction reads 8 full
//     lines of B for 8 different rows; (j, a) are loaded coalesced 8 entries at a time per row and
//     broadcast inside the 8-lane group with ds_swizzle.
// Summation order inside a row is still CSR storage order (one FMA per entry), as in the reference.
// =====================================================================================================
constexpr int SLAB_BLOCK = 256;
constexpr int SLAB_GROUP = 8;                       // lanes per row
constexpr int SLAB_GROUPS = SLAB_BLOCK / SLAB_GROUP;

template <int T>
__device__ __forceinline__ int group8_bcast(int v)
{
    // ds_swizzle bit-mask mode: src lane = ((lane & and) | or) ^ xor inside each 32-lane half;
    // and = 0b11000 keeps the 8-lane group, or = T picks entry T of the group.
    return __builtin_amdgcn_ds_swizzle(v, 0x18 | (T << 5));
}
template <int T>
__device__ __forceinline__ double group8_bcast(double v)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = group8_bcast<T>(u.i[0]);
    u.i[1] = group8_bcast<T>(u.i[1]);
    return u.d;
}

__device__ __forceinline__ unsigned group8_ballot(bool pred)
{
    const unsigned long long b = __ballot(pred);
    return (unsigned)(b >> (lane_id() & ~(SLAB_GROUP - 1))) & 0xFFu;
}

// One chunk (<= 8 entries, lane t of the group holds entry t) of one row: all B reads are issued before
// the first FMA so that 8 line reads per group are in flight (a branch per entry would serialise them
// behind s_waitcnt vmcnt(0)).  Entries past `cnt` read a valid address (entry 0's row) and are dropped
// by a select, never by arithmetic (0 * Inf would poison the sum).
template <typename real_t, int VEC, int T>
__device__ __forceinline__ void slab_load(int cnt, int jv, const real_t *__restrict__ B, size_t ldb, unsigned lcol,
                                          real_t (&b)[VEC])
{
    int j = group8_bcast<T>(jv);
    j = (T < cnt) ? j : 0;
    vload<real_t, VEC>(b, B + (size_t)j * ldb + lcol);
}
template <typename real_t, int VEC, int T>
__device__ __forceinline__ void slab_fma(int cnt, double av, const real_t (&b)[VEC], real_t (&acc)[VEC])
{
    const real_t a = (real_t)group8_bcast<T>(av);
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const real_t f = mx_fma(a, b[v], acc[v]);
        acc[v] = (T < cnt) ? f : acc[v];
    }
}
template <typename real_t, int VEC>
__device__ __forceinline__ void slab_chunk(int cnt, int jv, double av, const real_t *__restrict__ B, size_t ldb,
                                           unsigned lcol, real_t (&acc)[VEC])
{
    real_t b0[VEC], b1[VEC], b2[VEC], b3[VEC], b4[VEC], b5[VEC], b6[VEC], b7[VEC];
    slab_load<real_t, VEC, 0>(cnt, jv, B, ldb, lcol, b0);
    slab_load<real_t, VEC, 1>(cnt, jv, B, ldb, lcol, b1);
    slab_load<real_t, VEC, 2>(cnt, jv, B, ldb, lcol, b2);
    slab_load<real_t, VEC, 3>(cnt, jv, B, ldb, lcol, b3);
    slab_load<real_t, VEC, 4>(cnt, jv, B, ldb, lcol, b4);
    slab_load<real_t, VEC, 5>(cnt, jv, B, ldb, lcol, b5);
    slab_load<real_t, VEC, 6>(cnt, jv, B, ldb, lcol, b6);
    slab_load<real_t, VEC, 7>(cnt, jv, B, ldb, lcol, b7);
    slab_fma<real_t, VEC, 0>(cnt, av, b0, acc);
    slab_fma<real_t, VEC, 1>(cnt, av, b1, acc);
    slab_fma<real_t, VEC, 2>(cnt, av, b2, acc);
    slab_fma<real_t, VEC, 3>(cnt, av, b3, acc);
    slab_fma<real_t, VEC, 4>(cnt, av, b4, acc);
    slab_fma<real_t, VEC, 5>(cnt, av, b5, acc);
    slab_fma<real_t, VEC, 6>(cnt, av, b6, acc);
    slab_fma<real_t, VEC, 7>(cnt, av, b7, acc);
}

// Timing-only barrier among the workgroups that share blockIdx % 8 (the XCD group): it keeps them on the same
// column panel so that the panel stays L2-resident.  No data is handed over, so no release/acquire is needed
// and a timeout is harmless: the spin is bounded and falling through only costs locality, never correctness
// (all co-resident by grid sizing; a block that is not resident simply makes the others time out).
__device__ __forceinline__ void xcd_timing_barrier(unsigned *ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 4096)
            __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}

template <typename real_t, int RPG, bool COLMAJOR>
__global__ __launch_bounds__(SLAB_BLOCK)
void spmm_slab_kernel(int m, int n,
                      const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                      const double *__restrict__ values,
                      const real_t *__restrict__ B, size_t ldb,
                      real_t *__restrict__ C, size_t ldc,
                      int npanels, int panel_cols, int nslabs, int nrowblocks, int c_vec_ok,
                      unsigned *__restrict__ sync_ctr, int sync_mode, size_t slab_stride)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;             // columns per slab
    constexpr int RB = SLAB_GROUPS * RPG;           // rows per workgroup step
    const int lg = threadIdx.x & (SLAB_GROUP - 1);
    const int grp = threadIdx.x / SLAB_GROUP;
    const int xcd = blockIdx.x & 7;
    const int wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const long long total = (long long)nslabs * nrowblocks;
    const long long lo = total * xcd / 8, hi = total * (xcd + 1) / 8;

    // every workgroup of the group runs the same number of steps (idle ones only keep the barrier count right)
    const int niter = (int)((hi - lo + nwg - 1) / nwg);
    unsigned *const my_ctr = sync_ctr + xcd * 64;    // one counter per group, 256 B apart
    unsigned step = 0;
    for (int it = 0; it < niter; it++) {
        const long long item_raw = lo + wg + (long long)it * nwg;
        const bool have = item_raw < hi;
        const long long item = have ? item_raw : lo;
        const int slab = (int)(item / nrowblocks);
        const int rb = (int)(item % nrowblocks);
        const int row0 = rb * RB + grp * RPG;
        const int col = slab * W + lg * VEC;
        const bool active = col < n;
        // slab_stride != 0: B was repacked slab-major ([slab][K][W], zero padded) so that a slab is contiguous and
        // spreads over all L2 channels; the caller then passes ldb = W and this adds the slab's base.
        const unsigned lcol = slab_stride ? (unsigned)(lg * VEC) : (active ? (unsigned)col : (unsigned)(n - VEC));
        const real_t *__restrict__ Bs = B + (size_t)slab * slab_stride;

        int cur[RPG], end[RPG];
        real_t acc[RPG][VEC];
#pragma unroll
        for (int r = 0; r < RPG; r++) {
            const int row = row0 + r;
            cur[r] = 0; end[r] = 0;
            if (have && row < m) { cur[r] = indptr[row]; end[r] = indptr[row + 1]; }
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[r][v] = 0;
        }

        for (int p = 0; p < npanels; p++) {
            const int pend = (p == npanels - 1) ? INT_MAX : (p + 1) * panel_cols;
            if (sync_mode == 2 || (sync_mode == 1 && p == 0)) {
                step++;
                xcd_timing_barrier(my_ctr, step * (unsigned)nwg);
            }
            unsigned pending = (1u << RPG) - 1u;          // rows that may still have entries in this panel
            while (__ballot(pending != 0) != 0ULL) {
                int jv[RPG];
                double av[RPG];
#pragma unroll
                for (int r = 0; r < RPG; r++) {
                    // unconditional reads (clamped to entry 0) so that all 2*RPG loads are in flight together
                    const int k = cur[r] + lg;
                    const bool valid = ((pending >> r) & 1u) && k < end[r];
                    const int ks = valid ? k : 0;
                    const int jl = indices[ks];
                    const double al = values[ks];
                    jv[r] = valid ? jl : INT_MAX;
                    av[r] = al;
                }
#pragma unroll
                for (int r = 0; r < RPG; r++) {
                    const unsigned long long inpanel = __ballot(jv[r] < pend);
                    if (inpanel == 0ULL) { pending &= ~(1u << r); continue; }   // no group of this wave has entries here
                    // sorted row: in-panel entries are a prefix of the chunk
                    const int cnt = __popc((unsigned)(inpanel >> (lane_id() & ~(SLAB_GROUP - 1))) & 0xFFu);
                    // entry 0 of an empty chunk may be INT_MAX: slab_load only dereferences entries < cnt (else row 0)
                    slab_chunk<real_t, VEC>(cnt, jv[r], av[r], Bs, ldb, lcol, acc[r]);
                    cur[r] += cnt;
                    if (cnt < SLAB_GROUP) pending &= ~(1u << r);          // panel (or row) exhausted
                }
            }
        }

        // epilogue: lane holds columns col..col+VEC-1 of rows row0..row0+RPG-1
        if (active && have) {
            if constexpr (!COLMAJOR) {
#pragma unroll
                for (int r = 0; r < RPG; r++)
                    if (row0 + r < m) vstore<real_t, VEC>(C + (size_t)(row0 + r) * ldc + col, acc[r]);
            } else {
                constexpr int RV = 16 / (int)sizeof(real_t);               // rows per 16-B store
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    real_t *__restrict__ dst = C + (size_t)(col + v) * ldc + row0;
                    if (c_vec_ok && row0 + RPG <= m) {
#pragma unroll
                        for (int r = 0; r < RPG; r += RV) {
                            real_t tmp[RV];
#pragma unroll
                            for (int q = 0; q < RV; q++) tmp[q] = acc[r + q][v];
                            vstore<real_t, RV>(dst + r, tmp);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < RPG; r++)
                            if (row0 + r < m) dst[r] = acc[r][v];
                    }
                }
            }
        }
    }
}

// panels so that one slab-panel (K/npanels rows x 128 B) stays within `l2_budget` bytes
static int pick_panels(int K, size_t l2_budget)
{
    const size_t slab_bytes = (size_t)K * 128;
    int p = (int)((slab_bytes + l2_budget - 1) / l2_budget);
    if (p < 1) p = 1;
    if (p > 64) p = 64;
    return p;
}

// B (K x n row-major, leading dimension ldb) -> slab-major [nslabs][K][W], zero padded past column n.
// One thread per 16-byte piece; reads are row-contiguous, each 8-lane group writes one full 128-byte line.
template <typename real_t>
__global__ __launch_bounds__(256)
void repack_slabs_kernel(int K, int n, int nslabs, const real_t *__restrict__ B, size_t ldb, real_t *__restrict__ Bp)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    const long long pieces_per_row = (long long)nslabs * SLAB_GROUP;
    const long long total = (long long)K * pieces_per_row;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(t / pieces_per_row);
        const int piece = (int)(t % pieces_per_row);
        const int slab = piece / SLAB_GROUP, lg = piece % SLAB_GROUP;
        const int col = slab * W + lg * VEC;
        real_t v[VEC];
#pragma unroll
        for (int q = 0; q < VEC; q++) v[q] = 0;
        if (col < n) vload<real_t, VEC>(v, B + (size_t)j * ldb + col);       // n % VEC == 0 (slab_ok)
        vstore<real_t, VEC>(Bp + ((size_t)slab * K + j) * W + lg * VEC, v);
    }
}

// grow-only per-device scratch for the packed copy of B
static void *slab_pack_workspace(size_t bytes)
{
    static thread_local void *ws[64] = {};
    static thread_local size_t cap[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (cap[dev] < bytes) {
        if (ws[dev]) (void)hipFree(ws[dev]);
        ws[dev] = nullptr; cap[dev] = 0;
        if (hipMalloc(&ws[dev], bytes) != hipSuccess) return nullptr;
        cap[dev] = bytes;
    }
    return ws[dev];
}

// per-device counters for the timing barrier (8 groups x 256 B), allocated once
static unsigned *slab_sync_workspace()
{
    static thread_local unsigned *ws[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ws[dev] && hipMalloc((void **)&ws[dev], 8 * 64 * sizeof(unsigned)) != hipSuccess) ws[dev] = nullptr;
    return ws[dev];
}

template <typename real_t, int RPG>
static int launch_spmm_slab_rpg(int m, int n, int K, const int32_t *indptr, const int32_t *indices,
                                const double *values, const real_t *B, size_t ldb, real_t *C, size_t ldc,
                                int colmajor, int npanels, int wg_per_cu, int sync_mode, hipStream_t stream)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    constexpr int RB = SLAB_GROUPS * RPG;
    const int nslabs = (int)ceil_div(n, W);
    const int nrowblocks = (int)ceil_div(m, RB);
    if (npanels < 1) npanels = 1;
    const int panel_cols = (int)ceil_div(K > 0 ? K : 1, npanels);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    long long grid = (long long)cus * wg_per_cu;
    const long long total = (long long)nslabs * nrowblocks;
    if (grid > total + 7) grid = total + 7;
    grid = (grid / 8) * 8;
    if (grid < 8) grid = 8;
    const int c_vec_ok = colmajor && (ldc % VEC == 0) && ((uintptr_t)C % 16 == 0);
    unsigned *sync = slab_sync_workspace();
    if (!sync) sync_mode = 0;
    if (sync_mode) MX_HIP(hipMemsetAsync(sync, 0, 8 * 64 * sizeof(unsigned), stream));
    // slab-major copy of B (MXGPU_SLAB_PACK=0 disables): with B row-major a slab is 128 B out of every ldb*s
    // bytes — a power-of-two stride that lands on a fraction of the L2 channels
    size_t slab_stride = 0;
    int pack = 1;
    if (const char *e = getenv("MXGPU_SLAB_PACK")) pack = atoi(e);
    if (pack) {
        const size_t bytes = (size_t)nslabs * (size_t)K * W * sizeof(real_t);
        real_t *Bp = (real_t *)slab_pack_workspace(bytes);
        if (Bp) {
            const long long pieces = (long long)K * nslabs * SLAB_GROUP;
            const unsigned g = (unsigned)(ceil_div(pieces, 256) < 8192 ? ceil_div(pieces, 256) : 8192);
            hipLaunchKernelGGL((repack_slabs_kernel<real_t>), dim3(g), dim3(256), 0, stream, K, n, nslabs, B, ldb, Bp);
            MX_LAUNCH_CHECK();
            B = Bp; ldb = W; slab_stride = (size_t)K * W;
        }
    }
    if (colmajor)
        hipLaunchKernelGGL((spmm_slab_kernel<real_t, RPG, true>), dim3((unsigned)grid), dim3(SLAB_BLOCK), 0, stream,
                           m, n, indptr, indices, values, B, ldb, C, ldc, npanels, panel_cols, nslabs, nrowblocks,
                           c_vec_ok, sync, sync_mode, slab_stride);
    else
        hipLaunchKernelGGL((spmm_slab_kernel<real_t, RPG, false>), dim3((unsigned)grid), dim3(SLAB_BLOCK), 0, stream,
                           m, n, indptr, indices, values, B, ldb, C, ldc, npanels, panel_cols, nslabs, nrowblocks,
                           c_vec_ok, sync, sync_mode, slab_stride);
    MX_LAUNCH_CHECK();
    return 0;
}

template <typename real_t>
static int launch_spmm_slab(int m, int n, int K, const int32_t *indptr, const int32_t *indices,
                            const double *values, const real_t *B, size_t ldb, real_t *C, size_t ldc,
                            int colmajor, int npanels, int wg_per_cu, hipStream_t stream)
{
    // experiment knobs (tuning only): MXGPU_SLAB_SYNC 0 none / 1 per row-block / 2 per panel; MXGPU_SLAB_RPG 8 / 16
    int sync_mode = 2, rpg = 8;
    if (const char *e = getenv("MXGPU_SLAB_SYNC")) sync_mode = atoi(e);
    if (const char *e = getenv("MXGPU_SLAB_RPG")) rpg = atoi(e);
    if (npanels <= 1) sync_mode = 0;
    if (rpg == 16)
        return launch_spmm_slab_rpg<real_t, 16>(m, n, K, indptr, indices, values, B, ldb, C, ldc, colmajor, npanels,
                                                wg_per_cu, sync_mode, stream);
    return launch_spmm_slab_rpg<real_t, 8>(m, n, K, indptr, indices, values, B, ldb, C, ldc, colmajor, npanels,
                                           wg_per_cu, sync_mode, stream);
}

// can the slab kernel take these operands?  (16-B aligned rows of B, whole vectors per row;
// row-major C additionally needs 16-B aligned rows of C)
template <typename real_t>
static bool slab_ok(int n, const real_t *B, size_t ldb, const real_t *C, size_t ldc, int colmajor)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    if (n < VEC || n % VEC || ldb % VEC || (uintptr_t)B % 16) return false;
    if (!colmajor && (ldc % VEC || (uintptr_t)C % 16)) return false;
    return true;
}

}  // namespace mx

static thread_local const char *g_last_spmm_kernel = "none";
extern "C" const char *mxd_spmm_last_kernel(void) { return g_last_spmm_kernel; }

extern "C" int mxd_spmm_csr_dense_ex(int m, int n, int K,
                                     const int32_t *indptr, const int32_t *indices, const double *values,
                                     const void *B, size_t ldb, void *C, size_t ldc,
                                     int dense_dtype, int colmajor_out, int algo, int rows_sorted,
                                     int npanels, int wg_per_cu, void *stream)
{
    MX_REQUIRE(m >= 0 && n >= 0 && K >= 0, "mxd_spmm_csr_dense_ex: negative dimension");
    if (m == 0 || n == 0) return 0;
    MX_REQUIRE(indptr && B && C, "mxd_spmm_csr_dense_ex: null pointer");
    MX_REQUIRE(dense_dtype == MX_F64 || dense_dtype == MX_F32, "mxd_spmm_csr_dense_ex: unsupported dense dtype %d", dense_dtype);
    hipStream_t st = mx::as_stream(stream);
    const bool ok = dense_dtype == MX_F64
        ? mx::slab_ok<double>(n, (const double *)B, ldb, (const double *)C, ldc, colmajor_out)
        : mx::slab_ok<float>(n, (const float *)B, ldb, (const float *)C, ldc, colmajor_out);
    bool auto_pick = false;
    if (algo == MX_SPMM_AUTO) {
        // Measured on MI355X, headline config (profiles/r01_*): row-wave 4.45 ms; slab kernel with B repacked
        // slab-major, one panel, no barrier 4.05 ms (the slab-major copy spreads a slab over all L2 channels);
        // slab + column panels + XCD barrier 5.9 ms (L2 hits 28 -> 60 %, but every panel visit re-reads the row's
        // (j, a) and the kernel turns VALU-bound).  So AUTO = one-panel packed slab kernel when B outgrows one
        // XCD's L2 and there is enough work to fill the persistent grid, else the row-wave kernel.
        const size_t b_bytes = (size_t)K * (size_t)n * (dense_dtype == MX_F64 ? 8 : 4);
        auto_pick = ok && b_bytes > ((size_t)8 << 20) && (long long)m * n >= (1LL << 24);
        algo = auto_pick ? MX_SPMM_SLAB : MX_SPMM_ROWWAVE;
        if (auto_pick) { npanels = 1; if (wg_per_cu <= 0) wg_per_cu = 4; }
    }
    if (algo == MX_SPMM_SLAB) {
        MX_REQUIRE(ok, "mxd_spmm_csr_dense_ex: operands do not meet the slab kernel's 16-byte alignment rules");
        g_last_spmm_kernel = "spmm_slab_kernel";
        if (npanels <= 0) npanels = rows_sorted ? mx::pick_panels(K, (size_t)2560 << 10) : 1;
        if (!rows_sorted) npanels = 1;                 // panels need column-sorted rows
        if (wg_per_cu <= 0) wg_per_cu = 4;
        if (dense_dtype == MX_F64)
            return mx::launch_spmm_slab<double>(m, n, K, indptr, indices, values, (const double *)B, ldb, (double *)C,
                                                ldc, colmajor_out, npanels, wg_per_cu, st);
        return mx::launch_spmm_slab<float>(m, n, K, indptr, indices, values, (const float *)B, ldb, (float *)C, ldc,
                                           colmajor_out, npanels, wg_per_cu, st);
    }
    g_last_spmm_kernel = "spmm_rowwave_kernel";
    return mxd_spmm_csr_dense(m, n, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor_out, stream);
}

extern "C" int mxd_spmm_csr_dense(int m, int n,
                                  const int32_t *indptr, const int32_t *indices, const double *values,
                                  const void *B, size_t ldb, void *C, size_t ldc,
                                  int dense_dtype, int colmajor_out, void *stream)
{
    MX_REQUIRE(m >= 0 && n >= 0, "mxd_spmm_csr_dense: negative dimension (m=%d, n=%d)", m, n);
    if (m == 0 || n == 0) return 0;
    MX_REQUIRE(indptr && B && C, "mxd_spmm_csr_dense: null pointer");
    hipStream_t st = mx::as_stream(stream);
    if (dense_dtype == MX_F64)
        return mx::dispatch_spmm<double, 2>(m, n, indptr, indices, values, (const double *)B, ldb,
                                            (double *)C, ldc, colmajor_out, st);
    if (dense_dtype == MX_F32)
        return mx::dispatch_spmm<float, 4>(m, n, indptr, indices, values, (const float *)B, ldb,
                                           (float *)C, ldc, colmajor_out, st);
    return mx::set_error("mxd_spmm_csr_dense: unsupported dense dtype %d", dense_dtype);
}
