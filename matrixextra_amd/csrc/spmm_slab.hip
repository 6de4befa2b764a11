// spmm_slab.hip — the slab / panel-sweep SpMM kernel (v2, opt-in: MX_SPMM_SLAB), the slab-major repack of B and the
// per-device workspaces shared with the planned kernel.
#include "spmm_common.h"

namespace mx {

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
int pick_panels(int K, size_t l2_budget)
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
void repack_slabs_kernel(int K, int Kp, int n, int nslabs, const real_t *__restrict__ B, size_t ldb, real_t *__restrict__ Bp)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    const long long pieces_per_row = (long long)nslabs * SLAB_GROUP;
    const long long total = (long long)Kp * pieces_per_row;          // rows K..Kp-1 of every slab are zero (plan padding)
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(t / pieces_per_row);
        const int piece = (int)(t % pieces_per_row);
        const int slab = piece / SLAB_GROUP, lg = piece % SLAB_GROUP;
        const int col = slab * W + lg * VEC;
        real_t v[VEC];
#pragma unroll
        for (int q = 0; q < VEC; q++) v[q] = 0;
        if (col < n && j < K) vload<real_t, VEC>(v, B + (size_t)j * ldb + col);       // n % VEC == 0 (slab_ok)
        vstore<real_t, VEC>(Bp + ((size_t)slab * Kp + j) * W + lg * VEC, v);
    }
}

// grow-only per-device scratch for the packed copy of B
void *slab_pack_workspace(size_t bytes, bool release)
{
    static thread_local void *ws[64] = {};
    static thread_local size_t cap[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (release) { if (ws[dev]) (void)hipFree(ws[dev]); ws[dev] = nullptr; cap[dev] = 0; return nullptr; }
    if (cap[dev] < bytes) {
        if (ws[dev]) (void)hipFree(ws[dev]);
        ws[dev] = nullptr; cap[dev] = 0;
        if (hipMalloc(&ws[dev], bytes) != hipSuccess) return nullptr;
        cap[dev] = bytes;
    }
    return ws[dev];
}

// per-device counters for the timing barrier (8 groups x 256 B), allocated once
unsigned *slab_sync_workspace()
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
            scratch_acquire(MX_SCRATCH_PACKED_B, stream);
            if (launch_repack<real_t>(K, K, n, B, ldb, Bp, stream)) return 1;
            B = Bp; ldb = W; slab_stride = (size_t)K * W;
        }
    }
    kt_begin(stream);
    if (colmajor)
        hipLaunchKernelGGL((spmm_slab_kernel<real_t, RPG, true>), dim3((unsigned)grid), dim3(SLAB_BLOCK), 0, stream,
                           m, n, indptr, indices, values, B, ldb, C, ldc, npanels, panel_cols, nslabs, nrowblocks,
                           c_vec_ok, sync, sync_mode, slab_stride);
    else
        hipLaunchKernelGGL((spmm_slab_kernel<real_t, RPG, false>), dim3((unsigned)grid), dim3(SLAB_BLOCK), 0, stream,
                           m, n, indptr, indices, values, B, ldb, C, ldc, npanels, panel_cols, nslabs, nrowblocks,
                           c_vec_ok, sync, sync_mode, slab_stride);
    kt_end(stream);
    scratch_done(MX_SCRATCH_PACKED_B, stream);
    MX_LAUNCH_CHECK();
    return 0;
}

template <typename real_t>
int launch_repack(int K, int Kp, int n, const real_t *B, size_t ldb, real_t *Bp, hipStream_t st)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    const int nslabs = (int)ceil_div(n, W);
    const long long pieces = (long long)Kp * nslabs * SLAB_GROUP;
    const unsigned gsz = (unsigned)(ceil_div(pieces, 256) < 8192 ? ceil_div(pieces, 256) : 8192);
    hipLaunchKernelGGL((repack_slabs_kernel<real_t>), dim3(gsz), dim3(256), 0, st, K, Kp, n, nslabs, B, ldb, Bp);
    MX_LAUNCH_CHECK();
    return 0;
}
template int launch_repack<double>(int, int, int, const double *, size_t, double *, hipStream_t);
template int launch_repack<float>(int, int, int, const float *, size_t, float *, hipStream_t);

template <typename real_t>
int slab_spmm(int m, int n, int K, const int32_t *indptr, const int32_t *indices,
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
template int slab_spmm<double>(int, int, int, const int32_t *, const int32_t *, const double *, const double *, size_t,
                               double *, size_t, int, int, int, hipStream_t);
template int slab_spmm<float>(int, int, int, const int32_t *, const int32_t *, const double *, const float *, size_t,
                              float *, size_t, int, int, int, hipStream_t);

}  // namespace mx
