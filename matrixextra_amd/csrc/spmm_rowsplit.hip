// spmm_rowsplit.hip — CSR x dense SpMM for gfx950 (MI355X): the kernel for SHORT-AND-FAT products (few rows and / or
// long rows, narrow dense operands), the shape of the one workload the reference publishes a number for: dense 100 x 1e4
// %*% CSC 1e4 x 1e4, density .05 (vignettes/Introducing_MatrixExtra.Rmd:247-251 -> matmul_dense_csc_numeric ->
// gemm_csr_drm_as_drm, src/matmul.cpp:118-142, :188-235) — 1e4 rows of ~500 entries against a 100-column B.
//
// Why another kernel: the row-wave kernel (spmm_rowwave.hip) gives 8 consecutive rows to one wavefront and 64 * VEC output
// columns to its 64 lanes.  With m = 1e4 that is 313 workgroups for 256 CUs — one or two wavefronts per SIMD, every one of
// them a serial chain of 4,000 dependent-latency B-row reads — with n = 16 or 64 most of the lanes of every read are idle,
// and the whole of B (8 MB there) is gathered by every XCD through a 4 MiB L2: half of the line reads come from the
// Infinity Cache (measured: 0.343 ms = 11.7 TB/s of line reads; an L2-resident gather runs at 20-28 TB/s).
// The planned sweep (spmm_plan.hip) needs m / 8 octets to fill its persistent grid.  Here:
//
//   * the unit of work is one SEGMENT of one row: a workgroup is 8 wavefronts = 8 / S rows x S segments per row
//     (S in {1, 2, 4, 8}, chosen on the host so that the launch has a few wavefronts for every slot of the machine while
//     a segment keeps >= 64 entries);
//   * inside a wavefront, G lanes (G = 8 .. 64, G * VEC >= the columns of the pass) own one row of B, so one load
//     instruction covers 64 / G entries; the segment's (j, a) are loaded coalesced, 64 entries at a time, and handed to
//     the groups by ds_bpermute (G < 64) or v_readlane into SGPRs (G = 64: no per-entry address arithmetic);
//   * COLUMN PANELS by launches: when B outgrows an XCD's L2, [0, K) is cut into P ranges of ~2.5 MB of B and the product
//     runs as P launches, launch p taking from every row the entries whose column lies in panel p and continuing the sums
//     left in C by launch p - 1 — so that at any moment every wavefront of the machine gathers from the same few MB.  Rows
//     sorted by column (what R's CSR / CSC classes hold) have their panel bounds found once by a cursor kernel (one
//     wavefront per row: sortedness check + P binary searches); a row that is NOT sorted is simply taken whole by launch 0
//     — always correct, no host round trip;
//   * partial sums meet in LDS: groups inside a wave by a butterfly, segments of a row in a fixed order when the tile is
//     written out — run-to-run reproducible, no atomics.  S = 1 and G = 64 with row-major C is the reference's
//     storage-order FMA chain bit for bit, panels included (the chain continues from the value stored in C); everything
//     else reassociates (<= 1e-12 relative, the bar of the planned kernel);
//   * both layouts of C: row-major rows straight from the tile, column-major as 8 / S-row segments per column (C is small
//     wherever this kernel is chosen);
//   * a second form, ROW GROUPS (spmm_rowgroup_kernel below), for the opposite corner — many SHORT rows against a narrow B:
//     several rows per wavefront, each summed by its own lane group in storage order.
//
// Roofline: HBM-bound by the contract's algorithmic bytes; what limits it in practice is the L2 -> L1 gather
// of nnz * n * s bytes (4 GB for the vignette product), reported as `l2_to_l1_gather` in bench.py.
#include "spmm_common.h"

namespace mx {

constexpr int RS_WAVES = 8;        // wavefronts per workgroup
constexpr int RS_UNROLL = 8;       // B-row reads in flight per wavefront (G = 64)
constexpr int RS_MAX_PANELS = 32;

// G = 64: one chunk of <= 64 entries, lane k holds (jv, av) of entry k — the row-wave kernel's inner loop
template <typename real_t, int VEC>
__device__ __forceinline__ void rs_chunk(int cnt, int jv, double av, const real_t *__restrict__ B, size_t ldb, unsigned col,
                                         real_t (&acc)[VEC])
{
    int k = 0;
    if (cnt == MX_WAVE) {
        // a full chunk: two register sets, the loads of one batch issued before the FMAs of the batch before it — 8 to 16
        // B-row reads in flight all the way through the chunk instead of draining to zero at every batch (same FMA order)
        real_t b0[RS_UNROLL][VEC], b1[RS_UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < RS_UNROLL; u++) vload<real_t, VEC>(b0[u], B + (size_t)__builtin_amdgcn_readlane(jv, u) * ldb + col);
#pragma unroll
        for (int kk = 0; kk < MX_WAVE; kk += 2 * RS_UNROLL) {
#pragma unroll
            for (int u = 0; u < RS_UNROLL; u++)
                vload<real_t, VEC>(b1[u], B + (size_t)__builtin_amdgcn_readlane(jv, kk + RS_UNROLL + u) * ldb + col);
#pragma unroll
            for (int u = 0; u < RS_UNROLL; u++) {
                const real_t a = (real_t)readlane_f64(av, kk + u);
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b0[u][v], acc[v]);
            }
            if (kk + 2 * RS_UNROLL < MX_WAVE) {
#pragma unroll
                for (int u = 0; u < RS_UNROLL; u++)
                    vload<real_t, VEC>(b0[u], B + (size_t)__builtin_amdgcn_readlane(jv, kk + 2 * RS_UNROLL + u) * ldb + col);
            }
#pragma unroll
            for (int u = 0; u < RS_UNROLL; u++) {
                const real_t a = (real_t)readlane_f64(av, kk + RS_UNROLL + u);
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b1[u][v], acc[v]);
            }
        }
        return;
    }
    for (; k + RS_UNROLL <= cnt; k += RS_UNROLL) {
        real_t b[RS_UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < RS_UNROLL; u++) {
            const int j = __builtin_amdgcn_readlane(jv, k + u);
            vload<real_t, VEC>(b[u], B + (size_t)j * ldb + col);
        }
#pragma unroll
        for (int u = 0; u < RS_UNROLL; u++) {
            const real_t a = (real_t)readlane_f64(av, k + u);       // narrowed per entry for f32 (matmul.cpp:53-57)
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[u][v], acc[v]);
        }
    }
    for (; k < cnt; k++) {
        const int j = __builtin_amdgcn_readlane(jv, k);
        const real_t a = (real_t)readlane_f64(av, k);
        real_t b[VEC];
        vload<real_t, VEC>(b, B + (size_t)j * ldb + col);
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[v], acc[v]);
    }
}

__device__ __forceinline__ double bperm_f64(double v, int src_lane)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[0]);
    u.i[1] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[1]);
    return u.d;
}

__device__ __forceinline__ double bperm_real(double v, int src_lane) { return bperm_f64(v, src_lane); }
__device__ __forceinline__ float bperm_real(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// G < 64: one chunk of <= 64 entries, lane k holds entry k; group g = lane / G takes entries g, g + NG, ... — one load
// instruction reads NG rows of B, 16 bytes per lane.  U instructions are in flight; entries past `cnt` re-read entry 0's
// row (valid memory) and are dropped by a select, never by arithmetic (0 * Inf would poison the sum).
template <typename real_t, int VEC, int G>
__device__ __forceinline__ void rs_chunk_groups(int cnt, int jv, double av, const real_t *__restrict__ B, size_t ldb, unsigned col,
                                                int g, real_t (&acc)[VEC])
{
    constexpr int NG = MX_WAVE / G;
    constexpr int U = G >= 32 ? 8 : 4;
    for (int t = 0; t * NG < cnt; t += U) {
        int jj[U];
        real_t bb[U][VEC];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = (t + u) * NG + g;
            jj[u] = __builtin_amdgcn_ds_bpermute((k < cnt ? k : 0) << 2, jv);
        }
#pragma unroll
        for (int u = 0; u < U; u++) vload<real_t, VEC>(bb[u], B + (size_t)jj[u] * ldb + col);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = (t + u) * NG + g;
            const real_t x = (real_t)bperm_f64(av, k < cnt ? k : 0);
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = k < cnt ? mx_fma(x, bb[u][v], acc[v]) : acc[v];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// COLUMN PANELS IN ONE LAUNCH.  The grid is 1-D, panel-major: workgroup id = (p * passes + pass) * nbx + rb, so the dispatcher
// hands out panel 0's workgroups, then panel 1's, ... exactly in the order the P launches of rounds 3-4 ran them — without
// the P - 1 launch gaps and without draining the machine at the end of every panel (the tail of panel p runs beside the
// head of panel p + 1).  Panel p continues the sums panel p - 1 left in C, and the two workgroups of one row block sit on
// different XCDs (id % 8), whose L2s do not see each other's plain stores: the hand-over goes through agent-scope
// (write-through / re-fetching) stores and loads of C, ordered by a counter per (pass, row block): a workgroup adds one when
// its part of C has left (s_waitcnt vmcnt(0)), its successor waits for the count to reach p before it reads C.  Workgroups
// are dispatched in id order, so a waiting workgroup's predecessor is always running or done: no deadlock (the same
// assumption as the look-back scans of mx_common.h).
struct PanelWhere { int rb, pass, p; };
__device__ __forceinline__ PanelWhere panel_where(int nbx, int passes)
{
    PanelWhere w;
    const unsigned id = blockIdx.x, t = id / (unsigned)nbx;
    w.rb = (int)(id - t * (unsigned)nbx);
    w.p = (int)(t / (unsigned)passes);
    w.pass = (int)(t - (unsigned)w.p * (unsigned)passes);
    return w;
}
// (The predecessor — the same row block of the previous panel — has a lower workgroup id and is dispatched first; the hardware
// dispatches in order, which no specification promises.  Should that ever fail with every CU held by waiting workgroups the wait
// would never end: it is BOUNDED — 2^24 polls of ~130 cycles, about a second, three orders of magnitude beyond the longest
// workgroup — and then traps: the launch fails loudly (the next synchronisation reports it) instead of hanging the device.
// MXGPU_ROWSPLIT_LAUNCHES=1 runs one launch per panel, without any wait.)
__device__ __forceinline__ void panel_wait(const unsigned *word, int p)
{
    unsigned polls = 0;
    while ((int)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > (1u << 24)) __builtin_trap();
    }
    asm volatile("" ::: "memory");            // the reads of C stay behind the last poll
}
// every thread's stores of C have left, then one thread counts the workgroup in
__device__ __forceinline__ void panel_signal(unsigned *word)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename real_t, int VEC>
__device__ __forceinline__ void c_load(real_t (&v)[VEC], const real_t *p, bool agent)
{
    if (agent) {
#pragma unroll
        for (int q = 0; q < VEC; q++) v[q] = __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else vload<real_t, VEC>(v, p);
}
template <typename real_t, int VEC>
__device__ __forceinline__ void c_store(real_t *p, const real_t (&v)[VEC], bool agent)
{
    if (agent) {
#pragma unroll
        for (int q = 0; q < VEC; q++) __hip_atomic_store(p + q, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else vstore<real_t, VEC>(p, v);
}

// ---------------------------------------------------------------------------------------------------------------
// LONG ROWS (round 5, tools/cliff_hunt.py).  One wavefront per row makes the longest row the kernel's tail: m = 1e5, 64 per
// row, n = 64 ran 0.16 ms with equal rows, 0.40 ms with log-normal rows (sigma 1.5) or four rows of 10,000 entries.  When
// the caller knows that long rows exist (the matrix profile's longest / mean row, csrc/profile.hip), rows longer than `piece`
// entries are treated as EMPTY by the kernels below — the wavefront that meets one in panel 0 registers it in a list, one
// 64-bit atomic handing out the list slot and the row's range of pieces together — and are then summed by
// spmm_longrows_kernel, one wavefront per piece of `piece` consecutive entries into a scratch row, and
// spmm_longrows_combine_kernel, which adds a row's pieces IN ORDER and overwrites the row of C: the same bits on every run
// whatever the order of the list (a regrouping of the row's sum, like the segments).
static thread_local unsigned long long *g_longrows_last = nullptr;   // counter of the thread's last product (nullptr: path off)
struct LongRows {
    unsigned long long *counter;     // (slots << 32) | pieces handed out so far
    int *rows;                       // [slot] row
    unsigned *base;                  // [slot] first piece of the row
    int2 *piece_of;                  // [piece] (row, number of the piece inside the row)
    const int *fit;                  // written by longrows_fit_kernel before the product: 1 = this launch's long rows fit the scratch
    int piece;                       // 0 = off
};
// the piece length the kernels of one launch go by: 0 when the path is off or the launch's long rows would not fit the
// scratch that was sized from the caller's hint (a scalar load per wavefront; every wavefront of the launch sees the same)
__device__ __forceinline__ int long_piece_of(const LongRows &lr) { return lr.piece && *lr.fit ? lr.piece : 0; }

// Before the product kernels: counts the rows longer than `piece` and their pieces from indptr and decides — on the device, for
// every kernel of the launch alike — whether they fit the scratch (`cap_rows` list slots, `cap_pieces` partial-sum rows).
// The scratch is sized from what the matrix profile knows (LongHint), capped; a caller whose hint was for another matrix, or a
// matrix with more long rows than the cap holds, gets the plain kernels: slower, never wrong, nothing written out of bounds.
// head: [0] the registration counter (u64), [2] fit, [3] workgroups done, [4..5] needed rows / pieces (for the debug entry),
// [16 ..] one (rows, pieces) partial per workgroup
__global__ __launch_bounds__(256)
void longrows_fit_kernel(int m, const int32_t *__restrict__ indptr, int piece, unsigned cap_rows, unsigned cap_pieces, unsigned *head)
{
    unsigned rows = 0, pieces = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < m; r += gridDim.x * blockDim.x) {
        const int len = indptr[r + 1] - indptr[r];
        if (len > piece) { rows++; pieces += (unsigned)((len + piece - 1) / piece); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { rows += __shfl_xor(rows, o, 64); pieces += __shfl_xor(pieces, o, 64); }
    // One partial per WORKGROUP into its own two words (head[16 + 2 b], up to 256 workgroups), added up by the last one to finish:
    // same-address atomics are served one after the other at the memory side — the first version's two per wavefront of 256
    // workgroups cost ~20 us (cliff hunt: 0.131 -> 0.157 ms on a product with long rows), 32 workgroups with one pair each were
    // too few threads for a million row pointers (0.137 -> 0.18 ms).  What remains is one counter increment per workgroup.
    __shared__ unsigned s_rows[4], s_pieces[4];
    __shared__ bool last;
    if (lane_id() == 0) { s_rows[threadIdx.x >> 6] = rows; s_pieces[threadIdx.x >> 6] = pieces; }
    __syncthreads();
    if (threadIdx.x == 0) {
        head[16 + 2 * blockIdx.x] = s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3];
        head[17 + 2 * blockIdx.x] = s_pieces[0] + s_pieces[1] + s_pieces[2] + s_pieces[3];
        __threadfence();
        last = atomicAdd(&head[3], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    unsigned nr = 0, np = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
        nr += __hip_atomic_load(&head[16 + 2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        np += __hip_atomic_load(&head[17 + 2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nr += __shfl_xor(nr, o, 64); np += __shfl_xor(np, o, 64); }
    if (lane_id() == 0) { s_rows[threadIdx.x >> 6] = nr; s_pieces[threadIdx.x >> 6] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        nr = s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3]; np = s_pieces[0] + s_pieces[1] + s_pieces[2] + s_pieces[3];
        head[4] = nr; head[5] = np;
        head[2] = nr <= cap_rows && np <= cap_pieces ? 1u : 0u;
    }
}
// wave-uniform `row` and `len` (the WHOLE row, not a panel's part of it): true = the row goes to the long-rows kernels
__device__ __forceinline__ bool long_row_divert(const LongRows &lr, int row, int len, bool registers)
{
    const int piece = long_piece_of(lr);
    if (piece == 0 || len <= piece) return false;
    if (registers) {
        const int np = (len + piece - 1) / piece;
        unsigned long long old = 0;
        if (lane_id() == 0) {
            old = atomicAdd(lr.counter, (1ULL << 32) | (unsigned long long)np);
            lr.rows[old >> 32] = row;
            lr.base[old >> 32] = (unsigned)old;
        }
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old);
        for (int k = lane_id(); k < np; k += MX_WAVE) lr.piece_of[base + k] = make_int2(row, k);
    }
    return true;
}

// cursors: (P + 1) x m per-row entry bounds of the column panels (rowsplit_cursors_kernel), nullptr = one panel, the whole
// row; done: one counter per (pass, row block), zeroed by the cursor kernel; P: the panels THIS launch runs, from p0 on
template <typename real_t, int VEC, int G, bool COLMAJOR>
__global__ __launch_bounds__(RS_WAVES * MX_WAVE)
void spmm_rowsplit_kernel(int m, int n, int S, int nbx, int passes, int P, int p0,
                          const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                          const double *__restrict__ values,
                          const int32_t *__restrict__ cursors, unsigned *__restrict__ done,
                          const real_t *__restrict__ B, size_t ldb,
                          real_t *__restrict__ C, size_t ldc, LongRows lr)
{
    constexpr int W = G * VEC;                // output columns per pass
    constexpr int LS = W + 1;                 // odd LDS row stride (elements)
    __shared__ real_t tile[RS_WAVES * LS];

    PanelWhere at = panel_where(nbx, passes);
    at.p += p0;                               // (p0 > 0: one launch per panel, the A/B form — then P = 1 here)
    const int accumulate = at.p > 0;
    const bool handover = P > 1;              // C goes through agent-scope accesses (the last panel's stores excepted)
    const int32_t *lo = cursors ? cursors + (size_t)at.p * m : nullptr;
    const int32_t *hi = cursors ? cursors + (size_t)(at.p + 1) * m : nullptr;
    unsigned *word = done + (size_t)at.pass * nbx + at.rb;

    const int lane = lane_id();
    const int wave = uniform(threadIdx.x / MX_WAVE);
    const int RW = RS_WAVES / S;              // rows per workgroup
    const int row0 = at.rb * RW;
    const int row = row0 + wave / S;
    const int seg = wave % S;
    const int c0 = at.pass * W;
    const int lg = lane % G;
    const int col = c0 + lg * VEC;
    const bool active = col < n;
    const unsigned lcol = active ? (unsigned)col : (unsigned)(n - VEC);     // clamped, valid column for idle lanes
    const bool direct = !COLMAJOR && S == 1;  // one wavefront = one whole row of row-major C: no tile

    real_t acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0;

    if (row < m) {
        const int s = uniform(lo ? lo[row] : indptr[row]);
        int e = uniform(hi ? hi[row] : indptr[row + 1]);
        if (lr.piece) {                        // a long row is left to spmm_longrows_kernel: empty here, in every panel
            const int len = lo ? uniform(indptr[row + 1]) - uniform(indptr[row]) : e - s;
            if (long_row_divert(lr, row, len, at.p == 0 && at.pass == 0 && seg == 0)) e = s;
        }
        int a = s, b = e;
        if (S > 1) {                           // whole 64-entry chunks per segment; the row's tail goes to the last busy one
            const int L = (((e - s + S - 1) / S + MX_WAVE - 1) / MX_WAVE) * MX_WAVE;
            a = min(e, s + seg * L);
            b = min(e, a + L);
        }
        int jv = 0;
        double av = 0.0;
        if (a + lane < b) { jv = indices[a + lane]; av = values[a + lane]; }
        // the storage-order chain goes on from what the earlier panels left in C
        if (direct && accumulate) {
            if (handover) panel_wait(word, at.p);
            if (active && lane < G) c_load<real_t, VEC>(acc, C + (size_t)row * ldc + col, true);
        }
        for (int k0 = a; k0 < b; k0 += MX_WAVE) {
            int jn = 0;
            double an = 0.0;                    // the next chunk's (j, a) are in flight while this one streams B
            if (k0 + MX_WAVE + lane < b) { jn = indices[k0 + MX_WAVE + lane]; an = values[k0 + MX_WAVE + lane]; }
            if constexpr (G == MX_WAVE) rs_chunk<real_t, VEC>(min(MX_WAVE, b - k0), jv, av, B, ldb, lcol, acc);
            else rs_chunk_groups<real_t, VEC, G>(min(MX_WAVE, b - k0), jv, av, B, ldb, lcol, lane / G, acc);
            jv = jn; av = an;
        }
    }
    if constexpr (G < MX_WAVE) {               // the groups' partial sums: butterfly over the group index
        // (with `accumulate` only group 0 started from C's value, the others from zero)
#pragma unroll
        for (int off = G; off < MX_WAVE; off <<= 1) {
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] += __shfl_xor(acc[v], off, MX_WAVE);
        }
    }

    const bool publish = handover && at.p + 1 < P;      // a later panel reads what this one stores
    if (direct) {
        if (row < m && active && lane < G) c_store<real_t, VEC>(C + (size_t)row * ldc + col, acc, publish);
        if (publish) panel_signal(word);
        return;
    }
    if (lane < G) {
        real_t *t = tile + wave * LS + lg * VEC;
#pragma unroll
        for (int v = 0; v < VEC; v++) t[v] = acc[v];
    }
    __syncthreads();
    if (accumulate && handover) panel_wait(word, at.p);  // (the gather above ran beside the earlier panel's)
    const int ncols = min(W, n - c0);
    const int total = RW * ncols;
    for (int idx = threadIdx.x; idx < total; idx += RS_WAVES * MX_WAVE) {
        // column-major: consecutive threads take consecutive rows of one column; row-major: consecutive columns of one row
        const int r = COLMAJOR ? idx % RW : idx / ncols;
        const int c = COLMAJOR ? idx / RW : idx % ncols;
        if (row0 + r >= m) continue;
        real_t *dst = COLMAJOR ? C + (size_t)(c0 + c) * ldc + row0 + r : C + (size_t)(row0 + r) * ldc + c0 + c;
        real_t sum = tile[(r * S) * LS + c];
        if (accumulate) sum = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + sum;
        for (int q = 1; q < S; q++) sum += tile[(r * S + q) * LS + c];       // segments in order
        if (publish) __hip_atomic_store(dst, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = sum;
    }
    if (publish) panel_signal(word);
}

// ---------------------------------------------------------------------------------------------------------------
// ROW GROUPS: the form for MANY SHORT rows against a narrow B (m = 1e6, 8-32 entries per row, n = 16-64: sparse features
// times a small weight matrix).  One wavefront per row is then 1e6 wavefronts of a few instructions each, most lanes idle
// in the (j, a) load and a butterfly at the end (0.28-0.36 ms where the bytes ask for 0.10-0.15).  Here the G lanes that own
// a row of B also own ONE ROW OF A: a wavefront carries 64 / G rows at once, group g walks its own row G entries at a time
// (lane l of the group loads entry k0 + l, every entry is handed round the group by ds_bpermute), one load instruction
// reads one row of B for each of the 64 / G rows, U of them in flight.  A row's terms are added by its own group in
// storage order: bit for bit the reference's FMA chain (src/matmul.cpp:118-142), column panels included for row-major C.
// Row-major C: each group stores its G * 16 bytes; column-major C: the workgroup's 8 * 64 / G rows go through an LDS tile
// and leave as column segments of that many consecutive rows.
template <typename real_t, int VEC, int G, bool COLMAJOR>
__global__ __launch_bounds__(RS_WAVES * MX_WAVE)
void spmm_rowgroup_kernel(int m, int n, int nbx, int passes, int P, int p0,
                          const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                          const double *__restrict__ values,
                          const int32_t *__restrict__ cursors, unsigned *__restrict__ done,
                          const real_t *__restrict__ B, size_t ldb,
                          real_t *__restrict__ C, size_t ldc, LongRows lr)
{
    static_assert(G < MX_WAVE, "one row per wavefront is the row-split form");
    constexpr int NG = MX_WAVE / G;           // rows per wavefront
    constexpr int RW = RS_WAVES * NG;         // rows per workgroup
    constexpr int W = G * VEC;                // output columns per pass
    constexpr int LS = W + 1;
    constexpr int U = G < 8 ? G : 8;          // B-row reads in flight per lane
    __shared__ real_t tile[COLMAJOR ? RW * LS : 1];

    PanelWhere at = panel_where(nbx, passes);                       // (column panels in one launch: see spmm_rowsplit_kernel)
    at.p += p0;
    const int accumulate = at.p > 0;
    const bool handover = P > 1, publish = handover && at.p + 1 < P;
    const int32_t *lo = cursors ? cursors + (size_t)at.p * m : nullptr;
    const int32_t *hi = cursors ? cursors + (size_t)(at.p + 1) * m : nullptr;
    unsigned *word = done + (size_t)at.pass * nbx + at.rb;

    const int lane = lane_id();
    const int wave = uniform(threadIdx.x / MX_WAVE);
    const int g = lane / G, lg = lane % G, gbase = g * G;
    const int row0 = at.rb * RW;
    const int row = row0 + wave * NG + g;
    const int c0 = at.pass * W;
    const int col = c0 + lg * VEC;
    const bool active = col < n;
    const unsigned lcol = active ? (unsigned)col : (unsigned)(n - VEC);

    real_t acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0;
    int s = 0, e = 0;
    if (row < m) {
        s = lo ? lo[row] : indptr[row];
        e = hi ? hi[row] : indptr[row + 1];
        if (const int piece = long_piece_of(lr)) {     // (long rows: see spmm_rowsplit_kernel; here per lane group)
            const int len = lo ? indptr[row + 1] - indptr[row] : e - s;
            if (len > piece) {
                if (at.p == 0 && at.pass == 0 && lg == 0) {
                    const unsigned np = (unsigned)((len + piece - 1) / piece);
                    const unsigned long long old = atomicAdd(lr.counter, (1ULL << 32) | (unsigned long long)np);
                    lr.rows[old >> 32] = row;
                    lr.base[old >> 32] = (unsigned)old;
                    for (unsigned k = 0; k < np; k++) lr.piece_of[(unsigned)old + k] = make_int2(row, (int)k);
                }
                e = s;
            }
        }
    }
    if (!COLMAJOR && accumulate) {
        if (handover) panel_wait(word, at.p);
        if (row < m && active) c_load<real_t, VEC>(acc, C + (size_t)row * ldc + col, true);
    }
    // the wavefront runs as long as its longest row (wave-uniform trip count, full EXEC for the permutes); a group whose
    // row has ended re-reads row 0 of B (valid memory, in cache) and drops the products by a select
    int len = e - s;
#pragma unroll
    for (int off = G; off < MX_WAVE; off <<= 1) len = max(len, __shfl_xor(len, off, MX_WAVE));
    const int longest = uniform(__shfl(len, 0, MX_WAVE));
    const int trips = (longest + G - 1) / G;
    // (the f32 kind narrows a value once, where it is loaded — matmul.cpp:53-57 narrows per entry: the same number — and
    // hands 4 bytes round the group instead of 8)
    int jv = 0;
    real_t av = 0;
    if (s + lg < e) { jv = indices[s + lg]; av = (real_t)values[s + lg]; }
    for (int t = 0; t < trips; t++) {
        const int k0 = s + t * G;
        int jn = 0;
        real_t an = 0;                          // the next G entries of the row are in flight while these stream B
        if (k0 + G + lg < e) { jn = indices[k0 + G + lg]; an = (real_t)values[k0 + G + lg]; }
        const int cnt = min(G, max(e - k0, 0));
        const int most = longest - t * G;       // wave-uniform: entries the longest row still has in this trip
#pragma unroll
        for (int q = 0; q < G; q += U) {
            if (q >= most) break;               // (rows shorter than the group: no batch of reads that nobody needs)
            int jj[U];
            real_t bb[U][VEC];
#pragma unroll
            for (int u = 0; u < U; u++) jj[u] = __builtin_amdgcn_ds_bpermute((gbase + (q + u < cnt ? q + u : 0)) << 2, jv);
#pragma unroll
            for (int u = 0; u < U; u++) vload<real_t, VEC>(bb[u], B + (size_t)jj[u] * ldb + lcol);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const real_t x = bperm_real(av, gbase + (q + u < cnt ? q + u : 0));
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v] = q + u < cnt ? mx_fma(x, bb[u][v], acc[v]) : acc[v];
            }
        }
        jv = jn; av = an;
    }

    if constexpr (!COLMAJOR) {
        if (row < m && active) c_store<real_t, VEC>(C + (size_t)row * ldc + col, acc, publish);
    } else {
        real_t *tl = tile + (wave * NG + g) * LS + lg * VEC;
#pragma unroll
        for (int v = 0; v < VEC; v++) tl[v] = acc[v];
        __syncthreads();
        if (accumulate && handover) panel_wait(word, at.p);
        const int ncols = min(W, n - c0);
        const int total = RW * ncols;
        for (int idx = threadIdx.x; idx < total; idx += RS_WAVES * MX_WAVE) {
            const int r = idx % RW, c = idx / RW;               // consecutive threads: consecutive rows of one column
            if (row0 + r >= m) continue;
            real_t *dst = C + (size_t)(c0 + c) * ldc + row0 + r;
            real_t sum = tile[r * LS + c];
            if (accumulate) sum = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + sum;
            if (publish) __hip_atomic_store(dst, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = sum;
        }
    }
    if (publish) panel_signal(word);
}

// cursors[p * m + row], p = 0 .. P: where panel p's entries of `row` begin (panel p = columns [p * panel_cols, (p + 1) *
// panel_cols)).  One wavefront per row: is the row sorted by column? (the same test as check_is_sorted, src/misc.cpp:118-128)
// — then lanes 1 .. P - 1 each find one bound by binary search; otherwise panel 0 takes the whole row.
__global__ __launch_bounds__(RS_WAVES * MX_WAVE)
void rowsplit_cursors_kernel(int m, int P, int panel_cols, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                             int32_t *__restrict__ cursors, unsigned *__restrict__ done, int ndone)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ndone; i += gridDim.x * blockDim.x) done[i] = 0u;
    const int lane = lane_id();
    const int row = blockIdx.x * RS_WAVES + uniform(threadIdx.x / MX_WAVE);
    if (row >= m) return;
    const int s = uniform(indptr[row]), e = uniform(indptr[row + 1]);
    bool bad = false;
    for (int k = s + lane; k + 1 < e; k += MX_WAVE) bad |= indices[k] > indices[k + 1];
    const bool sorted = __ballot(bad) == 0;
    for (int p = lane; p <= P; p += MX_WAVE) {
        int at = e;
        if (p == 0) at = s;
        else if (p < P && sorted) at = s + lower_bound_dev(indices + s, e - s, p * panel_cols);
        cursors[(size_t)p * m + row] = at;
    }
}

// One wavefront per PIECE of a long row (see LongRows): the sums of entries [s + k * piece, s + (k + 1) * piece) x B, all the
// columns of one pass, into scratch row E[piece index]; the piece -> (row, k) table was written when the row was registered
// (a binary search over the slots' first pieces instead cost 14 dependent loads per piece: more than the piece itself).
// Persistent grid: the number of pieces is only known on the device.
template <typename real_t, int VEC, int G>
__global__ __launch_bounds__(RS_WAVES * MX_WAVE)
void spmm_longrows_kernel(int n, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                          const double *__restrict__ values, const real_t *__restrict__ B, size_t ldb, real_t *__restrict__ E,
                          LongRows lr)
{
    constexpr int W = G * VEC;
    const unsigned total = (unsigned)*lr.counter;
    const int lane = lane_id(), lg = lane % G;
    const int col = blockIdx.y * W + lg * VEC;
    const bool active = col < n;
    const unsigned lcol = active ? (unsigned)col : (unsigned)(n - VEC);
    const unsigned wave0 = blockIdx.x * RS_WAVES + uniform(threadIdx.x / MX_WAVE), nwaves = gridDim.x * RS_WAVES;
    for (unsigned w = wave0; w < total; w += nwaves) {
        const int2 rk = lr.piece_of[w];
        const int row = uniform(rk.x), k = uniform(rk.y);
        const int s = uniform(indptr[row]), e = uniform(indptr[row + 1]);
        const int a = s + (int)k * lr.piece, b = min(e, a + lr.piece);
        real_t acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = 0;
        int jv = 0;
        double av = 0.0;
        if (a + lane < b) { jv = indices[a + lane]; av = values[a + lane]; }
        for (int k0 = a; k0 < b; k0 += MX_WAVE) {
            int jn = 0;
            double an = 0.0;
            if (k0 + MX_WAVE + lane < b) { jn = indices[k0 + MX_WAVE + lane]; an = values[k0 + MX_WAVE + lane]; }
            if constexpr (G == MX_WAVE) rs_chunk<real_t, VEC>(min(MX_WAVE, b - k0), jv, av, B, ldb, lcol, acc);
            else rs_chunk_groups<real_t, VEC, G>(min(MX_WAVE, b - k0), jv, av, B, ldb, lcol, lane / G, acc);
            jv = jn; av = an;
        }
        if constexpr (G < MX_WAVE) {
#pragma unroll
            for (int off = G; off < MX_WAVE; off <<= 1) {
#pragma unroll
                for (int v = 0; v < VEC; v++) acc[v] += __shfl_xor(acc[v], off, MX_WAVE);
            }
        }
        if (active && lane < G) {
#pragma unroll
            for (int v = 0; v < VEC; v++) E[(size_t)w * n + col + v] = acc[v];
        }
    }
}

// One wavefront per long row: its pieces added in order, the row of C overwritten (the product kernels left zeros there)
template <typename real_t, bool COLMAJOR>
__global__ __launch_bounds__(RS_WAVES * MX_WAVE)
void spmm_longrows_combine_kernel(int n, const int32_t *__restrict__ indptr, const real_t *__restrict__ E, real_t *__restrict__ C,
                                  size_t ldc, LongRows lr)
{
    const int nslots = (int)(*lr.counter >> 32);
    const int lane = lane_id();
    for (int i = blockIdx.x * RS_WAVES + uniform(threadIdx.x / MX_WAVE); i < nslots; i += gridDim.x * RS_WAVES) {
        const int row = uniform(lr.rows[i]);
        const unsigned base = (unsigned)uniform((int)lr.base[i]);
        const int len = uniform(indptr[row + 1]) - uniform(indptr[row]);
        const int np = (len + lr.piece - 1) / lr.piece;
        for (int c = lane; c < n; c += MX_WAVE) {
            real_t sum = E[(size_t)base * n + c];
            for (int k = 1; k < np; k++) sum += E[((size_t)base + k) * n + c];
            C[COLMAJOR ? (size_t)c * ldc + row : (size_t)row * ldc + c] = sum;
        }
    }
}

// S = 0: the row-group form (several rows per wavefront, G < 64); the launch runs panels p0 .. p0 + P - 1
template <typename real_t, int VEC, int G, bool COLMAJOR>
static void launch_one(int m, int n, int S, int nbx, int passes, int P, int p0, const int32_t *indptr, const int32_t *indices,
                       const double *values, const int32_t *cursors, unsigned *done, const real_t *B, size_t ldb, real_t *C, size_t ldc,
                       const LongRows &lr, hipStream_t stream)
{
    const dim3 grid((unsigned)((size_t)nbx * passes * P));
    if constexpr (G < MX_WAVE) {
        if (S == 0) {
            set_last_spmm_kernel("spmm_rowgroup_kernel");            // (named where the form is final: after the alignment fallback)
            hipLaunchKernelGGL((spmm_rowgroup_kernel<real_t, VEC, G, COLMAJOR>), grid, dim3(RS_WAVES * MX_WAVE), 0, stream,
                               m, n, nbx, passes, P, p0, indptr, indices, values, cursors, done, B, ldb, C, ldc, lr);
            return;
        }
    }
    set_last_spmm_kernel("spmm_rowsplit_kernel");
    hipLaunchKernelGGL((spmm_rowsplit_kernel<real_t, VEC, G, COLMAJOR>), grid, dim3(RS_WAVES * MX_WAVE), 0, stream,
                       m, n, S, nbx, passes, P, p0, indptr, indices, values, cursors, done, B, ldb, C, ldc, lr);
}

// One launch for all the panels, or one per panel (rounds 3-4)?  Measured (tools/rowsplit_fused_probe.py, ms, one launch /
// P launches): where the sums of a panel go through the LDS tile (column-major C, segments) the wait for the earlier panel
// comes AFTER the gather and one launch wins everywhere tried (m = 3e3, 6 panels 0.298 / 0.323; m = 1e5, 4 panels 1.662 /
// 1.711; m = 1e6 row groups 1.008 / 1.029).  The storage-order chain of row-major C must START from C's value: poll, then
// the read of C, then the gather — two serial round trips per workgroup, and C through 4- / 8-byte agent-scope accesses.
// That still wins while a panel is a few rounds of the machine and the launch gaps count (m = 1e4: 3 panels 0.196 / 0.200,
// 15 panels 0.313 / 0.366) and loses when a panel is many rounds of small workgroups (m = 1e5: 0.628 / 0.555, f32 n = 256
// 2.09 / 1.53; m = 1e6 row groups 1.16 / 1.03): there one launch per panel stays.
// MXGPU_ROWSPLIT_LAUNCHES=1 / =0 forces one launch per panel / one launch (the A/B and the tests of both forms).
static bool rowsplit_one_launch_per_panel(bool through_tile, long long workgroups_per_panel)
{
    const char *e = getenv("MXGPU_ROWSPLIT_LAUNCHES");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return !through_tile && workgroups_per_panel > 4096;
}

template <typename real_t, int VEC, int G, bool COLMAJOR>
static int launch_rowsplit(int m, int n, int K, int S, int P, const int32_t *indptr, const int32_t *indices, const double *values,
                           const real_t *B, size_t ldb, real_t *C, size_t ldc, LongHint lh, long long nnz, hipStream_t stream)
{
    constexpr int W = G * VEC;
    if (S == 0 && G == MX_WAVE) S = 1;                      // rows of B that fill the wavefront: one row per wavefront anyway
    const int nbx = (int)ceil_div(m, S == 0 ? RS_WAVES * (MX_WAVE / G) : RS_WAVES / S), passes = (int)ceil_div(n, W);
    if ((long long)nbx * passes * P >= (1LL << 31)) return set_error("rowsplit_spmm: %d x %d x %d workgroups exceed the grid", nbx, passes, P);
    // LONG ROWS: the list and the pieces' scratch rows.  Sized by what the matrix profile knows (LongHint: the rows longer than
    // the canonical piece and their entries — at most that many rows are longer than THIS piece, with at most entries / piece
    // + one piece per row), else by what nnz allows (at most nnz / piece long rows); capped at LONGROWS_CAP_BYTES either way.
    // Whether the launch's long rows really fit is decided on the device (longrows_fit_kernel): the hint only sizes.  The
    // piece length is the caller's — fixed once per product — and never changes here (round 5 doubled it when the scratch
    // bound passed 1 GiB, which made the grouping of a long row's sum depend on a block's nnz and n).
    constexpr size_t LONGROWS_CAP_BYTES = (size_t)256 << 20;
    const int long_piece = lh.piece > 0 ? lh.piece : 0;
    LongRows lr = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    real_t *E = nullptr;
    g_longrows_last = nullptr;
    char *buf = nullptr;
    size_t head = 2560, rows_b = 0, po_b = 0, slots = 0, pieces = 0;       // (256 bytes of counters + 256 x 2 partials of the fit kernel)
    if (long_piece > 0 && nnz > long_piece) {
        slots = (size_t)(nnz / long_piece) + 1;
        pieces = 2 * slots;
        if (lh.rows >= 0 && lh.pieces >= 0) {
            // (float counts: 24 bits of mantissa — a margin of 2^-20 of the value and 8 on top)
            slots = std::min(slots, (size_t)((double)lh.rows * (1.0 + 1e-6)) + 8);
            pieces = std::min(pieces, (size_t)((double)lh.pieces * (1.0 + 1e-6)) + slots + 8);
        }
        const size_t per_piece = (size_t)n * sizeof(real_t) + 8, per_slot = 8;
        if (slots * per_slot + pieces * per_piece > LONGROWS_CAP_BYTES) {
            pieces = LONGROWS_CAP_BYTES / (per_piece + per_slot);
            slots = std::min(slots, pieces);
        }
        slots = std::min<size_t>(slots, 0x7fffffffu); pieces = std::min<size_t>(pieces, 0x7fffffffu);
        rows_b = (slots * 4 + 255) & ~(size_t)255;
        po_b = (pieces * 8 + 255) & ~(size_t)255;
        buf = (char *)scratch_buffer(MX_SCRATCH_LONGROWS, head + 2 * rows_b + po_b + pieces * (size_t)n * sizeof(real_t));
        if (!buf) (void)hipGetLastError();
    }
    if (buf) {
        scratch_acquire(MX_SCRATCH_LONGROWS, stream);
        lr.counter = (unsigned long long *)buf;
        lr.fit = (const int *)(buf + 8);
        lr.rows = (int *)(buf + head);
        lr.base = (unsigned *)(buf + head + rows_b);
        lr.piece_of = (int2 *)(buf + head + 2 * rows_b);
        lr.piece = long_piece;
        E = (real_t *)(buf + head + 2 * rows_b + po_b);
        MX_HIP(hipMemsetAsync(buf, 0, 32, stream));
        hipLaunchKernelGGL(longrows_fit_kernel, dim3((unsigned)std::min<long long>(256, ceil_div(m, 1024))), dim3(256), 0, stream,
                           m, indptr, long_piece, (unsigned)slots, (unsigned)pieces, (unsigned *)buf);
        g_longrows_last = lr.counter;
    }
    // after the product kernels: the pieces, then the rows (both persistent: the counts live on the device)
    auto long_rows = [&]() {
        if (!lr.piece) return;
        const unsigned blocks = (unsigned)std::min<size_t>(2048, (size_t)(nnz / long_piece) / RS_WAVES + 1);
        hipLaunchKernelGGL((spmm_longrows_kernel<real_t, VEC, G>), dim3(blocks, (unsigned)passes), dim3(RS_WAVES * MX_WAVE), 0, stream,
                           n, indptr, indices, values, B, ldb, E, lr);
        hipLaunchKernelGGL((spmm_longrows_combine_kernel<real_t, COLMAJOR>), dim3(std::min(blocks, 512u)), dim3(RS_WAVES * MX_WAVE), 0, stream,
                           n, indptr, E, C, ldc, lr);
        scratch_done(MX_SCRATCH_LONGROWS, stream);
    };
    if (P <= 1) {
        kt_begin(stream);
        launch_one<real_t, VEC, G, COLMAJOR>(m, n, S, nbx, passes, 1, 0, indptr, indices, values, nullptr, nullptr, B, ldb, C, ldc, lr, stream);
        long_rows();
        kt_end(stream);
        MX_LAUNCH_CHECK();
        return 0;
    }
    // grow-only per-thread scratch; a caller that comes back on another stream waits for the previous product (scratch_acquire)
    const size_t cur_bytes = ((size_t)(P + 1) * (size_t)m * sizeof(int32_t) + 255) & ~(size_t)255;
    const int ndone = nbx * passes;
    int32_t *cur = (int32_t *)scratch_buffer(MX_SCRATCH_ROWSPLIT, cur_bytes + (size_t)ndone * sizeof(unsigned));
    if (!cur) return set_error("rowsplit_spmm: cannot allocate %zu bytes of panel cursors", cur_bytes + (size_t)ndone * sizeof(unsigned));
    unsigned *done = (unsigned *)((char *)cur + cur_bytes);
    scratch_acquire(MX_SCRATCH_ROWSPLIT, stream);
    const int panel_cols = (int)ceil_div(K, P);
    kt_begin(stream);
    hipLaunchKernelGGL(rowsplit_cursors_kernel, dim3((unsigned)ceil_div(m, RS_WAVES)), dim3(RS_WAVES * MX_WAVE), 0, stream,
                       m, P, panel_cols, indptr, indices, cur, done, ndone);
    if (rowsplit_one_launch_per_panel(COLMAJOR || S > 1, (long long)nbx * passes)) {
        for (int p = 0; p < P; p++)
            launch_one<real_t, VEC, G, COLMAJOR>(m, n, S, nbx, passes, 1, p, indptr, indices, values, cur, done, B, ldb, C, ldc, lr, stream);
    } else {
        launch_one<real_t, VEC, G, COLMAJOR>(m, n, S, nbx, passes, P, 0, indptr, indices, values, cur, done, B, ldb, C, ldc, lr, stream);
    }
    scratch_done(MX_SCRATCH_ROWSPLIT, stream);
    long_rows();
    kt_end(stream);
    MX_LAUNCH_CHECK();
    return 0;
}

template <typename real_t, int VEC, bool COLMAJOR>
static int pick_group_rowsplit(int m, int n, int K, int S, int P, const int32_t *indptr, const int32_t *indices, const double *values,
                               const real_t *B, size_t ldb, real_t *C, size_t ldc, LongHint lh, long long nnz, hipStream_t st)
{
    if constexpr (VEC > 1) {
        if (n <= 8 * VEC) return launch_rowsplit<real_t, VEC, 8, COLMAJOR>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, st);
        if (n <= 16 * VEC) return launch_rowsplit<real_t, VEC, 16, COLMAJOR>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, st);
        if (n <= 32 * VEC) return launch_rowsplit<real_t, VEC, 32, COLMAJOR>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, st);
    }
    return launch_rowsplit<real_t, VEC, 64, COLMAJOR>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, st);
}

// Column panels (tools/auto_map.py, profiles/r04_auto_map.json): none while an XCD's L2 still holds most of B (5 MB: 80 %
// of the reads hit, and three launches of a third of each row were SLOWER: 0.156 vs 0.122 ms at m = 1e4, 500 entries per
// row, n = 64); from 7 MB on, ~3 MB of B per panel (vignette shape, 8 MB: 3 panels 0.182 ms, 4: 0.189, 1: 0.193) as long as a row keeps
// enough entries per panel to pay for a wavefront of its own: 32 when 32 or 64 lanes own a row of B (m = 1e5, K = 1e4,
// 128 per row, n = 128: 4 panels 0.576 ms, one 0.971), 64 / 96 with 16- / 8-lane groups, whose load instruction covers 4 / 8
// entries (n = 16, 32 per panel: 2.36 vs 1.75 ms at m = 1e6)
static int rowsplit_panels_by_size(int m, int n, int K, int dense_bytes, double avg_len)
{
    const double b_bytes = (double)K * n * dense_bytes;
    if (b_bytes < 7e6) return 1;
    // a few thousand rows are ONE round of the machine: wavefronts that start together walk their (sorted) rows together
    // and share the lines they gather without any panel (m = 500, 2000 entries per row: one panel 0.042 ms, three 0.061)
    if ((long long)m * ((n * dense_bytes + 1023) / 1024) < 4096) return 1;
    int P = (int)((b_bytes + 3e6 - 1) / 3e6);
    const int row_bytes = n * dense_bytes;
    const double per_panel = row_bytes <= 128 ? 96.0 : (row_bytes <= 256 ? 64.0 : 32.0);
    const int by_len = (int)(avg_len / per_panel);
    if (P > by_len) P = by_len;
    if (P > RS_MAX_PANELS) P = RS_MAX_PANELS;
    return P < 1 ? 1 : P;
}

// What the kernel costs with P column panels, microseconds (the model AUTO compares with the planned sweep's, csrc/spmm.hip;
// constants fitted to tools/auto_map.py's map, profiles/r04_auto_map.json): the gather of nnz * n * s bytes at the L2 -> L1
// rate of the lane-group width for the share of a panel an XCD's L2 holds and at the Infinity Cache's for the rest, a cost
// per (row, panel, pass) wavefront, ~6 us per launch, and — when C outgrows the Infinity Cache — the re-read and re-write
// of C by every launch after the first.
double rowsplit_est_us(int m, int n, int K, int dense_bytes, double avg_len, int P)
{
    const int sz = dense_bytes, vec = 16 / sz;
    const double nnz = avg_len * m, c_bytes = (double)m * n * sz;
    const double passes = (double)((n + 64 * vec - 1) / (64 * vec));
    const int G = n <= 8 * vec ? 8 : (n <= 16 * vec ? 16 : (n <= 32 * vec ? 32 : 64));
    // TB/s: 8-lane groups read one line per row of B; wider groups 28 — of the lanes that HAVE a column: n = 100 leaves 7 of
    // 32 (f32) / 14 of 64 (f64) lanes of every read idle, and the useful rate is 28 x 0.78 = 21.9 (measured: vignette shape
    // f32 2 GB in 0.107 ms, f64 4 GB in 0.19 ms; n = 64 in f64 fills its 32 lanes: 3.3 GB in 0.164 ms)
    // (applied to the 16- / 32-lane groups only: with a full wavefront per row the round-4 map — m = 1e6, K = 1e4, 128 per row,
    // n = 100 — is priced better without it)
    const double lanes_used = G < 64 ? (double)n * sz / ((double)G * 16.0) : 1.0;
    double l2_rate = n * sz <= 128 ? 17.0 : 28.0 * (lanes_used < 1.0 ? lanes_used : 1.0), mall_rate = 8.5;
    double row_us = 0.2e-3 * passes;                                              // one wavefront per (row, panel, pass)
    double lockstep = 1.0;
    if (rowsplit_segments(m, n, sz, avg_len / P) == 0) {                          // the row-group form (tools/rowgroup_probe.py)
        l2_rate = G == 8 ? 18.0 : 21.0;
        mall_rate = 6.5;
        row_us = G == 8 ? 0.0 : (G == 16 ? 0.07e-3 : 0.15e-3);
        // 64 / G rows per wavefront, as long as the longest of them (real row lengths are skewed: tools/zipf_map.py — m = 1e6,
        // 9 per row, log-normal sigma .5, n = 16: 0.131 ms where equal rows take 0.061)
        lockstep = lockstep_factor(profile_cv(), 64 / G);
    } else if (profile_cv() > 0.15 && m < 50000) {
        // one wavefront per row (segment) and too few rows to fill the machine: the long rows are the launch's tail even with the
        // long-rows pieces — measured on log-normal rows (tools/cliff_hunt.py, round 5; sigma 1 / 1.5 = cv 1.1 / 1.8): 2.9-3.2x the
        // equal-rows rate at m = 1e4 with 500 per row.  (Without this term AUTO kept the dense-ish skewed products on this kernel —
        // 0.31 ms — when the tile kernel with dealt rows takes 0.175.  From 5e4 rows on the slowdown is 1.5x and the model was
        // fitted on skewed data there — real-sim's shape, tests/test_gpu_zipf.py — so nothing is added.)
        const double cv = profile_cv() < 1.5 ? profile_cv() : 1.5;
        lockstep = 1.0 + cv * 0.6;
    }
    // the hit rate: the mass of the entries whose row of B an XCD's L2 (4 MiB) holds — the hottest 4 MiB / (n s) rows of the
    // panel —, which for uniform columns is the share of the panel's bytes (round 4's term); hot columns sit anywhere, so P
    // panels each hold 1 / P of every popularity class: the mass of the P times as many hottest columns of the whole matrix
    const double hit = profile_mass(4.0 * 1048576.0 / ((double)n * sz) * P, K);
    const double rate = 1e6 / (hit / l2_rate + (1.0 - hit) / mall_rate) / lockstep;   // bytes per microsecond
    const double c_traffic_us = c_bytes > 128e6 ? (2.0 * P - 2.0) * c_bytes / 5e6 : 0.0;
    return nnz * n * sz / rate + row_us * m * P + 6.0 * P + 4.0 + c_traffic_us;
}
// the panel count the sizes suggest, or none at all when the model says that the launches cost more than the locality buys
// (m = 1e4, K = 1e5, 500 per row, n = 16: five panels 0.078 ms, one 0.059; n = 100: fifteen 0.350, one 0.491)
int rowsplit_panels(int m, int n, int K, int dense_bytes, double avg_len)
{
    const int P = rowsplit_panels_by_size(m, n, K, dense_bytes, avg_len);
    if (P <= 1) return 1;
    return rowsplit_est_us(m, n, K, dense_bytes, avg_len, P) < rowsplit_est_us(m, n, K, dense_bytes, avg_len, 1) ? P : 1;
}

// Segments per row: only when the rows alone cannot fill HALF the machine's wavefront slots (256 CUs x 32), and only while
// a segment keeps a full chunk of 64 entries.  Splitting costs the LDS combine and the direct store path: measured on the
// vignette's shape (m = 1e4, 500 per row, n = 100, one panel) S = 1 / 2 / 4 / 8 = 0.193 / 0.220 / 0.282 / 0.285 ms — one
// wavefront per row wins as soon as there are ~4k rows (tools/rowsplit_sweep.py; round 4's first rule split up to 32k rows)
//
// 0 = the ROW-GROUP form (several rows per wavefront) for many short rows against a narrow B — measured against one
// wavefront per row at m = 1e6, K = 1e4 (tools/rowgroup_probe.py, ms, f64): n = 16 (8 lanes per row), 8 / 16 / 32 / 48 / 64
// entries per row: 0.061 / 0.148 / 0.267 / 0.439 / 0.564 against 0.301 / 0.346 / 0.345 / 0.491 / 0.520; n = 32 (16 lanes),
// 8 / 32 / 48: 0.164 / 0.454 / 0.702 against 0.334 / 0.478 / 0.663; n = 64 (32 lanes), 8 / 16 / 32 / 64: 0.348 / 0.549 /
// 0.891 / 1.695 against 0.520 / 0.585 / 0.889 / 1.627 — i.e. up to 56 / 40 / 24 entries per row with 8 / 16 / 32 lanes,
// and only when the rows still fill the machine's wavefront slots at 64 / G rows per wavefront.
int rowsplit_segments(int m, int n, int dense_bytes, double avg_len)
{
    {
        const int vec = 16 / dense_bytes;
        const int G = n <= 8 * vec ? 8 : (n <= 16 * vec ? 16 : (n <= 32 * vec ? 32 : 64));
        const double limit = G == 8 ? 56.0 : (G == 16 ? 40.0 : 24.0);
        if (G < 64 && n % vec == 0 && avg_len <= limit && (long long)m * G >= 8192LL * 64) return 0;
    }
    const int W = 64 * (16 / dense_bytes);
    const long long passes = (n + W - 1) / W;
    int S = 1;
    while (S < RS_WAVES && (long long)m * S * passes < 3000 && avg_len / (2 * S) >= 64.0) S <<= 1;
    return S;
}

// The piece length of the long-rows path, 0 = off.  Only with a matrix profile in scope (DeviceCSR.profile(), the exports'
// host-side profile, AUTO's own pass), and only when the longest row is at least two pieces long: rows of even length never
// pay the two extra launches.  Measured (tools/longrow_sweep.py, ms, off / best piece): m = 1e5, 64 per row, n = 64: four
// rows of 10,000 entries 0.394 / 0.185 (256), log-normal sigma 1.5 0.331 / 0.211 (256-512); m = 2e5, 100 per row, n = 32:
// 1.514 / 0.550 (512) and 1.076 / 0.542 (512-1024); m = 1e4, 500 per row, n = 100: 0.380 / 0.244 (512), 0.413 / 0.305
// (1024); m = 1e6, 12 per row, n = 16 (row groups): 0.601 / 0.140, 0.572 / 0.277 (128).  Pieces of 128 entries cost more
// than they balance once the mean row is 64 or longer (0.92 ms where 512 gives 0.53): ~6 mean rows per piece, a power of
// two in [128, 1024].  Rows SORTED by length (longest first) are the one case that loses (0.164 -> 0.186, 0.479 -> 0.536):
// the launch order already is the longest-first schedule.
LongHint rowsplit_long_hint(int m, long long nnz)
{
    LongHint h;
    const double ratio = profile_longest_over_mean();
    if (ratio <= 0.0 || m <= 0 || nnz <= 0) return h;
    const double mean = (double)nnz / m;
    int piece = canonical_long_piece(mean);
    if (const char *e = getenv("MXGPU_LONG_PIECE")) piece = atoi(e);       // (tools/longrow_sweep.py)
    if (!(piece > 0 && ratio * mean >= 2.0 * piece)) return h;
    h.piece = piece;
    double entries = 0.0, rows = 0.0;
    // (the profile counted the rows longer than the canonical piece: a bound for any piece at least that long)
    if (piece >= canonical_long_piece(mean) && profile_long_rows(&entries, &rows)) { h.rows = (long long)rows; h.pieces = (long long)(entries / piece) + 1; }
    return h;
}

template <typename real_t>
int rowsplit_spmm(int m, int n, int K, int S, int P, const int32_t *indptr, const int32_t *indices, const double *values,
                  const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream, LongHint lh, long long nnz)
{
    constexpr int VECMAX = 16 / (int)sizeof(real_t);
    if (S != 0 && S != 1 && S != 2 && S != 4 && S != 8)
        return set_error("rowsplit_spmm: segments per row must be 1, 2, 4 or 8, or 0 for the row-group form (got %d)", S);
    if (P < 1 || P > RS_MAX_PANELS) return set_error("rowsplit_spmm: 1 .. %d column panels (got %d)", RS_MAX_PANELS, P);
    // widest per-lane access the operands allow (16 B when the rows of B are 16-B aligned); row-major C is read / written
    // by vector accesses only on the S = 1 path, which needs its rows aligned as well
    const bool b_vec = (n % VECMAX == 0) && (ldb % VECMAX == 0) && ((uintptr_t)B % 16 == 0);
    const bool c_vec = colmajor || S > 1 || ((ldc % VECMAX == 0) && ((uintptr_t)C % 16 == 0));
    if (S == 0 && !(b_vec && c_vec)) S = 1;                 // (the row-group form needs the 16-byte accesses)
    if (b_vec && c_vec)
        return colmajor ? pick_group_rowsplit<real_t, VECMAX, true>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, stream)
                        : pick_group_rowsplit<real_t, VECMAX, false>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, stream);
    return colmajor ? pick_group_rowsplit<real_t, 1, true>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, stream)
                    : pick_group_rowsplit<real_t, 1, false>(m, n, K, S, P, indptr, indices, values, B, ldb, C, ldc, lh, nnz, stream);
}
template int rowsplit_spmm<double>(int, int, int, int, int, const int32_t *, const int32_t *, const double *, const double *, size_t,
                                   double *, size_t, int, hipStream_t, LongHint, long long);
template int rowsplit_spmm<float>(int, int, int, int, int, const int32_t *, const int32_t *, const double *, const float *, size_t,
                                  float *, size_t, int, hipStream_t, LongHint, long long);

}  // namespace mx

// diagnostic: how many rows (and pieces) the calling thread's last row-split product on the current device handed to its
// long-rows path — 0 / 0 when the path was off.  Synchronises the device.
extern "C" int mxd_debug_rowsplit_long_rows(long long *rows, long long *pieces)
{
    MX_REQUIRE(rows && pieces, "mxd_debug_rowsplit_long_rows: null argument");
    *rows = *pieces = 0;
    if (!mx::g_longrows_last) return 0;
    unsigned long long cnt = 0;
    MX_HIP(hipDeviceSynchronize());
    MX_HIP(hipMemcpy(&cnt, mx::g_longrows_last, sizeof(cnt), hipMemcpyDeviceToHost));
    *rows = (long long)(cnt >> 32);
    *pieces = (long long)(cnt & 0xffffffffULL);
    return 0;
}
// ... and what longrows_fit_kernel found before that product: the rows longer than the piece, their pieces, and whether they
// fitted the scratch (0: the product kernels summed them in line).  Synchronises the device.
extern "C" int mxd_debug_rowsplit_long_fit(long long *rows_needed, long long *pieces_needed, int *fit)
{
    MX_REQUIRE(rows_needed && pieces_needed && fit, "mxd_debug_rowsplit_long_fit: null argument");
    *rows_needed = *pieces_needed = 0; *fit = 0;
    if (!mx::g_longrows_last) return 0;
    unsigned head[8] = {0};
    MX_HIP(hipDeviceSynchronize());
    MX_HIP(hipMemcpy(head, mx::g_longrows_last, sizeof(head), hipMemcpyDeviceToHost));
    *rows_needed = head[4]; *pieces_needed = head[5]; *fit = (int)head[2];
    return 0;
}
