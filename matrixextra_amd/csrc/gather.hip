// gather.hip — CSR row gather (X[rows, ]), index-vector classification, and
// the per-row sort precondition, for gfx950.
//
// Replaces:
//   copy_csr_rows_template   src/slice.cpp:225-274   (serial size pass + std::copy per row)
//   check_is_seq / _rev_seq  src/slice.cpp:25-47
//   check_is_sorted + sort_sparse_indices_known_ncol  src/misc.cpp:118-128, :261-298 (§8f rank 1)
//
// Gather: lengths -> exclusive scan -> one G-lane group per output row copies
// indices and values (contiguous source and destination segments, so both
// sides are coalesced inside a row).  HBM-bound: 4r + 8r + 4(r+1) + 2*12*nnz_out bytes.
#include "mx_common.h"
#include <algorithm>
#include <cstdlib>

namespace mx {

int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace,
                       hipStream_t st);
size_t scan_workspace_bytes(int64_t n);

constexpr int GATHER_BLOCK = 256;

__global__ __launch_bounds__(GATHER_BLOCK)
void gather_lengths_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ rows,
                           int32_t *__restrict__ lens)
{
    const int i = blockIdx.x * GATHER_BLOCK + threadIdx.x;
    if (i < r) { const int row = rows[i]; lens[i] = indptr[row + 1] - indptr[row]; }
}

// new_indptr of the gather in ONE launch: row lengths, their scan (decoupled look-back over the tiles' totals) and the
// grand total — lengths -> three scan launches cost more in launch gaps than in work for r = 200 k.  A thread sizes
// GC_ITEMS consecutive output rows (all 2 * GC_ITEMS row-pointer gathers in flight).
constexpr int GC_ITEMS = 8;
constexpr int GC_TILE = GATHER_BLOCK * GC_ITEMS;
__global__ __launch_bounds__(GATHER_BLOCK)
void gather_count_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ rows,
                         int32_t *__restrict__ new_indptr, unsigned long long *__restrict__ tile_state,
                         unsigned *__restrict__ ticket, long long *__restrict__ total_out, int ntiles)
{
    __shared__ int s_tile;
    __shared__ long long s_base;
    __shared__ int wave_tot[GATHER_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    const long long i0 = (long long)tile * GC_TILE + (long long)tid * GC_ITEMS;
    int row[GC_ITEMS], len[GC_ITEMS];
#pragma unroll
    for (int k = 0; k < GC_ITEMS; k++) row[k] = i0 + k < r ? rows[i0 + k] : -1;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < GC_ITEMS; k++) {
        const int rr = row[k] >= 0 ? row[k] : 0;
        const int a = indptr[rr], b = indptr[rr + 1];
        len[k] = row[k] >= 0 ? b - a : 0;
    }
#pragma unroll
    for (int k = 0; k < GC_ITEMS; k++) sum += len[k];
    int incl = sum;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int up = __shfl_up(incl, o2, 64); if (lane >= o2) incl += up; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wbase = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < GATHER_BLOCK / 64; w++) { wbase += w < wave ? wave_tot[w] : 0; tile_total += wave_tot[w]; }
    if (wave == 0) {
        const long long excl = lookback_exclusive<4>(tile_state, tile, tile_total);
        if (lane == 0) {
            s_base = excl;
            if (tile == ntiles - 1) {
                *total_out = excl + tile_total;
                new_indptr[r] = excl + tile_total <= (long long)INT_MAX ? (int32_t)(excl + tile_total) : INT_MAX;
            }
        }
    }
    __syncthreads();
    long long run = s_base + wbase + incl - sum;
#pragma unroll
    for (int k = 0; k < GC_ITEMS; k++) {
        if (i0 + k < r) new_indptr[i0 + k] = (int32_t)run;
        run += len[k];
    }
}

template <int G, typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(GATHER_BLOCK)
void gather_copy_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                        const VT *__restrict__ values, const int32_t *__restrict__ rows,
                        const int32_t *__restrict__ new_indptr, int32_t *__restrict__ new_indices,
                        VT *__restrict__ new_values, long long capacity)
{
    const int lg = threadIdx.x % G;
    const long long i = (long long)blockIdx.x * (GATHER_BLOCK / G) + threadIdx.x / G;
    if (i >= r) return;
    const int row = rows[i];
    const int src = indptr[row];
    const int len = indptr[row + 1] - src;
    const int dst = new_indptr[i];
    if ((long long)dst + len > capacity) return;                      // (behind the one-launch gather: arrays sized for an estimate)
    for (int k = lg; k < len; k += G) {
        new_indices[dst + k] = indices[src + k];
        if constexpr (HAS_VALUES) new_values[dst + k] = values[src + k];
    }
}

template <typename VT, bool HAS_VALUES>
static int launch_gather_copy(int G, int r, const int32_t *indptr, const int32_t *indices, const void *values,
                              const int32_t *rows, const int32_t *new_indptr, int32_t *new_indices,
                              void *new_values, hipStream_t st, long long capacity = LLONG_MAX)
{
#define MX_CASE(GG)                                                                                      \
    case GG: {                                                                                           \
        const unsigned grid = (unsigned)ceil_div(r, GATHER_BLOCK / GG);                                  \
        hipLaunchKernelGGL((gather_copy_kernel<GG, VT, HAS_VALUES>), dim3(grid), dim3(GATHER_BLOCK), 0,  \
                           st, r, indptr, indices, (const VT *)values, rows, new_indptr, new_indices,    \
                           (VT *)new_values, capacity);                                                  \
        break;                                                                                           \
    }
    switch (G) { MX_CASE(4) MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64)
                 default: return set_error("gather: bad group %d", G); }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
}

// The whole gather in ONE launch (round 3): a workgroup sizes a tile of GF_TILE output rows, scans their lengths, finds
// its place in the output with a decoupled look-back and copies its rows — into arrays that the caller sized for an
// ESTIMATE of the result (`capacity` entries).  new_indptr and the total are always complete and exact; rows that would end
// beyond `capacity` are not copied, the caller sees total > capacity and runs gather_copy_kernel into exactly sized
// arrays instead.  What this removes from a call, measured at cfg3 (two launches: 64 us per call, 43 us of kernels):
//   * the second launch and the host round trip for nnz_out BETWEEN the launches;
//   * the state memset: the look-back words carry a generation (lookback_exclusive_gen) and the ticket counter runs on
//     from launch to launch, so the library's state array is never cleared;
//   * the wait for the size: the workgroup of the LAST tile stores the total straight into pinned host memory as soon as
//     its look-back is through — the host has it while the copies are still running and returns (the copy completes in
//     stream order, like every device-level call).
// Copy: G lanes per row, GF_ROWS rows per group and trip with all their loads in flight before the first store; the
// first trip's loads are issued BEFORE the look-back (they need no output position), so its latency hides the wait.
constexpr int GF_TILE = GATHER_BLOCK;
constexpr int GF_ROWS = 8;
template <int G, typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(GATHER_BLOCK)
void gather_fused_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                         const VT *__restrict__ values, const int32_t *__restrict__ rows,
                         int32_t *__restrict__ new_indptr, int32_t *__restrict__ new_indices, VT *__restrict__ new_values,
                         long long capacity, unsigned long long *__restrict__ tile_state, unsigned *__restrict__ ticket,
                         unsigned ticket_base, unsigned gen, long long *__restrict__ total_out,
                         unsigned long long *__restrict__ host_word, int ntiles)
{
    __shared__ int s_tile;
    __shared__ long long s_base;
    __shared__ int src_l[GF_TILE], len_l[GF_TILE], off_l[GF_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (int)(atomicAdd(ticket, 1u) - ticket_base);
    __syncthreads();
    const int tile = s_tile;
    const long long i = (long long)tile * GF_TILE + tid;
    int src = 0, len = 0;
    if (i < r) {
        const int row = rows[i];
        src = indptr[row];
        len = indptr[row + 1] - src;
    }
    long long incl = len;                                             // (a tile of 256 rows can hold more than 2^31 entries)
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const long long up = __shfl_up(incl, o2, 64); if (lane >= o2) incl += up; }
    __shared__ long long wave_tot64[GATHER_BLOCK / 64];
    __shared__ int wave_max[GATHER_BLOCK / 64];
    int lmax = len;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) lmax = max(lmax, __shfl_xor(lmax, o2, 64));
    if (lane == 63) { wave_tot64[wave] = incl; wave_max[wave] = lmax; }
    src_l[tid] = src;
    len_l[tid] = len;
    __syncthreads();
    long long wbase = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < GATHER_BLOCK / 64; w++) { wbase += w < wave ? wave_tot64[w] : 0; tile_total += wave_tot64[w]; }
    const long long my_off = wbase + incl - len;                      // my row's offset inside the tile
    // the first trip's source loads: issued now, consumed behind the look-back
    constexpr int NG = GATHER_BLOCK / G;                              // lane groups per workgroup
    const int lg = tid % G, grp = tid / G;
    int s[GF_ROWS], n[GF_ROWS], jv[GF_ROWS];
    VT xv[GF_ROWS];
    auto rows_of = [&](int r0) {
#pragma unroll
        for (int q = 0; q < GF_ROWS; q++) { s[q] = src_l[r0 + q]; n[q] = len_l[r0 + q]; }
    };
    auto load_at = [&](int k) {
#pragma unroll
        for (int q = 0; q < GF_ROWS; q++) {
            jv[q] = 0; xv[q] = VT();
            if (k < n[q]) {
                jv[q] = indices[s[q] + k];
                if constexpr (HAS_VALUES) xv[q] = values[s[q] + k];
            }
        }
    };
    if (grp * GF_ROWS < GF_TILE) {                                    // (G = 4: more lane groups than trips)
        rows_of(grp * GF_ROWS);
        load_at(lg);
    }
    if (wave == 0) {
        const long long excl = lookback_exclusive_gen<4>(tile_state, tile, tile_total, gen);
        if (lane == 0) {
            s_base = excl;
            if (tile == ntiles - 1) {
                const long long total = excl + tile_total;
                *total_out = total;
                new_indptr[r] = total <= (long long)INT_MAX ? (int32_t)total : INT_MAX;
                if (host_word)                                        // the size, straight to the waiting host
                    __hip_atomic_store(host_word, lbg_pack(0, gen, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    const long long base = s_base;
    {
        const long long dst = base + my_off;
        if (i < r) new_indptr[i] = dst <= (long long)INT_MAX ? (int32_t)dst : INT_MAX;
        // rows that end beyond the caller's arrays are not copied; offsets inside a tile fit int32 whenever they matter
        off_l[tid] = dst + len <= capacity ? (int)my_off : -1;
    }
    __syncthreads();
    // A tile whose rows are very uneven (round 5, tools/cliff_hunt_ops.py: log-normal row lengths, sigma 1.5: 0.28 ms where
    // equal rows take 0.067; a row of 50,000 entries: one lane group for 1,500 trips): the lane groups below walk 8 rows in
    // lockstep, as long as the longest.  Such a tile is copied FLAT instead: thread t takes positions t, t + 256, ... of the
    // tile's output range and finds the row of a position by a binary search over the rows' offsets in LDS — the same work
    // for every thread whatever the row lengths.  (Even tiles keep the lane groups: no search per entry.)
    {
        int tmax = 0;
#pragma unroll
        for (int w = 0; w < GATHER_BLOCK / 64; w++) tmax = max(tmax, wave_max[w]);
        if (tmax > 64 && (long long)tmax * 64 > tile_total && base + tile_total <= capacity) {   // longest > 4 x mean (256 rows)
            for (long long p = tid; p < tile_total; p += GATHER_BLOCK) {
                int lo = 0, hi = GF_TILE - 1;                        // the last row whose offset is <= p
#pragma unroll
                for (int step = 0; step < 8; step++) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (off_l[mid] <= (int)p) lo = mid; else hi = mid - 1;
                }
                const int at = src_l[lo] + ((int)p - off_l[lo]);
                new_indices[base + p] = indices[at];
                if constexpr (HAS_VALUES) new_values[base + p] = values[at];
            }
            return;
        }
    }
    for (int r0 = grp * GF_ROWS; r0 < GF_TILE; r0 += NG * GF_ROWS) {
        if (r0 != grp * GF_ROWS) rows_of(r0);
        long long d[GF_ROWS];
        int maxn = 0;
#pragma unroll
        for (int q = 0; q < GF_ROWS; q++) {
            const int o = off_l[r0 + q];
            if (o < 0) n[q] = 0;
            d[q] = base + o;
            maxn = max(maxn, n[q]);
        }
        for (int k = lg; k < maxn; k += G) {                          // (uniform per group: all rows of the trip advance together)
            if (r0 != grp * GF_ROWS || k != lg) load_at(k);
#pragma unroll
            for (int q = 0; q < GF_ROWS; q++) {
                if (k < n[q]) {
                    new_indices[d[q] + k] = jv[q];
                    if constexpr (HAS_VALUES) new_values[d[q] + k] = xv[q];
                }
            }
        }
    }
}

template <typename VT, bool HAS_VALUES>
static int launch_gather_fused(int G, int r, const int32_t *indptr, const int32_t *indices, const void *values,
                               const int32_t *rows, int32_t *new_indptr, int32_t *new_indices, void *new_values,
                               long long capacity, unsigned long long *state, unsigned *ticket, unsigned ticket_base, unsigned gen,
                               long long *total_dev, unsigned long long *host_word, int ntiles, hipStream_t st)
{
#define MX_CASE(GG)                                                                                                   \
    case GG:                                                                                                          \
        hipLaunchKernelGGL((gather_fused_kernel<GG, VT, HAS_VALUES>), dim3((unsigned)ntiles), dim3(GATHER_BLOCK), 0, st, r,   \
                           indptr, indices, (const VT *)values, rows, new_indptr, new_indices, (VT *)new_values, capacity,    \
                           state, ticket, ticket_base, gen, total_dev, host_word, ntiles);                            \
        break;
    switch (G) { MX_CASE(4) MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64)
                 default: return set_error("gather: bad group %d", G); }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
}

// The fused gather's state, per thread and device (one stream per thread and device at a time, like AUTO's plan): the
// look-back words + ticket counter + total in device memory (zeroed when allocated, never cleared afterwards), one pinned
// host word the last tile's workgroup stores the total into, and the launch counters that make the generation / ticket base.
struct GatherState {
    char *dev = nullptr;                     // [int64 total][uint32 ticket][pad][uint64 tile_state[cap_tiles]]
    size_t cap_tiles = 0;
    unsigned ticket_base = 0, gen = 0;
    unsigned long long *host_word = nullptr;
    hipStream_t last_stream = nullptr;       // the stream of the previous launch (a caller that switches streams)
    hipEvent_t last_done = nullptr;          // recorded behind the previous launch (an event outlives its stream)
    bool launched = false;
};
static GatherState *gather_state(int ntiles, hipStream_t st)
{
    static thread_local GatherState gs[16];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    GatherState &g = gs[d];
    if (!g.host_word) {
        if (hipHostMalloc((void **)&g.host_word, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            g.host_word = nullptr;
            return nullptr;
        }
        *g.host_word = 0;
    }
    if (g.cap_tiles < (size_t)ntiles) {
        if (g.dev) { (void)hipStreamSynchronize(st); (void)hipFree(g.dev); g.dev = nullptr; g.cap_tiles = 0; }
        const size_t want = (size_t)ntiles + (size_t)ntiles / 2 + 1024;
        if (hipMalloc((void **)&g.dev, 16 + want * 8) != hipSuccess) { (void)hipGetLastError(); g.dev = nullptr; return nullptr; }
        if (hipMemsetAsync(g.dev, 0, 16 + want * 8, st) != hipSuccess) return nullptr;
        g.cap_tiles = want;
        g.ticket_base = 0;                                            // a fresh ticket counter; generations run on
    }
    return &g;
}

// ---- check_is_seq / check_is_rev_seq ---------------------------------------------------------
// flag[0] starts at 0 and is set by any violating pair.
__global__ __launch_bounds__(256)
void is_seq_kernel(const int32_t *__restrict__ idx, int64_t n, int step, int32_t *__restrict__ flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= idx[i] != idx[i - 1] + step;
    if (__ballot(bad) != 0ULL && lane_id() == 0) atomicOr(flag, 1);
}

// ---- per-row sortedness / sort -----------------------------------------------------------------
// check_is_sorted over every row (misc.cpp:118-128, as sort_sparse_indices_known_ncol applies it row by row, :283): a
// row is unsorted iff some entry is smaller than its predecessor IN THE SAME ROW, i.e. iff some DESCENT
// (indices[k] < indices[k-1]) sits at a position that is not the start of a row.  With
//     D = #{descents},   S = #{non-empty rows whose first entry is a descent}
// all rows are sorted  <=>  D == S.  Both are counted in ONE pass over the indices (round 3; rounds 1-2 counted S in a
// second pass over indptr that re-read one 128-byte line of `indices` per row: 256 MB on top of the 408 MB of cfg4):
// a workgroup owns a contiguous range of quads (4 entries = one 16-byte load per lane), streams it in sub-chunks of
// RS_SUB_Q quads, leaves each quad's four descent bits in a byte of an LDS bitmap, and resolves the rows that start inside
// the sub-chunk against that bitmap — the row pointers are the only other thing read (coalesced, once).  The rows of a
// sub-chunk are found by a cursor that the workgroup carries along (rows start in ascending order); the first cursor
// of a workgroup comes from a 64-ary search of indptr by one wavefront while the others already stream.
// Entry positions are counted from a 16-byte aligned base (`shift` = elements between that base and indices[0]), so any
// int32-aligned array takes the vector path; a quad that holds at least one valid entry lies inside one aligned 16-byte
// granule of the array's pages, so partial quads at either end are loaded whole and masked.  nnz and the first entry
// (indptr[0] > 0: a row-block view with absolute offsets) are read on the device — no host round trip before the launch.
constexpr int RS_BLOCK = 256, RS_U = 4, RS_SUB_Q = RS_BLOCK * RS_U, RS_NB = 2048;

__global__ __launch_bounds__(RS_BLOCK)
void rows_sorted_tile_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ idx_al, int shift,
                             unsigned *__restrict__ partial, unsigned *__restrict__ done, unsigned *__restrict__ out2,
                             unsigned long long *__restrict__ host_word, unsigned gen)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned char bits[2][RS_SUB_Q];
    __shared__ int cur[3];
    __shared__ unsigned wsum[2][RS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = indptr[0];
    const long long lo = (long long)p0 + shift, hi = (long long)indptr[m] + shift;     // descents count at lo < k' < hi
    const long long Q0 = lo >> 2, Q1 = (hi + 3) >> 2;
    long long cq = (Q1 - Q0 + gridDim.x - 1) / gridDim.x;
    cq = (cq + RS_SUB_Q - 1) / RS_SUB_Q * RS_SUB_Q;
    const long long Qa = Q0 + (long long)blockIdx.x * cq, Qb = Qa + cq < Q1 ? Qa + cq : Q1;
    unsigned D = 0, S = 0;
    if (Qa < Qb) {                                                            // (uniform per workgroup)
        if (wave == 0) {
            // first row r in [0, m] with indptr[r] >= key: 64 probes per round
            const long long key = 4 * Qa - shift;
            int a = 0, b = m;                                                 // indptr[m] >= key because 4 Qa < hi
            while (a < b) {
                const int stride = (b - a + 63) >> 6;
                const long long idx = (long long)a + (long long)lane * stride;
                const bool lt = idx < b && (long long)indptr[idx] < key;
                const int cnt = __popcll(__ballot(lt));                       // probes 0 .. cnt-1 are below the key
                const long long nb = (long long)a + (long long)cnt * stride;  // the first probe that is not
                if (cnt > 0) a = a + (cnt - 1) * stride + 1;
                if (nb < b) b = (int)nb;
                if (cnt == 0) b = a;
            }
            if (lane == 0) { cur[0] = a; cur[1] = m; cur[2] = m; }
        }
        i4 c[RS_U];
        int pv[RS_U];
        long long Qs = Qa;
        auto issue = [&](long long Qstart) {
#pragma unroll
            for (int u = 0; u < RS_U; u++) {
                const long long Q = Qstart + u * RS_BLOCK + tid;
                c[u] = (i4){0, 0, 0, 0};
                pv[u] = 0;
                if (Q < Qb) {
                    c[u] = *reinterpret_cast<const i4 *>(idx_al + Q * 4);
                    if (lane == 0 && Q > 0) pv[u] = idx_al[Q * 4 - 1];       // the wavefront's first quad: predecessor from memory
                }
            }
        };
        auto mark = [&](long long Qstart, int buf) {
#pragma unroll
            for (int u = 0; u < RS_U; u++) {
                const int qi = u * RS_BLOCK + tid;
                const long long k = (Qstart + qi) * 4;
                const int up = __shfl_up(c[u][3], 1, 64);
                const int prev = lane == 0 ? pv[u] : up;
                unsigned nib = 0;
                nib |= (unsigned)(c[u][0] < prev && k > lo && k < hi);
                nib |= (unsigned)(c[u][1] < c[u][0] && k + 1 > lo && k + 1 < hi) << 1;
                nib |= (unsigned)(c[u][2] < c[u][1] && k + 2 > lo && k + 2 < hi) << 2;
                nib |= (unsigned)(c[u][3] < c[u][2] && k + 3 > lo && k + 3 < hi) << 3;
                if (Qstart + qi >= Qb) nib = 0;
                bits[buf][qi] = (unsigned char)nib;
                D += __popc(nib);
            }
        };
        issue(Qs);
        mark(Qs, 0);
        for (int it = 0;; it++) {
            const long long Qn = Qs + RS_SUB_Q;
            __syncthreads();                                                  // bitmap `it` and cursor `it` are complete
            const int rc = cur[it % 3];
            if (tid == 0) cur[(it + 2) % 3] = m;
            const long long sbase = 4 * Qs - shift;                           // row start s lies in this sub-chunk iff 0 <= s - sbase < 4 RS_SUB_Q
            // first round of row pointers, then the next sub-chunk's entries: both in flight together
            long long row = (long long)rc + 64 * wave + lane;
            int s = 0, e = 0;
            if (row < m) { s = indptr[row]; e = indptr[row + 1]; }
            if (Qn < Qb) issue(Qn);
            for (;;) {
                const bool in_m = row < m;
                const long long off = (long long)s - sbase;
                const bool exceed = !in_m || off >= 4 * RS_SUB_Q;
                if (!exceed && e > s && s > p0) S += (bits[it & 1][off >> 2] >> (off & 3)) & 1u;
                const unsigned long long ex = __ballot(exceed);
                if (ex) {
                    if (lane == 0) {
                        const long long cand = row + __builtin_ctzll(ex);
                        atomicMin(&cur[(it + 1) % 3], cand < m ? (int)cand : m);
                    }
                    break;
                }
                row += RS_BLOCK;
                s = e = 0;
                if (row < m) { s = indptr[row]; e = indptr[row + 1]; }
            }
            if (Qn >= Qb) break;
            mark(Qn, (it + 1) & 1);
            Qs = Qn;
        }
    }
    // one plain store per workgroup and count, summed by rows_sorted_finish_kernel: same-address atomics serialise in L2
    // (~10 ns each; one atomicAdd per wavefront of a 3072-workgroup grid was 130 us of a 180 us kernel)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { D += __shfl_xor(D, off, 64); S += __shfl_xor(S, off, 64); }
    if (lane == 0) { wsum[0][wave] = D; wsum[1][wave] = S; }
    __syncthreads();
    // The workgroup that finishes LAST adds the partial counts up and reports (no second launch, no copy packet): every
    // workgroup publishes its two counts with agent-scope stores, then takes a number; the last one reads them all back
    // with agent-scope loads (another CU's plain stores are not visible through this CU's L1 otherwise).
    __shared__ bool s_last;
    if (tid == 0) {
        __hip_atomic_store(&partial[blockIdx.x], wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partial[gridDim.x + blockIdx.x], wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the counts have left before the number is taken
        // The workgroups finish within a few microseconds of each other: 2048 adds on ONE word queued up for ~20 us at
        // the end of the kernel (same-address atomics serialise in L2, ~10 ns each).  Eight shard counters (blockIdx % 8,
        // one cache line each) take the adds side by side; the last arriver of a shard moves on to the top counter.
        const unsigned shard = blockIdx.x & 7u, in_shard = (gridDim.x - shard + 7u) >> 3;
        bool last = false;
        if (__hip_atomic_fetch_add(done + 32 * (1 + shard), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            const unsigned nshards = gridDim.x < 8u ? gridDim.x : 8u;
            last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                         // (once per launch; the loads below bypass L1 anyway)
    unsigned d = 0, sct = 0;
    for (unsigned i = tid; i < gridDim.x; i += RS_BLOCK) {
        d += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sct += __hip_atomic_load(&partial[gridDim.x + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d += __shfl_xor(d, off, 64); sct += __shfl_xor(sct, off, 64); }
    __syncthreads();
    if (lane == 0) { wsum[0][wave] = d; wsum[1][wave] = sct; }
    __syncthreads();
    if (tid == 0) {
        const unsigned dt = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3], stt = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
        out2[0] = dt; out2[1] = stt;                                           // [0] descents, [1] descents at row starts
        done[0] = 0;                                                           // ready for the next launch
        for (int k = 1; k <= 8; k++) done[32 * k] = 0;
        if (host_word)
            __hip_atomic_store(host_word, ((unsigned long long)gen << 32) | (dt == stt ? 1ULL : 0ULL), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Per-row sort by column id into tmp (stable: ties keep their storage order), one wavefront per row; rows that are
// already non-decreasing are copied through, as the reference skips them (misc.cpp:283).
//   rows of up to SORT_LDS_MAX entries: bitonic network over (column, position) keys held in the wavefront's slice of LDS
//   longer rows: rank sort — every lane ranks its entries against the whole row, O(len^2 / 64)
constexpr int SORT_LDS_MAX = 512;
constexpr int SORT_WAVES = GATHER_BLOCK / 64;

template <typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(GATHER_BLOCK)
void sort_rows_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                      const VT *__restrict__ values, int32_t *__restrict__ tmp_idx, VT *__restrict__ tmp_val)
{
    __shared__ unsigned long long keys_all[SORT_WAVES][SORT_LDS_MAX];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * SORT_WAVES + wave;
    if (row >= m) return;
    const int s = indptr[row], len = indptr[row + 1] - s;
    if (len == 0) return;
    const int32_t *__restrict__ keys = indices + s;
    // already sorted?  (one pass, as check_is_sorted)
    bool bad = false;
    for (int i = lane + 1; i < len; i += 64) bad |= keys[i] < keys[i - 1];
    if (__ballot(bad) == 0ULL) {
        for (int i = lane; i < len; i += 64) {
            tmp_idx[s + i] = keys[i];
            if constexpr (HAS_VALUES) tmp_val[s + i] = values[s + i];
        }
        return;
    }
    if (len <= SORT_LDS_MAX) {
        unsigned long long *kb = keys_all[wave];
        int n2 = 64;
        while (n2 < len) n2 <<= 1;                                            // network size (power of two >= len)
        for (int i = lane; i < n2; i += 64)                                   // pad with +inf keys
            kb[i] = i < len ? ((unsigned long long)(unsigned)keys[i] << 32) | (unsigned)i : ~0ULL;
        // column ids are non-negative ints: unsigned order == signed order.  One wavefront's LDS operations execute in
        // order, so the stages need no barrier among the lanes of the wavefront beyond the data dependence.
        for (int k = 2; k <= n2; k <<= 1) {
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int t = lane; t < (n2 >> 1); t += 64) {
                    const int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));       // lower index of comparator t
                    const int l = i | jj;
                    const bool up = (i & k) == 0;
                    const unsigned long long a = kb[i], b = kb[l];
                    if ((a > b) == up) { kb[i] = b; kb[l] = a; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        for (int i = lane; i < len; i += 64) {
            const unsigned long long kv = kb[i];
            tmp_idx[s + i] = (int32_t)(kv >> 32);
            if constexpr (HAS_VALUES) tmp_val[s + i] = values[s + (int)(kv & 0xFFFFFFFFu)];
        }
        return;
    }
    for (int i = lane; i < len; i += 64) {
        const int key = keys[i];
        int rank = 0;
        for (int k = 0; k < len; k++) {
            const int other = keys[k];
            rank += (other < key) || (other == key && k < i);
        }
        tmp_idx[s + rank] = key;
        if constexpr (HAS_VALUES) tmp_val[s + rank] = values[s + i];
    }
}

template <typename VT, bool HAS_VALUES>
static int launch_sort_rows(int m, const int32_t *indptr, const int32_t *indices, const void *values,
                            int32_t *tmp_idx, void *tmp_val, hipStream_t st)
{
    const unsigned grid = (unsigned)ceil_div(m, SORT_WAVES);
    hipLaunchKernelGGL((sort_rows_kernel<VT, HAS_VALUES>), dim3(grid), dim3(GATHER_BLOCK), 0, st, m, indptr, indices,
                       (const VT *)values, tmp_idx, (VT *)tmp_val);
    MX_LAUNCH_CHECK();
    return 0;
}

}  // namespace mx

extern "C" size_t mxd_gather_workspace_bytes(int r)
{
    const size_t lens = ((size_t)(r > 0 ? r : 1) * sizeof(int32_t) + 15) & ~(size_t)15;
    return lens + mx::scan_workspace_bytes(r);
}

extern "C" int mxd_csr_gather_count(int r, const int32_t *indptr, const int32_t *rows_take, int32_t *new_indptr,
                                    void *workspace, int64_t *nnz_out_host, void *stream)
{
    MX_REQUIRE(r >= 0, "mxd_csr_gather_count: negative r");
    MX_REQUIRE(new_indptr && workspace, "mxd_csr_gather_count: null pointer");
    hipStream_t st = mx::as_stream(stream);
    // workspace: [int64 total][uint32 ticket, pad][uint64 tile_state[ntiles]]  (fits mxd_gather_workspace_bytes(r))
    int64_t *total_dev = (int64_t *)workspace;
    if (r == 0) {
        MX_HIP(hipMemsetAsync(new_indptr, 0, sizeof(int32_t), st));
        MX_HIP(hipMemsetAsync(total_dev, 0, sizeof(int64_t), st));
    } else {
        const int ntiles = (int)mx::ceil_div(r, mx::GC_TILE);
        MX_HIP(hipMemsetAsync(workspace, 0, 16 + (size_t)ntiles * 8, st));
        hipLaunchKernelGGL(mx::gather_count_kernel, dim3((unsigned)ntiles), dim3(mx::GATHER_BLOCK), 0, st, r, indptr, rows_take,
                           new_indptr, (unsigned long long *)((char *)workspace + 16), (unsigned *)((char *)workspace + 8),
                           (long long *)total_dev, ntiles);
        MX_LAUNCH_CHECK();
    }
    if (nnz_out_host) {
        if (mx::read_back_small(nnz_out_host, total_dev, sizeof(int64_t), st)) return 1;
        MX_REQUIRE(*nnz_out_host <= (int64_t)INT_MAX, "result has %lld entries: exceeds R's int32 index range",
                   (long long)*nnz_out_host);
    }
    return 0;
}

extern "C" int mxd_csr_gather_fill(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                                   const int32_t *rows_take, const int32_t *new_indptr, int32_t *new_indices,
                                   void *new_values, int value_dtype, int64_t nnz_out, void *stream)
{
    MX_REQUIRE(r >= 0, "mxd_csr_gather_fill: negative r");
    if (r == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const int G = nnz_out < 0 ? 32 : mx::pick_group((double)nnz_out / (double)r);
    switch (value_dtype) {
        case MX_F64: return mx::launch_gather_copy<double, true>(G, r, indptr, indices, values, rows_take, new_indptr,
                                                                 new_indices, new_values, st);
        case MX_LGL: return mx::launch_gather_copy<int32_t, true>(G, r, indptr, indices, values, rows_take, new_indptr,
                                                                  new_indices, new_values, st);
        case MX_NONE: return mx::launch_gather_copy<int32_t, false>(G, r, indptr, indices, nullptr, rows_take,
                                                                    new_indptr, new_indices, nullptr, st);
        default: return mx::set_error("mxd_csr_gather_fill: unsupported value dtype %d", value_dtype);
    }
}

extern "C" int mxd_csr_gather_fused(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                                    const int32_t *rows_take, int32_t *new_indptr, int32_t *new_indices, void *new_values,
                                    int value_dtype, int64_t capacity, double avg_row_len, void *workspace,
                                    int64_t *nnz_out_host, void *stream)
{
    (void)workspace;                                                 // (kept in the signature; the state lives in the library)
    MX_REQUIRE(r >= 0 && capacity >= 0, "mxd_csr_gather_fused: negative size");
    MX_REQUIRE(new_indptr && nnz_out_host, "mxd_csr_gather_fused: null pointer");
    hipStream_t st = mx::as_stream(stream);
    if (r == 0) {
        MX_HIP(hipMemsetAsync(new_indptr, 0, sizeof(int32_t), st));
        *nnz_out_host = 0;
        return 0;
    }
    const int ntiles = (int)mx::ceil_div(r, mx::GF_TILE);
    mx::GatherState *g = mx::gather_state(ntiles, st);
    MX_REQUIRE(g, "mxd_csr_gather_fused: cannot allocate the look-back state");
    // One launch at a time uses the state.  A call returns only after the LAST tile (in ticket order) has published the
    // total — every ticket of that launch is then taken and every state word written, so the next launch's tickets
    // (ticket_base + ...) and generation cannot mix with it even while its copy phase still runs.  The one exit that
    // returns earlier is an error; and a caller may come back on ANOTHER stream (a torch stream switch): wait for the
    // previous stream then, so that the state is never shared by two launches in flight.
    // (through an EVENT: the previous stream may be gone by now — the small-call path's stream is destroyed by
    // mxd_release_workspaces — and synchronising a destroyed stream is a crash, an event recorded on it is not)
    if (!g->last_done && hipEventCreateWithFlags(&g->last_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g->last_done = nullptr; }
    if (g->launched && g->last_stream != st) {
        if (g->last_done) MX_HIP(hipStreamWaitEvent(st, g->last_done, 0));
        else MX_HIP(hipDeviceSynchronize());
    }
    g->last_stream = st;
    g->launched = true;
    g->gen = (g->gen + 1) & 0x3FFFFFFFu;
    if (g->gen == 0) g->gen = 1;
    const unsigned gen = g->gen, ticket_base = g->ticket_base;
    long long *total_dev = (long long *)g->dev;
    unsigned *ticket = (unsigned *)(g->dev + 8);
    unsigned long long *state = (unsigned long long *)(g->dev + 16);
    const int G = mx::pick_group(avg_row_len > 0 ? avg_row_len : 32.0);
    // A tile is 256 rows and ONE workgroup copies it: a selection of a few thousand long rows is a handful of workgroups for
    // 256 CUs (2,000 rows of 500 entries: 8 tiles, 0.118 ms; tools/cliff_hunt_ops.py).  Below 128 tiles of rows that average
    // 32 entries or more the launch only SIZES the result (capacity -1: every row "does not fit") and the per-row copy
    // kernel — one lane group per row, r groups — follows in the same stream; new_indptr and the total are the same.
    const bool few_tiles = ntiles < 128 && avg_row_len >= 32.0 && (double)r * avg_row_len >= 65536.0;   // (small calls: one launch)
    const long long cap_k = few_tiles ? -1 : (long long)capacity;
    int rc;
    switch (value_dtype) {
        case MX_F64: rc = mx::launch_gather_fused<double, true>(G, r, indptr, indices, values, rows_take, new_indptr, new_indices,
                                                                new_values, cap_k, state, ticket, ticket_base, gen, total_dev,
                                                                g->host_word, ntiles, st); break;
        case MX_LGL: rc = mx::launch_gather_fused<int32_t, true>(G, r, indptr, indices, values, rows_take, new_indptr, new_indices,
                                                                 new_values, cap_k, state, ticket, ticket_base, gen, total_dev,
                                                                 g->host_word, ntiles, st); break;
        case MX_NONE: rc = mx::launch_gather_fused<int32_t, false>(G, r, indptr, indices, nullptr, rows_take, new_indptr,
                                                                   new_indices, nullptr, cap_k, state, ticket, ticket_base, gen,
                                                                   total_dev, g->host_word, ntiles, st); break;
        default: return mx::set_error("mxd_csr_gather_fused: unsupported value dtype %d", value_dtype);
    }
    if (rc) return rc;
    if (few_tiles) {
        switch (value_dtype) {
            case MX_F64: rc = mx::launch_gather_copy<double, true>(G, r, indptr, indices, values, rows_take, new_indptr, new_indices,
                                                                   new_values, st, (long long)capacity); break;
            case MX_LGL: rc = mx::launch_gather_copy<int32_t, true>(G, r, indptr, indices, values, rows_take, new_indptr, new_indices,
                                                                    new_values, st, (long long)capacity); break;
            default: rc = mx::launch_gather_copy<int32_t, false>(G, r, indptr, indices, nullptr, rows_take, new_indptr, new_indices,
                                                                  nullptr, st, (long long)capacity); break;
        }
        if (rc) return rc;
    }
    if (g->last_done) (void)hipEventRecord(g->last_done, st);
    g->ticket_base = ticket_base + (unsigned)ntiles;                  // (the launch was accepted: its tiles will take their tickets)
    // the size arrives in the pinned word while the copies still run: a short spin, then the ordinary wait for the stream
    volatile unsigned long long *hw = g->host_word;
    unsigned long long w = 0;
    bool got = false;
    for (int spin = 0; spin < 200000; spin++) {
        w = *hw;
        if (((w >> 32) & 0x3FFFFFFFu) == gen) { got = true; break; }
    }
    if (!got) {
        MX_HIP(hipStreamSynchronize(st));
        w = *hw;
        MX_REQUIRE(((w >> 32) & 0x3FFFFFFFu) == gen, "mxd_csr_gather_fused: the kernel did not report its size");
    }
    const unsigned long long total = w & mx::LBG_VALUE;
    MX_REQUIRE(total <= (unsigned long long)INT_MAX, "result has %llu or more entries: exceeds R's int32 index range", total);
    *nnz_out_host = (int64_t)total;
    return 0;
}

extern "C" int mxd_check_is_seq(const int32_t *idx, int64_t n, int reversed, int32_t *workspace4, int *flag_host,
                                void *stream)
{
    MX_REQUIRE(flag_host, "mxd_check_is_seq: null flag pointer");
    if (n < 2) { *flag_host = 1; return 0; }     // slice.cpp:27,39
    MX_REQUIRE(idx && workspace4, "mxd_check_is_seq: null pointer");
    hipStream_t st = mx::as_stream(stream);
    MX_HIP(hipMemsetAsync(workspace4, 0, sizeof(int32_t), st));
    const unsigned grid = (unsigned)(mx::ceil_div(n, 256) < 2048 ? mx::ceil_div(n, 256) : 2048);
    hipLaunchKernelGGL(mx::is_seq_kernel, dim3(grid), dim3(256), 0, st, idx, n, reversed ? -1 : 1, workspace4);
    MX_LAUNCH_CHECK();
    int32_t flag = 1;
    if (mx::read_back_small(&flag, workspace4, sizeof(flag), st)) return 1;
    *flag_host = flag == 0;
    return 0;
}

extern "C" int mxd_csr_rows_sorted(int m, const int32_t *indptr, const int32_t *indices, int32_t *workspace4,
                                   int *flag_host, void *stream)
{
    MX_REQUIRE(flag_host, "mxd_csr_rows_sorted: null flag pointer");
    if (m <= 0) { *flag_host = 1; return 0; }
    MX_REQUIRE(indptr && workspace4, "mxd_csr_rows_sorted: null pointer");
    hipStream_t st = mx::as_stream(stream);
    const int nb = mx::RS_NB;
    // [2 nb partial counts][pad][top counter + 8 shard counters, a cache line each]: zeroed when the buffer is (re)allocated,
    // reset by every launch
    unsigned *partial = (unsigned *)mx::scratch_buffer_zeroed(mx::MX_SCRATCH_PARTIALS, (size_t)(2 * nb + 32 * 9 + 32) * sizeof(unsigned), st, nullptr);
    MX_REQUIRE(partial, "mxd_csr_rows_sorted: cannot allocate the partial counts");
    MX_REQUIRE(((uintptr_t)indices & 3) == 0, "mxd_csr_rows_sorted: indices not int32-aligned");
    const int shift = (int)(((uintptr_t)indices & 15) >> 2);                 // entries between the 16-byte aligned base and indices[0]
    mx::HostSignal sig;
    if (mx::host_signal_next(&sig)) return 1;
    hipLaunchKernelGGL(mx::rows_sorted_tile_kernel, dim3(nb), dim3(mx::RS_BLOCK), 0, st, m, indptr, indices - shift, shift, partial,
                       partial + 2 * nb + 32, (unsigned *)workspace4, sig.word, sig.gen);
    MX_LAUNCH_CHECK();
    unsigned sorted = 0;
    if (mx::host_signal_wait(sig, &sorted, st)) return 1;
    *flag_host = sorted != 0;
    return 0;
}

extern "C" int mxd_csr_sort_rows(int m, int64_t nnz, const int32_t *indptr, int32_t *indices, void *values,
                                 int value_dtype, int32_t *tmp_indices, void *tmp_values, void *stream)
{
    MX_REQUIRE(m >= 0 && nnz >= 0, "mxd_csr_sort_rows: negative size");
    if (m == 0 || nnz == 0) return 0;
    MX_REQUIRE(indptr && indices && tmp_indices, "mxd_csr_sort_rows: null pointer");
    hipStream_t st = mx::as_stream(stream);
    int rc;
    size_t vbytes = 0;
    switch (value_dtype) {
        case MX_F64: rc = mx::launch_sort_rows<double, true>(m, indptr, indices, values, tmp_indices, tmp_values, st);
                     vbytes = 8; break;
        case MX_LGL: case MX_I32:
                     rc = mx::launch_sort_rows<int32_t, true>(m, indptr, indices, values, tmp_indices, tmp_values, st);
                     vbytes = 4; break;
        case MX_NONE: rc = mx::launch_sort_rows<int32_t, false>(m, indptr, indices, nullptr, tmp_indices, nullptr, st);
                     break;
        default: return mx::set_error("mxd_csr_sort_rows: unsupported value dtype %d", value_dtype);
    }
    if (rc) return rc;
    MX_HIP(hipMemcpyAsync(indices, tmp_indices, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (vbytes) MX_HIP(hipMemcpyAsync(values, tmp_values, (size_t)nnz * vbytes, hipMemcpyDeviceToDevice, st));
    return 0;
}
