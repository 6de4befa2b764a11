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
                        VT *__restrict__ new_values)
{
    const int lg = threadIdx.x % G;
    const long long i = (long long)blockIdx.x * (GATHER_BLOCK / G) + threadIdx.x / G;
    if (i >= r) return;
    const int row = rows[i];
    const int src = indptr[row];
    const int len = indptr[row + 1] - src;
    const int dst = new_indptr[i];
    for (int k = lg; k < len; k += G) {
        new_indices[dst + k] = indices[src + k];
        if constexpr (HAS_VALUES) new_values[dst + k] = values[src + k];
    }
}

template <typename VT, bool HAS_VALUES>
static int launch_gather_copy(int G, int r, const int32_t *indptr, const int32_t *indices, const void *values,
                              const int32_t *rows, const int32_t *new_indptr, int32_t *new_indices,
                              void *new_values, hipStream_t st)
{
#define MX_CASE(GG)                                                                                      \
    case GG: {                                                                                           \
        const unsigned grid = (unsigned)ceil_div(r, GATHER_BLOCK / GG);                                  \
        hipLaunchKernelGGL((gather_copy_kernel<GG, VT, HAS_VALUES>), dim3(grid), dim3(GATHER_BLOCK), 0,  \
                           st, r, indptr, indices, (const VT *)values, rows, new_indptr, new_indices,    \
                           (VT *)new_values);                                                            \
        break;                                                                                           \
    }
    switch (G) { MX_CASE(4) MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64)
                 default: return set_error("gather: bad group %d", G); }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
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
// row is unsorted iff some entry is smaller than its predecessor IN THE SAME ROW.  Counted without knowing which row an
// entry belongs to:   D = #{k >= 1 : indices[k] < indices[k-1]}  (one pure stream over the indices, 16 B per lane)
//                     S = #{non-empty rows r with start s > 0 : indices[s] < indices[s-1]}  (one pass over indptr)
// every descent that is not at a row start is inside a row, so all rows are sorted  <=>  D == S.  The first
// `nb_entries` workgroups count D, the others S; nnz is read from indptr[m] on the device (no host round trip before
// the launch).  (The element-parallel kernel this replaces searched indptr for every descent — 20 dependent loads — and
// read 4 B per lane: 0.26 TB/s.)
template <bool VEC>
__global__ __launch_bounds__(256)
void rows_sorted_count_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                              unsigned *__restrict__ counters, int nb_entries)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    const long long nnz = indptr[m];
    const int lane = lane_id();
    unsigned cnt = 0;
    const bool entry_part = (int)blockIdx.x < nb_entries;
    if (entry_part) {
        if constexpr (VEC) {
            const long long nq = nnz >> 2;                                    // whole quads
            constexpr int U = 4;                                              // quads per thread and trip: 4 loads in flight
            const long long stride = (long long)nb_entries * 256;
            for (long long q0 = (long long)blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += stride * U) {
                i4 c[U];
                int pv[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const long long q = q0 + u * stride;
                    const long long qs = q < nq ? q : q0;                     // clamped: every load is issued
                    c[u] = *reinterpret_cast<const i4 *>(indices + qs * 4);
                    pv[u] = 0;
                    if (lane == 0) pv[u] = indices[qs > 0 ? qs * 4 - 1 : 0];  // the wavefront's first quad: predecessor from memory
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const long long q = q0 + u * stride;
                    const int up = __shfl_up(c[u][3], 1, 64);                 // lane - 1 holds quad q - 1 of the same trip
                    if (q < nq) {
                        const int prev = q > 0 ? (lane == 0 ? pv[u] : up) : c[u][0];
                        cnt += (c[u][0] < prev) + (c[u][1] < c[u][0]) + (c[u][2] < c[u][1]) + (c[u][3] < c[u][2]);
                    }
                }
            }
            if (blockIdx.x == 0 && threadIdx.x < (unsigned)(nnz & 3)) {       // the last partial quad
                const long long k = (nq << 2) + threadIdx.x;
                if (k > 0) cnt += indices[k] < indices[k - 1];
            }
        } else {
            for (long long k = (long long)blockIdx.x * 256 + threadIdx.x + 1; k < nnz; k += (long long)nb_entries * 256)
                cnt += indices[k] < indices[k - 1];
        }
    } else {
        const long long nb_rows = gridDim.x - nb_entries;
        constexpr int U = 4;                                                  // rows per thread and trip
        const long long stride = nb_rows * 256;
        for (long long r0 = (long long)(blockIdx.x - nb_entries) * 256 + threadIdx.x; r0 < m; r0 += stride * U) {
            int s[U], e[U], a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long long r = r0 + u * stride < m ? r0 + u * stride : r0;
                s[u] = indptr[r]; e[u] = indptr[r + 1];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool ok = e[u] > s[u] && s[u] > 0;
                a[u] = indices[ok ? s[u] : 0]; b[u] = indices[ok ? s[u] - 1 : 0];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (r0 + u * stride < m && e[u] > s[u] && s[u] > 0) cnt += a[u] < b[u];
        }
    }
    // one plain store per workgroup, summed by rows_sorted_finish_kernel: same-address atomics serialise in L2 (~10 ns
    // each; one atomicAdd per wavefront of a 3072-workgroup grid was 130 us of a 180 us kernel)
    __shared__ unsigned wsum[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counters[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// out[0] = sum of the first nb_entries partial counts (D), out[1] = sum of the rest (S)
__global__ __launch_bounds__(256)
void rows_sorted_finish_kernel(const unsigned *__restrict__ partial, int nb, int nb_entries, unsigned *__restrict__ out)
{
    __shared__ unsigned wsum[2][4];
    unsigned d = 0, s = 0;
    for (int i = threadIdx.x; i < nb; i += 256) { const unsigned v = partial[i]; if (i < nb_entries) d += v; else s += v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d += __shfl_xor(d, off, 64); s += __shfl_xor(s, off, 64); }
    if (lane_id() == 0) { wsum[0][threadIdx.x >> 6] = d; wsum[1][threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
        out[1] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
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
    const int nb_entries = 2048, nb_rows = (int)std::min<int64_t>(512, mx::ceil_div(m, 256));
    unsigned *partial = (unsigned *)mx::scratch_buffer(mx::MX_SCRATCH_PARTIALS, (size_t)(nb_entries + nb_rows) * sizeof(unsigned));
    MX_REQUIRE(partial, "mxd_csr_rows_sorted: cannot allocate the partial counts");
    if (((uintptr_t)indices & 15) == 0)
        hipLaunchKernelGGL((mx::rows_sorted_count_kernel<true>), dim3(nb_entries + nb_rows), dim3(256), 0, st, m, indptr,
                           indices, partial, nb_entries);
    else
        hipLaunchKernelGGL((mx::rows_sorted_count_kernel<false>), dim3(nb_entries + nb_rows), dim3(256), 0, st, m, indptr,
                           indices, partial, nb_entries);
    hipLaunchKernelGGL(mx::rows_sorted_finish_kernel, dim3(1), dim3(256), 0, st, partial, nb_entries + nb_rows, nb_entries,
                       (unsigned *)workspace4);                              // [0] descents, [1] descents at row starts
    MX_LAUNCH_CHECK();
    uint32_t counts[2] = {0, 0};
    if (mx::read_back_small(counts, workspace4, sizeof(counts), st)) return 1;
    *flag_host = counts[0] == counts[1];
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
