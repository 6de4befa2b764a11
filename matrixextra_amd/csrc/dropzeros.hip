// dropzeros.hip — the two CSR utilities either side of the merges, for gfx950:
//
//   remove_zero_valued_csr<>   src/misc.cpp:553-664 (exports :667-698; R caller `remove_zeros`, R/utils.R:286-312): what
//                              follows A - B, whose cancelled entries stay in the result as explicit zeros
//                              (operators.cpp:477-495).  Entries whose value is zero leave the matrix; with remove_NAs the
//                              missing ones too.  Count per row -> scan -> ordered compaction per row.
//   check_valid_csr_matrix     src/misc.cpp:970-1016 (R caller `check_sparse_matrix`, R/utils.R:448): index range and
//                              index-pointer monotony, one pass over each array.
//
// Bandwidth work: 8 (or 4) bytes per entry for the count, 12 (8) in + 12 (8) per kept entry out for the fill.
#include "mx_common.h"
#include "spmm_common.h"

#include <algorithm>
#include <climits>

namespace mx {

constexpr int DZ_BLOCK = 256;

// MODE: 0 numeric, NAs stay; 1 numeric, NAs leave; 2 R logical, NAs stay; 3 R logical, NAs leave.
// `keep` is the predicate of the reference's copy loop, `dirty` that of its first scan (is there anything to remove at
// all?): they differ for MODE 3, where the scan looks for zeros OR NAs but the loop only drops the NAs (misc.cpp:580-582
// against :642) — a logical matrix with zeros and remove_NAs = TRUE keeps its zeros.  Kept as it is.
template <int MODE, typename VT> __device__ __forceinline__ bool dz_keep(VT v)
{
    if constexpr (MODE == 0) return v != 0.0;                      // (NaN != 0: missing values are "true" and stay)
    else if constexpr (MODE == 1) return v != 0.0 && v == v;
    else if constexpr (MODE == 2) return v != 0;
    else return v != MX_NA_INT;
}
template <int MODE, typename VT> __device__ __forceinline__ bool dz_dirty(VT v)
{
    if constexpr (MODE == 0) return v == 0.0;
    else if constexpr (MODE == 1) return v == 0.0 || v != v;
    else if constexpr (MODE == 2) return v == 0;
    else return v == 0 || v == MX_NA_INT;
}

// A lane group counts DZ_ROWS rows at a time: the loads of all of them are issued before any is looked at (a row of 50
// doubles is two loads per lane: one row per group left the memory pipeline mostly empty — 0.39 ms for 0.8 GB).
constexpr int DZ_ROWS = 4;
template <int G, int MODE, typename VT>
__global__ __launch_bounds__(DZ_BLOCK)
void drop_count_kernel(int m, const int32_t *__restrict__ indptr, const VT *__restrict__ values, int32_t *__restrict__ counts,
                       unsigned *__restrict__ dirty)
{
    const int lg = threadIdx.x % G;
    const long long row0 = ((long long)blockIdx.x * (DZ_BLOCK / G) + threadIdx.x / G) * DZ_ROWS;
    int s[DZ_ROWS], n[DZ_ROWS], kept[DZ_ROWS];
    int maxn = 0;
#pragma unroll
    for (int q = 0; q < DZ_ROWS; q++) {
        const long long row = row0 + q;
        s[q] = row < m ? indptr[row] : 0;
        n[q] = row < m ? indptr[row + 1] - s[q] : 0;
        kept[q] = 0;
        maxn = max(maxn, n[q]);
    }
    bool d = false;
    for (int k = lg; k < maxn; k += G) {
        VT v[DZ_ROWS];
#pragma unroll
        for (int q = 0; q < DZ_ROWS; q++) v[q] = k < n[q] ? values[s[q] + k] : VT(1);     // (1: kept and clean, not counted below)
#pragma unroll
        for (int q = 0; q < DZ_ROWS; q++) {
            if (k < n[q]) {
                kept[q] += dz_keep<MODE, VT>(v[q]) ? 1 : 0;
                d |= dz_dirty<MODE, VT>(v[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < DZ_ROWS; q++) {
#pragma unroll
        for (int sft = G / 2; sft > 0; sft >>= 1) kept[q] += __shfl_xor(kept[q], sft, MX_WAVE);
        if (lg == 0 && row0 + q < m) counts[row0 + q] = kept[q];
    }
    // (one flag for the whole grid: same-address atomics serialise at ~10 ns each, and 2M wavefronts setting it cost 20 ms
    // at cfg4 size — a wavefront first looks whether somebody else has already done it)
    if (__ballot(d) != 0ull && lane_id() == 0 && __hip_atomic_load(dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        atomicOr(dirty, 1u);
}

// (the fill with DZ_ROWS rows per group in flight like the count was measured too: slower, 0.91 against 0.73 ms per call —
// four rows' worth of scattered stores per round; it stays at one row per group)
template <int G, int MODE, typename VT>
__global__ __launch_bounds__(DZ_BLOCK)
void drop_fill_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const VT *__restrict__ values,
                      const int32_t *__restrict__ out_indptr, int32_t *__restrict__ out_indices, VT *__restrict__ out_values)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (DZ_BLOCK / G) + threadIdx.x / G;
    // (no early return: the ballots below want every lane of the wave)
    const int s0 = row < m ? indptr[row] : 0, e = row < m ? indptr[row + 1] : 0;
    int base = row < m ? out_indptr[row] : 0;
    const int shift = (lane_id() / G) * G;                         // this group's bits inside the wave's ballot
    constexpr unsigned long long gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
    // the groups of a wave run as many rounds as the longest of their rows
    int rounds = (e - s0 + G - 1) / G;
#pragma unroll
    for (int s = 32; s >= G && s > 0; s >>= 1) rounds = max(rounds, __shfl_xor(rounds, s, MX_WAVE));
    for (int r = 0; r < rounds; r++) {
        const int k = s0 + r * G + lg;
        const bool valid = k < e;
        VT v = VT(0);
        int j = 0;
        if (valid) { v = values[k]; j = indices[k]; }
        const bool keep = valid && dz_keep<MODE, VT>(v);
        const unsigned long long mine = (__ballot(keep) >> shift) & gmask;
        if (keep) {
            const int pos = base + __popcll(mine & ((1ull << lg) - 1ull));
            out_indices[pos] = j;
            out_values[pos] = v;
        }
        base += __popcll(mine);
    }
}

template <int MODE, typename VT>
static int launch_drop_count(int G, int m, const int32_t *indptr, const void *values, int32_t *counts, unsigned *dirty, hipStream_t st)
{
    const VT *x = (const VT *)values;
#define MX_DZ_COUNT(GG)                                                                                                      \
    hipLaunchKernelGGL((drop_count_kernel<GG, MODE, VT>), dim3((unsigned)ceil_div((long long)m, (DZ_BLOCK / GG) * DZ_ROWS)), dim3(DZ_BLOCK), 0, \
                       st, m, indptr, x, counts, dirty)
    switch (G) {
        case 4: MX_DZ_COUNT(4); break;
        case 8: MX_DZ_COUNT(8); break;
        case 16: MX_DZ_COUNT(16); break;
        case 32: MX_DZ_COUNT(32); break;
        default: MX_DZ_COUNT(64); break;
    }
#undef MX_DZ_COUNT
    MX_LAUNCH_CHECK();
    return 0;
}

template <int MODE, typename VT>
static int launch_drop_fill(int G, int m, const int32_t *indptr, const int32_t *indices, const void *values, const int32_t *out_indptr,
                            int32_t *out_indices, void *out_values, hipStream_t st)
{
    const VT *x = (const VT *)values;
    VT *ox = (VT *)out_values;
#define MX_DZ_FILL(GG)                                                                                                       \
    hipLaunchKernelGGL((drop_fill_kernel<GG, MODE, VT>), dim3((unsigned)ceil_div((long long)m, DZ_BLOCK / GG)), dim3(DZ_BLOCK), 0,  \
                       st, m, indptr, indices, x, out_indptr, out_indices, ox)
    switch (G) {
        case 4: MX_DZ_FILL(4); break;
        case 8: MX_DZ_FILL(8); break;
        case 16: MX_DZ_FILL(16); break;
        case 32: MX_DZ_FILL(32); break;
        default: MX_DZ_FILL(64); break;
    }
#undef MX_DZ_FILL
    MX_LAUNCH_CHECK();
    return 0;
}

// lanes per row (MXGPU_DROP_G: A/B runs)
static int drop_group(int m, int64_t nnz)
{
    static const int forced = [] { const char *e = getenv("MXGPU_DROP_G"); return e ? atoi(e) : 0; }();
    if (forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
    // half the lanes the mean row length would fill: two rounds per row keep more rows — more independent loads — in a
    // wavefront (cfg4 size, 50 per row: 0.86 ms with 32 lanes, 1.11 with 64; 12 per row: 0.31 with 8, 0.38 with 16)
    return nnz < 0 ? 16 : std::max(4, pick_group((double)nnz / (double)(m > 0 ? m : 1)) / 2);
}

// ---- check_valid_csr_matrix -------------------------------------------------------------------------------------
// out[0] = min index, out[1] = max index, out[2] = NA in the index pointer, out[3] = index pointer decreases somewhere
__global__ __launch_bounds__(256)
void csr_valid_kernel(int m, int64_t nnz, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int *__restrict__ out)
{
    int lo = INT_MAX, hi = INT_MIN;
    bool na = false, dec = false;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t; i < nnz; i += stride) {
        const int j = indices[i];
        lo = min(lo, j);
        hi = max(hi, j);
    }
    for (int64_t r = t; r <= m; r += stride) {
        const int a = indptr[r];
        na |= a == MX_NA_INT;
        if (r < m) dec |= a > indptr[r + 1];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        lo = min(lo, __shfl_xor(lo, s, MX_WAVE));
        hi = max(hi, __shfl_xor(hi, s, MX_WAVE));
    }
    const bool any_na = __ballot(na) != 0ull, any_dec = __ballot(dec) != 0ull;
    // one set of atomics per workgroup (same-address atomics serialise: ~10 ns each)
    __shared__ int s_lo[4], s_hi[4], s_fl[4];
    const int w = threadIdx.x / MX_WAVE;
    if (lane_id() == 0) { s_lo[w] = lo; s_hi[w] = hi; s_fl[w] = (any_na ? 1 : 0) | (any_dec ? 2 : 0); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int fl = 0;
        for (int k = 0; k < 4; k++) { lo = min(lo, s_lo[k]); hi = max(hi, s_hi[k]); fl |= s_fl[k]; }
        if (lo != INT_MAX) atomicMin(out + 0, lo);
        if (hi != INT_MIN) atomicMax(out + 1, hi);
        if (fl & 1) atomicOr(out + 2, 1);
        if (fl & 2) atomicOr(out + 3, 1);
    }
}

__global__ void csr_valid_init_kernel(int *out) { out[0] = INT_MAX; out[1] = INT_MIN; out[2] = 0; out[3] = 0; }

}  // namespace mx

// workspace: [int64 total][uint32 dirty, pad][int32 counts[m]][scan workspace]
extern "C" size_t mxd_csr_drop_workspace_bytes(int m)
{
    const size_t mm = (size_t)(m > 0 ? m : 1);
    return 16 + ((4 * mm + 15) & ~(size_t)15) + mx::scan_workspace_bytes((int64_t)mm) + 16;
}

extern "C" int mxd_csr_drop_count(int m, int64_t nnz, const int32_t *indptr, const void *values, int value_dtype, int remove_NAs,
                                  int32_t *out_indptr, void *workspace, int64_t *nnz_out_host, int *dirty_host, void *stream)
{
    MX_REQUIRE(m >= 0 && indptr && out_indptr && workspace, "mxd_csr_drop_count: bad arguments");
    MX_REQUIRE(value_dtype == MX_F64 || value_dtype == MX_LGL, "mxd_csr_drop_count: values must be f64 or R logical");
    hipStream_t st = mx::as_stream(stream);
    int64_t *total_dev = (int64_t *)workspace;
    unsigned *dirty = (unsigned *)((char *)workspace + 8);
    int32_t *counts = (int32_t *)((char *)workspace + 16);
    void *scan_ws = (char *)workspace + 16 + ((4 * (size_t)(m > 0 ? m : 1) + 15) & ~(size_t)15);
    MX_HIP(hipMemsetAsync(workspace, 0, 16, st));
    if (m == 0) MX_HIP(hipMemsetAsync(out_indptr, 0, sizeof(int32_t), st));
    else {
        const int G = mx::drop_group(m, nnz);
        int rc;
        if (value_dtype == MX_F64)
            rc = remove_NAs ? mx::launch_drop_count<1, double>(G, m, indptr, values, counts, dirty, st)
                            : mx::launch_drop_count<0, double>(G, m, indptr, values, counts, dirty, st);
        else
            rc = remove_NAs ? mx::launch_drop_count<3, int32_t>(G, m, indptr, values, counts, dirty, st)
                            : mx::launch_drop_count<2, int32_t>(G, m, indptr, values, counts, dirty, st);
        if (rc) return rc;
        if (mx::exclusive_scan_i32(counts, m, out_indptr, total_dev, scan_ws, st)) return 1;
    }
    if (nnz_out_host || dirty_host) {
        long long both[2] = {0, 0};
        if (mx::read_back_small(both, workspace, 16, st)) return 1;
        if (nnz_out_host) *nnz_out_host = both[0];
        if (dirty_host) *dirty_host = (int)(both[1] & 0xFFFFFFFFll) != 0;
    }
    return 0;
}

extern "C" int mxd_csr_drop_fill(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const void *values,
                                 int value_dtype, int remove_NAs, const int32_t *out_indptr, int32_t *out_indices,
                                 void *out_values, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_csr_drop_fill: negative m");
    MX_REQUIRE(value_dtype == MX_F64 || value_dtype == MX_LGL, "mxd_csr_drop_fill: values must be f64 or R logical");
    if (m == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const int G = mx::drop_group(m, nnz);
    if (value_dtype == MX_F64)
        return remove_NAs ? mx::launch_drop_fill<1, double>(G, m, indptr, indices, values, out_indptr, out_indices, out_values, st)
                          : mx::launch_drop_fill<0, double>(G, m, indptr, indices, values, out_indptr, out_indices, out_values, st);
    return remove_NAs ? mx::launch_drop_fill<3, int32_t>(G, m, indptr, indices, values, out_indptr, out_indices, out_values, st)
                      : mx::launch_drop_fill<2, int32_t>(G, m, indptr, indices, values, out_indptr, out_indices, out_values, st);
}

// 0 valid; 1 negative index, 2 index >= ncols, 4 NA in the index pointer, 5 index pointer not monotone — in the reference's
// order of checks (its third check, NA among the indices, can never fire: NA_INTEGER is INT_MIN and fails the first).
// `flags_dev`: 4 ints of device scratch.
extern "C" int mxd_csr_check_valid(int m, int ncols, int64_t nnz, const int32_t *indptr, const int32_t *indices, int *flags_dev,
                                   int *code_host, void *stream)
{
    MX_REQUIRE(m >= 0 && indptr && flags_dev && code_host, "mxd_csr_check_valid: bad arguments");
    hipStream_t st = mx::as_stream(stream);
    hipLaunchKernelGGL(mx::csr_valid_init_kernel, dim3(1), dim3(1), 0, st, flags_dev);
    const long long work = (long long)std::max<int64_t>(nnz, (int64_t)m + 1);
    const unsigned grid = (unsigned)std::min<long long>(2048, std::max<long long>(1, mx::ceil_div(work, (long long)(256 * 8))));
    hipLaunchKernelGGL(mx::csr_valid_kernel, dim3(grid), dim3(256), 0, st, m, nnz, indptr, indices, flags_dev);
    MX_LAUNCH_CHECK();
    int f[4];
    if (mx::read_back_small(f, flags_dev, sizeof(f), st)) return 1;
    int code = 0;
    if (nnz > 0 && f[0] < 0) code = 1;
    else if (nnz > 0 && f[1] >= ncols) code = 2;
    else if (f[2]) code = 4;
    else if (f[3]) code = 5;
    *code_host = code;
    return 0;
}
