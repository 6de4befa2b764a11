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
// flag[0] starts at 1 and is cleared by any violating pair.
__global__ __launch_bounds__(256)
void is_seq_kernel(const int32_t *__restrict__ idx, int64_t n, int step, int32_t *__restrict__ flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= idx[i] != idx[i - 1] + step;
    if (__ballot(bad) != 0ULL && lane_id() == 0) atomicAnd(flag, 0);
}

// ---- per-row sortedness / sort -----------------------------------------------------------------
__global__ __launch_bounds__(256)
void rows_sorted_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                        int64_t nnz, int32_t *__restrict__ flag)
{
    // element-parallel: entry k violates if it is not the first of its row and indices[k] < indices[k-1];
    // "first of its row" is found by a binary search of k in indptr.
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        if (indices[k] < indices[k - 1]) {
            // is k a row start?  find the last row whose indptr <= k
            int lo = 0, hi = m;      // indptr[lo] <= k < indptr[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (indptr[mid] <= k) lo = mid; else hi = mid; }
            if (indptr[lo] != k) bad = true;
        }
    }
    if (__ballot(bad) != 0ULL && lane_id() == 0) atomicAnd(flag, 0);
}

// Stable rank sort of each row into tmp: one G-lane group per row, each lane
// ranks its entries against the whole row (rows already non-decreasing are
// copied through, as the reference skips them, misc.cpp:283).
template <int G, typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(GATHER_BLOCK)
void sort_rows_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                      const VT *__restrict__ values, int32_t *__restrict__ tmp_idx, VT *__restrict__ tmp_val)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (GATHER_BLOCK / G) + threadIdx.x / G;
    if (row >= m) return;
    const int s = indptr[row], len = indptr[row + 1] - s;
    const int32_t *__restrict__ keys = indices + s;
    for (int i = lg; i < len; i += G) {
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
static int launch_sort_rows(int G, int m, const int32_t *indptr, const int32_t *indices, const void *values,
                            int32_t *tmp_idx, void *tmp_val, hipStream_t st)
{
#define MX_CASE(GG)                                                                                     \
    case GG: {                                                                                          \
        const unsigned grid = (unsigned)ceil_div(m, GATHER_BLOCK / GG);                                 \
        hipLaunchKernelGGL((sort_rows_kernel<GG, VT, HAS_VALUES>), dim3(grid), dim3(GATHER_BLOCK), 0, st, \
                           m, indptr, indices, (const VT *)values, tmp_idx, (VT *)tmp_val);             \
        break;                                                                                          \
    }
    switch (G) { MX_CASE(4) MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64)
                 default: return set_error("sort: bad group %d", G); }
#undef MX_CASE
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
    int32_t *lens = (int32_t *)workspace;
    const size_t lens_bytes = ((size_t)(r > 0 ? r : 1) * sizeof(int32_t) + 15) & ~(size_t)15;
    void *scan_ws = (char *)workspace + lens_bytes;
    if (r > 0) {
        hipLaunchKernelGGL(mx::gather_lengths_kernel, dim3((unsigned)mx::ceil_div(r, mx::GATHER_BLOCK)),
                           dim3(mx::GATHER_BLOCK), 0, st, r, indptr, rows_take, lens);
        MX_LAUNCH_CHECK();
    }
    int64_t *total_dev = (int64_t *)scan_ws;
    const int rc = mx::exclusive_scan_i32(lens, r, new_indptr, total_dev, scan_ws, st);
    if (rc) return rc;
    if (nnz_out_host) {
        MX_HIP(hipMemcpyAsync(nnz_out_host, total_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        MX_HIP(hipStreamSynchronize(st));
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
    const int32_t one = 1;
    MX_HIP(hipMemcpyAsync(workspace4, &one, sizeof(one), hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)(mx::ceil_div(n, 256) < 2048 ? mx::ceil_div(n, 256) : 2048);
    hipLaunchKernelGGL(mx::is_seq_kernel, dim3(grid), dim3(256), 0, st, idx, n, reversed ? -1 : 1, workspace4);
    MX_LAUNCH_CHECK();
    int32_t flag = 0;
    MX_HIP(hipMemcpyAsync(&flag, workspace4, sizeof(flag), hipMemcpyDeviceToHost, st));
    MX_HIP(hipStreamSynchronize(st));
    *flag_host = flag != 0;
    return 0;
}

extern "C" int mxd_csr_rows_sorted(int m, const int32_t *indptr, const int32_t *indices, int32_t *workspace4,
                                   int *flag_host, void *stream)
{
    MX_REQUIRE(flag_host, "mxd_csr_rows_sorted: null flag pointer");
    if (m <= 0) { *flag_host = 1; return 0; }
    MX_REQUIRE(indptr && workspace4, "mxd_csr_rows_sorted: null pointer");
    hipStream_t st = mx::as_stream(stream);
    int32_t ends[1];
    MX_HIP(hipMemcpyAsync(ends, indptr + m, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MX_HIP(hipStreamSynchronize(st));
    const int64_t nnz = ends[0];
    if (nnz < 2) { *flag_host = 1; return 0; }
    const int32_t one = 1;
    MX_HIP(hipMemcpyAsync(workspace4, &one, sizeof(one), hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)(mx::ceil_div(nnz, 256) < 4096 ? mx::ceil_div(nnz, 256) : 4096);
    hipLaunchKernelGGL(mx::rows_sorted_kernel, dim3(grid), dim3(256), 0, st, m, indptr, indices, nnz, workspace4);
    MX_LAUNCH_CHECK();
    int32_t flag = 0;
    MX_HIP(hipMemcpyAsync(&flag, workspace4, sizeof(flag), hipMemcpyDeviceToHost, st));
    MX_HIP(hipStreamSynchronize(st));
    *flag_host = flag != 0;
    return 0;
}

extern "C" int mxd_csr_sort_rows(int m, int64_t nnz, const int32_t *indptr, int32_t *indices, void *values,
                                 int value_dtype, int32_t *tmp_indices, void *tmp_values, void *stream)
{
    MX_REQUIRE(m >= 0 && nnz >= 0, "mxd_csr_sort_rows: negative size");
    if (m == 0 || nnz == 0) return 0;
    MX_REQUIRE(indptr && indices && tmp_indices, "mxd_csr_sort_rows: null pointer");
    hipStream_t st = mx::as_stream(stream);
    const int G = mx::pick_group((double)nnz / (double)m);
    int rc;
    size_t vbytes = 0;
    switch (value_dtype) {
        case MX_F64: rc = mx::launch_sort_rows<double, true>(G, m, indptr, indices, values, tmp_indices, tmp_values, st);
                     vbytes = 8; break;
        case MX_LGL: case MX_I32:
                     rc = mx::launch_sort_rows<int32_t, true>(G, m, indptr, indices, values, tmp_indices, tmp_values, st);
                     vbytes = 4; break;
        case MX_NONE: rc = mx::launch_sort_rows<int32_t, false>(G, m, indptr, indices, nullptr, tmp_indices, nullptr, st);
                     break;
        default: return mx::set_error("mxd_csr_sort_rows: unsupported value dtype %d", value_dtype);
    }
    if (rc) return rc;
    MX_HIP(hipMemcpyAsync(indices, tmp_indices, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (vbytes) MX_HIP(hipMemcpyAsync(values, tmp_values, (size_t)nnz * vbytes, hipMemcpyDeviceToDevice, st));
    return 0;
}
