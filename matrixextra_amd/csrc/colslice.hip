// colslice.hip — column-filtering CSR slices for gfx950.
//
// Replaces:
//   copy_csr_rows_col_seq_template   src/slice.cpp:326-383   X[rows, c0:c1]
//   copy_csr_arbitrary_template      src/slice.cpp:449-578   X[rows, cols]   (hash map + per-row argsort)
//   reverse_columns_inplace          src/slice.cpp:142-170
// (reverse_rows_template, src/slice.cpp:49-95, is the row gather of gather.hip with the row list n-1..0.)
//
// Same skeleton as the row gather: per-row output lengths -> exclusive scan -> fill.  Inside a row the kept
// entries must keep their input order, so the fill pass compacts with an in-group prefix (ballot for the 0/1
// column-range predicate, shuffle scan for the multiplicities of the arbitrary selector).
// The arbitrary selector uses a dense column map (start[c]..start[c+1] = positions of column c in cols_take,
// ascending) instead of the reference's hash map; rows are re-ordered by new column id afterwards unless
// cols_take is non-decreasing, as the reference does.
#include "mx_common.h"

namespace mx {

int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace,
                       hipStream_t st);
size_t scan_workspace_bytes(int64_t n);

constexpr int CS_BLOCK = 256;

template <int G>
__device__ __forceinline__ unsigned long long cs_group_ballot(bool pred)
{
    const unsigned long long b = __ballot(pred);
    if constexpr (G == 64) return b;
    else return (b >> (lane_id() & ~(G - 1))) & ((1ULL << G) - 1ULL);
}

// ---- column range -----------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(CS_BLOCK)
void colrange_count_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                           const int32_t *__restrict__ rows, int min_col, int max_col, int32_t *__restrict__ lens)
{
    const int lg = threadIdx.x % G;
    const long long i = (long long)blockIdx.x * (CS_BLOCK / G) + threadIdx.x / G;
    const bool valid = i < r;
    int s = 0, e = 0;
    if (valid) { const int row = rows[i]; s = indptr[row]; e = indptr[row + 1]; }
    int cnt = 0;
    for (int k0 = s; k0 < e; k0 += G) {
        const int k = k0 + lg;
        bool keep = false;
        if (k < e) { const int c = indices[k]; keep = c >= min_col && c <= max_col; }
        cnt += __popcll(cs_group_ballot<G>(keep));
    }
    if (valid && lg == 0) lens[i] = cnt;
}

// KIND: MX_NONE no values, MX_F64, MX_LGL (int32 -> double, slice.cpp:363,373)
template <int G, int KIND>
__global__ __launch_bounds__(CS_BLOCK)
void colrange_fill_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                          const void *__restrict__ values, const int32_t *__restrict__ rows, int min_col, int max_col,
                          const int32_t *__restrict__ new_indptr, int32_t *__restrict__ new_indices,
                          double *__restrict__ new_values)
{
    const int lg = threadIdx.x % G;
    const long long i = (long long)blockIdx.x * (CS_BLOCK / G) + threadIdx.x / G;
    const bool valid = i < r;
    int s = 0, e = 0, o = 0;
    if (valid) { const int row = rows[i]; s = indptr[row]; e = indptr[row + 1]; o = new_indptr[i]; }
    const unsigned long long below = (1ULL << lg) - 1ULL;
    for (int k0 = s; k0 < e; k0 += G) {
        const int k = k0 + lg;
        bool keep = false;
        int c = 0;
        if (k < e) { c = indices[k]; keep = c >= min_col && c <= max_col; }
        const unsigned long long kb = cs_group_ballot<G>(keep);
        if (keep) {
            const int pos = o + __popcll(kb & below);
            new_indices[pos] = c - min_col;
            if constexpr (KIND == MX_F64) new_values[pos] = ((const double *)values)[k];
            else if constexpr (KIND == MX_LGL) new_values[pos] = (double)((const int32_t *)values)[k];
        }
        o += __popcll(kb);
    }
}

// ---- arbitrary column selector: dense map ------------------------------------------------------------------
__global__ __launch_bounds__(256)
void colmap_count_kernel(const int32_t *__restrict__ cols, int64_t n, int ncol_map, int32_t *__restrict__ cnt)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int c = cols[p];
        if (c >= 0 && c < ncol_map) atomicAdd(&cnt[c], 1);
    }
}
__global__ __launch_bounds__(256)
void colmap_place_kernel(const int32_t *__restrict__ cols, int64_t n, int ncol_map, const int32_t *__restrict__ start,
                         int32_t *__restrict__ cursor, int32_t *__restrict__ pos)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int c = cols[p];
        if (c >= 0 && c < ncol_map) pos[start[c] + atomicAdd(&cursor[c], 1)] = (int)p;
    }
}
// repeated columns: put their positions in ascending order (lists are short: insertion sort by one lane)
__global__ __launch_bounds__(256)
void colmap_order_kernel(int ncol_map, const int32_t *__restrict__ start, int32_t *__restrict__ pos)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncol_map; c += gridDim.x * blockDim.x) {
        const int s = start[c], e = start[c + 1];
        for (int i = s + 1; i < e; i++) {
            const int key = pos[i];
            int k = i - 1;
            while (k >= s && pos[k] > key) { pos[k + 1] = pos[k]; k--; }
            pos[k + 1] = key;
        }
    }
}

template <int G>
__global__ __launch_bounds__(CS_BLOCK)
void colmap_rows_count_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                              const int32_t *__restrict__ rows, int ncol_map, const int32_t *__restrict__ start,
                              int32_t *__restrict__ lens)
{
    const int lg = threadIdx.x % G;
    const long long i = (long long)blockIdx.x * (CS_BLOCK / G) + threadIdx.x / G;
    const bool valid = i < r;
    int s = 0, e = 0;
    if (valid) { const int row = rows[i]; s = indptr[row]; e = indptr[row + 1]; }
    int cnt = 0;
    for (int k = s + lg; k < e; k += G) {
        const int c = indices[k];
        if (c >= 0 && c < ncol_map) cnt += start[c + 1] - start[c];
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, G);
    if (valid && lg == 0) lens[i] = cnt;
}

template <int G, typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(CS_BLOCK)
void colmap_rows_fill_kernel(int r, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                             const VT *__restrict__ values, const int32_t *__restrict__ rows, int ncol_map,
                             const int32_t *__restrict__ start, const int32_t *__restrict__ pos,
                             const int32_t *__restrict__ new_indptr, int32_t *__restrict__ new_indices,
                             VT *__restrict__ new_values)
{
    const int lg = threadIdx.x % G;
    const long long i = (long long)blockIdx.x * (CS_BLOCK / G) + threadIdx.x / G;
    const bool valid = i < r;
    int s = 0, e = 0, o = 0;
    if (valid) { const int row = rows[i]; s = indptr[row]; e = indptr[row + 1]; o = new_indptr[i]; }
    for (int k0 = s; k0 < e; k0 += G) {            // e - s is uniform inside the group
        const int k = k0 + lg;
        int q0 = 0, mult = 0;
        if (k < e) {
            const int c = indices[k];
            if (c >= 0 && c < ncol_map) { q0 = start[c]; mult = start[c + 1] - q0; }
        }
        int incl = mult;                            // inclusive scan of the multiplicities inside the group
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            const int up = __shfl_up(incl, off, G);
            if (lg >= off) incl += up;
        }
        const int total = __shfl(incl, G - 1, G);
        int dst = o + incl - mult;
        for (int t = 0; t < mult; t++) {
            new_indices[dst + t] = pos[q0 + t];
            if constexpr (HAS_VALUES) new_values[dst + t] = values[k];
        }
        o += total;
    }
}

// ---- reverse columns in place: col -> ncol-1-col and reverse each row (pairwise swaps, race-free) -----------
template <int G, typename VT, bool HAS_VALUES>
__global__ __launch_bounds__(CS_BLOCK)
void reverse_columns_kernel(int m, const int32_t *__restrict__ indptr, int32_t *__restrict__ indices,
                            VT *__restrict__ values, int ncol)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (CS_BLOCK / G) + threadIdx.x / G;
    if (row >= m) return;
    const int s = indptr[row], e = indptr[row + 1];
    const int len = e - s;
    for (int a = lg; a < (len + 1) / 2; a += G) {
        const int ia = s + a, ib = e - 1 - a;
        const int ca = ncol - indices[ia] - 1;
        if (ia == ib) { indices[ia] = ca; continue; }
        const int cb = ncol - indices[ib] - 1;
        indices[ia] = cb; indices[ib] = ca;
        if constexpr (HAS_VALUES) { const VT va = values[ia]; values[ia] = values[ib]; values[ib] = va; }
    }
}

__global__ __launch_bounds__(256)
void reversed_iota_kernel(int n, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = n - 1 - i;
}

static inline size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }

}  // namespace mx

#define MX_GROUP_SWITCH(G, ...)                                                              \
    switch (G) {                                                                             \
        case 4:  { constexpr int GG = 4;  __VA_ARGS__; break; }                              \
        case 8:  { constexpr int GG = 8;  __VA_ARGS__; break; }                              \
        case 16: { constexpr int GG = 16; __VA_ARGS__; break; }                              \
        case 32: { constexpr int GG = 32; __VA_ARGS__; break; }                              \
        case 64: { constexpr int GG = 64; __VA_ARGS__; break; }                              \
        default: return mx::set_error("bad lane-group size %d", G);                          \
    }

static int finish_count(int r, int32_t *lens, int32_t *new_indptr, void *scan_ws, int64_t *nnz_out_host, hipStream_t st)
{
    int64_t *total_dev = (int64_t *)scan_ws;
    const int rc = mx::exclusive_scan_i32(lens, r, new_indptr, total_dev, scan_ws, st);
    if (rc) return rc;
    if (nnz_out_host) {
        if (mx::read_back_small(nnz_out_host, total_dev, sizeof(int64_t), st)) return 1;
        MX_REQUIRE(*nnz_out_host <= (int64_t)INT_MAX, "result has %lld entries: exceeds R's int32 index range",
                   (long long)*nnz_out_host);
    }
    return 0;
}

// lanes per picked row: the source rows' mean length is what matters (nnz_src / nrows_src)
static int group_for(double avg) { return mx::pick_group(avg); }

extern "C" int mxd_csr_colrange_count(int r, const int32_t *indptr, const int32_t *indices, const int32_t *rows_take,
                                      int min_col, int max_col, double avg_row_len, int32_t *new_indptr,
                                      void *workspace, int64_t *nnz_out_host, void *stream)
{
    MX_REQUIRE(r >= 0 && new_indptr && workspace, "mxd_csr_colrange_count: bad arguments");
    hipStream_t st = mx::as_stream(stream);
    int32_t *lens = (int32_t *)workspace;
    void *scan_ws = (char *)workspace + mx::pad16((size_t)(r > 0 ? r : 1) * sizeof(int32_t));
    if (r > 0) {
        const int G = group_for(avg_row_len);
        MX_GROUP_SWITCH(G, hipLaunchKernelGGL((mx::colrange_count_kernel<GG>), dim3((unsigned)mx::ceil_div(r, mx::CS_BLOCK / GG)),
                                              dim3(mx::CS_BLOCK), 0, st, r, indptr, indices, rows_take, min_col, max_col, lens));
        MX_LAUNCH_CHECK();
    }
    return finish_count(r, lens, new_indptr, scan_ws, nnz_out_host, st);
}

extern "C" int mxd_csr_colrange_fill(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                                     int value_dtype, const int32_t *rows_take, int min_col, int max_col,
                                     double avg_row_len, const int32_t *new_indptr, int32_t *new_indices,
                                     double *new_values, void *stream)
{
    MX_REQUIRE(r >= 0, "mxd_csr_colrange_fill: negative r");
    if (r == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const int G = group_for(avg_row_len);
#define MX_CR_LAUNCH(KIND)                                                                                          \
    MX_GROUP_SWITCH(G, hipLaunchKernelGGL((mx::colrange_fill_kernel<GG, KIND>), dim3((unsigned)mx::ceil_div(r, mx::CS_BLOCK / GG)), \
                                          dim3(mx::CS_BLOCK), 0, st, r, indptr, indices, values, rows_take, min_col, \
                                          max_col, new_indptr, new_indices, new_values))
    switch (value_dtype) {
        case MX_F64: MX_CR_LAUNCH(MX_F64); break;
        case MX_LGL: MX_CR_LAUNCH(MX_LGL); break;
        case MX_NONE: MX_CR_LAUNCH(MX_NONE); break;
        default: return mx::set_error("mxd_csr_colrange_fill: unsupported value dtype %d", value_dtype);
    }
#undef MX_CR_LAUNCH
    MX_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t mxd_colmap_workspace_bytes(int ncol_map)
{
    // [cursor int32[ncol_map]][counts int32[ncol_map]][scan workspace]
    return 2 * mx::pad16((size_t)(ncol_map > 0 ? ncol_map : 1) * sizeof(int32_t)) + mx::scan_workspace_bytes(ncol_map);
}

extern "C" int mxd_colmap_build(const int32_t *cols_take, int64_t n, int ncol_map, int32_t *start, int32_t *pos,
                                void *workspace, void *stream)
{
    MX_REQUIRE(n >= 0 && ncol_map >= 0 && start && workspace, "mxd_colmap_build: bad arguments");
    hipStream_t st = mx::as_stream(stream);
    const size_t seg = mx::pad16((size_t)(ncol_map > 0 ? ncol_map : 1) * sizeof(int32_t));
    int32_t *cursor = (int32_t *)workspace;
    int32_t *cnt = (int32_t *)((char *)workspace + seg);
    void *scan_ws = (char *)workspace + 2 * seg;
    MX_HIP(hipMemsetAsync(workspace, 0, 2 * seg, st));
    const unsigned grid = (unsigned)(mx::ceil_div(n > 0 ? n : 1, 256) < 1024 ? mx::ceil_div(n > 0 ? n : 1, 256) : 1024);
    if (n > 0) {
        hipLaunchKernelGGL(mx::colmap_count_kernel, dim3(grid), dim3(256), 0, st, cols_take, n, ncol_map, cnt);
        MX_LAUNCH_CHECK();
    }
    const int rc = mx::exclusive_scan_i32(cnt, ncol_map, start, nullptr, scan_ws, st);
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(mx::colmap_place_kernel, dim3(grid), dim3(256), 0, st, cols_take, n, ncol_map, start, cursor, pos);
        MX_LAUNCH_CHECK();
        const unsigned g2 = (unsigned)(mx::ceil_div(ncol_map, 256) < 2048 ? mx::ceil_div(ncol_map, 256) : 2048);
        hipLaunchKernelGGL(mx::colmap_order_kernel, dim3(g2), dim3(256), 0, st, ncol_map, start, pos);
        MX_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mxd_csr_colmap_count(int r, const int32_t *indptr, const int32_t *indices, const int32_t *rows_take,
                                    int ncol_map, const int32_t *start, double avg_row_len, int32_t *new_indptr,
                                    void *workspace, int64_t *nnz_out_host, void *stream)
{
    MX_REQUIRE(r >= 0 && new_indptr && workspace, "mxd_csr_colmap_count: bad arguments");
    hipStream_t st = mx::as_stream(stream);
    int32_t *lens = (int32_t *)workspace;
    void *scan_ws = (char *)workspace + mx::pad16((size_t)(r > 0 ? r : 1) * sizeof(int32_t));
    if (r > 0) {
        const int G = group_for(avg_row_len);
        MX_GROUP_SWITCH(G, hipLaunchKernelGGL((mx::colmap_rows_count_kernel<GG>), dim3((unsigned)mx::ceil_div(r, mx::CS_BLOCK / GG)),
                                              dim3(mx::CS_BLOCK), 0, st, r, indptr, indices, rows_take, ncol_map, start, lens));
        MX_LAUNCH_CHECK();
    }
    return finish_count(r, lens, new_indptr, scan_ws, nnz_out_host, st);
}

extern "C" int mxd_csr_colmap_fill(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                                   int value_dtype, const int32_t *rows_take, int ncol_map, const int32_t *start,
                                   const int32_t *pos, double avg_row_len, const int32_t *new_indptr,
                                   int32_t *new_indices, void *new_values, void *stream)
{
    MX_REQUIRE(r >= 0, "mxd_csr_colmap_fill: negative r");
    if (r == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const int G = group_for(avg_row_len);
#define MX_CM_LAUNCH(VT, HV)                                                                                         \
    MX_GROUP_SWITCH(G, hipLaunchKernelGGL((mx::colmap_rows_fill_kernel<GG, VT, HV>), dim3((unsigned)mx::ceil_div(r, mx::CS_BLOCK / GG)), \
                                          dim3(mx::CS_BLOCK), 0, st, r, indptr, indices, (const VT *)values, rows_take, \
                                          ncol_map, start, pos, new_indptr, new_indices, (VT *)new_values))
    switch (value_dtype) {
        case MX_F64: MX_CM_LAUNCH(double, true); break;
        case MX_LGL: case MX_I32: MX_CM_LAUNCH(int32_t, true); break;
        case MX_NONE: MX_CM_LAUNCH(int32_t, false); break;
        default: return mx::set_error("mxd_csr_colmap_fill: unsupported value dtype %d", value_dtype);
    }
#undef MX_CM_LAUNCH
    MX_LAUNCH_CHECK();
    return 0;
}

extern "C" int mxd_csr_reverse_columns(int m, int64_t nnz, const int32_t *indptr, int32_t *indices, void *values,
                                       int value_dtype, int ncol, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_csr_reverse_columns: negative m");
    if (m == 0 || nnz == 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const int G = group_for(0.5 * (double)nnz / (double)m);
#define MX_RC_LAUNCH(VT, HV)                                                                                        \
    MX_GROUP_SWITCH(G, hipLaunchKernelGGL((mx::reverse_columns_kernel<GG, VT, HV>), dim3((unsigned)mx::ceil_div(m, mx::CS_BLOCK / GG)), \
                                          dim3(mx::CS_BLOCK), 0, st, m, indptr, indices, (VT *)values, ncol))
    switch (value_dtype) {
        case MX_F64: MX_RC_LAUNCH(double, true); break;
        case MX_LGL: case MX_I32: MX_RC_LAUNCH(int32_t, true); break;
        case MX_NONE: MX_RC_LAUNCH(int32_t, false); break;
        default: return mx::set_error("mxd_csr_reverse_columns: unsupported value dtype %d", value_dtype);
    }
#undef MX_RC_LAUNCH
    MX_LAUNCH_CHECK();
    return 0;
}

extern "C" int mxd_reversed_iota(int n, int32_t *out, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mx::reversed_iota_kernel, dim3((unsigned)mx::ceil_div(n, 256)), dim3(256), 0, mx::as_stream(stream), n, out);
    MX_LAUNCH_CHECK();
    return 0;
}
