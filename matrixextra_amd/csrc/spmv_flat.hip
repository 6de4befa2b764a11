// spmv_flat.hip — CSR x dense-vector SpMV, flat-stream kernel (what MX_SPMV_AUTO runs for large matrices).
//
// Replaces matmul_csr_dvec<> (src/matmul.cpp:381-419) like the lane-group kernel of spmv.hip, with the work cut by
// ENTRIES instead of rows:
//   * a 256-thread workgroup owns the rows that START inside its slice of ~3.8 k consecutive entries (the first row of
//     every slice comes from a small pass over indptr, slice_rows_kernel: a search inside every workgroup cost ~2 k
//     scattered index-pointer lines per workgroup, a quarter of the gather's own line traffic); the slice is read 16 B per
//     lane (indices) + 2 x 16 B per lane (values), 16 entries per thread, all loads issued before the first use;
//   * every entry then gathers its vector element (16 independent gathers per thread in flight) and forms the
//     separately rounded product a * v[j];
//   * the products are staged in LDS and each row is summed by ONE thread in STORAGE ORDER: y = ((0 + p0) + p1) + ...,
//     bit for bit the reference's loop without FMA contraction (float accumulate for the float32 kind, matmul.cpp:403;
//     NA_INTEGER / NA_LOGICAL terms are NA_REAL, :406-411).  Rows of more than 256 entries are summed by a wavefront
//     (reassociated; within 1e-12).  Skewed row lengths cost nothing: the slices are equally long whatever the rows are.
// What bounds it (tools/microbench/spmv_ceiling.hip, profiles/r02_spmv_ceiling.json): not the (j, a) stream (62.8 us
// alone = 6.1 TB/s) but the gather — 32 M random 8-byte reads of a 0.8 MB vector move 32 M 128-byte lines from L2 into
// the CUs' L1 (130 us alone, ~250 G lines/s = the L2's request rate).
#include "spmv_rows.h"

namespace mx {

constexpr int FL_THREADS = 256;
constexpr int FL_EPT = 16;                              // entries per thread
constexpr int FL_ROUNDS = FL_EPT / 4;
constexpr int FL_ROUND_ENTRIES = FL_THREADS * 4;        // 1024
constexpr int FL_CAP = FL_THREADS * FL_EPT;             // 4096 entries per pass
constexpr int FL_TARGET = FL_CAP - 256;                 // slices are cut every 3840 entries (room for the last row's tail)
constexpr int FL_STAGE = FL_CAP + FL_CAP / 32;

// slice_rows[t] = the first row r in [0, m] with indptr[r] >= t * FL_TARGET, for t = 0 .. nslices (one pass over indptr:
// thread r owns the cuts in (indptr[r-1], indptr[r]])
__global__ __launch_bounds__(256)
void slice_rows_kernel(int m, const int32_t *__restrict__ indptr, int32_t *__restrict__ slice_rows, int nslices)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r > m) return;
    const long long a = r == 0 ? -1 : indptr[r - 1], b = indptr[r];
    // floor(a / T) for a >= -1
    long long k = a < 0 ? 0 : a / FL_TARGET + 1;
    const long long k_end = b / FL_TARGET;
    for (; k <= k_end && k <= nslices; k++) slice_rows[k] = (int32_t)r;
    if (r == m)                                                        // cuts past the last entry: no row starts there
        for (long long kk = k_end + 1; kk <= nslices; kk++) slice_rows[kk] = m;
}

template <int KIND>
__global__ __launch_bounds__(FL_THREADS)
void spmv_flat_kernel(int m, long long nnz, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                      const double *__restrict__ values, const void *__restrict__ v_, void *__restrict__ y_,
                      const int32_t *__restrict__ slice_rows)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ double stage[FL_STAGE];
    const int tid = threadIdx.x;
    const long long cut0 = (long long)blockIdx.x * FL_TARGET;
    const long long lim4 = (nnz - 4) & ~3LL;                          // the arrays' last whole aligned quad (nnz >= 4)
    int col[FL_EPT];
    double x[FL_EPT];
    // entries [sb, sb + FL_CAP): clamped offsets, no branch between the loads; ownership is sorted out afterwards
    auto load_pass = [&](long long sb) {
        const bool any_whole = sb <= lim4;                            // uniform
        const int32_t *__restrict__ ip = indices + (any_whole ? sb : 0);
        const double *__restrict__ xp = values + (any_whole ? sb : 0);
        const int lim_rel = any_whole ? (int)min(lim4 - sb, (long long)FL_CAP) : 0;
#pragma unroll
        for (int r = 0; r < FL_ROUNDS; r++) {
            const int off = min(r * FL_ROUND_ENTRIES + tid * 4, lim_rel);
            const i4 c = *reinterpret_cast<const i4 *>(ip + off);
            const d2 a01 = *reinterpret_cast<const d2 *>(xp + off);
            const d2 a23 = *reinterpret_cast<const d2 *>(xp + off + 2);
            col[r * 4 + 0] = c[0]; col[r * 4 + 1] = c[1]; col[r * 4 + 2] = c[2]; col[r * 4 + 3] = c[3];
            x[r * 4 + 0] = a01[0]; x[r * 4 + 1] = a01[1]; x[r * 4 + 2] = a23[0]; x[r * 4 + 3] = a23[1];
        }
    };
    long long sb = cut0 & ~3LL;
    load_pass(sb);                                                    // in flight while the rows are looked up
    const int R0 = slice_rows[blockIdx.x], R1 = slice_rows[blockIdx.x + 1];
    if (R0 >= R1) return;                                             // no row starts in this slice
    const long long E0 = indptr[R0], E1 = indptr[R1];                 // uniform; E0 >= cut0 >= sb
    const int row_first = R0 + tid;
    int rs_first = 0, re_first = 0;
    if (row_first < R1) { rs_first = indptr[row_first]; re_first = indptr[row_first + 1]; }
    RowAcc<KIND> carry;
    int carry_row = -1;
    for (bool first = true;; first = false) {
        const long long he = min(sb + FL_CAP, E1);
        // ownership + the quads that straddle the arrays' end (clamped above): entries outside [max(sb, E0), he) are
        // never read back from the staging area, only their gather has to be safe
        const int head_rel = (int)(max(E0, sb) - sb), n_rel = (int)(he - sb);
        const int lim_rel = sb <= lim4 ? (int)min(lim4 - sb, (long long)FL_CAP) : -1;
        if (sb + FL_CAP > lim4) {                                     // uniform: the pass that holds the arrays' last quad
#pragma unroll
            for (int r = 0; r < FL_ROUNDS; r++) {
                const int er = r * FL_ROUND_ENTRIES + tid * 4;
                if (er > lim_rel && er < n_rel) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (er + q < n_rel) { col[r * 4 + q] = indices[sb + er + q]; x[r * 4 + q] = values[sb + er + q]; }
                }
            }
        }
        // ---- gather + product, 16 independent gathers per thread
        double f[FL_EPT];
#pragma unroll
        for (int r = 0; r < FL_ROUNDS; r++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int er = r * FL_ROUND_ENTRIES + tid * 4 + q;
                const bool mine = er >= head_rel && er < n_rel;
                f[r * 4 + q] = vec_factor<KIND>(v_, mine ? col[r * 4 + q] : 0);
            }
        }
        __syncthreads();                                              // the previous pass is done with the staging area
#pragma unroll
        for (int r = 0; r < FL_ROUNDS; r++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int pos = r * FL_ROUND_ENTRIES + tid * 4 + q;
                stage[pos + (pos >> 5)] = term<KIND>(x[r * 4 + q], f[r * 4 + q]);
            }
        }
        __syncthreads();
        reduce_rows<KIND, FL_THREADS>(stage, sb, he, E0, R0, R1, indptr, rs_first, re_first, carry, carry_row, first, y_);
        sb += FL_CAP;
        if (sb >= E1) break;                                          // uniform
        load_pass(sb);
    }
}

// 16-B aligned index / value arrays (the wide loads) and at least one whole quad
bool spmv_flat_ok(int m, int64_t nnz, const int32_t *indices, const double *values)
{
    return m > 0 && nnz >= 4 && !(((uintptr_t)indices | (uintptr_t)values) & 15);
}

int spmv_flat_launch(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                     const void *v, int v_dtype, void *y, hipStream_t st)
{
    const unsigned grid = (unsigned)(nnz / FL_TARGET + 1);            // slices cover the starts 0 .. nnz (trailing empty rows)
    int32_t *slice_rows = (int32_t *)scratch_buffer(MX_SCRATCH_SPMV_SLICES, ((size_t)grid + 1) * sizeof(int32_t));
    MX_REQUIRE(slice_rows, "spmv: cannot allocate the slice table");
    scratch_acquire(MX_SCRATCH_SPMV_SLICES, st);                     // (a caller that switched streams waits for the previous product)
    hipLaunchKernelGGL(slice_rows_kernel, dim3((unsigned)ceil_div((int64_t)m + 1, 256)), dim3(256), 0, st, m, indptr,
                       slice_rows, (int)grid);
#define MX_FL(KIND)                                                                                         \
    hipLaunchKernelGGL((spmv_flat_kernel<KIND>), dim3(grid), dim3(FL_THREADS), 0, st, m, (long long)nnz,    \
                       indptr, indices, values, v, y, slice_rows)
    switch (v_dtype) {
        case MX_F64: MX_FL(MX_F64); break;
        case MX_I32: MX_FL(MX_I32); break;
        case MX_LGL: MX_FL(MX_LGL); break;
        case MX_F32: MX_FL(MX_F32); break;
        default: return set_error("spmv: unsupported vector dtype %d", v_dtype);
    }
#undef MX_FL
    scratch_done(MX_SCRATCH_SPMV_SLICES, st);
    MX_LAUNCH_CHECK();
    return 0;
}

}  // namespace mx
