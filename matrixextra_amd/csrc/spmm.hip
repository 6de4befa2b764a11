// spmm.hip — CSR x dense SpMM for gfx950 (MI355X): the device-level entry points and the choice of kernel.
//
// Replaces the two OpenMP loops of the reference:
//   gemm_csr_drm_as_drm  src/matmul.cpp:118-142  (C row-major)
//   gemm_csr_drm_as_dcm  src/matmul.cpp:150-185  (C column-major, what R needs)
// Three kernels, one measured progression: spmm_rowwave.hip (v1, any operands), spmm_slab.hip (v2,
// opt-in), spmm_plan.hip (v3, what AUTO runs when B outgrows an XCD's L2).
#include "spmm_common.h"

namespace mx {

// Optional HIP-event ring around the dominant kernel of every SpMM launch (bench.py's roofline figure): events sit
// on the launch stream right before / after the kernel, nothing else in between.
struct KernelTimer {
    static constexpr int N = 256;
    hipEvent_t a[N], b[N];
    bool made = false, on = false;
    int count = 0;
};
// per thread and per device: events belong to the device that was current when they were created
static thread_local KernelTimer g_kts[16];
static inline KernelTimer &kt_cur()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    return g_kts[d];
}
void kt_begin(hipStream_t st)
{
    KernelTimer &kt = kt_cur();
    if (!kt.on || kt.count >= KernelTimer::N) return;
    if (!kt.made) {
        for (int i = 0; i < KernelTimer::N; i++) { (void)hipEventCreate(&kt.a[i]); (void)hipEventCreate(&kt.b[i]); }
        kt.made = true;
    }
    (void)hipEventRecord(kt.a[kt.count], st);
}
void kt_end(hipStream_t st)
{
    KernelTimer &kt = kt_cur();
    if (!kt.on || kt.count >= KernelTimer::N) return;
    (void)hipEventRecord(kt.b[kt.count], st);
    kt.count++;
}

static thread_local const char *g_last_spmm_kernel = "none";
void set_last_spmm_kernel(const char *name) { g_last_spmm_kernel = name; }

}  // namespace mx

namespace mx {
// this thread's grow-only scratch (AUTO's plan, the slab-major copy of B, the export scratch); re-created on demand
void release_thread_workspaces()
{
    plan_auto_release();
    slab_pack_workspace(0, true);
    scratch_release();
}
}  // namespace mx
// the calling thread's workspaces and every idle block of the pool (pool.hip) go back to the device
extern "C" int mxd_release_workspaces(void)
{
    mx::release_thread_workspaces();
    mx::pool_trim();
    return 0;
}

extern "C" int mxd_spmm_kernel_timing(int enable)
{
    mx::KernelTimer &kt = mx::kt_cur();
    kt.on = enable != 0;
    kt.count = 0;
    return 0;
}
// elapsed ms of the dominant kernel of each SpMM launch since mxd_spmm_kernel_timing(1) (synchronises on the events)
extern "C" int mxd_spmm_kernel_times(float *out_ms, int max_out, int *count)
{
    MX_REQUIRE(count, "mxd_spmm_kernel_times: null count pointer");
    mx::KernelTimer &kt = mx::kt_cur();
    const int n = kt.count < max_out ? kt.count : max_out;
    for (int i = 0; i < n; i++) {
        MX_HIP(hipEventSynchronize(kt.b[i]));
        MX_HIP(hipEventElapsedTime(&out_ms[i], kt.a[i], kt.b[i]));
    }
    *count = n;
    kt.count = 0;
    return 0;
}

extern "C" const char *mxd_spmm_last_kernel(void) { return mx::g_last_spmm_kernel; }

namespace mx {
// AUTO's choice, in one place: PLANNED when B outgrows one XCD's L2, the operands meet the 16-byte rules and there is
// enough work to fill the persistent grid (measured on MI355X, headline config, profiles/r01_*: row-wave 4.45 ms; one-panel
// slab kernel on the slab-major copy of B 4.05 ms; planned panel sweep 1.74 ms + 0.27 ms to build the plan from plain CSR),
// SLAB for K >= 2^25 (the plan's 32-bit slab offsets), else ROWWAVE.
static int spmm_auto_algo(int m, int n, int K, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc, int colmajor)
{
    const bool ok = dense_dtype == MX_F64 ? slab_ok<double>(n, (const double *)B, ldb, (const double *)C, ldc, colmajor)
                                          : slab_ok<float>(n, (const float *)B, ldb, (const float *)C, ldc, colmajor);
    const size_t b_bytes = (size_t)K * (size_t)n * (dense_dtype == MX_F64 ? 8 : 4);
    const bool big = ok && b_bytes > ((size_t)8 << 20) && (long long)m * n >= (1LL << 24);
    return big ? (K < (1 << 25) ? MX_SPMM_PLANNED : MX_SPMM_SLAB) : MX_SPMM_ROWWAVE;
}
int spmm_auto_family(int m, int n, int K, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc, int colmajor)
{
    return spmm_auto_algo(m, n, K, dense_dtype, B, ldb, C, ldc, colmajor);
}
// One block (rows or columns) of a product whose kernel family was chosen for the WHOLE product (the export pipelines).
// from_auto: the family is AUTO's choice — a planned block still falls back to the row-wave kernel when its plan would
// pad too much, as AUTO does.
int spmm_block(int family, bool from_auto, int m, int n, int K, const int32_t *indptr, const int32_t *indices, const double *values,
               const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype, int colmajor, int npanels, hipStream_t st)
{
    if (family == MX_SPMM_PLANNED && from_auto) {
        const bool ok = dense_dtype == MX_F64 ? slab_ok<double>(n, (const double *)B, ldb, (const double *)C, ldc, colmajor)
                                              : slab_ok<float>(n, (const float *)B, ldb, (const float *)C, ldc, colmajor);
        if (ok && K < (1 << 25)) {
            bool ready = false;
            if (plan_auto_build(m, K, indptr, indices, values, npanels, st, 1.55, &ready)) return 1;
            if (ready) return plan_auto_run(n, B, ldb, C, ldc, dense_dtype, colmajor, st);
        }
        family = MX_SPMM_ROWWAVE;
    }
    return mxd_spmm_csr_dense_ex(m, n, K, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor, family, 0, npanels, 0, st);
}
}  // namespace mx

extern "C" int mxd_spmm_auto_algo(int m, int n, int K, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc,
                                  int colmajor_out, int *algo)
{
    MX_REQUIRE(algo, "mxd_spmm_auto_algo: null pointer");
    MX_REQUIRE(dense_dtype == MX_F64 || dense_dtype == MX_F32, "mxd_spmm_auto_algo: unsupported dense dtype %d", dense_dtype);
    *algo = mx::spmm_auto_algo(m, n, K, dense_dtype, B, ldb, C, ldc, colmajor_out);
    return 0;
}

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
    bool auto_pick_planned = false;
    if (algo == MX_SPMM_AUTO) {
        // AUTO rebuilds the plan on every call (nothing is assumed about A between calls); callers that multiply one
        // matrix repeatedly keep a plan (mxd_spmm_plan_create_auto + mxd_spmm_plan_run)
        algo = mx::spmm_auto_algo(m, n, K, dense_dtype, B, ldb, C, ldc, colmajor_out);
        auto_pick_planned = algo == MX_SPMM_PLANNED;
        if (algo == MX_SPMM_SLAB) { npanels = 1; if (wg_per_cu <= 0) wg_per_cu = 4; }
    }
    if (algo == MX_SPMM_PLANNED) {
        MX_REQUIRE(ok, "mxd_spmm_csr_dense_ex: operands do not meet the planned kernel's 16-byte alignment rules");
        // Measured with log-normal row lengths (tools/skew_probe.py): up to ~1.55x the CSR the planned sweep still
        // beats the row-wave kernel even with the plan built per call; beyond that AUTO stops after the sizing pass
        // (count + scan, ~0.1 ms) and uses the row-wave kernel.
        bool ready = false;          // the plan's buffers are re-used from call to call (grow-only, per thread)
        if (mx::plan_auto_build(m, K, indptr, indices, values, npanels, st, auto_pick_planned ? 1.55 : 0.0, &ready)) return 1;
        if (ready) return mx::plan_auto_run(n, B, ldb, C, ldc, dense_dtype, colmajor_out, stream);
        algo = MX_SPMM_ROWWAVE;
    }
    if (algo == MX_SPMM_SLAB) {
        MX_REQUIRE(ok, "mxd_spmm_csr_dense_ex: operands do not meet the slab kernel's 16-byte alignment rules");
        mx::set_last_spmm_kernel("spmm_slab_kernel");
        if (npanels <= 0) npanels = rows_sorted ? mx::pick_panels(K, (size_t)2560 << 10) : 1;
        if (!rows_sorted) npanels = 1;                 // panels need column-sorted rows
        if (wg_per_cu <= 0) wg_per_cu = 4;
        if (dense_dtype == MX_F64)
            return mx::slab_spmm<double>(m, n, K, indptr, indices, values, (const double *)B, ldb, (double *)C,
                                                ldc, colmajor_out, npanels, wg_per_cu, st);
        return mx::slab_spmm<float>(m, n, K, indptr, indices, values, (const float *)B, ldb, (float *)C, ldc,
                                           colmajor_out, npanels, wg_per_cu, st);
    }
    mx::set_last_spmm_kernel("spmm_rowwave_kernel");
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
        return mx::rowwave_spmm<double>(m, n, indptr, indices, values, (const double *)B, ldb, (double *)C, ldc,
                                        colmajor_out, st);
    if (dense_dtype == MX_F32)
        return mx::rowwave_spmm<float>(m, n, indptr, indices, values, (const float *)B, ldb, (float *)C, ldc,
                                       colmajor_out, st);
    return mx::set_error("mxd_spmm_csr_dense: unsupported dense dtype %d", dense_dtype);
}
