// spmm.hip — CSR x dense SpMM for gfx950 (MI355X): the device-level entry points and the choice of kernel.
//
// Replaces the two OpenMP loops of the reference:
//   gemm_csr_drm_as_drm  src/matmul.cpp:118-142  (C row-major)
//   gemm_csr_drm_as_dcm  src/matmul.cpp:150-185  (C column-major, what R needs)
// Four kernels: spmm_rowwave.hip (v1, any operands: tiny products), spmm_slab.hip (v2: one-slab
// products of very short rows), spmm_plan.hip (v3: many rows against a wide B), spmm_rowsplit.hip (round 4: few or long
// rows, narrow B); AUTO chooses with a cost model fitted to a map of 228 shapes (spmm_auto_cost below).
#include "spmm_common.h"
#include <algorithm>
#include <cmath>

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

static thread_local const float *g_profile = nullptr;
ProfileScope::ProfileScope(const float *p) : saved(g_profile) { g_profile = p; }
ProfileScope::~ProfileScope() { g_profile = saved; }
double profile_mass(double top, int K)
{
    if (top <= 0.0) return 0.0;
    if (top >= (double)K) return 1.0;
    if (!g_profile || g_profile[39] < 0.0f) return top / (double)K;   // uniform columns
    const double l = std::log2(top < 1.0 ? 1.0 : top);
    const int i = (int)l;
    if (i >= 31) return g_profile[31];
    const double f = l - i, v = g_profile[i] + (g_profile[i + 1] - g_profile[i]) * f;
    return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}
double profile_cv() { return g_profile ? (double)g_profile[32] : 0.0; }
double profile_longest_over_mean() { return g_profile ? (double)g_profile[33] : 0.0; }
bool profile_long_rows(double *entries, double *rows)
{
    if (!g_profile || g_profile[37] != 1.0f) return false;            // ([37] = 1: [35] / [36] were filled by this build's passes)
    *entries = g_profile[35]; *rows = g_profile[36];
    return true;
}
bool profile_in_scope() { return g_profile != nullptr; }
// a profile that says what no profile says (uniform columns are recognised by the marker in [39]; equal rows)
const float *uniform_profile()
{
    static float u[MX_PROFILE_LEN] = {0};
    u[39] = -1.0f;
    return u;
}

static thread_local const char *g_last_spmm_kernel = "none";
void set_last_spmm_kernel(const char *name) { g_last_spmm_kernel = name; }

}  // namespace mx

namespace mx {
void small_stage_release();                  // api.hip: the small-call staging blocks (pinned host + device twin)
// this thread's grow-only scratch (AUTO's plan, the slab-major copy of B, the export scratch, the small-call blocks);
// re-created on demand
void release_thread_workspaces()
{
    plan_auto_release();
    slab_pack_workspace(0, true);
    scratch_release();
    small_stage_release();
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
// AUTO's choice, in one place.  Rounds 1-3 had one rule (PLANNED when B outgrows an XCD's L2 and m * n >= 2^24, else
// ROWWAVE) read off the headline configuration; round 4 mapped every kernel over 228 (later 272) shapes between the benchmarks
// (tools/auto_map.py -> profiles/r04_auto_map.json) and replaced it by a small cost model of the two kernels that
// matter there — the row-split kernel (column panels by launches) and the planned panel sweep — in the quantities that
// bound them on MI355X:
//   * both gather nnz rows of B through the CUs' L1s; what they read from is an XCD's L2 (4 MiB) when the panel of B they
//     work on fits it (~28 TB/s of line reads measured in the row-split kernel, 17 with 8-lane groups; 23.5 TB/s in the
//     planned sweep with one panel, 19 TB/s with several), the Infinity Cache otherwise (~8.5 TB/s); a uniformly gathered
//     panel of T bytes hits L2 with probability 4 MiB / T (guide: "Indexed rows: gather into LDS");
//   * the row-split kernel pays ~0.2 ns per (row, panel, column pass) — one wavefront each; its row-group form, several
//     rows per wavefront, 0 - 0.15 ns — and ~6 us per launch (rowsplit_est_us, spmm_rowsplit.hip, which also decides
//     between the panels its sizes suggest and none);
//     the planned kernel ~0.005 ns per (row, slab, panel), ~15 us of fixed cost per run, and, when the plan is not kept,
//     ~36 us + 6 ps per entry to build it; its persistent grid fills with (octet of 64 rows, slab) pairs — rate x
//     pairs / (pairs + 2400) — and below 32k rows it is never chosen (at m = 1e4 the row-split kernel won 38 of 40
//     points, the other two within 18 %).
// keep_plan: the caller keeps the plan across products (DeviceCSR, the exports' CSR cache) — the build is not charged.
// Callers that do not know nnz get the old rule.  Products below 2^22 multiply-adds are launch-bound whatever runs and
// stay on the row-wave kernel, whose sums are the reference's storage-order FMA chain bit for bit.
struct AutoCost { double rowsplit_us, planned_us, tile_us; int panels; };
static AutoCost spmm_auto_cost(int m, int n, int K, int64_t nnz, int sz, bool keep_plan, int colmajor = 0, bool rows_sorted = false)
{
    const double avg = m > 0 ? (double)nnz / m : 0.0;
    const int cols_per_line = 128 / sz;
    const double slabs = (double)((n + cols_per_line - 1) / cols_per_line);
    // (the hit rate of the gather is the MASS of the entries whose row of the 128-byte slab an XCD's L2 holds — 32,768 rows per
    // panel —, which for uniform columns is the share of the panel's bytes: 4 MiB / panel bytes)
    auto rate = [&](double panels, double l2_rate, double mall_rate) {         // bytes per us
        const double hit = profile_mass(32768.0 * panels, K);
        return 1e6 / (hit / l2_rate + (1.0 - hit) / mall_rate);
    };
    AutoCost c;
    c.panels = rowsplit_panels(m, n, K, sz, avg);
    c.rowsplit_us = rowsplit_est_us(m, n, K, sz, avg, c.panels);        // (spmm_rowsplit.hip: the model next to the heuristics it prices)
    const double plan_panels = std::max(1.0, std::ceil((double)K * 128.0 / 2.5e6));
    // the sweep's persistent grid fills with the number of (octet of 64 rows, slab) pairs: 1,563 of them (m = 1e5, one slab)
    // ran at 9 TB/s, 6,250 at 17.7, 12,500 at 19, 125,000 at 23.8
    const double pairs = std::ceil(m / 64.0) * slabs, fill = pairs / (pairs + 2400.0);
    const double sweep_rate = (plan_panels > 1.0 ? 19.0 : 23.5) * fill;
    c.planned_us = (double)nnz * slabs * 128.0 / rate(plan_panels, sweep_rate, 8.5 * fill) +
                   0.005e-3 * m * slabs * plan_panels + 15.0 +
                   // (rows of uneven length: the count pass reads the indices, octets are dealt and flagged — tools/zipf_map.py,
                   // m = 1e6, 33 per row, log-normal sigma 1: rebuilt 0.77 ms against 0.28 with the plan kept)
                   (keep_plan ? 0.0 : 36.0 + (profile_cv() > 0.2 ? 15e-6 : 6e-6) * (double)nnz);
    // the LDS-tile kernel (spmm_tile.hip, round 5): its own model; a caller that does not vouch for column-sorted rows pays
    // the sortedness pass (one read of the indices) on top
    c.tile_us = tile_est_us(m, n, K, sz, avg, colmajor, nullptr) + (rows_sorted ? 0.0 : 3.0 + (double)nnz * 4.0 / 3e6);
    return c;
}
static int spmm_auto_algo(int m, int n, int K, int64_t nnz, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc,
                          int colmajor, bool keep_plan)
{
    const bool ok = dense_dtype == MX_F64 ? slab_ok<double>(n, (const double *)B, ldb, (const double *)C, ldc, colmajor)
                                          : slab_ok<float>(n, (const float *)B, ldb, (const float *)C, ldc, colmajor);
    const int sz = dense_dtype == MX_F64 ? 8 : 4;
    const size_t b_bytes = (size_t)K * (size_t)n * sz;
    const bool big = ok && b_bytes > ((size_t)8 << 20) && (long long)m * n >= (1LL << 24);
    if (nnz < 0) return big ? (K < (1 << 25) ? MX_SPMM_PLANNED : MX_SPMM_SLAB) : MX_SPMM_ROWWAVE;     // rounds 1-3
    // (tiny products are launch-bound whatever runs — unless their rows are long: the row-wave kernel walks a row's entries
    // one dependent read after the other; m = 1000, 200 per row, n = 16: 0.074 ms there, 0.010 in the row-split kernel; 50 per row: 0.023 and 0.012)
    if (nnz * (long long)n < (1LL << 22) && nnz <= 32LL * m) return MX_SPMM_ROWWAVE;
    const bool tile_can = nnz <= (1LL << 29) && (dense_dtype == MX_F64 ? tile_ok<double>(n, (const double *)B, ldb) : tile_ok<float>(n, (const float *)B, ldb));
    if (!ok || m < 32768) {
        if (!tile_can) return MX_SPMM_ROWSPLIT;
        const AutoCost c = spmm_auto_cost(m, n, K, nnz, sz, keep_plan, colmajor, keep_plan);
        return c.tile_us * 1.1 < c.rowsplit_us ? MX_SPMM_TILE : MX_SPMM_ROWSPLIT;
    }
    if (K >= (1 << 25)) return big ? MX_SPMM_SLAB : MX_SPMM_ROWSPLIT;           // (the plan's 32-bit slab offsets)
    // one 128-byte slab, a handful of entries per row, a slab-panel of B that fits L2: 8 lanes per row and 8 rows per
    // wavefront (the slab kernel's shape) beat one wavefront per row and a plan that costs more than the product
    // (m = 1e6, 8 entries per row, n = 16: slab 0.096 ms, planned rebuilt 0.164, row-split 0.296; plan kept 0.064)
    // — unless the row-split kernel's row-group form applies (round 4, later: 0.061 ms there, and bit for bit the reference's sums)
    if (!keep_plan && n * sz <= 128 && nnz <= 16LL * m && (double)K * 128.0 <= 2.5e6 &&
        rowsplit_segments(m, n, sz, (double)nnz / m) != 0) return MX_SPMM_SLAB;
    const AutoCost c = spmm_auto_cost(m, n, K, nnz, sz, keep_plan, colmajor, keep_plan);
    if (tile_can && c.tile_us * 1.1 < std::min(c.planned_us, c.rowsplit_us)) return MX_SPMM_TILE;
    return c.planned_us < c.rowsplit_us ? MX_SPMM_PLANNED : MX_SPMM_ROWSPLIT;
}
// the row-split kernel's segments per row and column panels for the WHOLE product the calling thread is about to run block
// by block (spmm_block): column-major C, several panels, several segments and narrow lane groups all reassociate, so a
// geometry re-derived from every block's own m, n and nnz would make the last bits of one product depend on how the export
// pipeline, or the device list, cut it (round 4's advisor finding).  Returned BY VALUE (struct SpmmFamily, spmm_common.h) and
// handed to spmm_block as an argument: the sharded export runs its blocks on the ShardPool's worker threads, where round 5's
// thread_local copy of this geometry was never set (round 5's advisor finding; tests/test_gpu_export_path.py)
SpmmFamily spmm_auto_family(int m, int n, int K, int64_t nnz, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc,
                            int colmajor)
{
    SpmmFamily f;
    f.family = spmm_auto_algo(m, n, K, nnz, dense_dtype, B, ldb, C, ldc, colmajor, false);
    // (the caller's profile is in scope here, not in the blocks — and a PLANNED family whose plan is left for its imbalance
    // falls back to the row-split kernel in spmm_block: exactly the products with very long rows)
    f.lh = nnz > 0 && m > 0 ? rowsplit_long_hint(m, nnz) : LongHint();
    f.tile_cv = (float)profile_cv();
    if (f.family == MX_SPMM_ROWSPLIT && nnz >= 0 && m > 0) {
        const int sz = dense_dtype == MX_F64 ? 8 : 4;
        const double avg = (double)nnz / m;
        f.panels = rowsplit_panels(m, n, K, sz, avg);
        const int S = rowsplit_segments(m, n, sz, avg / f.panels);
        f.segments = S == 0 ? -1 : S;            // (run_rowsplit's code for the row-group form)
    }
    return f;
}
// entries of a device-resident CSR whose count the caller did not pass: indptr[m] (one 4-byte copy, synchronises `st`)
static int device_nnz(const int32_t *indptr, int m, hipStream_t st, int64_t *nnz)
{
    int32_t last = 0, first = 0;
    MX_HIP(hipMemcpyAsync(&last, indptr + m, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MX_HIP(hipMemcpyAsync(&first, indptr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MX_HIP(hipStreamSynchronize(st));
    *nnz = (int64_t)last - first;
    return 0;
}
// lh: the long-rows path as chosen for the WHOLE product (spmm_auto_family); piece -1 = from the profile in scope
static int run_rowsplit(int m, int n, int K, int64_t nnz, int segments, int panels, int rows_sorted_hint, const int32_t *indptr,
                        const int32_t *indices, const double *values, const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype,
                        int colmajor, hipStream_t st, LongHint lh_family = {-1, -1, -1})
{
    (void)rows_sorted_hint;                    // (the cursor kernel checks every row itself)
    const int sz = dense_dtype == MX_F64 ? 8 : 4;
    // segments: 1 / 2 / 4 / 8 wavefronts per row, -1 = the row-group form (several rows per wavefront), 0 = chosen here
    if (segments == 0 || panels <= 0) {
        if (nnz < 0 && device_nnz(indptr, m, st, &nnz)) return 1;
        const double avg = m > 0 ? (double)nnz / m : 0.0;
        if (panels <= 0) panels = rowsplit_panels(m, n, K, sz, avg);
        if (segments == 0) segments = rowsplit_segments(m, n, sz, avg / panels);
    }
    if (segments < 0) segments = 0;           // rowsplit_spmm's code for the row-group form
    const LongHint lh = nnz <= 0 ? LongHint() : (lh_family.piece >= 0 ? lh_family : rowsplit_long_hint(m, nnz));   // (off without a profile)
    // (mxd_spmm_last_kernel: set by the launch itself, spmm_rowsplit.hip launch_one — the form can still change there)
    if (dense_dtype == MX_F64)
        return rowsplit_spmm<double>(m, n, K, segments, panels, indptr, indices, values, (const double *)B, ldb, (double *)C, ldc, colmajor, st,
                                     lh, nnz);
    return rowsplit_spmm<float>(m, n, K, segments, panels, indptr, indices, values, (const float *)B, ldb, (float *)C, ldc, colmajor, st,
                                lh, nnz);
}
// One block (rows or columns) of a product whose kernel family was chosen for the WHOLE product (the export pipelines).
// from_auto: the family is AUTO's choice — a planned block still falls back to the row-wave kernel when its plan would
// pad too much, as AUTO does.
int spmm_block(const SpmmFamily &fam, bool from_auto, int m, int n, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices,
               const double *values, const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype, int colmajor, int npanels,
               hipStream_t st)
{
    int family = fam.family;
    if (family == MX_SPMM_ROWSPLIT)
        return run_rowsplit(m, n, K, nnz, from_auto ? fam.segments : 0, from_auto ? fam.panels : npanels, 0, indptr, indices, values,
                            B, ldb, C, ldc, dense_dtype, colmajor, st, from_auto ? fam.lh : LongHint{-1, -1, -1});
    if (family == MX_SPMM_TILE && from_auto) {
        TileDealScope deal(fam.tile_cv);
        return mxd_spmm_csr_dense_ex2(m, n, K, nnz, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor, family, 0, 0, 0, st);
    }
    if (family == MX_SPMM_PLANNED && from_auto) {
        const bool ok = dense_dtype == MX_F64 ? slab_ok<double>(n, (const double *)B, ldb, (const double *)C, ldc, colmajor)
                                              : slab_ok<float>(n, (const float *)B, ldb, (const float *)C, ldc, colmajor);
        if (ok && K < (1 << 25)) {
            bool ready = false;
            if (plan_auto_build(m, K, indptr, indices, values, npanels, st, 1.55, &ready)) return 1;
            if (ready && nnz >= 0 && plan_auto_imbalance(n, dense_dtype == MX_F64 ? 8 : 4) > plan_max_imbalance(dense_dtype == MX_F64 ? 8 : 4)) ready = false;
            if (ready) return plan_auto_run(n, B, ldb, C, ldc, dense_dtype, colmajor, st);
        }
        family = nnz >= 0 ? MX_SPMM_ROWSPLIT : MX_SPMM_ROWWAVE;       // (very uneven rows: the plan would pad too much)
        if (family == MX_SPMM_ROWSPLIT)
            return run_rowsplit(m, n, K, nnz, 0, 0, 0, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor, st,
                                from_auto ? fam.lh : LongHint{-1, -1, -1});
    }
    return mxd_spmm_csr_dense_ex2(m, n, K, nnz, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor, family, 0, npanels, 0, st);
}
}  // namespace mx

extern "C" int mxd_spmm_auto_algo2(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, const void *B, size_t ldb,
                                   const void *C, size_t ldc, int colmajor_out, int *algo)
{
    MX_REQUIRE(algo, "mxd_spmm_auto_algo: null pointer");
    MX_REQUIRE(dense_dtype == MX_F64 || dense_dtype == MX_F32, "mxd_spmm_auto_algo: unsupported dense dtype %d", dense_dtype);
    *algo = mx::spmm_auto_algo(m, n, K, nnz, dense_dtype, B, ldb, C, ldc, colmajor_out, keep_plan != 0);
    return 0;
}
// the model's two estimates (microseconds) and the row-split kernel's panel count, for tools/auto_map.py
extern "C" int mxd_spmm_auto_cost(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, double *rowsplit_us, double *planned_us,
                                  int *panels)
{
    MX_REQUIRE(nnz >= 0 && rowsplit_us && planned_us && panels, "mxd_spmm_auto_cost: nnz and three result pointers are required");
    const mx::AutoCost c = mx::spmm_auto_cost(m, n, K, nnz, dense_dtype == MX_F64 ? 8 : 4, keep_plan != 0);
    *rowsplit_us = c.rowsplit_us; *planned_us = c.planned_us; *panels = c.panels;
    return 0;
}
extern "C" int mxd_spmm_auto_cost2(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, int colmajor_out, int rows_sorted,
                                   double *rowsplit_us, double *planned_us, double *tile_us, int *panels, int *tile_cpl)
{
    MX_REQUIRE(nnz >= 0 && rowsplit_us && planned_us && tile_us && panels && tile_cpl, "mxd_spmm_auto_cost2: nnz and five result pointers are required");
    const int sz = dense_dtype == MX_F64 ? 8 : 4;
    const mx::AutoCost c = mx::spmm_auto_cost(m, n, K, nnz, sz, keep_plan != 0, colmajor_out, rows_sorted != 0);
    *rowsplit_us = c.rowsplit_us; *planned_us = c.planned_us; *tile_us = c.tile_us; *panels = c.panels;
    mx::tile_est_us(m, n, K, sz, m > 0 ? (double)nnz / m : 0.0, colmajor_out, tile_cpl);
    return 0;
}
extern "C" int mxd_spmm_auto_algo3(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, const void *B, size_t ldb, const void *C,
                                   size_t ldc, int colmajor_out, const float *profile, int *algo)
{
    mx::ProfileScope scope(profile);
    return mxd_spmm_auto_algo2(m, n, K, nnz, keep_plan, dense_dtype, B, ldb, C, ldc, colmajor_out, algo);
}
extern "C" int mxd_spmm_auto_cost3(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, int colmajor_out, int rows_sorted,
                                   const float *profile, double *rowsplit_us, double *planned_us, double *tile_us, int *panels, int *tile_cpl)
{
    mx::ProfileScope scope(profile);
    return mxd_spmm_auto_cost2(m, n, K, nnz, keep_plan, dense_dtype, colmajor_out, rows_sorted, rowsplit_us, planned_us, tile_us, panels, tile_cpl);
}
extern "C" int mxd_spmm_csr_dense_ex3(int m, int n, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                                      const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype, int colmajor_out, int algo,
                                      int rows_sorted, int npanels, int wg_per_cu, const float *profile, void *stream)
{
    mx::ProfileScope scope(profile);
    return mxd_spmm_csr_dense_ex2(m, n, K, nnz, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor_out, algo, rows_sorted, npanels,
                                  wg_per_cu, stream);
}
extern "C" int mxd_spmm_auto_algo(int m, int n, int K, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc,
                                  int colmajor_out, int *algo)
{
    return mxd_spmm_auto_algo2(m, n, K, -1, 0, dense_dtype, B, ldb, C, ldc, colmajor_out, algo);
}

extern "C" int mxd_spmm_csr_dense_ex(int m, int n, int K,
                                     const int32_t *indptr, const int32_t *indices, const double *values,
                                     const void *B, size_t ldb, void *C, size_t ldc,
                                     int dense_dtype, int colmajor_out, int algo, int rows_sorted,
                                     int npanels, int wg_per_cu, void *stream)
{
    return mxd_spmm_csr_dense_ex2(m, n, K, -1, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor_out, algo, rows_sorted,
                                  npanels, wg_per_cu, stream);
}

extern "C" int mxd_spmm_csr_dense_ex2(int m, int n, int K, int64_t nnz,
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
    // One-shot AUTO on a matrix nobody profiled: where the planned sweep is a candidate (it syncs to size its plan anyway) the
    // choice between it and the gather kernels hangs on how popular the hot columns are — tools/zipf_map.py: power-law columns,
    // real-sim's shape, n = 128: planned rebuilt 0.257 ms, row-split 0.163 — so the ~40 us profile pass comes first.
    float own_profile[MX_PROFILE_LEN];
    // (~25 us: only for products the model prices at 0.2 ms or more)
    // (and only up to 2^24 columns: the pass clears and scans 8 K bytes of counters — beyond that the columns count as uniform)
    bool profile_here = algo == MX_SPMM_AUTO && !mx::profile_in_scope() && nnz >= (1LL << 21) && m >= 32768 && indices && K <= (1 << 24);
    if (profile_here) {
        const mx::AutoCost c0 = mx::spmm_auto_cost(m, n, K, nnz, dense_dtype == MX_F64 ? 8 : 4, false, colmajor_out, rows_sorted != 0);
        profile_here = std::min(c0.rowsplit_us, c0.planned_us) >= 200.0;
    }
    if (profile_here) {
        void *ws = mx::scratch_buffer(mx::MX_SCRATCH_PROFILE, mxd_csr_profile_workspace_bytes(K));
        mx::scratch_acquire(mx::MX_SCRATCH_PROFILE, st);
        const int rc = ws ? mxd_csr_profile(m, K, nnz, indptr, indices, own_profile, ws, st) : 1;
        if (ws) mx::scratch_done(mx::MX_SCRATCH_PROFILE, st);
        if (rc) return mxd_spmm_csr_dense_ex3(m, n, K, nnz, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor_out, algo, rows_sorted,
                                              npanels, wg_per_cu, mx::uniform_profile(), stream);       // (no memory for the counters: sizes only)
        return mxd_spmm_csr_dense_ex3(m, n, K, nnz, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor_out, algo, rows_sorted, npanels,
                                      wg_per_cu, own_profile, stream);
    }
    if (algo == MX_SPMM_AUTO) {
        // AUTO rebuilds the plan on every call (nothing is assumed about A between calls); callers that multiply one
        // matrix repeatedly keep a plan (mxd_spmm_plan_create_auto + mxd_spmm_plan_run)
        algo = mx::spmm_auto_algo(m, n, K, nnz, dense_dtype, B, ldb, C, ldc, colmajor_out, false);
        auto_pick_planned = algo == MX_SPMM_PLANNED;
        if (algo == MX_SPMM_SLAB) { npanels = 1; if (wg_per_cu <= 0) wg_per_cu = 4; }
        if (algo == MX_SPMM_ROWSPLIT) { npanels = 0; wg_per_cu = 0; }   // (column panels / segments per row: AUTO picks both)
        if (algo == MX_SPMM_TILE) { npanels = 0; wg_per_cu = 0; }       // (geometry and slab width: the tile kernel's own choice)
    }
    if (algo == MX_SPMM_ROWSPLIT)
        return mx::run_rowsplit(m, n, K, nnz, wg_per_cu, npanels, rows_sorted, indptr, indices, values, B, ldb, C, ldc, dense_dtype,
                                colmajor_out, st);
    if (algo == MX_SPMM_TILE) {
        // (nnz unknown: indptr[m] is read back, as for the row-split kernel — the 2^29-entry limit and the slab-width model need it)
        if (nnz < 0 && mx::device_nnz(indptr, m, st, &nnz)) return 1;
        mx::set_last_spmm_kernel("spmm_tile_kernel");
        if (dense_dtype == MX_F64)
            return mx::tile_spmm<double>(m, n, K, nnz, wg_per_cu, npanels, rows_sorted, indptr, indices, values, (const double *)B, ldb,
                                         (double *)C, ldc, colmajor_out, st);
        return mx::tile_spmm<float>(m, n, K, nnz, wg_per_cu, npanels, rows_sorted, indptr, indices, values, (const float *)B, ldb,
                                    (float *)C, ldc, colmajor_out, st);
    }
    if (algo == MX_SPMM_PLANNED) {
        MX_REQUIRE(ok, "mxd_spmm_csr_dense_ex: operands do not meet the planned kernel's 16-byte alignment rules");
        // Measured with log-normal row lengths (tools/skew_probe.py): up to ~1.55x the CSR the planned sweep still
        // beats the row-wave kernel even with the plan built per call; beyond that AUTO stops after the sizing pass
        // (count + scan, ~0.1 ms) and uses the row-wave kernel.
        bool ready = false;          // the plan's buffers are re-used from call to call (grow-only, per thread)
        if (mx::plan_auto_build(m, K, indptr, indices, values, npanels, st, auto_pick_planned ? 1.55 : 0.0, &ready)) return 1;
        // (one octet that outlasts the rest of the sweep: the row-split kernel and its long-rows path instead)
        if (ready && auto_pick_planned && nnz >= 0 && mx::plan_auto_imbalance(n, dense_dtype == MX_F64 ? 8 : 4) > mx::plan_max_imbalance(dense_dtype == MX_F64 ? 8 : 4))
            ready = false;
        if (ready) return mx::plan_auto_run(n, B, ldb, C, ldc, dense_dtype, colmajor_out, stream);
        if (nnz >= 0)                                                // (very uneven rows: the plan would pad too much)
            return mx::run_rowsplit(m, n, K, nnz, 0, 0, 0, indptr, indices, values, B, ldb, C, ldc, dense_dtype, colmajor_out, st);
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

// the imbalance (mxd_spmm_plan_imbalance) above which AUTO leaves a kept plan for the row-split kernel, by element type
extern "C" double mxd_spmm_plan_imbalance_limit(int dense_dtype) { return mx::plan_max_imbalance(dense_dtype == MX_F64 ? 8 : 4); }
