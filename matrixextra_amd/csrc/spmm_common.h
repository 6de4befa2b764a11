// spmm_common.h — pieces shared by the three SpMM kernels (spmm_rowwave.hip, spmm_slab.hip, spmm_plan.hip) and the
// dispatcher (spmm.hip): vector load/store helpers, the XCD timing barrier, and the host-side entry points each
// translation unit exposes to the dispatcher.
#pragma once
#include "mx_common.h"
#include <cstdlib>
#include <new>

namespace mx {

constexpr int SLAB_GROUP = 8;                       // lanes that own one 128-byte line of a slab (one row of C)

template <typename T, int N> struct VecT;
template <> struct VecT<double, 1> { using type = double; };
template <> struct VecT<double, 2> { using type = double __attribute__((ext_vector_type(2))); };
template <> struct VecT<float, 1>  { using type = float; };
template <> struct VecT<float, 2>  { using type = float __attribute__((ext_vector_type(2))); };
template <> struct VecT<float, 4>  { using type = float __attribute__((ext_vector_type(4))); };

template <typename real_t, int VEC>
__device__ __forceinline__ void vload(real_t (&dst)[VEC], const real_t *__restrict__ p)
{
    using V = typename VecT<real_t, VEC>::type;
    if constexpr (VEC == 1) {
        dst[0] = *p;
    } else {
        const V v = *reinterpret_cast<const V *>(p);
#pragma unroll
        for (int i = 0; i < VEC; i++) dst[i] = v[i];
    }
}

template <typename real_t, int VEC>
__device__ __forceinline__ void vstore(real_t *__restrict__ p, const real_t (&src)[VEC])
{
    using V = typename VecT<real_t, VEC>::type;
    if constexpr (VEC == 1) {
        *p = src[0];
    } else {
        V v;
#pragma unroll
        for (int i = 0; i < VEC; i++) v[i] = src[i];
        *reinterpret_cast<V *>(p) = v;
    }
}

// streaming store: C is written once and not read again by the kernel — keep it from displacing the packed B in L2 /
// the Infinity Cache (measured on the planned kernel: 2.05 -> 1.98 ms)
template <typename real_t, int VEC>
__device__ __forceinline__ void vstore_nt(real_t *__restrict__ p, const real_t (&src)[VEC])
{
    using V = typename VecT<real_t, VEC>::type;
    if constexpr (VEC == 1) {
        __builtin_nontemporal_store(src[0], p);
    } else {
        V v;
#pragma unroll
        for (int i = 0; i < VEC; i++) v[i] = src[i];
        __builtin_nontemporal_store(v, reinterpret_cast<V *>(p));
    }
}

__device__ __forceinline__ double mx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float mx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Timing-only barrier among the workgroups that share blockIdx % 8 (the XCD group): it keeps them on the same
// column panel so that the panel stays L2-resident.  No data is handed over, so no release/acquire is needed
// and a timeout is harmless: the spin is bounded and falling through only costs locality, never correctness
// (all co-resident by grid sizing; a block that is not resident simply makes the others time out).
__device__ __forceinline__ void xcd_timing_barrier(unsigned *ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 4096)
            __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}

// can the slab / planned kernels take these operands?  (16-B aligned rows of B, whole vectors per row;
// row-major C additionally needs 16-B aligned rows of C)
template <typename real_t>
inline bool slab_ok(int n, const real_t *B, size_t ldb, const real_t *C, size_t ldc, int colmajor)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    if (n < VEC || n % VEC || ldb % VEC || (uintptr_t)B % 16) return false;
    if (!colmajor && (ldc % VEC || (uintptr_t)C % 16)) return false;
    return true;
}

// ---- spmm.hip: the profile of the matrix the calling thread's current AUTO decision is about (mxd_csr_profile, csrc/profile.hip),
// set for the duration of an entry point that was handed one; absent: uniform columns, equal rows.
// profile_mass(top, K): share of the entries whose column is among the `top` hottest columns (what an L2 holding `top` rows
// of B serves); profile_cv(): coefficient of variation of the row lengths
struct ProfileScope { const float *saved; explicit ProfileScope(const float *p); ~ProfileScope(); };
double profile_mass(double top, int K);
double profile_cv();
double profile_longest_over_mean();                       // 0 without a profile in scope
bool profile_long_rows(double *entries, double *rows);    // [35], [36]: entries / number of the rows longer than canonical_long_piece(mean)
bool profile_in_scope();
// The tile kernel's row dealing for a block of a product whose family was chosen for the whole product (the export pipelines, whose
// workers have no profile in scope): cv by value.  Under such a hint long rows are never cut into parts — a block's geometry,
// and with it the part length, depends on the block: the exports keep the storage-order bits whatever the device list.
struct TileDealScope { float saved; explicit TileDealScope(float cv); ~TileDealScope(); };
const float *uniform_profile();
inline double lockstep_factor(double cv, int rows_together)
{
    // a wavefront that walks `rows_together` rows at once runs as long as the longest: E[max of k] / mean ~ 1 + cv * z(k)
    const double z = rows_together >= 8 ? 1.42 : (rows_together >= 4 ? 1.03 : (rows_together >= 2 ? 0.56 : 0.0));
    return 1.0 + cv * z;
}

// ---- spmm.hip: HIP-event ring around the dominant kernel of every launch, name of the last kernel used
void kt_begin(hipStream_t st);
void kt_end(hipStream_t st);
void set_last_spmm_kernel(const char *name);

// ---- spmm_rowwave.hip
template <typename real_t>
int rowwave_spmm(int m, int n, const int32_t *indptr, const int32_t *indices, const double *values,
                 const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream);

// ---- spmm_rowsplit.hip: S segments per row (1, 2, 4 or 8), one wavefront per segment, G lanes per row of B
// and P column panels of B (one launch each)
template <typename real_t>
int rowsplit_spmm(int m, int n, int K, int S, int P, const int32_t *indptr, const int32_t *indices, const double *values,
                  const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream, LongHint lh = LongHint(),
                  long long nnz = -1);                     // lh.piece > 0 (and nnz known): rows longer than that are cut into pieces
LongHint rowsplit_long_hint(int m, long long nnz);        // from the matrix profile in scope, piece 0 = no long rows known
int rowsplit_segments(int m, int n, int dense_bytes, double avg_len);
double rowsplit_est_us(int m, int n, int K, int dense_bytes, double avg_len, int P);
int rowsplit_panels(int m, int n, int K, int dense_bytes, double avg_len);

// ---- spmm_tile.hip: row blocks x 256 / 512-byte column slabs, K-tiles of B staged in LDS by LDS-DMA (dense-ish operands)
template <typename real_t>
bool tile_ok(int n, const real_t *B, size_t ldb);
double tile_est_us(int m, int n, int K, int dense_bytes, double avg_len, int colmajor, int *cpl);
template <typename real_t>
int tile_spmm(int m, int n, int K, int64_t nnz, int variant, int nw, int rows_sorted, const int32_t *indptr, const int32_t *indices,
              const double *values, const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream);

// ---- spmm_slab.hip
int pick_panels(int K, size_t l2_budget);
void *slab_pack_workspace(size_t bytes, bool release = false);     // grow-only per-device scratch for the packed B
unsigned *slab_sync_workspace();                                   // counters of the XCD timing barrier
// B (K x n row-major) -> slab-major [nslabs][Kp][W], zero padded past column n and past row K
template <typename real_t>
int launch_repack(int K, int Kp, int n, const real_t *B, size_t ldb, real_t *Bp, hipStream_t st);
template <typename real_t>
int slab_spmm(int m, int n, int K, const int32_t *indptr, const int32_t *indices, const double *values,
              const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, int npanels, int wg_per_cu,
              hipStream_t stream);

// ---- spmm_plan.hip: the plan AUTO keeps per thread (buffers grow-only, rebuilt on every call)
// builds it; *ready = false when max_pad_ratio > 0 and the plan would hold more than ratio x nnz slots
int plan_auto_build(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values, int npanels,
                    hipStream_t st, double max_pad_ratio, bool *ready);
int plan_auto_run(int n, const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype, int colmajor, void *stream);
double plan_auto_imbalance(int n, int dense_bytes);       // of the plan plan_auto_build just built (spmm_plan.hip plan_imbalance)
// above this AUTO leaves a plan for the row-split kernel.  By element size (round 6, tools/plan_imbalance_probe.py): the tail of an
// unbalanced sweep is step-bound — as long in f32 as in f64 — while the row-split kernel moves half the bytes in f32: f32 products
// are better off there from 2.5 on (2.5-2.8: 0.107-0.13 ms against 0.125-0.19), f64 products only from ~4.5 (2e5 x 5e4, n = 32, imbalance
// 3.1 / 3.9: planned 0.43 / 0.40 ms, row-split 0.57 / 0.59; 3e5 x 1e5 at 5.5: planned 1.21, row-split 0.97)
inline double plan_max_imbalance(int dense_bytes) { return dense_bytes == 8 ? 4.5 : 2.5; }
void plan_auto_release();

// ---- scan.hip
int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace,
                       hipStream_t st);
size_t scan_workspace_bytes(int64_t n);

}  // namespace mx
