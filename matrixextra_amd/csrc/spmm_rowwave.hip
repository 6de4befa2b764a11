// spmm_rowwave.hip — CSR x dense SpMM for gfx950 (MI355X): the row-wave kernel (v1), any operands.
//
// Replaces the two OpenMP loops of the reference:
//   gemm_csr_drm_as_drm  src/matmul.cpp:118-142  (C row-major)
//   gemm_csr_drm_as_dcm  src/matmul.cpp:150-185  (C column-major, what R needs)
//
// Design (v1, "row-wave"): a workgroup of 4 wavefronts owns a tile of TR
// consecutive rows; each wavefront walks TR/4 rows.  For one row the 64 lanes
// span 64*VEC consecutive columns of the output, so every nonzero a_ij turns
// into ONE fully coalesced read of B[j, slab] (1 KiB for f64 n=128 / f32
// n=256: 16 B per lane) with the row base address in SGPRs (v_readlane of the
// column id), followed by VEC FMAs per lane.  (j, a) of the row are loaded
// coalesced by the wave, kept in one VGPR pair and broadcast lane by lane; the
// next row's first chunk is prefetched while the current row streams B.
// Accumulation runs in CSR storage order, one FMA per nonzero per column —
// the same order as the reference's axpy loop — so results differ from an
// FMA-enabled BLAS in nothing and from a non-FMA one by one rounding per term.
//
// Column-major epilogue: the tile's rows are parked in LDS (row stride odd ->
// conflict-free transposed reads) and written out as TR-row-contiguous
// segments per output column (256 B for f64, TR=32), instead of the
// stride-m scatter the CPU code does with dcopy.
//
// Roofline: HBM-bound.  Algorithmic bytes per launch =
//   4(m+1) + 12 nnz + s*K*n + s*m*n   (SURVEY §8d).  The gather of B rows is
// served by L2 / Infinity Cache (B = 102 MB for the headline config).
#include "spmm_common.h"

namespace mx {

constexpr int SPMM_WAVES = 4;     // wavefronts per workgroup
constexpr int SPMM_UNROLL = 8;    // B-row reads in flight per wavefront

// one chunk of <=64 nonzeros of the current row: lane k holds (jv, av) of entry k.
// B is wave-uniform and `col` a per-lane element offset, so each read is
// "SGPR row base + VGPR lane offset" (global_load ... s[base], no 64-bit VALU
// address arithmetic per nonzero).  Lanes past the last column read a clamped,
// valid column instead of branching; their results are never stored.
template <typename real_t, int VEC>
__device__ __forceinline__ void spmm_chunk(int cnt, int jv, double av,
                                           const real_t *__restrict__ B, size_t ldb, unsigned col,
                                           real_t (&acc)[VEC])
{
    int k = 0;
    for (; k + SPMM_UNROLL <= cnt; k += SPMM_UNROLL) {
        real_t b[SPMM_UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < SPMM_UNROLL; u++) {
            const int j = __builtin_amdgcn_readlane(jv, k + u);
            const real_t *__restrict__ rowp = B + (size_t)j * ldb;
            vload<real_t, VEC>(b[u], rowp + col);
        }
#pragma unroll
        for (int u = 0; u < SPMM_UNROLL; u++) {
            const real_t a = (real_t)readlane_f64(av, k + u);   // narrowed per nonzero for f32 (matmul.cpp:53-57)
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[u][v], acc[v]);
        }
    }
    for (; k < cnt; k++) {
        const int j = __builtin_amdgcn_readlane(jv, k);
        const real_t a = (real_t)readlane_f64(av, k);
        const real_t *__restrict__ rowp = B + (size_t)j * ldb;
        real_t b[VEC];
        vload<real_t, VEC>(b, rowp + col);
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[v], acc[v]);
    }
}

// TR rows per workgroup.  COLMAJOR: stage the tile in LDS and write transposed.
// VSTORE (COLMAJOR only): two consecutive rows per lane -> wider stores; needs
// even ldc and 2*sizeof(real_t)-aligned C.
template <typename real_t, int VEC, bool COLMAJOR, bool VSTORE, int TR>
__global__ __launch_bounds__(SPMM_WAVES * MX_WAVE)
void spmm_rowwave_kernel(int m, int n,
                         const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                         const double *__restrict__ values,
                         const real_t *__restrict__ B, size_t ldb,
                         real_t *__restrict__ C, size_t ldc)
{
    constexpr int W = MX_WAVE * VEC;          // output columns per workgroup pass
    constexpr int S = W + 1;                  // odd LDS row stride (elements)
    constexpr int ROWS_PER_WAVE = TR / SPMM_WAVES;
    __shared__ real_t tile[COLMAJOR ? TR * S : 1];

    const int lane = lane_id();
    const int wave = uniform(threadIdx.x / MX_WAVE);
    const int row0 = blockIdx.x * TR;
    const int c0 = blockIdx.y * W;
    const int col = c0 + lane * VEC;
    const bool active = col < n;
    // clamped column for the reads of inactive lanes (n >= VEC always holds here)
    const unsigned lcol = active ? (unsigned)col : (unsigned)(n - VEC);

    const int r_begin = row0 + wave * ROWS_PER_WAVE;
    const int r_end = min(r_begin + ROWS_PER_WAVE, m);

    int s = 0, e = 0, jv = 0;
    double av = 0.0;
    if (r_begin < r_end) {
        s = uniform(indptr[r_begin]);
        e = uniform(indptr[r_begin + 1]);
        if (s + lane < e) { jv = indices[s + lane]; av = values[s + lane]; }
    }
    for (int row = r_begin; row < r_end; row++) {
        // prefetch the first chunk of the next row
        int e2 = e, jv2 = 0;
        double av2 = 0.0;
        if (row + 1 < r_end) {
            e2 = uniform(indptr[row + 2]);
            if (e + lane < e2) { jv2 = indices[e + lane]; av2 = values[e + lane]; }
        }
        real_t acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = 0;

        spmm_chunk<real_t, VEC>(min(MX_WAVE, e - s), jv, av, B, ldb, lcol, acc);
        for (int k0 = s + MX_WAVE; k0 < e; k0 += MX_WAVE) {   // rows longer than one wavefront
            int jc = 0;
            double ac = 0.0;
            if (k0 + lane < e) { jc = indices[k0 + lane]; ac = values[k0 + lane]; }
            spmm_chunk<real_t, VEC>(min(MX_WAVE, e - k0), jc, ac, B, ldb, lcol, acc);
        }

        if constexpr (COLMAJOR) {
            real_t *t = tile + (row - row0) * S + lane * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v++) t[v] = acc[v];
        } else {
            if (active) vstore<real_t, VEC>(C + (size_t)row * ldc + col, acc);
        }
        s = e; e = e2; jv = jv2; av = av2;
    }

    if constexpr (COLMAJOR) {
        __syncthreads();
        const int ncols = min(W, n - c0);
        constexpr int RPL = VSTORE ? 2 : 1;        // rows per lane in the write-out
        constexpr int LPC = TR / RPL;              // lanes per output column
        constexpr int CPW = MX_WAVE / LPC;         // columns per wave-instruction
        const int q = lane % LPC;
        const int r = q * RPL;
        const int grow = row0 + r;
        for (int cb = wave * CPW; cb < ncols; cb += SPMM_WAVES * CPW) {
            const int c = cb + lane / LPC;
            if (c < ncols && grow < m) {
                real_t *dst = C + (size_t)(c0 + c) * ldc + grow;
                if constexpr (VSTORE) {
                    real_t two[2] = { tile[r * S + c], tile[(r + 1) * S + c] };
                    vstore<real_t, 2>(dst, two);   // m even & grow even => grow+1 < m
                } else {
                    *dst = tile[r * S + c];
                }
            }
        }
    }
}

template <typename real_t, int VEC, bool COLMAJOR, bool VSTORE>
static int launch_spmm(int m, int n, const int32_t *indptr, const int32_t *indices, const double *values,
                       const real_t *B, size_t ldb, real_t *C, size_t ldc, hipStream_t stream)
{
    constexpr int TR = 32;
    constexpr int W = MX_WAVE * VEC;
    dim3 grid((unsigned)ceil_div(m, TR), (unsigned)ceil_div(n, W));
    kt_begin(stream);
    hipLaunchKernelGGL((spmm_rowwave_kernel<real_t, VEC, COLMAJOR, VSTORE, TR>), grid,
                       dim3(SPMM_WAVES * MX_WAVE), 0, stream, m, n, indptr, indices, values, B, ldb, C, ldc);
    kt_end(stream);
    MX_LAUNCH_CHECK();
    return 0;
}

template <typename real_t, int VECMAX>
static int dispatch_spmm(int m, int n, const int32_t *indptr, const int32_t *indices, const double *values,
                         const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream)
{
    // widest per-lane access the operands allow (16 B when rows of B are 16-B aligned)
    const bool b_vec = (n % VECMAX == 0) && (ldb % VECMAX == 0) && ((uintptr_t)B % (VECMAX * sizeof(real_t)) == 0);
    if (colmajor) {
        const bool vs = (ldc % 2 == 0) && ((uintptr_t)C % (2 * sizeof(real_t)) == 0);
        if (b_vec) return vs ? launch_spmm<real_t, VECMAX, true, true>(m, n, indptr, indices, values, B, ldb, C, ldc, stream)
                             : launch_spmm<real_t, VECMAX, true, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
        return vs ? launch_spmm<real_t, 1, true, true>(m, n, indptr, indices, values, B, ldb, C, ldc, stream)
                  : launch_spmm<real_t, 1, true, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
    }
    const bool c_vec = b_vec && (ldc % VECMAX == 0) && ((uintptr_t)C % (VECMAX * sizeof(real_t)) == 0);
    if (c_vec) return launch_spmm<real_t, VECMAX, false, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
    return launch_spmm<real_t, 1, false, false>(m, n, indptr, indices, values, B, ldb, C, ldc, stream);
}

template <typename real_t>
int rowwave_spmm(int m, int n, const int32_t *indptr, const int32_t *indices, const double *values,
                 const real_t *B, size_t ldb, real_t *C, size_t ldc, int colmajor, hipStream_t stream)
{
    return dispatch_spmm<real_t, 16 / (int)sizeof(real_t)>(m, n, indptr, indices, values, B, ldb, C, ldc, colmajor, stream);
}
template int rowwave_spmm<double>(int, int, const int32_t *, const int32_t *, const double *, const double *, size_t,
                                  double *, size_t, int, hipStream_t);
template int rowwave_spmm<float>(int, int, const int32_t *, const int32_t *, const double *, const float *, size_t,
                                 float *, size_t, int, hipStream_t);

}  // namespace mx
