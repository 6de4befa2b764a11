// dvec_na.hip — CSR (op) dense vector when the vector holds "special" elements: NA / NaN, zeros under / %% %/% ^,
// negatives under ^, infinities under *.  R must then return a matrix whose PATTERN grows: every cell the recycled vector
// makes special becomes an explicit entry (0 * NA is NA, 0 / 0 is NaN, 0 ^ 0 is 1, 0 ^ -1 is Inf).
//
// Replaces multiply_csr_by_dvec_with_NAs, src/operators.cpp:2258-2856 (serial; per-row std::sort of what it appended).
// Here the result is put together from pieces the library already has:
//     result = X'  (+)  F        X' = X's pattern, values `x op v[cell]` (with branch A's own rules for special elements)
//                                F  = the special cells with their FILL values, as a CSR matrix with sorted rows
//     (+) = sorted union of the patterns, X' wins where both hold a cell      (merge.hip, MX_OP_FIRST: count -> scan -> fill)
// F is built in one of three ways, following the reference's three length branches:
//   A  length <= nrows and divides nrows (:2316-2518): a special element makes its rows FULL rows 0 .. ncols-1;
//   B  length >= nrows * ncols (:2520-2572): one thread per row walks the row's cells of the dense vector in column order
//      (coalesced across rows), counts, then appends in order — sorted rows without a sort;
//   C  any other length (:2574-2637): the special elements are compacted, every (element, repeat) pair is a cell
//      (row = position mod nrows, col = position / nrows); cells are counted per row with atomics, scattered, and the rows
//      sorted by the per-row sort kernel (gather.hip) — work proportional to the cells produced, as in the reference.
// The reference's quirks are kept (the CPU checker under tests restates them too): in B / C a cell under an NA_real_ element
// is filled with NaN and one under a plain NaN (or an infinity, for *) with NA_real_ — the other way round from branch A;
// the operation is always `value op element`.  When no cell is added in B / C the structure is reported as unchanged
// (the reference returns its input indptr / indices there).
#include "mx_common.h"
#include "r_arith.h"

namespace mx {

int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace, hipStream_t st);
size_t scan_workspace_bytes(int64_t n);

constexpr int DN_BLOCK = 256;
// op: 0 multiply, 1 powerto, 2 divide, 3 divrest, 4 intdiv (mx_dvec_op order)

__device__ __forceinline__ bool dn_is_na(double d)                 // R's ISNA: a NaN whose low word is 1954
{
    return d != d && (unsigned)(__double_as_longlong(d) & 0xFFFFFFFFLL) == 1954u;
}
__device__ __forceinline__ bool dn_special(int op, double d)       // :2530-2533
{
    if (d != d) return true;
    if (op == MX_DV_MULTIPLY) return isinf(d);
    if (op == MX_DV_POWERTO) return d <= 0;
    return d == 0;
}
__device__ __forceinline__ double dn_op(int op, double x, double d)
{
    switch (op) {
        case MX_DV_MULTIPLY: return x * d;
        case MX_DV_DIVIDE:   return x / d;
        case MX_DV_DIVREST:  return r_modulus(x, d);
        case MX_DV_INTDIV:   return r_intdiv(x, d);
        default:             return r_pow(x, d);
    }
}
// branch A: what a full row is filled with (:2352, :2374-2395, :2476-2487)
__device__ __forceinline__ double dn_fill_A(int op, double d)
{
    if (op == MX_DV_MULTIPLY) return dn_is_na(d) ? na_real() : dv_nan();
    if (op == MX_DV_POWERTO) return d != d ? d : (d == 0 ? 1.0 : __builtin_inf());
    return d == 0 ? dv_nan() : d;                                   // division family: 0 -> NaN, NA / NaN -> the element itself
}
// branch A: the value of an entry the matrix holds
__device__ __forceinline__ double dn_value_A(int op, double x, double d)
{
    if (op == MX_DV_MULTIPLY) return d != d ? dn_fill_A(op, d) : x * d;            // (an infinity: x * d over the fill, :2354-2358)
    if (op == MX_DV_POWERTO) return r_pow(x, d);
    return d != d ? d : dn_op(op, x, d);                                            // NaN element: the whole row is the element (:2389-2393)
}
// branches B / C: the fill of a cell the matrix does not hold (:2544-2563 -> :2817-2833)
__device__ __forceinline__ double dn_fill_BC(int op, double d)
{
    const bool is_div = op == MX_DV_DIVIDE || op == MX_DV_DIVREST || op == MX_DV_INTDIV;
    if ((is_div && d == 0) || dn_is_na(d)) return dv_nan();
    if (op == MX_DV_POWERTO && d == 0) return 1.0;
    if (op == MX_DV_POWERTO && d < 0) return __builtin_inf();
    return na_real();
}

// ---- branch A ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DN_BLOCK)
void dna_count_A_kernel(int nrows, int ncols, const double *__restrict__ dvec, unsigned long long len, int op,
                        int32_t *__restrict__ fcnt)
{
    const long long r = (long long)blockIdx.x * DN_BLOCK + threadIdx.x;
    if (r < nrows) fcnt[r] = dn_special(op, dvec[(unsigned long long)r % len]) ? ncols : 0;
}
// one wavefront per row: columns 0 .. ncols-1 and the fill
__global__ __launch_bounds__(DN_BLOCK)
void dna_fill_A_kernel(int nrows, int ncols, const double *__restrict__ dvec, unsigned long long len, int op,
                       const int32_t *__restrict__ Fp, int32_t *__restrict__ Fj, double *__restrict__ Fx)
{
    const long long r = (long long)blockIdx.x * (DN_BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int s = Fp[r], n = Fp[r + 1] - s;
    if (n == 0) return;
    const double fill = dn_fill_A(op, dvec[(unsigned long long)r % len]);
    for (int c = lane_id(); c < n; c += 64) { Fj[s + c] = c; Fx[s + c] = fill; }
}

// ---- branch B: one thread per row over the dense vector (cell (r, c) at r + c * nrows) ------------------------------
template <bool FILL>
__global__ __launch_bounds__(DN_BLOCK)
void dna_rows_B_kernel(int nrows, int ncols, const double *__restrict__ dvec, int op, int32_t *__restrict__ fcnt,
                       const int32_t *__restrict__ Fp, int32_t *__restrict__ Fj, double *__restrict__ Fx)
{
    const long long r = (long long)blockIdx.x * DN_BLOCK + threadIdx.x;
    if (r >= nrows) return;
    long long at = FILL ? Fp[r] : 0;
    int cnt = 0;
    for (int c = 0; c < ncols; c++) {
        const double d = dvec[(unsigned long long)r + (unsigned long long)c * (unsigned long long)nrows];
        if (dn_special(op, d)) {
            if constexpr (FILL) { Fj[at] = c; Fx[at] = dn_fill_BC(op, d); at++; }
            cnt++;
        }
    }
    if constexpr (!FILL) fcnt[r] = cnt;
}

// ---- branch C: special elements -> cells -------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(DN_BLOCK)
void dna_specials_kernel(const double *__restrict__ dvec, unsigned long long len, int op, unsigned long long *__restrict__ count,
                         unsigned long long *__restrict__ list)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * DN_BLOCK + threadIdx.x; i < len; i += (unsigned long long)gridDim.x * DN_BLOCK) {
        const bool sp = dn_special(op, dvec[i]);
        const unsigned long long b = __ballot(sp);                  // one atomic per wavefront
        if (b == 0) continue;
        unsigned long long base = 0;
        if (lane_id() == (int)__builtin_ctzll(b)) base = atomicAdd(count, (unsigned long long)__popcll(b));
        base = ((unsigned long long)__builtin_amdgcn_readlane((int)(base >> 32), (int)__builtin_ctzll(b)) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(base & 0xFFFFFFFFu), (int)__builtin_ctzll(b));
        if constexpr (FILL) { if (sp) list[base + __popcll(b & ((1ULL << lane_id()) - 1ULL))] = i; }
    }
}
// pair t = (special s, repeat rep): position = list[s] + rep * len; FILL: scatter (col, fill) behind the row's cursor
template <bool FILL>
__global__ __launch_bounds__(DN_BLOCK)
void dna_cells_C_kernel(int nrows, const double *__restrict__ dvec, unsigned long long len, unsigned long long cells, int op,
                        const unsigned long long *__restrict__ list, unsigned long long nspecial, unsigned long long nrep,
                        int32_t *__restrict__ fcnt, const int32_t *__restrict__ Fp, int32_t *__restrict__ cursor,
                        int32_t *__restrict__ Fj, double *__restrict__ Fx)
{
    const unsigned long long total = nspecial * nrep;
    for (unsigned long long t = (unsigned long long)blockIdx.x * DN_BLOCK + threadIdx.x; t < total; t += (unsigned long long)gridDim.x * DN_BLOCK) {
        const unsigned long long s = t / nrep, rep = t - s * nrep;
        const unsigned long long ix = list[s], pos = ix + rep * len;
        if (pos >= cells) continue;
        const int row = (int)(pos % (unsigned long long)nrows), col = (int)(pos / (unsigned long long)nrows);
        if constexpr (!FILL) atomicAdd(&fcnt[row], 1);
        else {
            const int at = Fp[row] + atomicAdd(&cursor[row], 1);
            Fj[at] = col;
            Fx[at] = dn_fill_BC(op, dvec[ix]);
        }
    }
}

// ---- X': the values of the entries the matrix holds ----------------------------------------------------------------------
// MODE 0: branch A (element = dvec[row % len], branch A's rules); 1: branch B (row + col * nrows); 2: branch C (modulo)
template <int G>
__global__ __launch_bounds__(DN_BLOCK)
void dna_values_kernel(int m, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                       const double *__restrict__ values, const double *__restrict__ dvec, unsigned long long len, int mode,
                       int op, double *__restrict__ out)
{
    const int lg = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * (DN_BLOCK / G) + threadIdx.x / G;
    if (row >= m) return;
    const int s = indptr[row], e = indptr[row + 1];
    const unsigned long long nr = (unsigned long long)m;
    for (int k = s + lg; k < e; k += G) {
        unsigned long long at;
        if (mode == 0) at = (unsigned long long)row % len;
        else if (mode == 1) at = (unsigned long long)row + nr * (unsigned long long)indices[k];
        else at = ((unsigned long long)row + nr * (unsigned long long)indices[k]) % len;
        const double d = dvec[at], x = values[k];
        out[k] = mode == 0 ? dn_value_A(op, x, d) : dn_op(op, x, d);
    }
}

struct DnBuf {                                                    // device memory released at scope exit unless taken
    void *p = nullptr;
    ~DnBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { MX_HIP(hipMalloc(&p, n ? n : 16)); return 0; }
    void *take() { void *q = p; p = nullptr; return q; }
};

}  // namespace mx

extern "C" int mxd_csr_by_dvec_with_NAs(int m, int ncols, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                                        const double *values, const double *dvec, int64_t dvec_len, int op,
                                        int32_t **out_indptr, int32_t **out_indices, double **out_values,
                                        int64_t *nnz_out, int *structure_unchanged, void *stream)
{
    using namespace mx;
    MX_REQUIRE(m >= 0 && ncols >= 0 && nnz >= 0 && dvec_len > 0, "mxd_csr_by_dvec_with_NAs: bad size");
    MX_REQUIRE(op >= MX_DV_MULTIPLY && op <= MX_DV_INTDIV, "mxd_csr_by_dvec_with_NAs: unknown operation %d", op);
    MX_REQUIRE(out_indptr && out_indices && out_values && nnz_out && structure_unchanged, "mxd_csr_by_dvec_with_NAs: null pointer");
    MX_REQUIRE(indptr && dvec && (nnz == 0 || (indices && values)), "mxd_csr_by_dvec_with_NAs: null pointer");
    hipStream_t st = as_stream(stream);
    *out_indptr = nullptr; *out_indices = nullptr; *out_values = nullptr; *nnz_out = 0; *structure_unchanged = 0;
    const unsigned long long len = (unsigned long long)dvec_len, cells = (unsigned long long)m * (unsigned long long)ncols;
    const int mode = (len <= (unsigned long long)m && m > 0 && (unsigned long long)m % len == 0) ? 0 : (len >= cells ? 1 : 2);
    // ---- F: counts per row -> indptr
    DnBuf fcnt, Fp, Fj, Fx, scan_ws, cursor, list, cnt64;
    if (fcnt.alloc(sizeof(int32_t) * ((size_t)m + 1)) || Fp.alloc(sizeof(int32_t) * ((size_t)m + 1)) ||
        scan_ws.alloc(scan_workspace_bytes(m) + 16))
        return 1;
    const unsigned rows_grid = (unsigned)ceil_div(m > 0 ? m : 1, DN_BLOCK);
    unsigned long long nspecial = 0, nrep = 1;
    if (m > 0 && mode == 0) {
        hipLaunchKernelGGL(dna_count_A_kernel, dim3(rows_grid), dim3(DN_BLOCK), 0, st, m, ncols, dvec, len, op, (int32_t *)fcnt.p);
    } else if (m > 0 && mode == 1) {
        hipLaunchKernelGGL((dna_rows_B_kernel<false>), dim3(rows_grid), dim3(DN_BLOCK), 0, st, m, ncols, dvec, op, (int32_t *)fcnt.p,
                           (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr);
    } else if (m > 0) {
        if (cnt64.alloc(16)) return 1;
        MX_HIP(hipMemsetAsync(cnt64.p, 0, 16, st));
        const unsigned g = (unsigned)std::min<unsigned long long>(4096, (len + DN_BLOCK - 1) / DN_BLOCK);
        hipLaunchKernelGGL((dna_specials_kernel<false>), dim3(g), dim3(DN_BLOCK), 0, st, dvec, len, op, (unsigned long long *)cnt64.p,
                           (unsigned long long *)nullptr);
        MX_LAUNCH_CHECK();
        if (read_back_small(&nspecial, cnt64.p, sizeof(nspecial), st)) return 1;
        nrep = (cells + len - 1) / len;
        // every (special, repeat) pair below `cells` is an entry of F: more than int32 can index -> the reference's own error
        MX_REQUIRE(nspecial == 0 || nspecial * (nrep > 1 ? nrep - 1 : 1) < (unsigned long long)INT_MAX,
                   "Error: the resulting matrix would have too many entries for a sparse CSR representation (int overflow).");
        if (list.alloc(sizeof(unsigned long long) * (size_t)(nspecial ? nspecial : 1))) return 1;
        MX_HIP(hipMemsetAsync(cnt64.p, 0, 16, st));
        hipLaunchKernelGGL((dna_specials_kernel<true>), dim3(g), dim3(DN_BLOCK), 0, st, dvec, len, op, (unsigned long long *)cnt64.p,
                           (unsigned long long *)list.p);
        MX_HIP(hipMemsetAsync(fcnt.p, 0, sizeof(int32_t) * ((size_t)m + 1), st));
        if (nspecial) {
            const unsigned long long pairs = nspecial * nrep;
            const unsigned g2 = (unsigned)std::min<unsigned long long>(16384, (pairs + DN_BLOCK - 1) / DN_BLOCK);
            hipLaunchKernelGGL((dna_cells_C_kernel<false>), dim3(g2), dim3(DN_BLOCK), 0, st, m, dvec, len, cells, op,
                               (const unsigned long long *)list.p, nspecial, nrep, (int32_t *)fcnt.p, (const int32_t *)nullptr,
                               (int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr);
        }
    }
    MX_LAUNCH_CHECK();
    int64_t *total_dev = (int64_t *)scan_ws.p;
    if (exclusive_scan_i32((const int32_t *)fcnt.p, m, (int32_t *)Fp.p, total_dev, scan_ws.p, st)) return 1;
    int64_t nnzF = 0;
    if (read_back_small(&nnzF, total_dev, sizeof(nnzF), st)) return 1;
    MX_REQUIRE(nnzF + nnz < (int64_t)INT_MAX || nnzF == 0,
               "Error: the resulting matrix would have too many entries for a sparse CSR representation (int overflow).");
    // ---- F: entries (sorted rows)
    if (Fj.alloc(sizeof(int32_t) * (size_t)nnzF) || Fx.alloc(sizeof(double) * (size_t)nnzF)) return 1;
    if (nnzF > 0 && mode == 0) {
        hipLaunchKernelGGL(dna_fill_A_kernel, dim3((unsigned)ceil_div(m, DN_BLOCK / 64)), dim3(DN_BLOCK), 0, st, m, ncols, dvec, len, op,
                           (const int32_t *)Fp.p, (int32_t *)Fj.p, (double *)Fx.p);
    } else if (nnzF > 0 && mode == 1) {
        hipLaunchKernelGGL((dna_rows_B_kernel<true>), dim3(rows_grid), dim3(DN_BLOCK), 0, st, m, ncols, dvec, op, (int32_t *)nullptr,
                           (const int32_t *)Fp.p, (int32_t *)Fj.p, (double *)Fx.p);
    } else if (nnzF > 0) {
        if (cursor.alloc(sizeof(int32_t) * ((size_t)m + 1))) return 1;
        MX_HIP(hipMemsetAsync(cursor.p, 0, sizeof(int32_t) * ((size_t)m + 1), st));
        const unsigned long long pairs = nspecial * nrep;
        const unsigned g2 = (unsigned)std::min<unsigned long long>(16384, (pairs + DN_BLOCK - 1) / DN_BLOCK);
        hipLaunchKernelGGL((dna_cells_C_kernel<true>), dim3(g2), dim3(DN_BLOCK), 0, st, m, dvec, len, cells, op,
                           (const unsigned long long *)list.p, nspecial, nrep, (int32_t *)nullptr, (const int32_t *)Fp.p,
                           (int32_t *)cursor.p, (int32_t *)Fj.p, (double *)Fx.p);
        MX_LAUNCH_CHECK();
        DnBuf tj, tx;                                               // the cells of a row arrive in any order: sort the rows by column
        if (tj.alloc(sizeof(int32_t) * (size_t)nnzF) || tx.alloc(sizeof(double) * (size_t)nnzF)) return 1;
        if (mxd_csr_sort_rows(m, nnzF, (const int32_t *)Fp.p, (int32_t *)Fj.p, Fx.p, MX_F64, (int32_t *)tj.p, tx.p, stream)) return 1;
        MX_HIP(hipStreamSynchronize(st));                           // (tj / tx go out of scope)
    }
    MX_LAUNCH_CHECK();
    // ---- X': values of the entries the matrix holds
    DnBuf Xv;
    if (Xv.alloc(sizeof(double) * (size_t)nnz)) return 1;
    if (nnz > 0 && m > 0) {
        const int G = pick_group((double)nnz / (double)m);
#define MX_DN_G(GG)                                                                                                          \
        case GG: hipLaunchKernelGGL((dna_values_kernel<GG>), dim3((unsigned)ceil_div(m, DN_BLOCK / GG)), dim3(DN_BLOCK), 0, st, m,  \
                                    indptr, indices, values, dvec, len, mode, op, (double *)Xv.p); break;
        switch (G) { MX_DN_G(4) MX_DN_G(8) MX_DN_G(16) MX_DN_G(32) default: MX_DN_G(64) }
#undef MX_DN_G
        MX_LAUNCH_CHECK();
    }
    // ---- result = X' (+) F, X' first
    DnBuf Op, Oj, Ox, mws;
    if (Op.alloc(sizeof(int32_t) * ((size_t)m + 1)) || mws.alloc(mxd_merge_workspace_bytes(m))) return 1;
    int64_t total = 0;
    if (mxd_csr_merge_count(MX_OP_FIRST, m, indptr, indices, nnz, (const int32_t *)Fp.p, (const int32_t *)Fj.p, nnzF,
                            (int32_t *)Op.p, mws.p, &total, stream))
        return 1;
    MX_REQUIRE(total < (int64_t)INT_MAX, "Error: the resulting matrix would have too many entries for a sparse CSR representation (int overflow).");
    if (mode != 0 && total == nnz) {                                // B / C without a new cell: the reference's early return (:2639-2647)
        MX_HIP(hipStreamSynchronize(st));
        *structure_unchanged = 1;
        *out_values = (double *)Xv.take();
        *nnz_out = nnz;
        return 0;
    }
    if (Oj.alloc(sizeof(int32_t) * (size_t)total) || Ox.alloc(sizeof(double) * (size_t)total)) return 1;
    if (mxd_csr_merge_fill(MX_OP_FIRST, m, indptr, indices, Xv.p, nnz, (const int32_t *)Fp.p, (const int32_t *)Fj.p, Fx.p, nnzF,
                           (const int32_t *)Op.p, (int32_t *)Oj.p, Ox.p, stream))
        return 1;
    MX_HIP(hipStreamSynchronize(st));                               // the temporaries above are freed on return
    *out_indptr = (int32_t *)Op.take(); *out_indices = (int32_t *)Oj.take(); *out_values = (double *)Ox.take();
    *nnz_out = total;
    return 0;
}
