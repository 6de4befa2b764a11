// merge.hip — elementwise CSR (+) CSR for gfx950: segmented per-row merge.
//
// Replaces the serial two-pointer loops of the reference:
//   add_csr_elemwise<>       src/operators.cpp:337-537  (+, -, |, xor : sorted UNION)
//   multiply_csr_elemwise<>  src/operators.cpp:99-207   (*, &        : sorted INTERSECTION)
//
// The CPU code walks every row with one running output cursor.  Here each row
// is owned by a group of G lanes (G from the mean row length) and the output
// position of every entry is computed directly, so all entries of all rows are
// placed in parallel:
//   for entry i of row A (column a):  lb  = lower_bound(row B, a)
//                                     hit = B[lb] == a
//        union position        = i + lb - #(hits among A[0..i))
//        intersection position =          #(hits among A[0..i))      (hits only)
//   for entry u of row B (column b, not a hit):
//        union position        = lower_bound(row A, b) + u - #(hits among B[0..u))
// With rows sorted and column ids unique inside a row (the precondition the R
// callers establish and the reference assumes, R/operators.R:58,64,748,754)
// this is exactly the sequence the two-pointer loop emits; structure (indptr,
// indices) is bit-identical and values are one IEEE operation each:
//   coincident: v1 + (sub ? -v2 : v2)  (operators.cpp:481-485; explicit zeros are kept)
//   B-only under subtraction: -v2      (operators.cpp:433-434,469-470,501)
//   product: v1 * v2                   (operators.cpp:172)
// Two passes (count -> exclusive scan -> fill) because the output offsets of a
// row depend on all rows before it.
//
// Roofline: HBM-bound; algorithmic bytes = 2*(12 nnz + 4(m+1)) read + 12 nnz_out + 4(m+1)
// written; the count pass re-reads the indices (8 nnz) on top of that.
#include "mx_common.h"

namespace mx {

int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace,
                       hipStream_t st);
size_t scan_workspace_bytes(int64_t n);

constexpr int MERGE_BLOCK = 256;

// ballot restricted to this lane's G-wide group
template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool pred)
{
    const unsigned long long b = __ballot(pred);
    if constexpr (G == 64) return b;
    else {
        const int shift = lane_id() & ~(G - 1);
        return (b >> shift) & ((1ULL << G) - 1ULL);
    }
}

// lower_bound(key) in a sorted row held one entry per lane (lane u of the group holds tbl = entry u, lanes past
// the row's end hold INT_MAX): binary search whose probes are cross-lane reads (ds_bpermute), not memory loads.
// Returns the count of entries < key (0..G) and whether entry[count] == key.
template <int G>
__device__ __forceinline__ int group_lower_bound(int tbl, int key, bool &hit)
{
    int lo = 0, len = G;
#pragma unroll
    for (int it = 0; it < 7; it++) {                  // ceil(log2(64)) + 1 probes at most
        if ((1 << it) > G) break;
        const int half = len >> 1;
        const int probe = __shfl(tbl, lo + half, G);  // all lanes take part; finished lanes ignore the value
        if (len > 0) {
            if (probe < key) { lo += half + 1; len -= half + 1; }
            else len = half;
        }
    }
    const int at = __shfl(tbl, lo < G ? lo : G - 1, G);
    hit = at == key;                                   // tbl[G-1] < key when lo == G, so no false hit
    return lo;
}

template <int G, bool INTERSECT>
__global__ __launch_bounds__(MERGE_BLOCK)
void merge_count_kernel(int m, const int32_t *__restrict__ p1, const int32_t *__restrict__ j1,
                        const int32_t *__restrict__ p2, const int32_t *__restrict__ j2,
                        int32_t *__restrict__ counts)
{
    const int lg = threadIdx.x % G;
    const long long row_ll = (long long)blockIdx.x * (MERGE_BLOCK / G) + threadIdx.x / G;
    const bool valid = row_ll < m;
    const int row = valid ? (int)row_ll : 0;
    int s1 = 0, e1 = 0, s2 = 0, e2 = 0;
    if (valid) { s1 = p1[row]; e1 = p1[row + 1]; s2 = p2[row]; e2 = p2[row + 1]; }
    const int n1 = e1 - s1, n2 = e2 - s2;
    if (__ballot(n1 > G || n2 > G) == 0ULL) {
        // every row of this wavefront fits its lane group: both rows live in registers
        const int a = lg < n1 ? j1[s1 + lg] : INT_MAX;
        const int b = lg < n2 ? j2[s2 + lg] : INT_MAX;
        bool hit;
        group_lower_bound<G>(b, a, hit);
        const int hits = __popcll(group_ballot<G>(hit && lg < n1));
        if (valid && lg == 0) counts[row] = INTERSECT ? hits : n1 + n2 - hits;
        return;
    }
    // search the shorter row's entries in the longer row
    const bool a_short = n1 <= n2;
    const int32_t *__restrict__ q = a_short ? j1 + s1 : j2 + s2;   // queries
    const int32_t *__restrict__ t = a_short ? j2 + s2 : j1 + s1;   // table
    const int nq = a_short ? n1 : n2, nt = a_short ? n2 : n1;
    int hits = 0;
    for (int i0 = 0; i0 < nq; i0 += G) {      // nq is uniform inside the group
        const int i = i0 + lg;
        bool hit = false;
        if (i < nq && nt > 0) {
            const int key = q[i];
            const int lb = lower_bound_dev(t, nt, key);
            hit = lb < nt && t[lb] == key;
        }
        hits += __popcll(group_ballot<G>(hit));
    }
    if (valid && lg == 0) counts[row] = INTERSECT ? hits : n1 + n2 - hits;
}

// OP: mx_merge_op.  VT = double (ADD/SUB/MUL) or int32_t (OR/XOR/AND).
template <int OP, typename VT>
__device__ __forceinline__ VT combine(VT a, VT b)
{
    if constexpr (OP == MX_OP_ADD) return a + b;
    else if constexpr (OP == MX_OP_SUB) return a + (-b);
    else if constexpr (OP == MX_OP_MUL) return a * b;
    else if constexpr (OP == MX_OP_OR)  return r_logical_or(a, b);
    else if constexpr (OP == MX_OP_XOR) return r_logical_xor(a, b);
    else return r_logical_and(a, b);
}

template <int G, int OP, typename VT>
__global__ __launch_bounds__(MERGE_BLOCK)
void merge_fill_kernel(int m, const int32_t *__restrict__ p1, const int32_t *__restrict__ j1,
                       const VT *__restrict__ x1,
                       const int32_t *__restrict__ p2, const int32_t *__restrict__ j2,
                       const VT *__restrict__ x2,
                       const int32_t *__restrict__ po, int32_t *__restrict__ jo, VT *__restrict__ xo)
{
    constexpr bool INTERSECT = (OP == MX_OP_MUL || OP == MX_OP_AND);
    const int lg = threadIdx.x % G;
    const long long row_ll = (long long)blockIdx.x * (MERGE_BLOCK / G) + threadIdx.x / G;
    const bool valid = row_ll < m;
    const int row = valid ? (int)row_ll : 0;
    int s1 = 0, e1 = 0, s2 = 0, e2 = 0, o = 0;
    if (valid) { s1 = p1[row]; e1 = p1[row + 1]; s2 = p2[row]; e2 = p2[row + 1]; o = po[row]; }
    const int n1 = e1 - s1, n2 = e2 - s2;
    const int32_t *__restrict__ a_idx = j1 + s1;
    const int32_t *__restrict__ b_idx = j2 + s2;
    const unsigned long long below = (1ULL << lg) - 1ULL;   // lg < 64 always

    if (__ballot(n1 > G || n2 > G) == 0ULL) {
        // register-resident rows: one entry of A and one of B per lane, searches by cross-lane probes
        const bool va = lg < n1, vb = lg < n2;
        const int a = va ? a_idx[lg] : INT_MAX;
        const int b = vb ? b_idx[lg] : INT_MAX;
        VT xa = VT(0), xb = VT(0);
        if (va) xa = x1[s1 + lg];
        if (vb) xb = x2[s2 + lg];
        bool hit_a, hit_b;
        const int lb_a = group_lower_bound<G>(b, a, hit_a);     // entries of B below a
        hit_a = hit_a && va;
        const VT partner = __shfl(xb, lb_a < G ? lb_a : G - 1, G);
        const int before_a = __popcll(group_ballot<G>(hit_a) & below);
        if constexpr (INTERSECT) {
            if (hit_a) {
                const int pos = o + before_a;
                jo[pos] = a;
                xo[pos] = combine<OP, VT>(xa, partner);
            }
        } else {
            if (va) {
                const int pos = o + lg + lb_a - before_a;
                jo[pos] = a;
                xo[pos] = hit_a ? combine<OP, VT>(xa, partner) : xa;
            }
            const int lb_b = group_lower_bound<G>(a, b, hit_b);  // entries of A below b
            hit_b = hit_b && vb;
            const int before_b = __popcll(group_ballot<G>(hit_b) & below);
            if (vb && !hit_b) {
                const int pos = o + lb_b + lg - before_b;
                jo[pos] = b;
                if constexpr (OP == MX_OP_SUB) xo[pos] = -xb; else xo[pos] = xb;
            }
        }
        return;
    }

    // entries of A
    int hits_before = 0;
    for (int i0 = 0; i0 < n1; i0 += G) {
        const int i = i0 + lg;
        bool hit = false;
        int lb = 0, key = 0;
        if (i < n1) {
            key = a_idx[i];
            lb = lower_bound_dev(b_idx, n2, key);
            hit = lb < n2 && b_idx[lb] == key;
        }
        const unsigned long long hb = group_ballot<G>(hit);
        const int before = hits_before + __popcll(hb & below);
        if (i < n1) {
            if constexpr (INTERSECT) {
                if (hit) {
                    const int pos = o + before;
                    jo[pos] = key;
                    xo[pos] = combine<OP, VT>(x1[s1 + i], x2[s2 + lb]);
                }
            } else {
                const int pos = o + i + lb - before;
                jo[pos] = key;
                const VT va = x1[s1 + i];
                xo[pos] = hit ? combine<OP, VT>(va, x2[s2 + lb]) : va;
            }
        }
        hits_before += __popcll(hb);
    }
    if constexpr (!INTERSECT) {
        // entries of B that have no partner in A
        hits_before = 0;
        for (int u0 = 0; u0 < n2; u0 += G) {
            const int u = u0 + lg;
            bool hit = false;
            int lb = 0, key = 0;
            if (u < n2) {
                key = b_idx[u];
                lb = lower_bound_dev(a_idx, n1, key);
                hit = lb < n1 && a_idx[lb] == key;
            }
            const unsigned long long hb = group_ballot<G>(hit);
            const int before = hits_before + __popcll(hb & below);
            if (u < n2 && !hit) {
                const int pos = o + lb + u - before;
                jo[pos] = key;
                const VT vb = x2[s2 + u];
                if constexpr (OP == MX_OP_SUB) xo[pos] = -vb; else xo[pos] = vb;
            }
            hits_before += __popcll(hb);
        }
    }
}

template <typename VT, int OP>
__global__ __launch_bounds__(256)
void values_elemwise_kernel(int64_t nnz, const VT *__restrict__ a, const VT *__restrict__ b, VT *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = combine<OP, VT>(a[i], b[i]);
}

static inline bool op_is_intersect(int op) { return op == MX_OP_MUL || op == MX_OP_AND; }

// G from the mean length of the longer operand's rows; nnz hints < 0 => 32
int merge_group(int m, int64_t nnz1, int64_t nnz2)
{
    if (nnz1 < 0 || nnz2 < 0 || m <= 0) return 32;
    const double avg = (double)(nnz1 > nnz2 ? nnz1 : nnz2) / (double)m;
    return pick_group(avg, 8);
}

int merge_count_launch(int op, int G, int m, const int32_t *p1, const int32_t *j1, const int32_t *p2,
                       const int32_t *j2, int32_t *counts, hipStream_t st)
{
    const bool isect = op_is_intersect(op);
#define MX_CASE(GG)                                                                                   \
    case GG: {                                                                                        \
        const unsigned grid = (unsigned)ceil_div(m, MERGE_BLOCK / GG);                                \
        if (isect) hipLaunchKernelGGL((merge_count_kernel<GG, true>), dim3(grid), dim3(MERGE_BLOCK), 0, st, \
                                      m, p1, j1, p2, j2, counts);                                     \
        else hipLaunchKernelGGL((merge_count_kernel<GG, false>), dim3(grid), dim3(MERGE_BLOCK), 0, st, \
                                m, p1, j1, p2, j2, counts);                                           \
        break;                                                                                        \
    }
    switch (G) { MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64) default: return set_error("merge: bad group %d", G); }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
}

template <int OP, typename VT>
static int merge_fill_op(int G, int m, const int32_t *p1, const int32_t *j1, const void *x1, const int32_t *p2,
                         const int32_t *j2, const void *x2, const int32_t *po, int32_t *jo, void *xo, hipStream_t st)
{
#define MX_CASE(GG)                                                                                   \
    case GG: {                                                                                        \
        const unsigned grid = (unsigned)ceil_div(m, MERGE_BLOCK / GG);                                \
        hipLaunchKernelGGL((merge_fill_kernel<GG, OP, VT>), dim3(grid), dim3(MERGE_BLOCK), 0, st, m,  \
                           p1, j1, (const VT *)x1, p2, j2, (const VT *)x2, po, jo, (VT *)xo);         \
        break;                                                                                        \
    }
    switch (G) { MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64) default: return set_error("merge: bad group %d", G); }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
}

int merge_fill_launch(int op, int G, int m, const int32_t *p1, const int32_t *j1, const void *x1, const int32_t *p2,
                      const int32_t *j2, const void *x2, const int32_t *po, int32_t *jo, void *xo, hipStream_t st)
{
    switch (op) {
        case MX_OP_ADD: return merge_fill_op<MX_OP_ADD, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        case MX_OP_SUB: return merge_fill_op<MX_OP_SUB, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        case MX_OP_MUL: return merge_fill_op<MX_OP_MUL, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        case MX_OP_OR:  return merge_fill_op<MX_OP_OR, int32_t>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        case MX_OP_XOR: return merge_fill_op<MX_OP_XOR, int32_t>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        case MX_OP_AND: return merge_fill_op<MX_OP_AND, int32_t>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        default: return set_error("merge: unknown op %d", op);
    }
}

}  // namespace mx

extern "C" size_t mxd_merge_workspace_bytes(int m)
{
    // [counts int32[m] padded to 16 B][scan workspace]
    const size_t counts = ((size_t)(m > 0 ? m : 1) * sizeof(int32_t) + 15) & ~(size_t)15;
    return counts + mx::scan_workspace_bytes(m);
}

extern "C" int mxd_csr_merge_count(int op, int m, const int32_t *indptr1, const int32_t *indices1, int64_t nnz1,
                                   const int32_t *indptr2, const int32_t *indices2, int64_t nnz2,
                                   int32_t *out_indptr, void *workspace, int64_t *nnz_out_host, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_csr_merge_count: negative m");
    MX_REQUIRE(out_indptr && workspace, "mxd_csr_merge_count: null pointer");
    hipStream_t st = mx::as_stream(stream);
    int32_t *counts = (int32_t *)workspace;
    const size_t counts_bytes = ((size_t)(m > 0 ? m : 1) * sizeof(int32_t) + 15) & ~(size_t)15;
    void *scan_ws = (char *)workspace + counts_bytes;
    if (m > 0) {
        const int rc = mx::merge_count_launch(op, mx::merge_group(m, nnz1, nnz2), m, indptr1, indices1, indptr2, indices2, counts, st);
        if (rc) return rc;
    }
    int64_t *total_dev = (int64_t *)scan_ws;   // first word of the scan workspace
    const int rc = mx::exclusive_scan_i32(counts, m, out_indptr, total_dev, scan_ws, st);
    if (rc) return rc;
    if (nnz_out_host) {
        if (mx::read_back_small(nnz_out_host, total_dev, sizeof(int64_t), st)) return 1;
        MX_REQUIRE(*nnz_out_host <= (int64_t)INT_MAX, "result has %lld entries: exceeds R's int32 index range",
                   (long long)*nnz_out_host);
    }
    return 0;
}

extern "C" int mxd_csr_merge_fill(int op, int m, const int32_t *indptr1, const int32_t *indices1, const void *values1,
                                  int64_t nnz1,
                                  const int32_t *indptr2, const int32_t *indices2, const void *values2,
                                  int64_t nnz2,
                                  const int32_t *out_indptr, int32_t *out_indices, void *out_values, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_csr_merge_fill: negative m");
    if (m == 0) return 0;
    return mx::merge_fill_launch(op, mx::merge_group(m, nnz1, nnz2), m, indptr1, indices1, values1, indptr2, indices2, values2,
                                 out_indptr, out_indices, out_values, mx::as_stream(stream));
}

extern "C" int mxd_values_elemwise(int op, int64_t nnz, const void *values1, const void *values2, void *out_values,
                                   void *stream)
{
    if (nnz <= 0) return 0;
    hipStream_t st = mx::as_stream(stream);
    const unsigned grid = (unsigned)(mx::ceil_div(nnz, 256) < 4096 ? mx::ceil_div(nnz, 256) : 4096);
#define MX_CASE(OPV, VT)                                                                             \
    case OPV:                                                                                         \
        hipLaunchKernelGGL((mx::values_elemwise_kernel<VT, OPV>), dim3(grid), dim3(256), 0, st, nnz,  \
                           (const VT *)values1, (const VT *)values2, (VT *)out_values);               \
        break;
    switch (op) {
        MX_CASE(MX_OP_ADD, double) MX_CASE(MX_OP_SUB, double) MX_CASE(MX_OP_MUL, double)
        MX_CASE(MX_OP_OR, int32_t) MX_CASE(MX_OP_XOR, int32_t) MX_CASE(MX_OP_AND, int32_t)
        default: return mx::set_error("mxd_values_elemwise: unknown op %d", op);
    }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
}
