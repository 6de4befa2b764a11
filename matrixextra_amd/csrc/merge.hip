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
#include <algorithm>
#include <cstdlib>
#include <type_traits>

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
// The count and fill kernels are bound by VALU issue, not by memory (a wave64 VALU instruction occupies its SIMD for
// four cycles; the first version of this search cost ~100 of them per row pair), so the search is branch-free and works
// on byte addresses: log2(G) probes "is entry[lo + s - 1] < key" for s = G/2 .. 1 leave lo = #entries < key among the
// first G - 1, one more probe at lo settles the last entry and the hit.  Three VALU instructions per probe
// (compare, select, add; the constant part of the address rides in the instruction's offset field).
template <int G>
__device__ __forceinline__ int group_lower_bound(int tbl, int key, bool &hit)
{
    const int base4 = G == 64 ? 0 : (lane_id() & ~(G - 1)) << 2;      // byte address of the group's lane 0
    int lo4 = base4;
#pragma unroll
    for (int s = G / 2; s >= 1; s >>= 1) {
        const int probe = __builtin_amdgcn_ds_bpermute(lo4 + (s - 1) * 4, tbl);
        lo4 += probe < key ? s * 4 : 0;
    }
    const int at = __builtin_amdgcn_ds_bpermute(lo4, tbl);             // all lanes of the group take part
    hit = at == key;
    return ((lo4 - base4) >> 2) + (at < key ? 1 : 0);
}

// Row pairs that do NOT fit the lane group (round 5, tools/cliff_hunt_ops.py; round 6: the windows slide).  G follows the MEAN
// row length and real row lengths are skewed: log-normal rows (sigma 1, mean 50, G = 64) put 60 % of the ENTRIES in rows that
// do not fit.  Such a pair is merged WINDOW AGAINST WINDOW: the group holds a window of G consecutive entries of each row in
// registers (one per lane; the next G of each already prefetched), searches window against window with the cross-lane probes
// of the fast path, and then BOTH windows slide past everything the step settled:
//   * an entry of A is settled once the B window reaches it (a <= the B window's last entry, or B is exhausted): every B entry
//     below it, and the one that could equal it, lies in this or an earlier window — its lower bound in B is known;
//   * likewise for B; one of the two windows is settled whole in every step (the one that ends first), the other up to there.
// So a step consumes G entries of one row AND the other row's entries up to the same column — ~1.8 G for rows of similar
// density — where round 5's block-against-block merge moved one block of G per step ((n1 + n2) / G steps per pair: half of every
// step's comparisons were against entries that stayed for the next one).  The windows never hold a settled entry: no
// "placed prefix", no hit masks carried from step to step.  A window slides by fa <= G entries: its remaining entries move down by
// two cross-lane reads (this window / the prefetched one), and the prefetch is re-issued at the new position (asynchronous:
// it is needed a step later; the re-read part hits in L1 / L2).
template <int G>
__device__ __forceinline__ void window_slide(int lg, int step, int &w, int &wn)
{
    if (step == 0) return;                                             // (uniform inside the group)
    const int src = (lg + step) & (G - 1);
    const int from_w = __shfl(w, src, G), from_n = __shfl(wn, src, G);
    w = lg + step < G ? from_w : from_n;
}
// number of coincidences (INTERSECT) or the length of the union
template <int G, bool INTERSECT>
__device__ __forceinline__ int count_row_slow(int lg, const int32_t *__restrict__ a_idx, int n1,
                                              const int32_t *__restrict__ b_idx, int n2)
{
    int ia = 0, ib = 0, hits = 0;
    int a = lg < n1 ? a_idx[lg] : INT_MAX, b = lg < n2 ? b_idx[lg] : INT_MAX;
    int an = G + lg < n1 ? a_idx[G + lg] : INT_MAX, bn = G + lg < n2 ? b_idx[G + lg] : INT_MAX;
    while (ia < n1 && ib < n2) {                                       // (uniform inside the group)
        const int ca = min(G, n1 - ia), cb = min(G, n2 - ib);
        bool hit;
        group_lower_bound<G>(b, a, hit);
        hits += __popcll(group_ballot<G>(hit && lg < ca));
        const int amax = __shfl(a, ca - 1, G), bmax = __shfl(b, cb - 1, G);
        const int fa = __popcll(group_ballot<G>(lg < ca && a <= bmax)), fb = __popcll(group_ballot<G>(lg < cb && b <= amax));
        window_slide<G>(lg, fa, a, an);
        window_slide<G>(lg, fb, b, bn);
        ia += fa; ib += fb;
        if (fa) an = ia + G + lg < n1 ? a_idx[ia + G + lg] : INT_MAX;
        if (fb) bn = ib + G + lg < n2 ? b_idx[ib + G + lg] : INT_MAX;
    }
    return INTERSECT ? hits : n1 + n2 - hits;
}

// VERY long row pairs (either row longer than MergeLong::T entries: a few rows of tens of thousands of entries, the tail of
// a log-normal or power-law length distribution) would keep one lane group busy long after the rest of the grid has
// finished (CSR + CSR, m = 1e6, 32 per row, four rows of 50,000 entries: 2.8 ms against 0.44).  The kernels below leave
// them out and append them to a list; merge_long_count_kernel / merge_long_fill_kernel then give every such pair a whole
// 512-thread workgroup, which cuts the pair into <= 32 pieces BY VALUE (evenly spaced entries of the longer row are pivots; the
// pieces of the other row follow by binary search, so equal column ids always meet inside one piece) and runs the blocked
// merge of count_row_slow / fill_row_slow on one piece per wavefront.
struct MergeLong {
    unsigned *count;     // entries of the list
    int *rows;
    int T;               // 0 = off
};
constexpr int MERGE_LONG_T = 1024;
constexpr int MERGE_LONG_BLOCK = 512;
constexpr int MERGE_LONG_MAXP = 32;                   // pieces per pair

// Every lane group sizes COUNT_U consecutive row pairs per call: the row pointers of all of them, then the index loads
// of all of them, are in flight together (one pair at a time — three dependent loads — ran at 1.8 TB/s).
// Measured at the cfg4 shape (2M x 2M, 50 + 50 per row pair, 888 MB read): 0.385 ms with the first, branchy search
// (VALU issue: ~100 instructions per row pair), 0.28 ms with group_lower_bound as it is now, 0.21-0.24 ms with the
// COUNT_U searches behind one fit test (below) — 0.21 ms is this load structure with the search taken out (4.2 TB/s).  Tried and dropped: staging the workgroup's index window
// through LDS with flat 16-byte loads (0.37-0.42 ms before the search was cheap; 0.44 ms as a persistent workgroup with
// the next tile's loads in registers — 85 % of the wave cycles waiting), COUNT_U = 8 (0.29 ms).
constexpr int COUNT_U = 4;
template <int G, bool INTERSECT>
__global__ __launch_bounds__(MERGE_BLOCK)
void merge_count_kernel(int m, const int32_t *__restrict__ p1, const int32_t *__restrict__ j1,
                        const int32_t *__restrict__ p2, const int32_t *__restrict__ j2,
                        int32_t *__restrict__ counts, MergeLong ml)
{
    const int lg = threadIdx.x % G;
    // G = 64: the lane group is the wavefront, so its rows — and their row pointers — are wave-uniform: read once through
    // the scalar cache instead of 64 identical lanes through the texture path (PMC: TA_BUSY 70 % of the count kernel)
    const int grp_in_block = G == 64 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : (int)threadIdx.x / G;
    const long long grp = (long long)blockIdx.x * (MERGE_BLOCK / G) + grp_in_block;
    int s1[COUNT_U], n1[COUNT_U], s2[COUNT_U], n2[COUNT_U], a[COUNT_U], b[COUNT_U];
    bool left[COUNT_U];
#pragma unroll
    for (int u = 0; u < COUNT_U; u++) {
        const long long row = grp * COUNT_U + u;
        const bool valid = row < m;
        const long long rs = valid ? row : 0;
        const int a0 = p1[rs], a1 = p1[rs + 1], b0 = p2[rs], b1 = p2[rs + 1];
        s1[u] = a0; n1[u] = valid ? a1 - a0 : 0;
        s2[u] = b0; n2[u] = valid ? b1 - b0 : 0;
    }
#pragma unroll
    for (int u = 0; u < COUNT_U; u++) {
        a[u] = lg < n1[u] && n1[u] <= G ? j1[s1[u] + lg] : INT_MAX;
        b[u] = lg < n2[u] && n2[u] <= G ? j2[s2[u] + lg] : INT_MAX;
    }
    // (very long pairs — merge_long_count_kernel's — are met in the slow branch below, which they take anyway: the straight-line
    // path of pairs that fit knows nothing of them.  Same box, cfg4: a branch with the list's atomic up here 207 -> 233 us,
    // branch-free selects on n1 / n2 here 207 -> 224)
#pragma unroll
    for (int u = 0; u < COUNT_U; u++) left[u] = false;
    // (Round 6 tried the straight-line searches for ALL pairs first and the sliding windows only for those that do not fit — so that
    // a wavefront with one big pair does not search its other pairs one after the other: log-normal rows, cfg4's shape, count
    // 495 -> 590 us, fill 1303 -> 1343: the wasted searches of the big pairs cost more than the interleaving saves.  Not kept.)
    // One test for all COUNT_U pairs: when every row fits its lane group (the normal case) the searches are straight-line
    // code, and the compiler interleaves their dependent chains of cross-lane reads (~100 cycles a probe).
    bool big = false;
#pragma unroll
    for (int u = 0; u < COUNT_U; u++) big = big || n1[u] > G || n2[u] > G;
    int c[COUNT_U];
    if (__ballot(big) == 0ULL) {
#pragma unroll
        for (int u = 0; u < COUNT_U; u++) {
            bool hit;
            group_lower_bound<G>(b[u], a[u], hit);
            const int hits = __popcll(group_ballot<G>(hit && lg < n1[u]));
            c[u] = INTERSECT ? hits : n1[u] + n2[u] - hits;
        }
    } else {
#pragma unroll
        for (int u = 0; u < COUNT_U; u++) {
            left[u] = ml.T != 0 && (n1[u] > ml.T || n2[u] > ml.T);   // (uniform inside the group)
            if (left[u]) {
                c[u] = 0;                                            // listed at the end of the kernel
            } else if (__ballot(n1[u] > G || n2[u] > G) == 0ULL) {
                bool hit;
                group_lower_bound<G>(b[u], a[u], hit);
                const int hits = __popcll(group_ballot<G>(hit && lg < n1[u]));
                c[u] = INTERSECT ? hits : n1[u] + n2[u] - hits;
            } else {
                c[u] = count_row_slow<G, INTERSECT>(lg, j1 + s1[u], n1[u], j2 + s2[u], n2[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < COUNT_U; u++) {
        const long long row = grp * COUNT_U + u;
        if (row < m && lg == 0 && !left[u]) counts[row] = c[u];
    }
    if (ml.T != 0) {
#pragma unroll
        for (int u = 0; u < COUNT_U; u++)
            if (left[u] && lg == 0) ml.rows[atomicAdd(ml.count, 1u)] = (int)(grp * COUNT_U + u);
    }
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
    else if constexpr (OP == MX_OP_FIRST) return a;                 // union, the left operand's value where both hold the cell
    else return r_logical_and(a, b);
}

// fill of such a row pair at output offset o: the sliding windows of count_row_slow with positions.  An entry is PLACED in the
// step that settles it: the entries of the other row below it are then that row's entries before its window (ib) plus the
// local lower bound, and the coincidences before it are those of earlier steps (H: a coincidence settles both its entries in
// one step) plus those among the window's entries in the lanes below.  Values are not carried in registers: the step's two
// value windows are requested when the windows are known (a step ahead of their use) and read by the lanes that place.
template <int G, int OP, typename VT>
__device__ __forceinline__ void fill_row_slow(int lg, const int32_t *__restrict__ a_idx, const VT *__restrict__ xa_, int n1,
                                              const int32_t *__restrict__ b_idx, const VT *__restrict__ xb_, int n2,
                                              long long o, int32_t *__restrict__ jo, VT *__restrict__ xo)
{
    constexpr bool INTERSECT = (OP == MX_OP_MUL || OP == MX_OP_AND);
    const unsigned long long below = (1ULL << lg) - 1ULL;
    int ia = 0, ib = 0, H = 0;
    int a = INT_MAX, b = INT_MAX, an = INT_MAX, bn = INT_MAX;
    VT xa = VT(0), xb = VT(0);
    if (lg < n1) { a = a_idx[lg]; xa = xa_[lg]; }
    if (lg < n2) { b = b_idx[lg]; xb = xb_[lg]; }
    if (G + lg < n1) an = a_idx[G + lg];
    if (G + lg < n2) bn = b_idx[G + lg];
    while (INTERSECT ? (ia < n1 && ib < n2) : (ia < n1 || ib < n2)) { // (uniform inside the group)
        const int ca = max(0, min(G, n1 - ia)), cb = max(0, min(G, n2 - ib));
        const bool va = lg < ca, vb = lg < cb;
        // (an exhausted row's window holds INT_MAX in every lane and a "last entry" of INT_MAX: everything of the other row is settled)
        const int amax = ca > 0 ? __shfl(a, ca - 1, G) : INT_MAX, bmax = cb > 0 ? __shfl(b, cb - 1, G) : INT_MAX;
        bool h;
        const int lb_a = group_lower_bound<G>(b, a, h);                // entries of the B window below a
        const bool set_a = va && a <= bmax, hit_a = h && set_a;
        const VT partner = __shfl(xb, lb_a < G ? lb_a : G - 1, G);
        const unsigned long long hita = group_ballot<G>(hit_a);
        if constexpr (INTERSECT) {
            if (hit_a) {
                const long long pos = o + H + __popcll(hita & below);
                jo[pos] = a;
                xo[pos] = combine<OP, VT>(xa, partner);
            }
        } else {
            if (set_a) {
                const long long pos = o + ia + lg + ib + lb_a - (H + __popcll(hita & below));
                jo[pos] = a;
                xo[pos] = hit_a ? combine<OP, VT>(xa, partner) : xa;
            }
            const int lb_b = group_lower_bound<G>(a, b, h);            // entries of the A window below b
            const bool set_b = vb && b <= amax, hit_b = h && set_b;
            const unsigned long long hitb = group_ballot<G>(hit_b);
            if (set_b && !hit_b) {
                const long long pos = o + ib + lg + ia + lb_b - (H + __popcll(hitb & below));
                jo[pos] = b;
                if constexpr (OP == MX_OP_SUB) xo[pos] = -xb; else xo[pos] = xb;
            }
        }
        H += __popcll(hita);
        const int fa = __popcll(group_ballot<G>(set_a)), fb = __popcll(group_ballot<G>(vb && b <= amax));
        window_slide<G>(lg, fa, a, an);
        window_slide<G>(lg, fb, b, bn);
        ia += fa; ib += fb;
        if (fa) {
            an = ia + G + lg < n1 ? a_idx[ia + G + lg] : INT_MAX;
            xa = ia + lg < n1 ? xa_[ia + lg] : VT(0);
        }
        if (fb) {
            bn = ib + G + lg < n2 ? b_idx[ib + G + lg] : INT_MAX;
            xb = ib + lg < n2 ? xb_[ib + lg] : VT(0);
        }
    }
}

// FILL_U consecutive row pairs per lane group: row pointers, then indices + values of all of them in flight together
constexpr int FILL_U = 2;
template <int G, int OP, typename VT>
__global__ __launch_bounds__(MERGE_BLOCK)
void merge_fill_kernel(int m, const int32_t *__restrict__ p1, const int32_t *__restrict__ j1,
                       const VT *__restrict__ x1,
                       const int32_t *__restrict__ p2, const int32_t *__restrict__ j2,
                       const VT *__restrict__ x2,
                       const int32_t *__restrict__ po, int32_t *__restrict__ jo, VT *__restrict__ xo, MergeLong ml)
{
    constexpr bool INTERSECT = (OP == MX_OP_MUL || OP == MX_OP_AND);
    const int lg = threadIdx.x % G;
    const int grp_in_block = G == 64 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : (int)threadIdx.x / G;   // see merge_count_kernel
    const long long grp = (long long)blockIdx.x * (MERGE_BLOCK / G) + grp_in_block;
    const unsigned long long below = (1ULL << lg) - 1ULL;   // lg < 64 always
    int s1[FILL_U], n1[FILL_U], s2[FILL_U], n2[FILL_U], o[FILL_U], a[FILL_U], b[FILL_U];
    VT xa[FILL_U], xb[FILL_U];
#pragma unroll
    for (int u = 0; u < FILL_U; u++) {
        const long long row = grp * FILL_U + u;
        const bool valid = row < m;
        const long long rs = valid ? row : 0;
        const int a0 = p1[rs], a1 = p1[rs + 1], b0 = p2[rs], b1 = p2[rs + 1];
        o[u] = po[rs];
        s1[u] = a0; n1[u] = valid ? a1 - a0 : 0;
        s2[u] = b0; n2[u] = valid ? b1 - b0 : 0;
    }
#pragma unroll
    for (int u = 0; u < FILL_U; u++) {
        const bool va = lg < n1[u] && n1[u] <= G, vb = lg < n2[u] && n2[u] <= G;
        a[u] = INT_MAX; b[u] = INT_MAX; xa[u] = VT(0); xb[u] = VT(0);
        if (va) { a[u] = j1[s1[u] + lg]; xa[u] = x1[s1[u] + lg]; }
        if (vb) { b[u] = j2[s2[u] + lg]; xb[u] = x2[s2[u] + lg]; }
    }
    bool left[FILL_U];                                               // very long pairs: merge_long_fill_kernel's (see merge_count_kernel)
#pragma unroll
    for (int u = 0; u < FILL_U; u++) left[u] = false;
#pragma unroll
    for (int u = 0; u < FILL_U; u++) {
        const long long row = grp * FILL_U + u;
        const bool fits = __ballot(n1[u] > G || n2[u] > G) == 0ULL;
        if (row >= m) continue;
        if (!fits) {
            left[u] = ml.T != 0 && (n1[u] > ml.T || n2[u] > ml.T);
            if (!left[u]) fill_row_slow<G, OP, VT>(lg, j1 + s1[u], x1 + s1[u], n1[u], j2 + s2[u], x2 + s2[u], n2[u], o[u], jo, xo);
            continue;
        }
        // register-resident rows: one entry of A and one of B per lane, searches by cross-lane probes
        const bool va = lg < n1[u], vb = lg < n2[u];
        bool hit_a, hit_b;
        const int lb_a = group_lower_bound<G>(b[u], a[u], hit_a);     // entries of B below a
        hit_a = hit_a && va;
        const VT partner = __shfl(xb[u], lb_a < G ? lb_a : G - 1, G);
        const int before_a = __popcll(group_ballot<G>(hit_a) & below);
        if constexpr (INTERSECT) {
            if (hit_a) {
                const int pos = o[u] + before_a;
                jo[pos] = a[u];
                xo[pos] = combine<OP, VT>(xa[u], partner);
            }
        } else {
            if (va) {
                const int pos = o[u] + lg + lb_a - before_a;
                jo[pos] = a[u];
                xo[pos] = hit_a ? combine<OP, VT>(xa[u], partner) : xa[u];
            }
            const int lb_b = group_lower_bound<G>(a[u], b[u], hit_b);  // entries of A below b
            hit_b = hit_b && vb;
            const int before_b = __popcll(group_ballot<G>(hit_b) & below);
            if (vb && !hit_b) {
                const int pos = o[u] + lb_b + lg - before_b;
                jo[pos] = b[u];
                if constexpr (OP == MX_OP_SUB) xo[pos] = -xb[u]; else xo[pos] = xb[u];
            }
        }
    }
    if (ml.T != 0) {
#pragma unroll
        for (int u = 0; u < FILL_U; u++)
            if (left[u] && lg == 0) ml.rows[atomicAdd(ml.count, 1u)] = (int)(grp * FILL_U + u);
    }
}

// ---- very long row pairs: one workgroup per pair (see MergeLong)
struct LongPiece { int as, ae, bs, be; };
// piece k of the pair: entries [k * L, (k + 1) * L) of the longer row and the entries of the other row in the same value range
__device__ __forceinline__ LongPiece long_piece(const int32_t *__restrict__ A, int n1, const int32_t *__restrict__ B, int n2, int L, int k)
{
    const bool a_pivot = n1 >= n2;
    const int32_t *__restrict__ X = a_pivot ? A : B, *__restrict__ Y = a_pivot ? B : A;
    const int nx = a_pivot ? n1 : n2, ny = a_pivot ? n2 : n1;
    const int xs = k * L, xe = min(nx, xs + L);
    const int ys = k == 0 ? 0 : lower_bound_dev(Y, ny, X[xs]);
    const int ye = xe >= nx ? ny : lower_bound_dev(Y, ny, X[xe]);
    LongPiece p;
    p.as = a_pivot ? xs : ys; p.ae = a_pivot ? xe : ye;
    p.bs = a_pivot ? ys : xs; p.be = a_pivot ? ye : xe;
    return p;
}
// ~32 pieces per pair (four rounds of the workgroup's 8 wavefronts), at least 256 entries each
__device__ __forceinline__ int long_piece_len(int n1, int n2)
{
    const int nx = max(n1, n2);
    return max(256, (nx + MERGE_LONG_MAXP - 1) / MERGE_LONG_MAXP);
}

template <bool INTERSECT>
__global__ __launch_bounds__(MERGE_LONG_BLOCK)
void merge_long_count_kernel(const int32_t *__restrict__ p1, const int32_t *__restrict__ j1, const int32_t *__restrict__ p2,
                             const int32_t *__restrict__ j2, int32_t *__restrict__ counts, MergeLong ml)
{
    __shared__ int s_hits;
    const int nlist = (int)*ml.count;
    const int lane = lane_id(), wave = uniform(threadIdx.x / MX_WAVE);
    for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int row = ml.rows[li];
        const int a0 = p1[row], n1 = p1[row + 1] - a0, b0 = p2[row], n2 = p2[row + 1] - b0;
        const int L = long_piece_len(n1, n2), P = (max(n1, n2) + L - 1) / L;
        if (threadIdx.x == 0) s_hits = 0;
        __syncthreads();
        for (int k = wave; k < P; k += MERGE_LONG_BLOCK / MX_WAVE) {
            const LongPiece pc = long_piece(j1 + a0, n1, j2 + b0, n2, L, k);
            const int h = count_row_slow<64, true>(lane, j1 + a0 + pc.as, pc.ae - pc.as, j2 + b0 + pc.bs, pc.be - pc.bs);
            if (lane == 0 && h) atomicAdd(&s_hits, h);
        }
        __syncthreads();
        if (threadIdx.x == 0) counts[row] = INTERSECT ? s_hits : n1 + n2 - s_hits;
        __syncthreads();
    }
}

template <int OP, typename VT>
__global__ __launch_bounds__(MERGE_LONG_BLOCK)
void merge_long_fill_kernel(const int32_t *__restrict__ p1, const int32_t *__restrict__ j1, const VT *__restrict__ x1,
                            const int32_t *__restrict__ p2, const int32_t *__restrict__ j2, const VT *__restrict__ x2,
                            const int32_t *__restrict__ po, int32_t *__restrict__ jo, VT *__restrict__ xo, MergeLong ml)
{
    constexpr bool INTERSECT = (OP == MX_OP_MUL || OP == MX_OP_AND);
    static_assert(MERGE_LONG_MAXP <= MX_WAVE, "one wavefront scans the pieces");
    __shared__ int s_ph[MX_WAVE];                                     // coincidences per piece (<= MERGE_LONG_MAXP), then their exclusive prefix
    const int nlist = (int)*ml.count;
    const int lane = lane_id(), wave = uniform(threadIdx.x / MX_WAVE);
    for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int row = ml.rows[li];
        const int a0 = p1[row], n1 = p1[row + 1] - a0, b0 = p2[row], n2 = p2[row + 1] - b0;
        const long long o = po[row];
        const int L = long_piece_len(n1, n2), P = (max(n1, n2) + L - 1) / L;
        for (int k = wave; k < P; k += MERGE_LONG_BLOCK / MX_WAVE) {
            const LongPiece pc = long_piece(j1 + a0, n1, j2 + b0, n2, L, k);
            const int h = count_row_slow<64, true>(lane, j1 + a0 + pc.as, pc.ae - pc.as, j2 + b0 + pc.bs, pc.be - pc.bs);
            if (lane == 0) s_ph[k] = h;
        }
        __syncthreads();
        if (wave == 0) {                                              // exclusive prefix over <= 2048 pieces, 64 at a time
            int carry = 0;
            for (int base = 0; base < P; base += MX_WAVE) {
                const int v = base + lane < P ? s_ph[base + lane] : 0;
                int incl = v;
#pragma unroll
                for (int off = 1; off < MX_WAVE; off <<= 1) { const int up = __shfl_up(incl, off, MX_WAVE); if (lane >= off) incl += up; }
                if (base + lane < P) s_ph[base + lane] = carry + incl - v;
                carry += __shfl(incl, MX_WAVE - 1, MX_WAVE);
            }
        }
        __syncthreads();
        for (int k = wave; k < P; k += MERGE_LONG_BLOCK / MX_WAVE) {
            const LongPiece pc = long_piece(j1 + a0, n1, j2 + b0, n2, L, k);
            const long long base = INTERSECT ? o + s_ph[k] : o + pc.as + pc.bs - s_ph[k];
            fill_row_slow<64, OP, VT>(lane, j1 + a0 + pc.as, x1 + a0 + pc.as, pc.ae - pc.as, j2 + b0 + pc.bs, x2 + b0 + pc.bs, pc.be - pc.bs,
                                      base, jo, xo);
        }
        __syncthreads();
    }
}

// =====================================================================================================
// One pass over the inputs: merge_fused_kernel.
//
// count -> scan -> fill reads the index arrays twice (the count pass re-reads 8 nnz bytes and, one row pair per lane
// group with three dependent loads, ran at 1.8 TB/s: 0.45 of the 1.55 ms of CSR + CSR at 2M x 2M / nnz 1e8).  Here a
// 512-thread workgroup takes a tile of consecutive rows, every wavefront K row steps of 64 / G rows each: all loads of
// the K steps are issued up front (both rows of a pair, one entry per lane, stay in registers), the rows' output lengths
// come from the cross-lane searches, and the tile's position in the output comes from a decoupled look-back over the
// tiles' totals (one 64-bit word per tile: 2 flag bits + value, relaxed agent-scope loads / stores — the value travels
// in the same word as its flag, so nothing else needs ordering; tiles are numbered by an atomic ticket so that every
// predecessor of a running tile has started).  The entries are then placed from the registers.  The caller provides
// out_indices / out_values for the upper bound (nnz1 + nnz2, or min(nnz1, nnz2) for the intersection) like the
// reference's own scratch arrays (operators.cpp:402-406, :139-143); nnz_out comes back with the last tile.
// Row pairs that do not fit a lane group take the searches in global memory (both phases), as in the two-pass kernels.
//
// Measured at cfg4 (2M x 2M, nnz 1e8 each): 1.40-1.46 ms for `+`, the two-pass form 1.07-1.11 ms — this kernel is the
// opt-in alternative (MXGPU_MERGE_FUSED=1), not the default.  A third form was tried and dropped: 64-row tiles whose index
// windows are staged in LDS by flat 16-byte loads and whose values are requested up front (every input byte read from
// memory once, searches out of LDS in batches of four): 1.55-1.60 ms.  Taken apart on the device: 1.14 ms with the
// look-back removed, 1.07-1.2 ms with count AND fill removed — the tile protocol itself (31 k tickets on one address at
// ~10 ns each, two dependent load round trips and three barriers per 134 KB tile, four 512-thread workgroups per CU)
// costs what the whole two-pass merge does.
// =====================================================================================================
constexpr int FUSED_WAVES = 8;
constexpr int FUSED_BLOCK = FUSED_WAVES * 64;
constexpr int FUSED_LOOK = 8;                         // windows of 64 predecessor tiles read per look-back round trip

// rows per tile: FUSED_STEPS row steps per wavefront (64 / G rows each)
constexpr int FUSED_STEPS = 32;
constexpr int FUSED_U = 2;                             // row steps whose loads are in flight together

template <int G, int OP, typename VT>
__global__ __launch_bounds__(FUSED_BLOCK)
void merge_fused_kernel(int m, const int32_t *__restrict__ p1, const int32_t *__restrict__ j1, const VT *__restrict__ x1,
                        const int32_t *__restrict__ p2, const int32_t *__restrict__ j2, const VT *__restrict__ x2,
                        int32_t *__restrict__ po, int32_t *__restrict__ jo, VT *__restrict__ xo,
                        unsigned long long *__restrict__ tile_state, unsigned *__restrict__ ticket,
                        long long *__restrict__ total_out, int ntiles)
{
    constexpr bool INTERSECT = (OP == MX_OP_MUL || OP == MX_OP_AND);
    constexpr int RPS = 64 / G;                                       // rows per wavefront and step
    constexpr int WAVE_ROWS = RPS * FUSED_STEPS;
    constexpr int TILE_ROWS = FUSED_WAVES * WAVE_ROWS;
    constexpr int PER = (TILE_ROWS + FUSED_BLOCK - 1) / FUSED_BLOCK;  // rows per thread in the tile's scan
    __shared__ int cnt[TILE_ROWS + 1];                                // output length of every row, then its exclusive offset
    __shared__ int s_tile;
    __shared__ long long s_base;
    __shared__ int wave_tot[FUSED_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane % G, grp = lane / G;
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);                 // tiles in starting order: predecessors are running or done
    __syncthreads();
    const int tile = s_tile;
    const long long row0 = (long long)tile * TILE_ROWS;
    const int wl0 = wave * WAVE_ROWS;                                 // this wavefront's first row inside the tile
    const unsigned long long below = (1ULL << lg) - 1ULL;

    // ---- phase 1: output length of every row of the tile (index arrays only)
    for (int k0 = 0; k0 < FUSED_STEPS; k0 += FUSED_U) {
        int s1[FUSED_U], n1[FUSED_U], s2[FUSED_U], n2[FUSED_U], a[FUSED_U], b[FUSED_U];
#pragma unroll
        for (int u = 0; u < FUSED_U; u++) {
            const long long row = row0 + wl0 + (k0 + u) * RPS + grp;
            const bool valid = row < m;
            const long long rs = valid ? row : 0;
            const int a0 = p1[rs], a1 = p1[rs + 1], b0 = p2[rs], b1 = p2[rs + 1];
            s1[u] = a0; n1[u] = valid ? a1 - a0 : 0;
            s2[u] = b0; n2[u] = valid ? b1 - b0 : 0;
        }
#pragma unroll
        for (int u = 0; u < FUSED_U; u++) {
            a[u] = lg < n1[u] && n1[u] <= G ? j1[s1[u] + lg] : INT_MAX;
            b[u] = lg < n2[u] && n2[u] <= G ? j2[s2[u] + lg] : INT_MAX;
        }
#pragma unroll
        for (int u = 0; u < FUSED_U; u++) {
            int c;
            if (__ballot(n1[u] > G || n2[u] > G) == 0ULL) {           // every pair of this step fits its lane group
                bool hit;
                group_lower_bound<G>(b[u], a[u], hit);
                const int hits = __popcll(group_ballot<G>(hit && lg < n1[u]));
                c = INTERSECT ? hits : n1[u] + n2[u] - hits;
            } else {
                c = count_row_slow<G, INTERSECT>(lg, j1 + s1[u], n1[u], j2 + s2[u], n2[u]);
            }
            if (lg == 0) cnt[wl0 + (k0 + u) * RPS + grp] = c;
        }
    }
    __syncthreads();
    // ---- exclusive scan of the tile's row lengths (in place), tile total
    {
        int v[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) { const int r = tid * PER + i; v[i] = r < TILE_ROWS ? cnt[r] : 0; sum += v[i]; }
        int incl = sum;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) { const int up = __shfl_up(incl, o2, 64); if (lane >= o2) incl += up; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wbase = 0;
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; w++) wbase += w < wave ? wave_tot[w] : 0;
        int run = wbase + incl - sum;
#pragma unroll
        for (int i = 0; i < PER; i++) { const int r = tid * PER + i; if (r < TILE_ROWS) cnt[r] = run; run += v[i]; }
        if (tid == FUSED_BLOCK - 1) cnt[TILE_ROWS] = run;             // the tile's total
    }
    __syncthreads();
    const long long tile_total = cnt[TILE_ROWS];

    // ---- decoupled look-back over the tiles' totals (mx_common.h)
    if (wave == 0) {
        const long long excl = lookback_exclusive<FUSED_LOOK>(tile_state, tile, tile_total);
        if (lane == 0) {
            s_base = excl;
            if (tile == ntiles - 1) {
                *total_out = excl + tile_total;
                if (excl + tile_total <= (long long)INT_MAX) po[m] = (int32_t)(excl + tile_total);
            }
        }
    }
    __syncthreads();
    const long long base = s_base;

    // ---- phase 2: place the entries (the index arrays come back from L2: the tile read them a moment ago)
    for (int k0 = 0; k0 < FUSED_STEPS; k0 += FUSED_U) {
        int s1[FUSED_U], n1[FUSED_U], s2[FUSED_U], n2[FUSED_U], a[FUSED_U], b[FUSED_U];
        VT xa[FUSED_U], xb[FUSED_U];
#pragma unroll
        for (int u = 0; u < FUSED_U; u++) {
            const long long row = row0 + wl0 + (k0 + u) * RPS + grp;
            const bool valid = row < m;
            const long long rs = valid ? row : 0;
            const int a0 = p1[rs], a1 = p1[rs + 1], b0 = p2[rs], b1 = p2[rs + 1];
            s1[u] = a0; n1[u] = valid ? a1 - a0 : 0;
            s2[u] = b0; n2[u] = valid ? b1 - b0 : 0;
        }
#pragma unroll
        for (int u = 0; u < FUSED_U; u++) {
            const bool va = lg < n1[u] && n1[u] <= G, vb = lg < n2[u] && n2[u] <= G;
            a[u] = INT_MAX; b[u] = INT_MAX; xa[u] = VT(0); xb[u] = VT(0);
            if (va) { a[u] = j1[s1[u] + lg]; xa[u] = x1[s1[u] + lg]; }
            if (vb) { b[u] = j2[s2[u] + lg]; xb[u] = x2[s2[u] + lg]; }
        }
#pragma unroll
        for (int u = 0; u < FUSED_U; u++) {
            const int lr = wl0 + (k0 + u) * RPS + grp;
            const long long row = row0 + lr;
            const bool fits = __ballot(n1[u] > G || n2[u] > G) == 0ULL;
            if (row >= m) continue;
            const long long o = base + cnt[lr];
            if (lg == 0) po[row] = (int32_t)o;
            if (!fits) {
                fill_row_slow<G, OP, VT>(lg, j1 + s1[u], x1 + s1[u], n1[u], j2 + s2[u], x2 + s2[u], n2[u], o, jo, xo);
                continue;
            }
            const bool va = lg < n1[u], vb = lg < n2[u];
            bool hit_a, hit_b;
            const int lb_a = group_lower_bound<G>(b[u], a[u], hit_a);  // entries of B below a
            hit_a = hit_a && va;
            const VT partner = __shfl(xb[u], lb_a < G ? lb_a : G - 1, G);
            const int before_a = __popcll(group_ballot<G>(hit_a) & below);
            if constexpr (INTERSECT) {
                if (hit_a) {
                    jo[o + before_a] = a[u];
                    xo[o + before_a] = combine<OP, VT>(xa[u], partner);
                }
            } else {
                if (va) {
                    const long long pos = o + lg + lb_a - before_a;
                    jo[pos] = a[u];
                    xo[pos] = hit_a ? combine<OP, VT>(xa[u], partner) : xa[u];
                }
                const int lb_b = group_lower_bound<G>(a[u], b[u], hit_b);  // entries of A below b
                hit_b = hit_b && vb;
                const int before_b = __popcll(group_ballot<G>(hit_b) & below);
                if (vb && !hit_b) {
                    const long long pos = o + lb_b + lg - before_b;
                    jo[pos] = b[u];
                    if constexpr (OP == MX_OP_SUB) xo[pos] = -xb[u]; else xo[pos] = xb[u];
                }
            }
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
// (also sets, for the launch that follows on this thread, from which length on a row pair is left to the one-workgroup-per-
// pair kernels: 8 mean rows, at least 1024 entries — with a mean row of 500, a 1,024-entry threshold sent every seventh
// pair of a log-normal matrix there, each to a workgroup with two busy wavefronts: 0.82 ms where the lane groups take 0.27)
static thread_local int g_merge_long_T = 0;
// The export level has the row pointers on the host and tells the launches that follow on this thread when many row pairs
// would not fit the width the MEAN row length suggests (api.hip mx_csr_elemwise_begin): one size up then — measured,
// CSR + CSR at 1e6 x 1e5, mean 32: rows of equal length 0.462 -> 0.51 ms, log-normal rows 0.78 -> 0.58 (sigma 0.5),
// 0.96 -> 0.74 (sigma 1).  Off by default and after the export.
static thread_local bool g_merge_widen = false;
void merge_group_widen(bool on) { g_merge_widen = on; }
int merge_group(int m, int64_t nnz1, int64_t nnz2)
{
    g_merge_long_T = 0;                                               // (off: sizes unknown, or a small merge — below)
    if (nnz1 < 0 || nnz2 < 0 || m <= 0) return 32;
    const double avg = (double)(nnz1 > nnz2 ? nnz1 : nnz2) / (double)m;
    // from 2^21 entries on: the list costs a memset, a launch and the scratch's stream bookkeeping per pass — ~20 us, which
    // showed as +15 % on a 0.29-ms merge of 1e5 entries (bench.py export_small_calls)
    if (nnz1 + nnz2 >= (1LL << 21)) g_merge_long_T = 8.0 * avg > (double)MERGE_LONG_T ? (avg < 1e8 ? (int)(8.0 * avg) : INT_MAX) : MERGE_LONG_T;
    const int G = pick_group(avg, 8);
    if (const char *e = getenv("MXGPU_MERGE_G")) {                    // (tools/merge_skew_probe.py: the lane-group width under skew)
        const int g = atoi(e);
        if (g == 8 || g == 16 || g == 32 || g == 64) return g;
    }
    return g_merge_widen && G < 64 ? 2 * G : G;
}

// the list of the very long pairs of one launch (per-thread grow-only scratch; off for small operands: one more launch
// would show in a 20-us call, and their tails are short)
static int merge_long_begin(int m, hipStream_t st, MergeLong *ml)
{
    ml->count = nullptr; ml->rows = nullptr; ml->T = 0;
    if (m < 2048 || g_merge_long_T == 0) return 0;
    char *buf = (char *)scratch_buffer(MX_SCRATCH_MERGE_LONG, 64 + (size_t)m * sizeof(int));
    if (!buf) return set_error("merge: cannot allocate %zu bytes for the list of long rows", 64 + (size_t)m * sizeof(int));
    scratch_acquire(MX_SCRATCH_MERGE_LONG, st);
    ml->count = (unsigned *)buf;
    ml->rows = (int *)(buf + 64);
    ml->T = g_merge_long_T;
    MX_HIP(hipMemsetAsync(ml->count, 0, sizeof(unsigned), st));
    return 0;
}

int merge_count_launch(int op, int G, int m, const int32_t *p1, const int32_t *j1, const int32_t *p2,
                       const int32_t *j2, int32_t *counts, hipStream_t st)
{
    const bool isect = op_is_intersect(op);
    MergeLong ml;
    if (merge_long_begin(m, st, &ml)) return 1;
#define MX_CASE(GG)                                                                                   \
    case GG: {                                                                                        \
        const unsigned grid = (unsigned)ceil_div(m, (MERGE_BLOCK / GG) * COUNT_U);                    \
        if (isect) hipLaunchKernelGGL((merge_count_kernel<GG, true>), dim3(grid), dim3(MERGE_BLOCK), 0, st, \
                                      m, p1, j1, p2, j2, counts, ml);                                 \
        else hipLaunchKernelGGL((merge_count_kernel<GG, false>), dim3(grid), dim3(MERGE_BLOCK), 0, st, \
                                m, p1, j1, p2, j2, counts, ml);                                       \
        break;                                                                                        \
    }
    switch (G) { MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64) default: return set_error("merge: bad group %d", G); }
#undef MX_CASE
    if (ml.T) {
        const unsigned grid = (unsigned)std::min(1024, m / 64 + 1);
        if (isect) hipLaunchKernelGGL((merge_long_count_kernel<true>), dim3(grid), dim3(MERGE_LONG_BLOCK), 0, st, p1, j1, p2, j2, counts, ml);
        else hipLaunchKernelGGL((merge_long_count_kernel<false>), dim3(grid), dim3(MERGE_LONG_BLOCK), 0, st, p1, j1, p2, j2, counts, ml);
        scratch_done(MX_SCRATCH_MERGE_LONG, st);
    }
    MX_LAUNCH_CHECK();
    return 0;
}

template <int OP, typename VT>
static int merge_fill_op(int G, int m, const int32_t *p1, const int32_t *j1, const void *x1, const int32_t *p2,
                         const int32_t *j2, const void *x2, const int32_t *po, int32_t *jo, void *xo, hipStream_t st)
{
    MergeLong ml;
    if (merge_long_begin(m, st, &ml)) return 1;
#define MX_CASE(GG)                                                                                   \
    case GG: {                                                                                        \
        const unsigned grid = (unsigned)ceil_div(m, (MERGE_BLOCK / GG) * FILL_U);                     \
        hipLaunchKernelGGL((merge_fill_kernel<GG, OP, VT>), dim3(grid), dim3(MERGE_BLOCK), 0, st, m,  \
                           p1, j1, (const VT *)x1, p2, j2, (const VT *)x2, po, jo, (VT *)xo, ml);     \
        break;                                                                                        \
    }
    switch (G) { MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64) default: return set_error("merge: bad group %d", G); }
    if (ml.T) {
        hipLaunchKernelGGL((merge_long_fill_kernel<OP, VT>), dim3((unsigned)std::min(1024, m / 64 + 1)), dim3(MERGE_LONG_BLOCK), 0, st,
                           p1, j1, (const VT *)x1, p2, j2, (const VT *)x2, po, jo, (VT *)xo, ml);
        scratch_done(MX_SCRATCH_MERGE_LONG, st);
    }
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
        case MX_OP_FIRST: return merge_fill_op<MX_OP_FIRST, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, st);
        default: return set_error("merge: unknown op %d", op);
    }
}

template <int OP, typename VT>
static int merge_fused_op(int G, int m, const int32_t *p1, const int32_t *j1, const void *x1, const int32_t *p2,
                          const int32_t *j2, const void *x2, int32_t *po, int32_t *jo, void *xo,
                          unsigned long long *tile_state, unsigned *ticket, long long *total, hipStream_t st)
{
#define MX_CASE(GG)                                                                                               \
    case GG: {                                                                                                    \
        const int ntiles = (int)ceil_div(m, FUSED_WAVES * (64 / GG) * FUSED_STEPS);                               \
        hipLaunchKernelGGL((merge_fused_kernel<GG, OP, VT>), dim3((unsigned)ntiles), dim3(FUSED_BLOCK), 0, st, m,  \
                           p1, j1, (const VT *)x1, p2, j2, (const VT *)x2, po, jo, (VT *)xo, tile_state, ticket,   \
                           total, ntiles);                                                                        \
        break;                                                                                                    \
    }
    switch (G) { MX_CASE(8) MX_CASE(16) MX_CASE(32) MX_CASE(64) default: return set_error("merge: bad group %d", G); }
#undef MX_CASE
    MX_LAUNCH_CHECK();
    return 0;
}

int merge_fused_launch(int op, int G, int m, const int32_t *p1, const int32_t *j1, const void *x1, const int32_t *p2,
                       const int32_t *j2, const void *x2, int32_t *po, int32_t *jo, void *xo, void *workspace,
                       hipStream_t st)
{
    // workspace: [int64 total][uint32 ticket, pad][uint64 tile_state[ntiles]]
    const int64_t ntiles = ceil_div(m, FUSED_WAVES * (64 / G) * FUSED_STEPS);
    long long *total = (long long *)workspace;
    unsigned *ticket = (unsigned *)(total + 1);
    unsigned long long *tile_state = (unsigned long long *)(total + 2);
    MX_HIP(hipMemsetAsync(workspace, 0, 16 + (size_t)ntiles * 8, st));
    switch (op) {
        case MX_OP_ADD: return merge_fused_op<MX_OP_ADD, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, tile_state, ticket, total, st);
        case MX_OP_SUB: return merge_fused_op<MX_OP_SUB, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, tile_state, ticket, total, st);
        case MX_OP_MUL: return merge_fused_op<MX_OP_MUL, double>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, tile_state, ticket, total, st);
        case MX_OP_OR:  return merge_fused_op<MX_OP_OR, int32_t>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, tile_state, ticket, total, st);
        case MX_OP_XOR: return merge_fused_op<MX_OP_XOR, int32_t>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, tile_state, ticket, total, st);
        case MX_OP_AND: return merge_fused_op<MX_OP_AND, int32_t>(G, m, p1, j1, x1, p2, j2, x2, po, jo, xo, tile_state, ticket, total, st);
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

// One pass (merge_fused_kernel).  out_indices / out_values must hold the upper bound of the result — nnz1 + nnz2 entries
// for ADD / SUB / OR / XOR, min(nnz1, nnz2) for MUL / AND; out_indptr[m+1].  workspace: mxd_merge_fused_workspace_bytes(m).
// *nnz_out_host is written after an internal synchronisation (the one host round trip).
extern "C" size_t mxd_merge_fused_workspace_bytes(int m)
{
    // tiles hold at least FUSED_WAVES * FUSED_STEPS rows (G = 64)
    return 16 + 8 * (size_t)mx::ceil_div(m > 0 ? m : 1, mx::FUSED_WAVES * mx::FUSED_STEPS) + 64;
}

extern "C" int mxd_csr_merge_fused(int op, int m, const int32_t *indptr1, const int32_t *indices1, const void *values1,
                                   int64_t nnz1, const int32_t *indptr2, const int32_t *indices2, const void *values2,
                                   int64_t nnz2, int32_t *out_indptr, int32_t *out_indices, void *out_values,
                                   void *workspace, int64_t *nnz_out_host, void *stream)
{
    MX_REQUIRE(m >= 0, "mxd_csr_merge_fused: negative m");
    MX_REQUIRE(out_indptr && workspace, "mxd_csr_merge_fused: null pointer");
    MX_REQUIRE((mx::op_is_intersect(op) ? (nnz1 < nnz2 ? nnz1 : nnz2) : nnz1 + nnz2) <= (int64_t)INT_MAX || nnz1 < 0 || nnz2 < 0,
               "mxd_csr_merge_fused: the result's upper bound exceeds R's int32 index range (use the count -> fill pair)");
    hipStream_t st = mx::as_stream(stream);
    if (m == 0) {
        MX_HIP(hipMemsetAsync(out_indptr, 0, sizeof(int32_t), st));
        if (nnz_out_host) *nnz_out_host = 0;
        return 0;
    }
    const int rc = mx::merge_fused_launch(op, mx::merge_group(m, nnz1, nnz2), m, indptr1, indices1, values1, indptr2, indices2,
                                          values2, out_indptr, out_indices, out_values, workspace, st);
    if (rc) return rc;
    if (nnz_out_host) {
        if (mx::read_back_small(nnz_out_host, workspace, sizeof(int64_t), st)) return 1;
        MX_REQUIRE(*nnz_out_host <= (int64_t)INT_MAX, "result has %lld entries: exceeds R's int32 index range",
                   (long long)*nnz_out_host);
    }
    return 0;
}

// Rows of uneven length at the DEVICE level (round 6): the lane-group width follows the mean row length; with skewed rows the pairs
// that do not fit take the sliding windows, and one width up is faster (1e6 x 1e5, 32 per row, log-normal sigma 1: `+` 0.866 -> 0.746 ms,
// `*` 0.618 -> 0.524; rows of equal length: 0.45 -> 0.51, hence a hint and not the default).  The export level decides this from the
// host row pointers by itself; a device-level caller that knows its operands (matrixextra_amd/device.py: the cached matrix profile's
// cv) says so here — for the merges this thread launches from now on, until told otherwise.
extern "C" int mxd_csr_merge_rows_uneven(int on)
{
    mx::merge_group_widen(on != 0);
    return 0;
}
