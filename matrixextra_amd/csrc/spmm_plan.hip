// spmm_plan.hip — the planned panel-sweep SpMM (v3, what MX_SPMM_AUTO runs for large operands): plan construction
// kernels, the sweep kernel, the plan object and its C-ABI (mxd_spmm_plan_*).
#include "spmm_common.h"
#include <algorithm>

namespace mx {

// =====================================================================================================
// v3 "planned panel sweep".
//
// PMC on v2 (profiles/r01_v2_*): with column panels + the XCD timing barrier the L2 hit rate only reaches
// 60 % because every (row, panel) visit re-reads the row's (j, a) chunk — with P panels the CSR arrays are
// streamed ~2P times per XCD and that traffic, not B, dominates and evicts the panel.  v3 fixes the data
// layout instead of the loop: a *plan* regroups A's entries by (octet of 8 row-bundles, panel) and
// interleaves the 8 bundles of an octet in batches of 8 steps (slot 64*batch + 8*g + u = step 8*batch + u of
// bundle g), so that
//   * one wavefront (8 lane groups = 8 bundles) reads 64 consecutive plan entries per 8 steps — every entry of A
//     is read exactly once per slab, coalesced, and reaches its lane group by a DPP row broadcast;
//   * entries of a bundle inside a panel are ordered by row, the group accumulates the current row in
//     registers and folds it into the bundle's accumulators in LDS when the row changes (only that group
//     touches those LDS rows: plain read-modify-write, no atomics);
//   * all workgroups of an XCD group stay close to the same panel (same code on statistically identical data;
//     optional timing barrier), whose slab-major copy of B
//     (K/P x 128 B, contiguous) fits the XCD's L2.
// Entry = int32 (col | local_row << 27; padding = zero row of the packed B, value 0) + f64 value; plan bytes ~ the CSR arrays (octet lengths rounded to 8 steps).
// Summation order: CSR order inside a (row, panel), panels added in ascending order — a regrouping of the
// reference's sequential sum (tolerance-level difference, not bitwise).  Works for unsorted rows too.
// =====================================================================================================
constexpr int PLAN_RB = 8;                         // rows per bundle (owned by one 8-lane group)
constexpr int PLAN_OCT_ROWS = PLAN_RB * 8;         // rows per octet (one wavefront)
// wavefronts per workgroup (template parameter WAVES): 16 = ONE 1024-thread workgroup with 128 KiB of LDS per CU,
// 8 = two 512-thread workgroups with 64 KiB each (one's epilogue overlaps the other's sweep), 4 = four.
constexpr int PLAN_MAXP = 64;
constexpr int PLAN_DEFAULT_WG_PER_CU = 1;
constexpr int PLAN_GEN_OCTS = 16;                  // octets one 16-wavefront workgroup sweeps together
constexpr int PLAN_CHUNK = 4;                      // batches of 8 steps fetched per plan read
constexpr int PLAN_TAIL_SLOTS = 512;               // readable padding behind the last octet (2 chunks)
constexpr int PLAN_ROW_SHIFT = 25;                 // entry = col (25 bits) | slot of the row in the octet (6 bits) << 25 | shared << 31

// Plan construction: one 512-thread workgroup per octet, wavefront g streams the entries of rows 8g..8g+7 (contiguous
// in the CSR arrays) 64 at a time, fully coalesced; the position of an entry inside a stream is a per-panel running
// count kept one panel per lane plus a ballot prefix.
//
// An octet's entries are dealt to its 8 streams (the 8 lane groups of the sweeping wavefront) in one of two layouts:
//   bundle layout  stream g = rows 8g..8g+7, panel after panel.  The octet is as long as its longest bundle.
//   dealt layout   per panel, the octet's entries of that panel (row after row) are cut into 8 equal pieces, stream g
//                  gets piece g: every stream is ceil(T_p / 8) long in panel p whatever the row lengths, a long row is
//                  shared by several lane groups (entries flagged `shared`: their folds must be atomic).  The octet is
//                  sum_p ceil(T_p / 8) long — at most 7 padding slots per panel.
// The count pass picks the dealt layout when it saves at least one chunk of 32 steps: rows of equal length stay in the
// bundle layout (no padding at all), log-normal row lengths (sigma 1: bundle layout 1.84x the CSR) are dealt.
constexpr int PLAN_LD = 8;
// col / panel_cols without the integer divide: float estimate (col < 2^25 is exact in float up to 2^24, so one
// correction step either way), clamped to the last panel
__device__ __forceinline__ int panel_of(int col, int panel_cols, float inv_pc, int npanels)
{
    int q = (int)((float)col * inv_pc);
    const int r = col - q * panel_cols;
    q += r >= panel_cols ? 1 : (r < 0 ? -1 : 0);
    return q < npanels ? q : npanels - 1;
}

// The lanes of the wavefront that hold an entry of the same panel as this lane: log2(P) ballots (one per bit of the panel
// id) instead of one ballot per panel.  Lanes without an entry (pan < 0) take lane 0's panel and are masked out of
// everybody's set by `valid`; their own result is not used.
__device__ __forceinline__ unsigned long long same_panel_lanes(int pan, int nbits, unsigned long long valid)
{
    unsigned long long same = valid;
    for (int b = 0; b < nbits; b++) {
        const bool bit = (pan >> b) & 1;
        const unsigned long long bb = __ballot(bit);
        same &= bit ? bb : ~bb;
    }
    return same;
}
// counts[q] += (entries of panel q in this chunk), counts held one panel per lane: every lane pushes its panel's count to
// lane `pan` (ds_permute: all pushers to one lane carry the same value; lanes nobody pushes to receive 0)
__device__ __forceinline__ int push_panel_counts(int pan, int cnt)
{
    return __builtin_amdgcn_ds_permute(pan << 2, cnt);
}

// pass 1, per octet:  steps[oct] = length in steps (whole batches of 8); layout[oct]; pstart[oct][0..P] = relative start of
// every panel in the streams (bundle layout: mean over the 8 bundles; [P] = unpadded length);
// bpo[oct][g][p] = bundle layout: start of panel p in stream g / dealt layout: entries of panel p in rows before 8g;
// bpo[oct][8][p] = entries of panel p in the octet (dealt layout).
__global__ __launch_bounds__(512)
void plan_count_kernel(int m, int npanels, int panel_cols, const int32_t *__restrict__ indptr,
                       const int32_t *__restrict__ indices, int32_t *__restrict__ steps,
                       int32_t *__restrict__ bpo, int noct, long long *__restrict__ nnz_out,
                       unsigned char *__restrict__ layout, int32_t *__restrict__ pstart,
                       long long *__restrict__ ndealt)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *nnz_out = indptr[m] - indptr[0];     // rides back with the step total (a row block of a larger CSR keeps absolute offsets)
    __shared__ int cnt_lds[8][64];                                   // entries per (stream, panel)
    __shared__ int exc_lds[8][64];                                   // start of the panel in the bundle's stream
    __shared__ int totals[8];
    __shared__ int dealt_flag;
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int oct = blockIdx.x;
    const int row0 = oct * PLAN_OCT_ROWS + g * PLAN_RB;
    const int s = indptr[min(row0, m)], e = indptr[min(row0 + PLAN_RB, m)];
    // Bundles that are already balanced (longest = ceil(total / 8), in whole batches — rows of equal length): no deal
    // can be shorter, so the layout is decided from the row pointers alone and the fill counts the panels itself
    // (layout 2) — this pass does not touch the 4 bytes per entry of `indices` for such octets.
    if (lane == 0) totals[g] = e - s;
    __syncthreads();
    {
        int bundle_len = 0, total = 0;
#pragma unroll
        for (int gg = 0; gg < 8; gg++) { bundle_len = max(bundle_len, totals[gg]); total += totals[gg]; }
        if (((bundle_len + 7) >> 3) == ((((total + 7) >> 3) + 7) >> 3)) {                // uniform over the workgroup
            if (threadIdx.x == 0) {
                steps[oct] = (bundle_len + 7) & ~7;
                layout[oct] = 2;
                pstart[(size_t)oct * (npanels + 1) + npanels] = bundle_len;
            }
            return;
        }
    }
    int mine = 0;                                                    // lane p accumulates the count of panel p
    const float inv_pc = 1.0f / (float)panel_cols;
    const int nbits = 32 - __builtin_clz((unsigned)(npanels > 1 ? npanels - 1 : 1));    // bits of a panel id
    // PLAN_LD chunks of 64 entries per pass: the loads of a pass are issued together (the kernel is latency-bound:
    // a bundle is only ~256 entries)
    for (int k0 = s; k0 < e; k0 += 64 * PLAN_LD) {
        int col[PLAN_LD];
#pragma unroll
        for (int c = 0; c < PLAN_LD; c++) {
            const int k = k0 + 64 * c + lane;
            col[c] = k < e ? indices[k] : -1;
        }
#pragma unroll
        for (int c = 0; c < PLAN_LD; c++) {
            if (k0 + 64 * c >= e) break;                             // uniform
            const bool has = col[c] >= 0;
            const unsigned long long valid = __ballot(has);
            int pan = has ? panel_of(col[c], panel_cols, inv_pc, npanels) : 0;
            pan = has ? pan : __builtin_amdgcn_readfirstlane(pan);   // lane 0 always holds an entry of a non-empty chunk
            const unsigned long long same = same_panel_lanes(pan, nbits, valid);
            mine += push_panel_counts(pan, __popcll(same));          // npanels <= 64: one lane per panel
        }
    }
    // exclusive prefix over the panels (lanes 0..npanels-1)
    int incl = lane < npanels ? mine : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    cnt_lds[g][lane] = lane < npanels ? mine : 0;
    exc_lds[g][lane] = incl - mine;
    __syncthreads();
    if (g == 0) {
        int T = 0, meanstart = 0, bundle_len = 0;
#pragma unroll
        for (int gg = 0; gg < 8; gg++) {
            T += cnt_lds[gg][lane];
            meanstart += exc_lds[gg][lane];
            bundle_len = max(bundle_len, totals[gg]);
        }
        const int L = (T + 7) >> 3;                                  // steps of panel `lane` in the dealt layout
        int S = L;                                                   // inclusive prefix of L over the panels
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(S, off, 64);
            if (lane >= off) S += up;
        }
        const int dealt_len = __shfl(S, 63, 64);
        const bool dealt = ((dealt_len + 7) >> 3) < ((bundle_len + 7) >> 3);
        const int len = dealt ? dealt_len : bundle_len;
        if (lane < npanels) pstart[(size_t)oct * (npanels + 1) + lane] = dealt ? S - L : meanstart / 8;
        if (lane == 0) {
            pstart[(size_t)oct * (npanels + 1) + npanels] = len;
            steps[oct] = (len + 7) & ~7;                                 // whole batches of 8 steps
            layout[oct] = dealt ? 1 : 0;
            dealt_flag = dealt ? 1 : 0;
            if (dealt) atomicAdd((unsigned long long *)ndealt, 1ULL);
        }
        if (lane < npanels) bpo[((size_t)oct * 9 + 8) * npanels + lane] = T;
    }
    __syncthreads();
    if (lane < npanels) {
        int v = incl - mine;                                         // bundle layout: my stream's panel start
        if (dealt_flag) {                                            // dealt layout: entries of the panel in earlier rows
            v = 0;
            for (int gg = 0; gg < g; gg++) v += cnt_lds[gg][lane];
        }
        bpo[((size_t)oct * 9 + g) * npanels + lane] = v;
    }
}

__device__ __forceinline__ bool plan_fits(long long total_steps, long long cap_slots)
{
    return total_steps >= 0 && total_steps <= (long long)INT_MAX && total_steps * 8 + PLAN_TAIL_SLOTS <= cap_slots;
}

// pass 2: scatter the entries to their interleaved slots (batch of 8 steps = 64 slots laid out [stream][step])
__global__ __launch_bounds__(512)
void plan_fill_kernel(int m, int npanels, int panel_cols, const int32_t *__restrict__ indptr,
                      const int32_t *__restrict__ indices, const double *__restrict__ values,
                      const int32_t *__restrict__ oct_off, const int32_t *__restrict__ bpo,
                      int32_t *__restrict__ pcol, double *__restrict__ pval, int noct, int pad_col,
                      int32_t *__restrict__ step_off, const unsigned char *__restrict__ layout,
                      int32_t *__restrict__ pstart, long long *__restrict__ ndealt, long long cap_slots,
                      const long long *__restrict__ total_steps)
{
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int oct = blockIdx.x;
    if (oct == 0 && threadIdx.x == 0) *ndealt = 0;                   // read back already: ready for the next build
    // launched before the host knew the plan's size (plan_build): nothing is written unless it fits.  The test reads
    // the scan's int64 total, not the int32 oct_off[noct] (which wraps for a corrupt indptr and could pass).
    if (!plan_fits(*total_steps, cap_slots)) return;
    const int lay = layout[oct];                                     // uniform: 0 bundle, 1 dealt, 2 bundle, not counted yet
    const bool dealt = lay == 1;
    const int row0 = oct * PLAN_OCT_ROWS + g * PLAN_RB;
    int rp[PLAN_RB + 1];                                             // the bundle's row pointers (wave-uniform)
#pragma unroll
    for (int r = 0; r <= PLAN_RB; r++) rp[r] = uniform(indptr[min(row0 + r, m)]);
    const int s = rp[0], e = rp[PLAN_RB];
    const long long base = oct_off[oct];
    // lane p: next free step of panel p in my stream (bundle layout) / next rank inside panel p (dealt layout)
    int nextstep = lane < npanels && lay != 2 ? bpo[((size_t)oct * 9 + g) * npanels + lane] : 0;
    // dealt layout, lane p: steps per stream in panel p and where the panel starts
    int Lp = 1, Sp = 0, Tp = 0;
    if (dealt && lane < npanels) {
        Tp = bpo[((size_t)oct * 9 + 8) * npanels + lane];
        Lp = (Tp + 7) >> 3;
        Sp = pstart[(size_t)oct * (npanels + 1) + lane];
    }
    const unsigned long long below = (1ULL << lane) - 1ULL;
    const float inv_pc = 1.0f / (float)panel_cols;
    const int nbits = 32 - __builtin_clz((unsigned)(npanels > 1 ? npanels - 1 : 1));    // bits of a panel id
    // tag = slot | shared << 6 (bit 31 of the entry).  Dealt octets: set afterwards, by plan_flag_kernel, on the LAST row
    // segment of every piece only (the padding entries below carry it from the start)
    const int shared_bit = 0;
    int colv[PLAN_LD];
    double av[PLAN_LD];
    auto load_pass = [&](int k0) {                                   // all loads of a pass in flight together
#pragma unroll
        for (int c = 0; c < PLAN_LD; c++) {
            const int k = k0 + 64 * c + lane;
            colv[c] = -1; av[c] = 0.0;
            if (k < e) { colv[c] = indices[k]; av[c] = values[k]; }
        }
    };
    if (lay == 2) {
        // not counted by pass 1 (bundles already balanced): count my stream's entries per panel here — from the
        // registers of the first pass, which the placement below re-uses (bundles of up to 256 entries are read once)
        __shared__ int exc2[8][64];
        int mine = 0;
        load_pass(s);
        auto count_pass = [&](int k0) {
#pragma unroll
            for (int c = 0; c < PLAN_LD; c++) {
                if (k0 + 64 * c >= e) break;                         // uniform
                const bool has = colv[c] >= 0;
                const unsigned long long valid = __ballot(has);
                int pan = has ? panel_of(colv[c], panel_cols, inv_pc, npanels) : 0;
                pan = has ? pan : __builtin_amdgcn_readfirstlane(pan);
                mine += push_panel_counts(pan, __popcll(same_panel_lanes(pan, nbits, valid)));
            }
        };
        count_pass(s);
        for (int k0 = s + 64 * PLAN_LD; k0 < e; k0 += 64 * PLAN_LD) { load_pass(k0); count_pass(k0); }
        int incl = lane < npanels ? mine : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        nextstep = incl - mine;                                      // start of panel `lane` in my stream
        exc2[g][lane] = nextstep;
        __syncthreads();
        if (g == 0 && lane < npanels) {                              // mean panel start over the 8 bundles (meeting points)
            int sum = 0;
#pragma unroll
            for (int gg = 0; gg < 8; gg++) sum += exc2[gg][lane];
            pstart[(size_t)oct * (npanels + 1) + lane] = sum / 8;
        }
        if (e - s > 64 * PLAN_LD) load_pass(s);                      // longer bundles: the first pass again
    }
    for (int k0 = s; k0 < e; k0 += 64 * PLAN_LD) {
        if (lay != 2 || k0 > s) load_pass(k0);
#pragma unroll
        for (int c = 0; c < PLAN_LD; c++) {
            if (k0 + 64 * c >= e) break;                             // uniform
            const int k = k0 + 64 * c + lane;
            const int col = colv[c];
            const bool has = col >= 0;
            const unsigned long long valid = __ballot(has);
            int pan = 0, slot = g * PLAN_RB;                         // slot: position of the entry's row in the octet
            if (has) {
                pan = panel_of(col, panel_cols, inv_pc, npanels);
#pragma unroll
                for (int r = 1; r < PLAN_RB; r++) slot += k >= rp[r];
            }
            pan = has ? pan : __builtin_amdgcn_readfirstlane(pan);   // lane 0 always holds an entry of a non-empty chunk
            const unsigned long long same = same_panel_lanes(pan, nbits, valid);
            // bundle: my step in the octet / dealt: rank in the panel  = my panel's running count + my rank in the chunk
            int t = __shfl(nextstep, pan, 64) + __popcll(same & below);
            nextstep += push_panel_counts(pan, __popcll(same));
            int stream = g;
            if (dealt) {                                             // rank t of panel `pan` -> (stream, step)
                const int L = max(__shfl(Lp, pan, 64), 1), S = __shfl(Sp, pan, 64);
                stream = (int)((float)t / (float)L);
                stream += (stream + 1) * L <= t ? 1 : (stream * L > t ? -1 : 0);
                t = S + t - stream * L;
            }
            if (has) {
                // ONE pair of stores per chunk (inside the panel loop it was one pair per panel, each with 1/P of the
                // lanes).  Slot layout inside a batch of 8 steps: [stream][step] — lane 8g+u of the reading wavefront
                // holds stream g's entry for step u, i.e. inside g's own lane group (DPP broadcast).
                const long long dst = (base + (t & ~7)) * 8 + stream * 8 + (t & 7);
                pcol[dst] = col | ((slot | shared_bit) << PLAN_ROW_SHIFT);
                pval[dst] = av[c];
            }
        }
    }
    // Padding: a no-op entry — value 0, column `pad_col` (the all-zero extra row of the packed B), and a row tag that
    // is at worst one harmless extra fold.  0 * 0 added to an accumulator that is never -0.0 changes no bit.
    const int steps_oct = oct_off[oct + 1] - (int)base;
    if (!dealt) {
        int last_slot = g * PLAN_RB;                                 // my stream's last row: not even a row switch
        if (e > s) {
#pragma unroll
            for (int r = 1; r < PLAN_RB; r++) last_slot += (e - 1) >= rp[r];
        }
        for (long long t = (e - s) + lane; t < steps_oct; t += 64) {
            const long long dst = (base + (t & ~7LL)) * 8 + g * 8 + (t & 7);
            pcol[dst] = pad_col | (last_slot << PLAN_ROW_SHIFT);
            pval[dst] = 0.0;
        }
    } else {
        const int padtag = (g * PLAN_RB) | 64;
        // holes at the end of every panel (the last pieces are up to 7 entries short): wavefront g takes the panels
        // p = g, g + 8, ...; lane j < 8 the j-th hole
        for (int pq = g; pq < npanels; pq += 8) {
            const int T = __shfl(Tp, pq, 64), L = __shfl(Lp, pq, 64), S = __shfl(Sp, pq, 64);
            const int q = T + lane;
            if (lane < 8 && q < 8 * L) {
                const int stream = q / (L > 0 ? L : 1), t = S + q - stream * L;
                const long long dst = (base + (t & ~7)) * 8 + stream * 8 + (t & 7);
                pcol[dst] = pad_col | (padtag << PLAN_ROW_SHIFT);
                pval[dst] = 0.0;
            }
        }
        // tail of all 8 streams up to the octet's (batch-rounded) length: wavefront g pads stream g
        const int used = pstart[(size_t)oct * (npanels + 1) + npanels];
        for (long long t = used + lane; t < steps_oct; t += 64) {
            const long long dst = (base + (t & ~7LL)) * 8 + g * 8 + (t & 7);
            pcol[dst] = pad_col | (padtag << PLAN_ROW_SHIFT);
            pval[dst] = 0.0;
        }
    }
    // PLAN_TAIL_SLOTS padding slots behind the last octet: the kernel's read-ahead runs two batches past an octet
    if (oct == noct - 1) {
        static_assert(PLAN_TAIL_SLOTS == 512, "one slot per thread of the last block");
        const long long dst = (long long)oct_off[noct] * 8 + threadIdx.x;
        pcol[dst] = pad_col;
        pval[dst] = 0.0;
    }
}

// Dealt octets, after the fill: which folds have to be atomic?  (Round 5: `shared` used to be set on every entry of a dealt
// octet, and the f32 sweep — whose LDS float atomics run at about one LANE per clock; ds_add_f64 does not have that
// problem — took 22 ms on the cfg5 shard with log-normal rows where equal rows take 3.6.  Flagging the last row segment of
// every piece: 6.1 ms; the rule below: 4.2.)
// The 8 lane groups of the sweeping wavefront run in lockstep and a panel starts at the same step in all 8 streams, so at
// any step all groups are inside the same panel, where a row's entries form one run of ranks cut at the piece boundaries.
// A group folds a row segment when it meets the next row:
//   * a segment that is not the last of its piece is folded INSIDE the panel (at a step >= 1 of it); a (row, panel) has at
//     most one such segment (the run's pieces in between are whole pieces, i.e. last segments), so two of these folds
//     never meet in one row;
//   * the last segment of a piece is folded at step 0 of the next panel — by all 8 groups at once, and only they fold at
//     that step; two of them meet in one row exactly when two pieces of the panel END with the same row (the later one
//     is then that row from start to end).  Rows are contiguous, so it is enough to compare with the neighbouring pieces;
//   * the last piece of a panel may end in padding holes (tag = pad, flagged by the fill): its last segment is then
//     folded at the first hole, inside the panel — it is the panel's last row, which has no non-last segment elsewhere.
// Only the second case needs an atomic fold.  An atomic fold and a plain read-modify-write of the same row in the same
// step are separate LDS instructions of one wavefront, which execute in order.  (A segment that continues with the same
// row in the next panel is not folded at the boundary at all: it becomes part of the next panel's segment.)
// One lane per (panel, stream) compares and, if need be, walks its piece backwards from its last entry.
__global__ __launch_bounds__(512)
void plan_flag_kernel(int noct, int npanels, const int32_t *__restrict__ oct_off, const int32_t *__restrict__ bpo,
                      const int32_t *__restrict__ pstart, const unsigned char *__restrict__ layout, int32_t *__restrict__ pcol,
                      long long cap_slots, const long long *__restrict__ total_steps)
{
    if (!plan_fits(*total_steps, cap_slots)) return;                 // the fill wrote nothing either
    const int oct = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (oct >= noct || layout[oct] != 1) return;
    const long long base = oct_off[oct];
    for (int idx = lane; idx < 8 * npanels; idx += 64) {
        const int p = idx >> 3, g = idx & 7;
        const int T = bpo[((size_t)oct * 9 + 8) * npanels + p], L = (T + 7) >> 3, S = pstart[(size_t)oct * (npanels + 1) + p];
        // slot (row position in the octet) of the last entry of stream gg's piece, -1 for an empty piece
        auto last_slot = [&](int gg) -> int {
            if (gg < 0 || gg > 7) return -1;
            const int first = gg * L, last = min(first + L, T) - 1;  // ranks of the piece's entries in the panel
            if (last < first) return -1;
            const int t = S + last - first;
            return (pcol[(base + (t & ~7)) * 8 + gg * 8 + (t & 7)] >> PLAN_ROW_SHIFT) & 63;
        };
        const int tag = last_slot(g);
        if (tag < 0 || (tag != last_slot(g - 1) && tag != last_slot(g + 1))) continue;
        const int first = g * L, last = min(first + L, T) - 1;
        for (int t = S + last - first; t >= S; t--) {
            const long long at = (base + (t & ~7)) * 8 + g * 8 + (t & 7);
            const int c = pcol[at];
            if (((c >> PLAN_ROW_SHIFT) & 63) != tag) break;
            pcol[at] = c | (int)0x80000000u;
        }
    }
}

// The order in which the sweep takes the octets (round 3).  A generation = the 16 octets one workgroup sweeps together, and
// its wavefronts meet at every panel boundary: a generation lasts as long as its LONGEST octet.  With rows of very uneven
// length (log-normal, sigma 1: octets of 64 rows differ by +-16 % in entries) the 16 neighbours of a generation differed by
// 1.3x between the shortest and the longest.  sched[] lists the octets by descending length (a counting sort over 4096
// length classes in one workgroup: ~20 us for 125 k octets), so that a generation's octets are equally long and the long
// generations start first; position t of the schedule is swept by wavefront t % 16 of generation t / 16.  Equal lengths
// (the headline matrix) keep the natural order.  Which octet a wavefront sweeps changes no bit of the result.
constexpr int PLAN_SCHED_BINS = 4096;
constexpr double PLAN_SYNC2_CV = 0.10;             // octet lengths more uneven than this: sync mode 2 by default
__global__ __launch_bounds__(1024)
void plan_sched_kernel(int noct, const int32_t *__restrict__ steps, int32_t *__restrict__ sched, long long *__restrict__ reordered)
{
    __shared__ int bins[PLAN_SCHED_BINS];
    __shared__ int smin, smax;
    const int tid = threadIdx.x;
    if (tid == 0) { smin = INT_MAX; smax = 0; }
    for (int b = tid; b < PLAN_SCHED_BINS; b += 1024) bins[b] = 0;
    __syncthreads();
    __shared__ double ssum, ssq;
    if (tid == 0) { ssum = 0.0; ssq = 0.0; }
    __syncthreads();
    int lo = INT_MAX, hi = 0;
    double sum1 = 0.0, sum2 = 0.0;
    for (int o = tid; o < noct; o += 1024) { const int v = steps[o]; lo = min(lo, v); hi = max(hi, v); sum1 += v; sum2 += (double)v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sum1 += __shfl_xor(sum1, off, 64); sum2 += __shfl_xor(sum2, off, 64); }
    atomicMin(&smin, lo);
    atomicMax(&smax, hi);
    if ((tid & 63) == 0) { atomicAdd(&ssum, sum1); atomicAdd(&ssq, sum2); }
    __syncthreads();
    const int vmin = smin, vmax = smax;
    // rides back to the host with the plan's size: 0 = octets of equal length, else 1 + 1000 x the coefficient of variation
    // of the octets' lengths (what the sweep's default sync mode is chosen from, mxd_spmm_plan_run_rows)
    if (tid == 0) {
        const double mean = ssum / noct, var = fmax(ssq / noct - mean * mean, 0.0);
        *reordered = vmax > vmin ? 1 + (long long)(1000.0 * sqrt(var) / fmax(mean, 1.0)) : 0;
        reordered[1] = vmax;                                             // the longest octet (plan_imbalance)
    }
    if (vmax <= vmin) {                                              // all octets equally long: natural order
        for (int o = tid; o < noct; o += 1024) sched[o] = o;
        return;
    }
    // class 0 = the longest octets
    const float scale = (float)(PLAN_SCHED_BINS - 1) / (float)(vmax - vmin);
    auto cls = [&](int v) { return (int)((float)(vmax - v) * scale); };
    for (int o = tid; o < noct; o += 1024) atomicAdd(&bins[cls(steps[o])], 1);
    __syncthreads();
    // exclusive scan of the 4096 class counts: 4 per thread
    __shared__ int wsum[16];
    int c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { c[k] = bins[tid * 4 + k]; sum += c[k]; }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += up; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
    int run = base + incl - sum;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) { bins[tid * 4 + k] = run; run += c[k]; }
    __syncthreads();
    for (int o = tid; o < noct; o += 1024) sched[atomicAdd(&bins[cls(steps[o])], 1)] = o;
}

// Where the sweep's wavefronts meet (locality only, any value is correct): the 16 octets that one workgroup sweeps
// together share the panel boundaries, as fractions of each octet's own length — the mean relative panel start over
// the group.  With per-octet boundaries every meeting waited for the wavefront whose panel happened to be longest
// (entries per octet and panel vary by ~4 %: the sum of the maxima is ~8 % more than the common length); with shared
// boundaries equally long octets arrive together.  Runs after the fill (which supplies pstart for layout-2 octets).
__global__ __launch_bounds__(256)
void plan_bounds_kernel(int noct, int npanels, const int32_t *__restrict__ oct_off, const int32_t *__restrict__ pstart,
                        int32_t *__restrict__ step_off, long long cap_slots,
                        const long long *__restrict__ total_steps, const int32_t *__restrict__ sched)
{
    if (!plan_fits(*total_steps, cap_slots)) return;                             // the fill wrote nothing either
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nn = (long long)noct * npanels;
    if (t > nn) return;
    if (t == nn) { step_off[nn] = oct_off[noct]; return; }
    const int pos = (int)(t / npanels), q = (int)(t % npanels);                  // position in the schedule
    const int oct = sched[pos];
    const int o0 = (pos / PLAN_GEN_OCTS) * PLAN_GEN_OCTS, o1 = min(o0 + PLAN_GEN_OCTS, noct);
    long long sum = 0, len = 0;
    for (int k = o0; k < o1; k++) {                                              // the octets swept together with this one
        const int o = sched[k];
        sum += pstart[(size_t)o * (npanels + 1) + q];
        len += pstart[(size_t)o * (npanels + 1) + npanels];
    }
    const int mine = oct_off[oct + 1] - oct_off[oct];
    const double frac = len > 0 ? (double)sum / (double)len : 0.0;
    int b = q == 0 ? 0 : (int)(frac * (double)mine);
    if (b > mine) b = mine;
    step_off[(size_t)oct * npanels + q] = oct_off[oct] + b;
}

// broadcast lane U of every 8-lane group: row_newbcast takes lane n of each 16-lane DPP row; bank_mask restricts the
// write to the low / high half of the row (banks of 4 lanes), so two moves serve the two groups of a row
template <int U>
__device__ __forceinline__ int group8_dpp_bcast(int v)
{
    int t = __builtin_amdgcn_mov_dpp(v, 0x150 + U, 0xF, 0x3, false);      // lanes of the other half: don't care
    return __builtin_amdgcn_update_dpp(t, v, 0x150 + 8 + U, 0xF, 0xC, false);
}
template <int U>
__device__ __forceinline__ void plan_bcast(int pcw, double pvw, int &pc, double &pv)
{
    union { double d; int i[2]; } a, b;
    a.d = pvw;
    pc = group8_dpp_bcast<U>(pcw);
    b.i[0] = group8_dpp_bcast<U>(a.i[0]);
    b.i[1] = group8_dpp_bcast<U>(a.i[1]);
    pv = b.d;
}

// f32 product: the value is narrowed to float once per batch, BEFORE the broadcast (matmul.cpp:53-57 narrows alpha per
// nonzero; the same float either way), so a step broadcasts two registers, not three, and converts nothing
template <int U>
__device__ __forceinline__ void plan_bcast(int pcw, float pvw, int &pc, float &pv)
{
    pc = group8_dpp_bcast<U>(pcw);
    pv = __builtin_bit_cast(float, group8_dpp_bcast<U>(__builtin_bit_cast(int, pvw)));
}
// acc += a * b over the lane's 16 bytes.  f32: two packed FMAs (v_pk_fma_f32) instead of four scalar ones; with the
// narrowed broadcast a step issues 9 VALU instructions instead of 14.  Same fused operation per element.  Measured at
// the cfg5 shard: sweep 3.66 -> 3.62 ms — VALU issue (a wave64 instruction holds its SIMD for four cycles; ~0.9 ms of
// them per launch) hides under the B-line gather, which is what bounds the sweep.
template <int VEC>
__device__ __forceinline__ void axpy_lane(double a, const double (&b)[VEC], double (&acc)[VEC])
{
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = mx_fma(a, b[v], acc[v]);
}
template <int VEC>
__device__ __forceinline__ void axpy_lane(float a, const float (&b)[VEC], float (&acc)[VEC])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(VEC % 2 == 0, "packed pairs");
    const f2 aa = {a, a};
#pragma unroll
    for (int v = 0; v < VEC; v += 2) {
        const f2 bb = {b[v], b[v + 1]};
        f2 cc = {acc[v], acc[v + 1]};
        cc = __builtin_elementwise_fma(aa, bb, cc);
        acc[v] = cc.x; acc[v + 1] = cc.y;
    }
}

// Fold a finished row's partial sums into its LDS accumulators.  A row is either owned by one lane group of the
// wavefront or marked `shared` in the plan (a long row dealt to several groups).  One wavefront's LDS operations execute
// in order, so for an owned row every form below is the same sequence of additions.  f64: two fire-and-forget
// ds_add_f64 (no return value, nothing to wait for — the read-modify-write cost an LDS round trip on ~70 % of the
// steps; atomic, so shared rows need nothing extra).  f32: read-modify-write of one 16-byte word for owned rows (four
// ds_add_f32 measured 2.4x slower overall), atomics for shared rows only.
template <int VEC>
__device__ __forceinline__ void lds_fold(double *d, double (&acc)[VEC], bool)
{
#pragma unroll
    for (int v = 0; v < VEC; v++) __hip_atomic_fetch_add(d + v, acc[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int VEC>
__device__ __forceinline__ void lds_fold(float *d, float (&acc)[VEC], bool shared)
{
    if (shared) {
#pragma unroll
        for (int v = 0; v < VEC; v++) __hip_atomic_fetch_add(d + v, acc[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; v++) d[v] += acc[v];
    }
}

// main kernel
// SHARED: the plan has octets in the dealt layout (folds of their rows must be atomic; only matters for f32)
template <typename real_t, bool COLMAJOR, int PLAN_WAVES, bool SHARED>
__global__ __launch_bounds__(PLAN_WAVES * 64)
void spmm_plan_kernel(int m, int n, int npanels, const int32_t *__restrict__ step_off,
                      const int32_t *__restrict__ pcol, const double *__restrict__ pval,
                      const real_t *__restrict__ Bp, size_t slab_stride,
                      real_t *__restrict__ C, size_t ldc, int nslabs, int ngens, int noct, int pad_col,
                      unsigned *__restrict__ sync_ctr, int sync_mode, const int32_t *__restrict__ sched)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    constexpr int U = 8;                                            // plan steps in flight per wavefront
    constexpr int PLAN_WG_ROWS = PLAN_OCT_ROWS * PLAN_WAVES;        // rows per workgroup generation
    // accumulator rows are padded by 8 (f64) / 16 (f32) bytes: the column-major epilogue reads one column of 64
    // consecutive rows per instruction, which at a 128-byte stride would hit a single LDS bank pair
    constexpr int S = W + 16 / (int)sizeof(real_t) / 2;
    __shared__ real_t accs[PLAN_WG_ROWS * S];                       // 16 waves: 1024 rows x 136 B = 136 KiB

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3, lg = lane & 7;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const long long total = (long long)nslabs * ngens;
    const long long lo = total * xcd / 8, hi = total * (xcd + 1) / 8;
    // Which items (slab, generation) this workgroup sweeps, in which order.  Natural order: XCD x takes the contiguous range
    // [lo, hi) of the slab-major item list, its workgroups round-robin.  WITH A SCHEDULE (generations by descending length)
    // and c = 8 / nslabs XCDs per slab that range would be the c-th part of the schedule — for 4 slabs the long half for one
    // XCD and the short half for its neighbour (round 5, tools/cliff_hunt.py: cfg2's shape in f32 with rows sorted by
    // length 2.32 ms where equal rows take 0.97; log-normal rows 1.29 ... 1.59x).  There the c XCDs of a slab deal its ROUNDS
    // (nwg consecutive generations: what an XCD's workgroups sweep together, equally long — which keeps them in the same
    // panel) among themselves, in snake order: round k * c + j in even octaves k, k * c + c - 1 - j in odd ones.  Every XCD
    // still sweeps its longest generations first.  (Dealing single generations instead put a quarter of the length
    // distribution into one round: the workgroups of an XCD drifted into different panels — 1.9 -> 3.0 ms.)
    const bool dealt_rounds = sched != nullptr && nslabs < 8 && (8 % nslabs) == 0;
    const int xcds_per_slab = dealt_rounds ? 8 / nslabs : 1, my_j = xcd % xcds_per_slab, my_slab = xcd / xcds_per_slab;
    const int nrounds = (ngens + nwg - 1) / nwg;
    const int niter = dealt_rounds ? (nrounds + xcds_per_slab - 1) / xcds_per_slab : (int)((hi - lo + nwg - 1) / nwg);
    auto item_of = [&](int it) -> long long {                       // -1: nothing in this iteration
        if (!dealt_rounds) {
            const long long raw = lo + wg + (long long)it * nwg;
            return raw < hi ? raw : -1;
        }
        const int round = it * xcds_per_slab + ((it & 1) ? xcds_per_slab - 1 - my_j : my_j);
        const long long gen = (long long)round * nwg + wg;
        return round < nrounds && gen < ngens ? (long long)my_slab * ngens + gen : -1;
    };
    unsigned *const my_ctr = sync_ctr + xcd * 64;
    real_t *const my_oct = accs + (size_t)wave * PLAN_OCT_ROWS * S;                     // this wavefront's 64 rows
    real_t *const my_rows = my_oct + lg * VEC;                                          // + slot * S: my 16 bytes of a row

    // Carried from one generation to the next: the stream bounds, the row map and the FIRST chunk of plan slots of the
    // wavefront's next octet are requested during the last chunk of the current one, so that a generation does not
    // start with two exposed memory latencies (bounds, then the first chunk: ~3-4 us of a ~58 us generation).
    bool primed = false;                                            // wave-uniform
    int bounds_c = 0, send_c = 0;
    // schedule look-ups run two generations ahead: a look-up consumed where it is issued would expose a load latency (and
    // drain the B-line pipeline) once per generation
    int oct_next_c = 0;                                             // next generation's octet (scalar, from the previous iteration)
    int octnn_v = 0;                                                // the one after: requested a generation ago, still a vector register
    int rc[PLAN_CHUNK];
    double rv[PLAN_CHUNK];
#pragma unroll
    for (int k = 0; k < PLAN_CHUNK; k++) { rc[k] = 0; rv[k] = 0.0; }
    for (int it = 0; it < niter; it++) {
        const long long item_raw = item_of(it);
        const bool have = item_raw >= 0;
        const long long item = have ? item_raw : lo;
        const int slab = (int)(item / ngens), gen = (int)(item % ngens);
        // position in the schedule -> octet (identity without a schedule); looked up one generation ahead
        const int pos = gen * PLAN_WAVES + wave;
        const bool oct_ok = have && pos < noct;
        // this wavefront's octet of the next generation (if any), and of the one after
        const long long item_n = it + 1 < niter ? item_of(it + 1) : -1, item_nn = it + 2 < niter ? item_of(it + 2) : -1;
        const int pos_n = (int)((item_n < 0 ? 0 : item_n) % ngens) * PLAN_WAVES + wave;
        const int pos_nn = (int)((item_nn < 0 ? 0 : item_nn) % ngens) * PLAN_WAVES + wave;
        const bool octn_ok = item_n >= 0 && pos_n < noct;
        const bool octnn_ok = item_nn >= 0 && pos_nn < noct;
        int oct = pos, oct_n = pos_n;
        if (sched) {
            if (it == 0) {                                           // the only look-ups that are waited for
                const int a = oct_ok ? sched[pos] : 0, b = octn_ok ? sched[pos_n] : 0;
                oct = __builtin_amdgcn_readfirstlane(a);
                oct_n = __builtin_amdgcn_readfirstlane(b);
            } else {
                oct = oct_next_c;
                oct_n = __builtin_amdgcn_readfirstlane(octnn_v);
            }
            oct_next_c = oct_n;
        }
        // slab base is wave-uniform (scalar registers), the per-lane part is a 32-bit byte offset: one VALU op per
        // address.  A slab is K x 128 B < 4 GiB because K < 2^27... checked on the host (K * 128 < 2^32).
        const char *__restrict__ Bbase = reinterpret_cast<const char *>(Bp + (size_t)slab * slab_stride);
        const unsigned lane_off = (unsigned)(lg * VEC * sizeof(real_t));

        // A wavefront's accumulator rows are touched by that wavefront only (zeroing, folds, epilogue): no
        // workgroup-wide synchronisation around a generation, the wavefronts only meet at the panel boundaries.
        for (int i = lane; i < PLAN_OCT_ROWS * S; i += 64) my_oct[i] = 0;
        if (sync_mode > 0) __syncthreads();                          // locality only: start the first panel together

        // One continuous, software-pipelined stream over the octet's entries of ALL panels (they are contiguous in
        // the plan).  Panel boundaries only matter for locality: when the stream crosses one, the 16 waves of the
        // CU's single workgroup meet at a __syncthreads (no global traffic, no pipeline restart: the prefetched
        // plan entries stay in flight).  Across the 32 CUs of the XCD group there is ONE global timing barrier per
        // generation (32 pollers per counter); in between the CUs run identical code on statistically identical
        // data and drift by a fraction of a panel.
        {
            if (sync_mode >= 2) xcd_timing_barrier(my_ctr, (unsigned)(it + 1) * (unsigned)nwg);
            // The octet's stream bounds and its slot -> row map (identity unless the plan balanced the bundles; used by
            // the epilogue only) are requested AFTER the barrier: a load in flight at a barrier makes all 16 wavefronts
            // wait for the slowest one (measured: +0.03 ms per launch for each of the two).
            // All panel boundaries of the octet come in with ONE load (lane p holds the start of the p-th panel) and
            // are picked out with v_readlane when the stream crosses a panel: a load at every boundary had to be
            // waited for with vmcnt(0), i.e. it drained the whole B-line pipeline once per panel.
            if (sched && octnn_ok) octnn_v = sched[pos_nn];          // consumed at the start of the next generation
            int bounds = 0, send = 0;
            if (primed) {                                           // requested during the previous generation
                bounds = bounds_c; send = send_c;
            } else if (oct_ok) {
                bounds = step_off[(size_t)oct * npanels + (lane < npanels ? lane : 0)];
                send = step_off[(size_t)oct * npanels + npanels];
            }
            int sbeg = __builtin_amdgcn_readfirstlane(bounds);      // wave-uniform: keep the loop control scalar
            send = __builtin_amdgcn_readfirstlane(send);
            int next_b = npanels > 1 ? __builtin_amdgcn_readlane(bounds, 1) : send;
            int p = 0;
            int cur = g * PLAN_RB;                                  // a slot to fold the initial zeros into (harmless)
            real_t acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = 0;
            // A batch = U = 8 steps = 64 consecutive plan slots, laid out [bundle g][step u]: lane l reads slot
            // (8 s + l) — one fully coalesced 256 B + 512 B read per batch — and step u's entry is broadcast from
            // lane u of each group.  (Reading the slot from all 8 lanes of a group instead costs the texture
            // addresser 8x the lane-bytes: PMC showed TA_BUSY 71 % and the kernel TA-bound.)
            // Every slot is a valid entry: padding is (zero row of B, value 0, current row) — no per-step validity
            // test, no clamp.
            static_assert(U == 8, "one batch = one wavefront of plan slots");
            static_assert(W * sizeof(real_t) == 128, "slab line");
            auto b_offset = [&](int c) -> unsigned {                // the row bits (27..29) fall off the 32-bit shift
                return ((unsigned)c * (unsigned)(W * sizeof(real_t))) + lane_off;
            };
            // The plan slots are fetched a CHUNK (PLAN_CHUNK = 4 batches = 32 steps) at a time, one chunk ahead.
            // Vector loads return in order, so a slot read that misses to HBM (the plan is a pure stream) holds back
            // every younger B-line load behind it; fetching one batch per iteration put that full latency into every
            // iteration (measured: 1.95 us per 8 steps per wave, whatever the locality of B).  Now it is paid once
            // per 32 steps.  Reads run one chunk past the octet (next octet's slots / the padding behind the last
            // octet): they only ever become addresses of valid B lines, never FMAs.
            // (Tried: an XCD-wide timing barrier at EVERY panel boundary instead of the workgroup's own meeting — 2.18 ms;
            //  one per generation — sync_mode 2 — 1.78; the workgroup-only meetings of sync_mode 1: 1.71.)
            // (Tried: nontemporal loads for this stream, so that it does not push B lines out of L2 — 1.90 ms instead
            //  of 1.72 at cfg2, 3.84 instead of 3.60 at the cfg5 shard.)
            int rn[PLAN_CHUNK];
            double rvn[PLAN_CHUNK];
            auto load_chunk = [&](int step, int (&c)[PLAN_CHUNK], double (&v)[PLAN_CHUNK]) {
                const long long e = (long long)step * 8 + lane;
#pragma unroll
                for (int k = 0; k < PLAN_CHUNK; k++) { c[k] = pcol[e + 64 * k]; v[k] = pval[e + 64 * k]; }
            };
            int pc[U];
            real_t pv[U];                                           // already narrowed for the f32 product
            real_t b[U][VEC];
#pragma unroll
            for (int u = 0; u < U; u++) {
                pc[u] = 0;
                pv[u] = 0;
#pragma unroll
                for (int v = 0; v < VEC; v++) b[u][v] = 0;
            }
            // consume step u of the batch in (pc, pv, b): row switch -> fold the finished row into LDS, then FMA
            auto consume = [&](int u) {
                const int tag = pc[u] >> PLAN_ROW_SHIFT;              // slot of the row (0..63), - 64 if this fold may meet another group's (plan_flag_kernel)
                if (tag != cur) {
                    lds_fold<VEC>(my_rows + (cur & 63) * S, acc, SHARED && cur < 0);
#pragma unroll
                    for (int v = 0; v < VEC; v++) acc[v] = 0;
                    cur = tag;
                }
                axpy_lane<VEC>(pv[u], b[u], acc);
                // keep the reload BEHIND the FMAs that read the old line (and the FMAs where they are): letting the two
                // cross renames b[u] and ends in a register copy at the back edge that waits for every load in flight
#pragma unroll
                for (int v = 0; v < VEC; v++) asm volatile("" : "+v"(acc[v]));
                __builtin_amdgcn_sched_barrier(0);
            };
            if (send > sbeg) {
                if (!primed) load_chunk(sbeg, rc, rv);
                // the first chunk has to be there before anything can start; with it complete at loop entry the
                // compiler's vmcnt bookkeeping is exact on both edges of the loop
#pragma unroll
                for (int k = 0; k < PLAN_CHUNK; k++) asm volatile("" : "+v"(rc[k]), "+v"(rv[k]));
            }
            // Consumption lags one batch behind the broadcast + B-line load: while batch t is consumed step by step,
            // the line of the same step of batch t+1 is requested into the registers the FMA just released, so 8
            // B-line loads per wavefront are in flight all the time.  The first pass consumes the no-op batch set up
            // above, the last batch is consumed after the loop.
            bool meta = false;                                      // next octet's bounds / row map requested
            primed = false;
            for (int s = sbeg; s < send; s += U * PLAN_CHUNK) {      // sbeg, send are wave-uniform
                const bool last = s + U * PLAN_CHUNK >= send;
                if (octn_ok && !meta && s + 2 * U * PLAN_CHUNK >= send) {        // one chunk before the last, if there is one
                    bounds_c = step_off[(size_t)oct_n * npanels + (lane < npanels ? lane : 0)];
                    send_c = step_off[(size_t)oct_n * npanels + npanels];
                    meta = true;
                }
                // the read-ahead of the last chunk fetches the first chunk of the next octet instead of running past
                // this one
                int ahead = s + U * PLAN_CHUNK;
                if (last && octn_ok) {
                    ahead = __builtin_amdgcn_readfirstlane(bounds_c);
                    primed = true;
                }
                load_chunk(ahead, rn, rvn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PLAN_CHUNK; k++) {
                    if (k > 0 && s + U * k >= send) break;             // octets end on a batch, not on a chunk (uniform)
                    const real_t rvk = (real_t)rv[k];
#define MX_PLAN_STEP(UU)                                                                                              \
                    consume(UU);                                                                                      \
                    plan_bcast<UU>(rc[k], rvk, pc[UU], pv[UU]);                                                       \
                    vload<real_t, VEC>(b[UU], reinterpret_cast<const real_t *>(Bbase + b_offset(pc[UU])));            \
                    __builtin_amdgcn_sched_barrier(0);
                    MX_PLAN_STEP(0) MX_PLAN_STEP(1) MX_PLAN_STEP(2) MX_PLAN_STEP(3)
                    MX_PLAN_STEP(4) MX_PLAN_STEP(5) MX_PLAN_STEP(6) MX_PLAN_STEP(7)
#undef MX_PLAN_STEP
                    if (sync_mode > 0) {
                        const int sn = s + U * k;                   // steps consumed so far
                        while (p < npanels - 1 && sn >= next_b) {   // the stream moved into the next panel
                            p++;
                            __syncthreads();
                            next_b = p < npanels - 1 ? __builtin_amdgcn_readlane(bounds, p + 1) : send;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < PLAN_CHUNK; k++) { rc[k] = rn[k]; rv[k] = rvn[k]; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) consume(u);
            lds_fold<VEC>(my_rows + (cur & 63) * S, acc, SHARED && cur < 0);
            if (sync_mode > 0)
                for (; p < npanels - 1; p++) __syncthreads();           // every wave meets npanels-1 times per generation
        }

        // each wavefront writes the 64 x W tile of C it accumulated (streaming stores: C is not read again); slot ->
        // row through the octet's row map (identity unless the plan balanced the bundles)
        if (oct_ok) {
            const int row_base = oct * PLAN_OCT_ROWS;
            const int ncols = min(W, n - slab * W);
            if constexpr (!COLMAJOR) {
#pragma unroll
                for (int rr = 0; rr < PLAN_OCT_ROWS / 8; rr++) {
                    const int r = rr * 8 + g;                        // slot
                    const int row = row_base + r;
                    if (row < m && lg * VEC < ncols) {
                        real_t t[VEC];
#pragma unroll
                        for (int v = 0; v < VEC; v++) t[v] = my_oct[(size_t)r * S + lg * VEC + v];
                        vstore_nt<real_t, VEC>(C + (size_t)row * ldc + slab * W + lg * VEC, t);
                    }
                }
            } else {
                // lane = slot: the 64 rows of the octet are one 512-byte (f64) segment of an output column
                const int row = row_base + lane;
                if (row < m) {
                    for (int c = 0; c < ncols; c++)
                        __builtin_nontemporal_store(my_oct[(size_t)lane * S + c], &C[(size_t)(slab * W + c) * ldc + row]);
                }
            }
        }
    }
}

}  // namespace mx

struct mx_spmm_plan {
    int m = 0, K = 0, npanels = 0, panel_cols = 0, noct = 0;
    long long total_steps = 0;
    long long nnz = 0;
    int32_t *step_off = nullptr; size_t step_off_cap = 0;
    int32_t *pcol = nullptr;     size_t pcol_cap = 0;
    double *pval = nullptr;      size_t pval_cap = 0;
    unsigned char *layout = nullptr; size_t layout_cap = 0;    // [noct]: 0 bundle layout, 1 dealt layout
    int32_t *pstart = nullptr;     size_t pstart_cap = 0;     // [noct][P + 1]: relative panel starts, unpadded length
    int32_t *sched = nullptr;      size_t sched_cap = 0;      // [noct]: the order the sweep takes the octets in
    bool reordered = false;                                    // sched is not the identity (octets of unequal length)
    double oct_cv = 0.0;                                       // coefficient of variation of the octets' lengths (steps)
    long long max_oct_steps = 0;                               // the longest octet
    long long *rbdev = nullptr;                                // [total steps][nnz][dealt octets], read back in one copy
    long long ndealt = 0;                                      // octets in the dealt layout (rows shared by lane groups)
    void *scratch = nullptr;     size_t scratch_cap = 0;       // rowpre + steps + scan workspace (build only)
    double build_ms = 0.0;
    bool ready = false;                                            // false: sized but not filled (rejected by AUTO)
};

namespace mx {


// AUTO's plan and the read-back landing zone are per thread AND per device (a thread may move between GPUs with
// mx_set_device / torch.cuda.set_device): indexed by hipGetDevice(), like the packed-B workspace.  One stream per
// (thread, device) at a time: two AUTO calls of one thread on different streams would share these buffers unordered.
constexpr int PLAN_MAX_DEVICES = 16;
static thread_local mx_spmm_plan *g_auto_plan[PLAN_MAX_DEVICES] = {};
static inline int cur_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= PLAN_MAX_DEVICES) d = 0;
    return d;
}

static int grow(void **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes && *p) return 0;
    if (*p) pool_free(*p);
    *p = nullptr; *cap = 0;
    MX_HIP(pool_malloc(p, bytes ? bytes : 16));
    *cap = bytes;
    return 0;
}

// pinned landing zone + event for the one host read-back of a plan build
struct PlanReadback {
    long long *host = nullptr;                                      // [0] total steps, [1] nnz, [2] dealt octets
    hipEvent_t ev = nullptr;
};
static PlanReadback *plan_readback()
{
    static thread_local PlanReadback rbs[PLAN_MAX_DEVICES];
    PlanReadback &rb = rbs[cur_device()];
    if (!rb.host) {
        if (hipHostMalloc((void **)&rb.host, 8 * sizeof(long long), hipHostMallocDefault) != hipSuccess) { rb.host = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(rb.host); rb.host = nullptr; return nullptr; }
    }
    return &rb;
}

// max_pad_ratio > 0: stop after the sizing pass when the plan would hold more than ratio x nnz slots (rows of very
// uneven length pad the 8-way interleave: an octet is as long as its longest bundle) — pl->ready stays false.
static int plan_build(mx_spmm_plan *pl, int m, int K, const int32_t *indptr, const int32_t *indices,
                      const double *values, int npanels, hipStream_t st, double max_pad_ratio = 0.0)
{
    pl->ready = false;
    MX_REQUIRE(K < (1 << 25), "spmm plan: more than 2^25 columns (32-bit slab offsets)");
    // measured (cfg2, after the shared panel boundaries): 1.6 MB panels (P = 8) are best for the kernel (1.77 vs 1.86 ms at
    // P = 5) and, by a hair, for kernel + plan build
    if (npanels <= 0) npanels = pick_panels(K, (size_t)1664 << 10);
    if (npanels > PLAN_MAXP) npanels = PLAN_MAXP;
    pl->m = m; pl->K = K; pl->npanels = npanels;
    pl->panel_cols = (int)ceil_div(K > 0 ? K : 1, npanels);
    pl->noct = (int)ceil_div(m, PLAN_OCT_ROWS);
    pl->total_steps = 0; pl->nnz = 0;
    if (m == 0) { pl->ready = true; return 0; }                     // nothing to plan (and no zero-sized launches)
    const size_t nop = (size_t)pl->noct * npanels;
    const size_t al = 255;
    const size_t steps_b = (((size_t)pl->noct * 4) + al) & ~al;
    const size_t octoff_b = ((((size_t)pl->noct + 1) * 4) + al) & ~al;
    const size_t bpo_b = ((nop * 9 * 4) + al) & ~al;
    if (grow(&pl->scratch, &pl->scratch_cap, steps_b + octoff_b + bpo_b + scan_workspace_bytes((int64_t)pl->noct))) return 1;
    if (!pl->rbdev) {                                               // [2] counts dealt octets: zero now, re-zeroed after every read-back
        MX_HIP(hipMalloc((void **)&pl->rbdev, 256));
        MX_HIP(hipMemsetAsync(pl->rbdev, 0, 256, st));
    }
    if (grow((void **)&pl->step_off, &pl->step_off_cap, (nop + 1) * 4)) return 1;
    if (grow((void **)&pl->layout, &pl->layout_cap, (size_t)pl->noct)) return 1;
    if (grow((void **)&pl->pstart, &pl->pstart_cap, (size_t)pl->noct * (npanels + 1) * 4)) return 1;
    if (grow((void **)&pl->sched, &pl->sched_cap, (size_t)pl->noct * 4)) return 1;
    int32_t *steps = (int32_t *)pl->scratch;
    int32_t *oct_off = (int32_t *)((char *)steps + steps_b);
    int32_t *bpo = (int32_t *)((char *)oct_off + octoff_b);
    long long *rb_dev = pl->rbdev;
    void *scan_ws = (char *)bpo + bpo_b;
    const unsigned blocks = (unsigned)pl->noct;
    hipLaunchKernelGGL(plan_count_kernel, dim3(blocks), dim3(512), 0, st, m, npanels, pl->panel_cols, indptr, indices,
                       steps, bpo, pl->noct, rb_dev + 1, pl->layout, pl->pstart, rb_dev + 2);
    MX_LAUNCH_CHECK();
    if (exclusive_scan_i32(steps, (int64_t)pl->noct, oct_off, (int64_t *)rb_dev, scan_ws, st)) return 1;
    hipLaunchKernelGGL(plan_sched_kernel, dim3(1), dim3(1024), 0, st, pl->noct, steps, pl->sched, rb_dev + 3);
    MX_LAUNCH_CHECK();
    PlanReadback *rb = plan_readback();
    MX_REQUIRE(rb, "spmm plan: cannot allocate the pinned read-back buffer");
    MX_HIP(hipMemcpyAsync(rb->host, rb_dev, 5 * sizeof(long long), hipMemcpyDeviceToHost, st));
    MX_HIP(hipEventRecord(rb->ev, st));
    // The host needs the plan's size to (re)allocate its arrays.  While it waits for the read-back the GPU would idle
    // (~25 us): when arrays from an earlier build exist, the fill is launched right away against their capacity — it
    // checks the size itself (on the device) and writes nothing if the plan does not fit; the host then re-allocates
    // and launches it again.  (Packing B in that gap instead is worse: the fill that follows pushes the packed B out of
    // the Infinity Cache and the sweep runs 2.75 ms instead of 2.05 ms.  B is packed right before the sweep.)
    const size_t cap_slots = pl->pcol && pl->pval ? std::min(pl->pcol_cap / 4, pl->pval_cap / 8) : 0;
    if (cap_slots) {
        hipLaunchKernelGGL(plan_fill_kernel, dim3(blocks), dim3(512), 0, st, m, npanels, pl->panel_cols, indptr, indices,
                           values, oct_off, bpo, pl->pcol, pl->pval, pl->noct, K, pl->step_off, pl->layout, pl->pstart,
                           rb_dev + 2, (long long)cap_slots, rb_dev);
        hipLaunchKernelGGL(plan_flag_kernel, dim3((unsigned)ceil_div(pl->noct, 8)), dim3(512), 0, st, pl->noct, npanels, oct_off, bpo,
                           pl->pstart, pl->layout, pl->pcol, (long long)cap_slots, rb_dev);
        hipLaunchKernelGGL(plan_bounds_kernel, dim3((unsigned)ceil_div((long long)nop + 1, 256)), dim3(256), 0, st, pl->noct,
                           npanels, oct_off, pl->pstart, pl->step_off, (long long)cap_slots, rb_dev, pl->sched);
        MX_LAUNCH_CHECK();
    }
    MX_HIP(hipEventSynchronize(rb->ev));
    const long long total = rb->host[0];
    pl->nnz = (int32_t)rb->host[1];
    pl->ndealt = rb->host[2];
    pl->reordered = rb->host[3] != 0;
    pl->oct_cv = rb->host[3] > 0 ? (double)(rb->host[3] - 1) * 1e-3 : 0.0;
    pl->max_oct_steps = rb->host[4];
    MX_REQUIRE(total >= 0 && total * 8 <= (long long)INT_MAX * 4LL, "spmm plan: too many steps (%lld)", total);
    MX_REQUIRE(total <= (long long)INT_MAX, "spmm plan: step offsets exceed int32");
    pl->total_steps = total;
    if (max_pad_ratio > 0.0 && (double)total * 8.0 > (double)pl->nnz * max_pad_ratio + 65536.0) {
        MX_HIP(hipMemsetAsync(rb_dev + 2, 0, sizeof(long long), st));                    // (the fill does it when it runs)
        return 0;
    }
    const size_t slots = (size_t)total * 8 + PLAN_TAIL_SLOTS;
    if (slots > cap_slots) {                                        // first build, or the early fill found the arrays too small
        if (grow((void **)&pl->pcol, &pl->pcol_cap, slots * 4)) return 1;
        if (grow((void **)&pl->pval, &pl->pval_cap, slots * 8)) return 1;
        hipLaunchKernelGGL(plan_fill_kernel, dim3(blocks), dim3(512), 0, st, m, npanels, pl->panel_cols, indptr, indices,
                           values, oct_off, bpo, pl->pcol, pl->pval, pl->noct, K, pl->step_off, pl->layout, pl->pstart,
                           rb_dev + 2, (long long)slots, rb_dev);
        hipLaunchKernelGGL(plan_flag_kernel, dim3((unsigned)ceil_div(pl->noct, 8)), dim3(512), 0, st, pl->noct, npanels, oct_off, bpo,
                           pl->pstart, pl->layout, pl->pcol, (long long)slots, rb_dev);
        hipLaunchKernelGGL(plan_bounds_kernel, dim3((unsigned)ceil_div((long long)nop + 1, 256)), dim3(256), 0, st, pl->noct,
                           npanels, oct_off, pl->pstart, pl->step_off, (long long)slots, rb_dev, pl->sched);
        MX_LAUNCH_CHECK();
    }
    pl->ready = true;
    return 0;
}

// slab-major copy of B with one extra all-zero row (index K) per slab: the plan's padding slots point at it
template <typename real_t>
static int plan_repack(int K, int n, const real_t *B, size_t ldb, hipStream_t st, real_t **Bp_out)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    const int nslabs = (int)ceil_div(n, W);
    const int Kp = K + 1;
    real_t *Bp = (real_t *)slab_pack_workspace((size_t)nslabs * (size_t)Kp * W * sizeof(real_t));
    MX_REQUIRE(Bp, "spmm plan: cannot allocate the packed copy of B");
    if (launch_repack<real_t>(K, Kp, n, B, ldb, Bp, st)) return 1;
    *Bp_out = Bp;
    return 0;
}

// rows [row0, row0 + m) of the planned matrix (row0 a multiple of 64: whole octets); C points at the block's first row
template <typename real_t>
static int plan_run(const mx_spmm_plan *pl, int row0, int m, int n, const real_t *B, size_t ldb, real_t *C, size_t ldc,
                    int colmajor, int wg_per_cu, int sync_mode, hipStream_t st)
{
    constexpr int VEC = 16 / (int)sizeof(real_t);
    constexpr int W = SLAB_GROUP * VEC;
    const int K = pl->K;
    const int oct0 = row0 / PLAN_OCT_ROWS, noct = (int)ceil_div(m, PLAN_OCT_ROWS);
    const int32_t *step_off = pl->step_off + (size_t)oct0 * pl->npanels;
    // the schedule (octets by descending length) covers the whole matrix: row ranges, and matrices whose octets are all
    // equally long, are swept in natural order
    const int32_t *sched = pl->reordered && row0 == 0 && m == pl->m ? pl->sched : nullptr;
    const int nslabs = (int)ceil_div(n, W);
    const int Kp = K + 1;
    real_t *Bp = nullptr;
    scratch_acquire(MX_SCRATCH_PACKED_B, st);                        // the packed copy of B is per thread: wait for a sweep on another stream
    if (plan_repack<real_t>(K, n, B, ldb, st, &Bp)) return 1;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (wg_per_cu != 1 && wg_per_cu != 2 && wg_per_cu != 4) wg_per_cu = PLAN_DEFAULT_WG_PER_CU;
    const int waves = 16 / wg_per_cu;
    const int ngens = (int)ceil_div(m, PLAN_OCT_ROWS * waves);
    long long grid = (long long)cus * wg_per_cu;
    const long long total = (long long)nslabs * ngens;
    if (grid > total + 7) grid = total + 7;
    grid = (grid / 8) * 8;
    if (grid < 8) grid = 8;
    unsigned *sync = slab_sync_workspace();
    if (!sync || pl->npanels <= 1) sync_mode = 0;
    if (sync_mode >= 2) MX_HIP(hipMemsetAsync(sync, 0, 8 * 64 * sizeof(unsigned), st));   // counters of the XCD timing barrier
    kt_begin(st);
    // f64 folds are atomic anyway: one variant; f32 keeps its cheaper read-modify-write when no row is shared
    const bool shared = sizeof(real_t) == 4 && pl->ndealt > 0;
#define MX_PLAN_LAUNCH2(CM, WV, SH)                                                                                          \
    hipLaunchKernelGGL((spmm_plan_kernel<real_t, CM, WV, SH>), dim3((unsigned)grid), dim3(WV * 64), 0, st, m, n, pl->npanels, \
                       step_off, pl->pcol, pl->pval, Bp, (size_t)Kp * W, C, ldc, nslabs, ngens, noct, K,                     \
                       sync, sync_mode, sched)
#define MX_PLAN_LAUNCH(CM, WV)                                                                                               \
    do { if constexpr (sizeof(real_t) == 4) { if (shared) MX_PLAN_LAUNCH2(CM, WV, true); else MX_PLAN_LAUNCH2(CM, WV, false); } \
         else MX_PLAN_LAUNCH2(CM, WV, false); } while (0)
    if (colmajor) {
        if (waves == 16) MX_PLAN_LAUNCH(true, 16); else if (waves == 8) MX_PLAN_LAUNCH(true, 8); else MX_PLAN_LAUNCH(true, 4);
    } else {
        if (waves == 16) MX_PLAN_LAUNCH(false, 16); else if (waves == 8) MX_PLAN_LAUNCH(false, 8); else MX_PLAN_LAUNCH(false, 4);
    }
#undef MX_PLAN_LAUNCH
#undef MX_PLAN_LAUNCH2
    kt_end(st);
    scratch_done(MX_SCRATCH_PACKED_B, st);
    MX_LAUNCH_CHECK();
    return 0;
}

}  // namespace mx

extern "C" int mxd_spmm_plan_create(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                                    int npanels, void *stream, mx_spmm_plan **plan_out)
{
    MX_REQUIRE(plan_out && m >= 0 && K >= 0, "mxd_spmm_plan_create: bad arguments");
    mx_spmm_plan *pl = *plan_out ? *plan_out : new (std::nothrow) mx_spmm_plan();      // pass an old plan to reuse its buffers
    MX_REQUIRE(pl, "out of host memory");
    if (mx::plan_build(pl, m, K, indptr, indices, values, npanels, mx::as_stream(stream))) {
        if (!*plan_out) { mxd_spmm_plan_destroy(pl); }
        return 1;
    }
    *plan_out = pl;
    return 0;
}

// the padding limit beyond which AUTO prefers the row-wave kernel (tools/skew_probe.py)
#define MX_PLAN_AUTO_PAD_RATIO 1.55

extern "C" int mxd_spmm_plan_create_auto(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                                         int npanels, void *stream, mx_spmm_plan **plan_out, int *ready)
{
    MX_REQUIRE(plan_out && ready && m >= 0 && K >= 0, "mxd_spmm_plan_create_auto: bad arguments");
    mx_spmm_plan *pl = *plan_out ? *plan_out : new (std::nothrow) mx_spmm_plan();
    MX_REQUIRE(pl, "out of host memory");
    if (mx::plan_build(pl, m, K, indptr, indices, values, npanels, mx::as_stream(stream), MX_PLAN_AUTO_PAD_RATIO)) {
        if (!*plan_out) { mxd_spmm_plan_destroy(pl); }
        return 1;
    }
    *plan_out = pl;
    *ready = pl->ready ? 1 : 0;
    return 0;
}

extern "C" int mxd_spmm_plan_destroy(mx_spmm_plan *pl)
{
    if (!pl) return 0;
    if (pl->step_off) mx::pool_free(pl->step_off);
    if (pl->pcol) mx::pool_free(pl->pcol);
    if (pl->pval) mx::pool_free(pl->pval);
    if (pl->scratch) mx::pool_free(pl->scratch);
    if (pl->layout) mx::pool_free(pl->layout);
    if (pl->pstart) mx::pool_free(pl->pstart);
    if (pl->sched) mx::pool_free(pl->sched);
    if (pl->rbdev) (void)hipFree(pl->rbdev);
    delete pl;
    return 0;
}

extern "C" int mxd_spmm_plan_info(const mx_spmm_plan *pl, int *npanels, int64_t *padded_entries)
{
    MX_REQUIRE(pl, "mxd_spmm_plan_info: null plan");
    if (npanels) *npanels = pl->npanels;
    if (padded_entries) *padded_entries = pl->total_steps * 8;
    return 0;
}

// How much longer than its share of the machine the sweep's longest work item runs.  An item is one octet x one 128-byte
// slab, swept by ONE wavefront; 4096 of them run at a time (256 CUs x 16).  With rows sorted by length, or a few rows of
// tens of thousands of entries against a narrow B, the longest octet alone outlasts everything else (tools/cliff_hunt.py:
// m = 2e5, 100 per row, n = 32, rows sorted by length: 1.87 ms against 0.33 for equal rows — imbalance 11.8; the row-split
// kernel with its long-rows path takes 0.54).  A wide B hides it (more items per octet: cfg2's shape, sorted rows, 8 slabs:
// 1.6).  AUTO leaves the plan for the row-split kernel above MX_PLAN_MAX_IMBALANCE.
namespace mx {
double plan_imbalance(const mx_spmm_plan *pl, int n, int dense_bytes)
{
    if (!pl || pl->total_steps <= 0 || pl->noct <= 0) return 0.0;
    const double nslabs = (double)((n + 128 / dense_bytes - 1) / (128 / dense_bytes));
    const double items = (double)pl->noct * nslabs, at_a_time = items < 4096.0 ? items : 4096.0;
    return (double)pl->max_oct_steps * at_a_time / ((double)pl->total_steps * nslabs);
}
double plan_auto_imbalance(int n, int dense_bytes) { return plan_imbalance(g_auto_plan[cur_device()], n, dense_bytes); }
}  // namespace mx

extern "C" int mxd_spmm_plan_imbalance(const mx_spmm_plan *pl, int n, int dense_dtype, double *imbalance)
{
    MX_REQUIRE(pl && imbalance && n > 0, "mxd_spmm_plan_imbalance: bad arguments");
    *imbalance = mx::plan_imbalance(pl, n, dense_dtype == MX_F64 ? 8 : 4);
    return 0;
}

extern "C" int mxd_spmm_plan_octet_cv(const mx_spmm_plan *pl, double *cv)
{
    MX_REQUIRE(pl && cv, "mxd_spmm_plan_octet_cv: null argument");
    *cv = pl->oct_cv;
    return 0;
}

extern "C" int mxd_spmm_plan_run(const mx_spmm_plan *pl, int n, const void *B, size_t ldb, void *C, size_t ldc,
                                 int dense_dtype, int colmajor_out, int wg_per_cu, int sync_mode, void *stream)
{
    MX_REQUIRE(pl, "mxd_spmm_plan_run: null plan");
    return mxd_spmm_plan_run_rows(pl, 0, pl->m, n, B, ldb, C, ldc, dense_dtype, colmajor_out, wg_per_cu, sync_mode, stream);
}

extern "C" int mxd_spmm_plan_run_rows(const mx_spmm_plan *pl, int row0, int nrows, int n, const void *B, size_t ldb, void *C,
                                      size_t ldc, int dense_dtype, int colmajor_out, int wg_per_cu, int sync_mode, void *stream)
{
    MX_REQUIRE(pl && n >= 0, "mxd_spmm_plan_run: bad arguments");
    MX_REQUIRE(pl->ready, "mxd_spmm_plan_run: the plan was sized but not built");
    MX_REQUIRE(row0 >= 0 && nrows >= 0 && row0 % mx::PLAN_OCT_ROWS == 0 && (long long)row0 + nrows <= pl->m,
               "mxd_spmm_plan_run_rows: rows [%d, %d + %d) of a %d-row plan (the first row must be a multiple of 64)", row0, row0,
               nrows, pl->m);
    if (nrows == 0 || n == 0) return 0;
    MX_REQUIRE(B && C, "mxd_spmm_plan_run: null pointer");
    hipStream_t st = mx::as_stream(stream);
    // 1: panel meetings inside the CU's workgroup; 2 adds one XCD-wide timing barrier per generation — which costs 3 % on
    // octets of equal length (cfg2: 1.74 -> 1.80 ms) and pays once the generations differ in length and the workgroups of an
    // XCD drift into different panels (log-normal rows, sigma 1 = octets +-16 %: 2.24 -> 1.94 ms; tools/skew_probe.py)
    if (sync_mode < 0) sync_mode = pl->oct_cv > mx::PLAN_SYNC2_CV ? 2 : 1;
    mx::set_last_spmm_kernel("spmm_plan_kernel");
    if (dense_dtype == MX_F64) {
        MX_REQUIRE(mx::slab_ok<double>(n, (const double *)B, ldb, (const double *)C, ldc, colmajor_out),
                   "mxd_spmm_plan_run: operands do not meet the 16-byte alignment rules");
        return mx::plan_run<double>(pl, row0, nrows, n, (const double *)B, ldb, (double *)C, ldc, colmajor_out, wg_per_cu, sync_mode, st);
    }
    if (dense_dtype == MX_F32) {
        MX_REQUIRE(mx::slab_ok<float>(n, (const float *)B, ldb, (const float *)C, ldc, colmajor_out),
                   "mxd_spmm_plan_run: operands do not meet the 16-byte alignment rules");
        return mx::plan_run<float>(pl, row0, nrows, n, (const float *)B, ldb, (float *)C, ldc, colmajor_out, wg_per_cu, sync_mode, st);
    }
    return mx::set_error("mxd_spmm_plan_run: unsupported dense dtype %d", dense_dtype);
}

// ---- the plan AUTO keeps (spmm.hip)
namespace mx {

int plan_auto_build(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values, int npanels,
                    hipStream_t st, double max_pad_ratio, bool *ready)
{
    mx_spmm_plan *&pl = g_auto_plan[cur_device()];
    if (!pl) pl = new (std::nothrow) mx_spmm_plan();
    MX_REQUIRE(pl, "out of host memory");
    // the plan's buffers are reused call after call: a caller that comes back on another stream waits for the sweep that
    // still reads them (scan.hip scratch_acquire / scratch_done)
    scratch_acquire(MX_SCRATCH_AUTO_PLAN, st);
    const int rc = plan_build(pl, m, K, indptr, indices, values, npanels, st, max_pad_ratio);
    // the build kernels wrote the plan's buffers on `st` whatever comes next — a sweep (plan_auto_run marks its end), a
    // rejected plan or an error: the next user, possibly on another stream, waits behind THIS mark (round 4's advisor finding)
    scratch_done(MX_SCRATCH_AUTO_PLAN, st);
    if (rc) return 1;
    *ready = pl->ready;
    return 0;
}

int plan_auto_run(int n, const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype, int colmajor, void *stream)
{
    const int rc = mxd_spmm_plan_run(g_auto_plan[cur_device()], n, B, ldb, C, ldc, dense_dtype, colmajor, 0, -1, stream);
    scratch_done(MX_SCRATCH_AUTO_PLAN, as_stream(stream));
    return rc;
}

void plan_auto_release()
{
    for (int d = 0; d < PLAN_MAX_DEVICES; d++)
        if (g_auto_plan[d]) { mxd_spmm_plan_destroy(g_auto_plan[d]); g_auto_plan[d] = nullptr; }
}

}  // namespace mx
