// profile.hip — what AUTO has to know about a CSR matrix beyond its sizes: real dgRMatrix data has
// power-law COLUMNS and skewed ROW lengths (the vignette's own application is LibSVM real-sim, Rmd:442-502), and both move
// the kernels apart:
//   * the gather kernels (row-wave, row-split) read a row of B per entry through an XCD's 4 MiB L2: what counts is the share
//     of the ENTRIES whose column is among the rows of B that L2 holds — the probability mass of the hottest columns — not
//     the share of B's bytes (uniform columns: the same number; Zipf columns, K = 1e5, n = 128: 0.68 against 0.04);
//   * the kernels that walk several rows in lockstep (the row-group form, the LDS-tile kernel) run as long as the longest
//     of their rows: the coefficient of variation of the row lengths prices that.
// mxd_csr_profile: one launch over a SAMPLE of the column ids (<= 2^18 entries, evenly spaced 256-entry runs: hot columns
// are what matters and they show in any sample) into per-column counters, with the row statistics from indptr beside it;
// one launch over the counters into a histogram of counts, whose last workgroup turns it into mass(top) for top = 1, 2, 4,
// ... hottest columns.  160 bytes come back to the host (one wait on an event: not capturable).
#include "spmm_common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace mx {

constexpr int PF_BINS = 1024;            // counts 0 .. 1022 exactly, 1023 = "at least 1023" (all of them hot)
constexpr int PF_RUN = 256;              // entries per sampled run
constexpr int PF_MAX_RUNS = 1024;        // 2^18 sampled entries
constexpr int PF_STATS = 5;              // per row block: sum of lengths, of squares, longest, entries / number of the long rows

// Kernel 1 — blocks 0 .. runs - 1: two independent half-samples of the column ids (even / odd runs): half A RANKS the
// columns, half B MEASURES the entries they hold — ranking and measuring on the same counts would credit the top ranks with
// their sampling noise (uniform columns, K = 1e6: the 32,768 "hottest" columns of one 2^18-entry sample hold 27 % of that
// sample and 3 % of the matrix).  Blocks runs .. : the row statistics (sum of lengths, of squares, longest row).
__global__ __launch_bounds__(256)
void profile_sample_kernel(int m, int64_t nnz, const int32_t *__restrict__ indptr, int runs, const int32_t *__restrict__ indices, int K,
                           unsigned *__restrict__ count_a, unsigned *__restrict__ count_b, double *__restrict__ stats, int long_len)
{
    const int run = blockIdx.x;
    if (run < runs) {
        const int64_t first = indptr[0];                // (a row block of a larger matrix: indptr[0] need not be 0)
        // run r covers entries [r * nnz / runs, ...): evenly spaced over the matrix
        const int64_t at = first + (int64_t)((double)run * (double)nnz / (double)runs) + threadIdx.x;
        if (at < first + nnz) {
            const int c = indices[at];
            if ((unsigned)c < (unsigned)K) atomicAdd((run & 1) ? &count_b[c] : &count_a[c], 1u);
        }
        return;
    }
    double s = 0.0, q = 0.0, le = 0.0, lc = 0.0;         // le / lc: entries / number of the rows longer than long_len
    unsigned mx_ = 0;
    const int nb = gridDim.x - runs;
    for (int r = (blockIdx.x - runs) * blockDim.x + threadIdx.x; r < m; r += nb * blockDim.x) {
        const int len = indptr[r + 1] - indptr[r];
        s += (double)len; q += (double)len * (double)len;
        mx_ = max(mx_, (unsigned)len);
        if (len > long_len) { le += (double)len; lc += 1.0; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64);
        le += __shfl_xor(le, o, 64); lc += __shfl_xor(lc, o, 64);
        mx_ = max(mx_, (unsigned)__shfl_xor((int)mx_, o, 64));
    }
    // one partial per workgroup, added up by the last workgroup of the second launch (a thousand wavefronts adding into the
    // same three words took 45 us: same-address atomics are served one after the other at the memory side)
    __shared__ double ps[4], pq[4], ple[4], plc[4];
    __shared__ unsigned pm[4];
    if (lane_id() == 0) { const int w = threadIdx.x / 64; ps[w] = s; pq[w] = q; pm[w] = mx_; ple[w] = le; plc[w] = lc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *dst = stats + PF_STATS * (size_t)(blockIdx.x - runs);
        dst[0] = ps[0] + ps[1] + ps[2] + ps[3];
        dst[1] = pq[0] + pq[1] + pq[2] + pq[3];
        dst[2] = (double)max(max(pm[0], pm[1]), max(pm[2], pm[3]));
        dst[3] = ple[0] + ple[1] + ple[2] + ple[3];
        dst[4] = plc[0] + plc[1] + plc[2] + plc[3];
    }
}

// Kernel 2 — bins[b] = number of columns whose count in half A is b (clipped), bins[PF_BINS + b] = their counts in half B;
// the LAST workgroup to finish turns the histogram into the profile: mass(top) for top = 2^i, walking the bins from the most
// frequent columns down (the columns of one bin carry equal counts: a bin is taken pro rata).
__global__ __launch_bounds__(256)
void profile_bins_kernel(int m, int K, const unsigned *__restrict__ count_a, const unsigned *__restrict__ count_b, unsigned *__restrict__ bins,
                         const double *__restrict__ stats, int row_blocks, unsigned *__restrict__ done, float *__restrict__ profile)
{
    __shared__ unsigned cols[PF_BINS], sums[PF_BINS];
    __shared__ bool last;
    for (int b = threadIdx.x; b < PF_BINS; b += blockDim.x) { cols[b] = 0; sums[b] = 0; }
    __syncthreads();
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
        const unsigned v = count_a[c], w = count_b[c];
        if (v | w) {
            const unsigned b = v < PF_BINS - 1 ? v : PF_BINS - 1;
            atomicAdd(&cols[b], 1u);
            atomicAdd(&sums[b], w);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < PF_BINS; b += blockDim.x) {
        if (cols[b]) { atomicAdd(&bins[b], cols[b]); atomicAdd(&bins[PF_BINS + b], sums[b]); }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // ---- the last workgroup: the histogram -> mass(top), in parallel (one thread walking 1,024 bins took ~40 us, reading them
    // from L2 one after the other 0.37 ms).  cols / sums are reused: the totals of all bins, then suffix sums from the top bin
    // down (cumulative columns / cumulative mass of the columns at least that frequent), then one thread per level.
    __shared__ double cum_cols[PF_BINS + 1], cum_mass[PF_BINS + 1], red[8];
    for (int b = threadIdx.x; b < PF_BINS; b += blockDim.x) {
        cols[b] = __hip_atomic_load(&bins[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sums[b] = __hip_atomic_load(&bins[PF_BINS + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    double t = 0.0, me = 0.0, ta = 0.0;
    for (int b = threadIdx.x; b < PF_BINS; b += blockDim.x) {
        t += (double)sums[b];
        if (b) { me += (double)cols[b]; ta += (double)cols[b] * b; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o, 64); me += __shfl_xor(me, o, 64); ta += __shfl_xor(ta, o, 64); }
    if (lane_id() == 0) { red[threadIdx.x / 64] = t; red[4 + threadIdx.x / 64] = me; }
    __shared__ double red_a[4];
    if (lane_id() == 0) red_a[threadIdx.x / 64] = ta;
    __syncthreads();
    double total = red[0] + red[1] + red[2] + red[3];
    const double met = red[4] + red[5] + red[6] + red[7], total_a = red_a[0] + red_a[1] + red_a[2] + red_a[3];
    const bool one_half = total <= 0.0;                 // a one-run sample: no second half — rank and measure on half A
    if (one_half) total = total_a > 0.0 ? total_a : 1.0;
    // position k = 0 is the top bin: cum_*[k + 1] = columns / mass of bins PF_BINS - 1 .. PF_BINS - 1 - k (bin 0: the columns half
    // A never met — as many as K minus the rest; only those half B met were counted there)
    if (threadIdx.x < 64) {                             // one wavefront: 16 positions per lane, a scan over the lanes
        constexpr int PER = PF_BINS / 64;
        double nc[PER], ns[PER], cc = 0.0, cm = 0.0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int b = PF_BINS - 1 - (threadIdx.x * PER + u);
            nc[u] = b == 0 ? fmax(0.0, (double)K - met) : (double)cols[b];
            ns[u] = one_half ? (b == 0 ? 0.0 : nc[u] * b) : (double)sums[b];
            cc += nc[u]; cm += ns[u];
        }
        double pc = cc, pm_ = cm;                       // inclusive scan of the lanes' totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double a = __shfl_up(pc, o, 64), b2 = __shfl_up(pm_, o, 64);
            if ((int)threadIdx.x >= o) { pc += a; pm_ += b2; }
        }
        double rc = pc - cc, rm = pm_ - cm;             // exclusive
        if (threadIdx.x == 0) cum_cols[0] = cum_mass[0] = 0.0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            rc += nc[u]; rm += ns[u];
            cum_cols[threadIdx.x * PER + u + 1] = rc; cum_mass[threadIdx.x * PER + u + 1] = rm;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const double top = (double)(1ULL << threadIdx.x);
        float v = 1.0f;
        if (top < cum_cols[PF_BINS]) {
            int lo = 0, hi = PF_BINS;                   // the first k with cum_cols[k + 1] >= top
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum_cols[mid + 1] >= top) hi = mid; else lo = mid + 1; }
            const double nc = cum_cols[lo + 1] - cum_cols[lo], ns = cum_mass[lo + 1] - cum_mass[lo];
            // (the columns of one bin carry equal counts: the bin is taken pro rata)
            v = (float)((cum_mass[lo] + (nc > 0.0 ? ns * ((top - cum_cols[lo]) / nc) : 0.0)) / total);
        }
        profile[threadIdx.x] = v;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 128) {       // the second wavefront: the row statistics' partials
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
        for (int k = threadIdx.x - 64; k < row_blocks; k += 64) {
            const double *b = stats + PF_STATS * k;
            s0 += b[0]; s1 += b[1]; s2 = fmax(s2, b[2]); s3 += b[3]; s4 += b[4];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 = fmax(s2, __shfl_xor(s2, o, 64));
            s3 += __shfl_xor(s3, o, 64); s4 += __shfl_xor(s4, o, 64);
        }
        if (threadIdx.x != 64) return;
        const double mean = s0 / m, var = s1 / m - mean * mean;
        profile[32] = mean > 0.0 ? (float)(sqrt(var > 0.0 ? var : 0.0) / mean) : 0.0f;
        profile[33] = mean > 0.0 ? (float)(s2 / mean) : 0.0f;
        profile[34] = (float)mean;
        // entries / number of the rows longer than canonical_long_piece(mean) — what sizes the row-split kernel's long-rows
        // scratch; floats round to nearest: rounded UP here (nextafter) so that they stay bounds
        profile[35] = nextafterf((float)s3, INFINITY); profile[36] = nextafterf((float)s4, INFINITY); profile[37] = 1.0f;
        for (int i = 38; i < MX_PROFILE_LEN; i++) profile[i] = 0.0f;
    }
}

}  // namespace mx

extern "C" size_t mxd_csr_profile_workspace_bytes(int K)
{
    return 2 * (size_t)(K > 0 ? K : 1) * sizeof(unsigned) + 2 * mx::PF_BINS * sizeof(unsigned) + 64 + mx::PF_STATS * 256 * sizeof(double) +
           MX_PROFILE_LEN * sizeof(float) + 64;
}

// profile[0 .. 31]: share of the entries whose column is among the 2^i most frequent columns (1.0 from 2^i >= the number of
// columns that occur at all); profile[32]: coefficient of variation of the row lengths; profile[33]: longest row / mean row;
// profile[34]: mean row length; profile[35] / [36]: entries / number of the rows longer than ~6 mean rows (the canonical piece
// of the row-split kernel's long-rows path, rounded up to floats), [37] = 1 when those two are filled; [38]: reserved (0);
// [39]: < 0 marks "uniform columns" (mx::uniform_profile).  One memset, two launches, 160 bytes through the pinned
// landing zone of read_back_small (~25 us; a pageable 8 KB copy of the bins cost 100 us more).
extern "C" int mxd_csr_profile(int m, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices, float *profile_host,
                               void *workspace, void *stream)
{
    MX_REQUIRE(profile_host, "mxd_csr_profile: null result pointer");
    for (int i = 0; i < MX_PROFILE_LEN; i++) profile_host[i] = 0.0f;
    if (m <= 0 || K <= 0 || nnz <= 0) { for (int i = 0; i < 32; i++) profile_host[i] = 1.0f; return 0; }
    MX_REQUIRE(indptr && indices && workspace, "mxd_csr_profile: null pointer");
    hipStream_t st = mx::as_stream(stream);
    unsigned *count_a = (unsigned *)workspace, *count_b = count_a + K;
    unsigned *bins = count_b + K;
    double *stats = (double *)(((uintptr_t)(bins + 2 * mx::PF_BINS) + 15) & ~(uintptr_t)15);
    unsigned *done = (unsigned *)(stats + mx::PF_STATS * 256);
    float *profile = (float *)(stats + mx::PF_STATS * 256 + 2);
    MX_HIP(hipMemsetAsync(workspace, 0, mxd_csr_profile_workspace_bytes(K), st));
    int runs = (int)std::min<int64_t>(mx::PF_MAX_RUNS, mx::ceil_div(nnz, mx::PF_RUN));
    if (runs > 1) runs &= ~1;                            // two halves of equal size
    const int row_blocks = (int)std::min<int64_t>(256, mx::ceil_div(m, 256));
    hipLaunchKernelGGL(mx::profile_sample_kernel, dim3((unsigned)(runs + row_blocks)), dim3(mx::PF_RUN), 0, st, m, nnz, indptr, runs, indices, K,
                       count_a, count_b, stats, mx::canonical_long_piece((double)nnz / (double)m));
    hipLaunchKernelGGL(mx::profile_bins_kernel, dim3((unsigned)std::min<int64_t>(256, mx::ceil_div(K, 256))), dim3(256), 0, st, m, K, count_a, count_b,
                       bins, stats, row_blocks, done, profile);
    MX_LAUNCH_CHECK();
    return mx::read_back_small(profile_host, profile, MX_PROFILE_LEN * sizeof(float), st);
}
