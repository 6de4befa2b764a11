// profile.hip — what AUTO has to know about a CSR matrix beyond its sizes: real dgRMatrix data has
// power-law COLUMNS and skewed ROW lengths (the vignette's own application is LibSVM real-sim, Rmd:442-502), and both move
// the kernels apart:
//   * the gather kernels (row-wave, row-split) read a row of B per entry through an XCD's 4 MiB L2: what counts is the share
//     of the ENTRIES whose column is among the rows of B that L2 holds — the probability mass of the hottest columns — not
//     the share of B's bytes (uniform columns: the same number; Zipf columns, K = 1e5, n = 128: 0.68 against 0.04);
//   * the kernels that walk several rows in lockstep (the row-group form, the LDS-tile kernel) run as long as the longest
//     of their rows: the coefficient of variation of the row lengths prices that.
// mxd_csr_profile: one pass over a SAMPLE of the column ids (<= 2^18 entries, evenly spaced 256-entry runs: hot columns
// are what matters and they show in any sample) into per-column counters, one pass over the counters into a histogram of
// counts, one pass over indptr for the row statistics; the host turns the histogram into mass(top) for top = 1, 2, 4, ...
// hottest columns.  ~40 us for cfg2's matrix; brings 8 KB back to the host (one stream synchronisation: not capturable).
#include "mx_common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace mx {

constexpr int PF_BINS = 1024;            // counts 0 .. 1022 exactly, 1023 = "at least 1023" (all of them hot)
constexpr int PF_RUN = 256;              // entries per sampled run
constexpr int PF_MAX_RUNS = 1024;        // 2^18 sampled entries

// Two independent half-samples (even / odd runs): half A RANKS the columns, half B MEASURES the entries they hold — ranking
// and measuring on the same counts would credit the top ranks with their sampling noise (uniform columns, K = 1e6: the
// 32,768 "hottest" columns of one 2^18-entry sample hold 27 % of that sample and 3 % of the matrix).
__global__ __launch_bounds__(256)
void profile_sample_kernel(int64_t nnz, const int32_t *__restrict__ indptr, int runs, const int32_t *__restrict__ indices, int K,
                           unsigned *__restrict__ count_a, unsigned *__restrict__ count_b)
{
    const int run = blockIdx.x;
    if (run >= runs) return;
    const int64_t first = indptr[0];                    // (a row block of a larger matrix: indptr[0] need not be 0)
    // run r covers entries [r * nnz / runs, ...): evenly spaced over the matrix
    const int64_t at = first + (int64_t)((double)run * (double)nnz / (double)runs) + threadIdx.x;
    if (at < first + nnz) {
        const int c = indices[at];
        if ((unsigned)c < (unsigned)K) atomicAdd((run & 1) ? &count_b[c] : &count_a[c], 1u);
    }
}

// bins[b] = number of columns whose count in half A is b (clipped; 0 included), bins[PF_BINS + b] = their counts in half B
__global__ __launch_bounds__(256)
void profile_bins_kernel(int K, const unsigned *__restrict__ count_a, const unsigned *__restrict__ count_b, unsigned *__restrict__ bins)
{
    __shared__ unsigned cols[PF_BINS], sums[PF_BINS];
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
}

// stats[0] = sum of lengths, [1] = sum of squares, [2] = longest row (as doubles; one atomic per wavefront)
__global__ __launch_bounds__(256)
void profile_rows_kernel(int m, const int32_t *__restrict__ indptr, double *__restrict__ stats, unsigned *__restrict__ longest)
{
    double s = 0.0, q = 0.0;
    unsigned mx_ = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < m; r += gridDim.x * blockDim.x) {
        const int len = indptr[r + 1] - indptr[r];
        s += (double)len; q += (double)len * (double)len;
        mx_ = max(mx_, (unsigned)len);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64);
        mx_ = max(mx_, (unsigned)__shfl_xor((int)mx_, o, 64));
    }
    if (lane_id() == 0) { atomicAdd(&stats[0], s); atomicAdd(&stats[1], q); atomicMax(longest, mx_); }
}

}  // namespace mx

extern "C" size_t mxd_csr_profile_workspace_bytes(int K)
{
    return 2 * (size_t)(K > 0 ? K : 1) * sizeof(unsigned) + 2 * mx::PF_BINS * sizeof(unsigned) + 64;
}

// profile[0 .. 31]: share of the entries whose column is among the 2^i most frequent columns (1.0 from 2^i >= the number of
// columns that occur at all); profile[32]: coefficient of variation of the row lengths; profile[33]: longest row / mean row;
// profile[34]: mean row length; profile[35 .. 39]: reserved (0).
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
    unsigned *longest = (unsigned *)(stats + 2);
    MX_HIP(hipMemsetAsync(workspace, 0, mxd_csr_profile_workspace_bytes(K), st));
    int runs = (int)std::min<int64_t>(mx::PF_MAX_RUNS, mx::ceil_div(nnz, mx::PF_RUN));
    if (runs > 1) runs &= ~1;                            // two halves of equal size
    hipLaunchKernelGGL(mx::profile_sample_kernel, dim3((unsigned)runs), dim3(mx::PF_RUN), 0, st, nnz, indptr, runs, indices, K, count_a, count_b);
    hipLaunchKernelGGL(mx::profile_bins_kernel, dim3((unsigned)std::min<int64_t>(256, mx::ceil_div(K, 256))), dim3(256), 0, st, K, count_a, count_b,
                       bins);
    hipLaunchKernelGGL(mx::profile_rows_kernel, dim3((unsigned)std::min<int64_t>(512, mx::ceil_div(m, 256))), dim3(256), 0, st, m, indptr, stats,
                       longest);
    MX_LAUNCH_CHECK();
    std::vector<unsigned> hb(2 * mx::PF_BINS + 8);
    MX_HIP(hipMemcpyAsync(hb.data(), bins, (2 * mx::PF_BINS) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    double hs[3] = {0, 0, 0};
    MX_HIP(hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, st));
    MX_HIP(hipStreamSynchronize(st));
    unsigned hl = 0;
    memcpy(&hl, &hs[2], sizeof(unsigned));
    // mass(top): walk the histogram of half A's counts from the most frequent columns down, adding up what half B saw of
    // them; the columns half A never met (bin 0: only those half B met are counted there) are as many as K minus the rest
    double total = 0.0, met = 0.0;
    for (int b = 0; b < mx::PF_BINS; b++) total += hb[mx::PF_BINS + b];
    for (int b = 1; b < mx::PF_BINS; b++) met += hb[b];
    if (total <= 0.0) {                                  // (a one-run sample: no second half — rank and measure on half A)
        for (int b = 1; b < mx::PF_BINS; b++) { hb[mx::PF_BINS + b] = hb[b] * (unsigned)b; total += hb[mx::PF_BINS + b]; }
        if (total <= 0.0) total = 1.0;
    }
    hb[0] = (unsigned)std::max(0.0, (double)K - met);
    int level = 0;
    double cols_seen = 0.0, mass_seen = 0.0;
    for (int b = mx::PF_BINS - 1; b >= 0 && level < 32; b--) {
        const double nc = hb[b], ns = hb[mx::PF_BINS + b];
        if (nc <= 0.0) continue;
        // the columns of one bin carry equal counts (the top bin: taken whole — at most a few hundred columns)
        while (level < 32 && (double)(1ULL << level) <= cols_seen + nc) {
            const double take = (double)(1ULL << level) - cols_seen;
            profile_host[level] = (float)((mass_seen + ns * (take / nc)) / total);
            level++;
        }
        cols_seen += nc; mass_seen += ns;
    }
    for (; level < 32; level++) profile_host[level] = 1.0f;
    const double mean = hs[0] / m, var = hs[1] / m - mean * mean;
    profile_host[32] = mean > 0.0 ? (float)(std::sqrt(var > 0.0 ? var : 0.0) / mean) : 0.0f;
    profile_host[33] = mean > 0.0 ? (float)((double)hl / mean) : 0.0f;
    profile_host[34] = (float)mean;
    return 0;
}
