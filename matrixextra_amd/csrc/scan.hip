// scan.hip — exclusive prefix sum of int32 row lengths into an int32 indptr,
// carried in int64 so that an overflow of R's int32 nnz limit is detected
// instead of wrapping (the reference keeps `size_t curr` and stores it into an
// int indptr, operators.cpp:414,521).
//
// Three launches (reduce tiles -> scan tile sums in one workgroup -> scan tiles
// with carried offset); vectors of up to 2^15 entries take one single-workgroup launch instead.  Row-count vectors are <= 32 MB here; the scan is a
// bandwidth-trivial step between the count and fill passes of merge / gather.
#include "mx_common.h"

#include <algorithm>
#include <cstring>

namespace mx {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ long long wave_incl_scan(long long v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < MX_WAVE; off <<= 1) {
        const long long o = __shfl_up(v, off, MX_WAVE);
        if (lane >= off) v += o;
    }
    return v;
}

// inclusive scan of one value per thread across the workgroup; returns the
// thread's inclusive value, *block_total = sum over the workgroup
__device__ __forceinline__ long long block_incl_scan(long long v, long long *block_total)
{
    __shared__ long long wave_sums[SCAN_BLOCK / MX_WAVE];
    const int lane = lane_id(), wave = threadIdx.x / MX_WAVE;
    long long incl = wave_incl_scan(v);
    if (lane == MX_WAVE - 1) wave_sums[wave] = incl;
    __syncthreads();
    long long base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / MX_WAVE; w++) {
        const long long ws = wave_sums[w];
        if (w < wave) base += ws;
        total += ws;
    }
    __syncthreads();
    *block_total = total;
    return incl + base;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void scan_reduce_kernel(const int32_t *__restrict__ counts, int64_t n, long long *__restrict__ tile_sums)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    long long sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) sum += counts[base + i];
    long long total;
    block_incl_scan(sum, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the tile sums in place; grand total out
__global__ __launch_bounds__(SCAN_BLOCK)
void scan_sums_kernel(long long *__restrict__ tile_sums, int64_t ntiles, long long *__restrict__ total_out)
{
    long long carry = 0;
    for (int64_t b0 = 0; b0 < ntiles; b0 += SCAN_BLOCK) {
        const int64_t i = b0 + threadIdx.x;
        const long long v = i < ntiles ? tile_sums[i] : 0;
        long long total;
        const long long incl = block_incl_scan(v, &total);
        if (i < ntiles) tile_sums[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void scan_tiles_kernel(const int32_t *__restrict__ counts, int64_t n, const long long *__restrict__ tile_offsets,
                       const long long *__restrict__ total, int32_t *__restrict__ out)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t c[SCAN_ITEMS];
    long long sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        c[i] = base + i < n ? counts[base + i] : 0;
        sum += c[i];
    }
    long long block_total;
    const long long incl = block_incl_scan(sum, &block_total);
    long long run = tile_offsets[blockIdx.x] + incl - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = (int32_t)run;
        run += c[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (int32_t)*total;
}

// Short vectors (the plan's per-octet step counts: 16 k entries for 1 M rows): ONE launch, one 1024-thread
// workgroup walking the vector in tiles of 16 k elements with a carried offset.
constexpr int SCAN1_BLOCK = 1024;
constexpr int64_t SCAN1_MAX = (int64_t)1 << 15;         // two tiles: beyond that the three-launch scan is faster (200 k elements: 0.2 ms in one workgroup)

__global__ __launch_bounds__(SCAN1_BLOCK)
void scan_single_kernel(const int32_t *__restrict__ counts, int64_t n, int32_t *__restrict__ out,
                        long long *__restrict__ total_out)
{
    __shared__ long long wave_sums[SCAN1_BLOCK / MX_WAVE];
    const int lane = lane_id(), wave = threadIdx.x / MX_WAVE;
    long long carry = 0;
    for (int64_t t0 = 0; t0 < n; t0 += (int64_t)SCAN1_BLOCK * SCAN_ITEMS) {
        const int64_t base = t0 + (int64_t)threadIdx.x * SCAN_ITEMS;
        int32_t c[SCAN_ITEMS];
        long long sum = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            c[i] = base + i < n ? counts[base + i] : 0;
            sum += c[i];
        }
        const long long incl = wave_incl_scan(sum);
        if (lane == MX_WAVE - 1) wave_sums[wave] = incl;
        __syncthreads();
        long long before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SCAN1_BLOCK / MX_WAVE; w++) {
            const long long ws = wave_sums[w];
            if (w < wave) before += ws;
            total += ws;
        }
        __syncthreads();
        long long run = carry + before + incl - sum;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < n) out[base + i] = (int32_t)run;
            run += c[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0) { out[n] = (int32_t)carry; *total_out = carry; }
}

namespace {
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;               // recorded behind the last user (scratch_done); an event outlives its stream
    hipStream_t last = nullptr;
    bool used = false;
    unsigned gen = 0;                        // counts the (re)allocations: what a user keeps IN the buffer between calls is gone when it changes
};
thread_local Scratch g_scratch[16][MX_SCRATCH_SLOTS];
Scratch &scratch_slot(int slot)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    return g_scratch[d][slot];
}
}
// A scratch slot is per thread and device, but a thread may come back on ANOTHER stream while the previous user's kernels
// are still queued (a torch stream switch; ADVICE r3 found this for the gather's look-back state): scratch_acquire makes
// `st` wait for the previous user when the stream changed, scratch_done marks the end of this user's work on `st`.
void scratch_acquire(int slot, hipStream_t st)
{
    Scratch &w = scratch_slot(slot);
    if (w.used && w.last != st) {
        if (w.done) (void)hipStreamWaitEvent(st, w.done, 0);
        else (void)hipDeviceSynchronize();
    }
}
void scratch_done(int slot, hipStream_t st)
{
    Scratch &w = scratch_slot(slot);
    if (!w.done && hipEventCreateWithFlags(&w.done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); w.done = nullptr; }
    if (w.done) (void)hipEventRecord(w.done, st);
    w.last = st;
    w.used = true;
}
void *scratch_buffer(int slot, size_t bytes)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    Scratch &w = g_scratch[d][slot];
    if (w.cap < bytes || !w.p) {
        if (w.p) (void)hipFree(w.p);
        w.p = nullptr; w.cap = 0;
        const size_t want = bytes + std::min<size_t>(bytes / 2, (size_t)256 << 20) + 4096;   // (fresh VRAM is cleared by the copy engines: 12 GB for an 8 GB result delayed that call's uploads by ~100 ms)
        if (hipMalloc(&w.p, want) != hipSuccess) { w.p = nullptr; return nullptr; }
        w.cap = want;
        w.gen++;
    }
    return w.p;
}
unsigned scratch_generation(int slot) { return scratch_slot(slot).gen; }
// the same buffer, zero-filled whenever it is (re)allocated (state that kernels reset themselves afterwards)
void *scratch_buffer_zeroed(int slot, size_t bytes, hipStream_t st, bool *fresh)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    Scratch &w = g_scratch[d][slot];
    const void *before = w.p;
    const size_t cap_before = w.cap;
    void *p = scratch_buffer(slot, bytes);
    const bool is_new = p && (p != before || w.cap != cap_before);
    if (is_new && hipMemsetAsync(p, 0, w.cap, st) != hipSuccess) return nullptr;
    if (fresh) *fresh = is_new;
    return p;
}
void scratch_release()
{
    for (auto &dev : g_scratch)
        for (auto &w : dev) { if (w.p) (void)hipFree(w.p); w.p = nullptr; w.cap = 0; }     // (hipFree waits for the device: nobody uses it any more)
}

int read_back_small(void *host_dst, const void *dev_src, size_t bytes, hipStream_t st)
{
    struct Rb { void *host = nullptr; hipEvent_t ev = nullptr; };
    static thread_local Rb rbs[16];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    Rb &rb = rbs[d];
    MX_REQUIRE(bytes <= 256, "read_back_small: %zu bytes", bytes);
    if (!rb.host) {
        MX_HIP(hipHostMalloc(&rb.host, 256, hipHostMallocDefault));
        if (hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipHostFree(rb.host);
            rb.host = nullptr;
            return set_error("read_back_small: cannot create an event");
        }
    }
    MX_HIP(hipMemcpyAsync(rb.host, dev_src, bytes, hipMemcpyDeviceToHost, st));
    MX_HIP(hipEventRecord(rb.ev, st));
    MX_HIP(hipEventSynchronize(rb.ev));
    memcpy(host_dst, rb.host, bytes);
    return 0;
}

int host_signal_next(HostSignal *s)
{
    struct Hs { unsigned long long *word = nullptr; unsigned gen = 0; };
    static thread_local Hs hs[16];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    Hs &h = hs[d];
    if (!h.word) {
        MX_HIP(hipHostMalloc((void **)&h.word, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *h.word = 0;
    }
    h.gen = (h.gen + 1) & 0x3FFFFFFFu;
    if (h.gen == 0) h.gen = 1;
    s->word = h.word;
    s->gen = h.gen;
    return 0;
}

int host_signal_wait(const HostSignal &s, unsigned *value, hipStream_t st)
{
    volatile unsigned long long *w = s.word;
    unsigned long long v = 0;
    bool got = false;
    for (int spin = 0; spin < 400000; spin++) {
        v = *w;
        if ((unsigned)(v >> 32) == s.gen) { got = true; break; }
    }
    if (!got) {
        MX_HIP(hipStreamSynchronize(st));
        v = *w;
        MX_REQUIRE((unsigned)(v >> 32) == s.gen, "the kernel did not report its result");
    }
    *value = (unsigned)(v & 0xFFFFFFFFu);
    return 0;
}

// workspace layout: [int64 total][int64 tile_sums[ntiles]]
size_t scan_workspace_bytes(int64_t n)
{
    return sizeof(long long) * (size_t)(1 + ceil_div(n > 0 ? n : 1, SCAN_TILE));
}

int exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev, void *workspace,
                       hipStream_t st)
{
    long long *ws = (long long *)workspace;
    long long *total = total_dev ? (long long *)total_dev : ws;
    long long *tile_sums = ws + 1;
    if (n <= 0) {
        MX_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), st));
        MX_HIP(hipMemsetAsync(total, 0, sizeof(long long), st));
        return 0;
    }
    if (n <= SCAN1_MAX) {
        hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(SCAN1_BLOCK), 0, st, counts, n, out, total);
        MX_LAUNCH_CHECK();
        return 0;
    }
    const int64_t ntiles = ceil_div(n, SCAN_TILE);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, st, counts, n, tile_sums);
    MX_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, tile_sums, ntiles, total);
    MX_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, st, counts, n, tile_sums,
                       total, out);
    MX_LAUNCH_CHECK();
    return 0;
}

}  // namespace mx

extern "C" size_t mxd_scan_workspace_bytes(int64_t n) { return mx::scan_workspace_bytes(n); }

extern "C" int mxd_exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out, int64_t *total_dev,
                                      void *workspace, void *stream)
{
    MX_REQUIRE(out && workspace, "mxd_exclusive_scan_i32: null pointer");
    return mx::exclusive_scan_i32(counts, n, out, total_dev, workspace, mx::as_stream(stream));
}
