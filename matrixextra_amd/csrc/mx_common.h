// mx_common.h — shared host/device helpers of libmxgpu (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <climits>
#include "../../include/mxgpu.h"

#define MX_WAVE 64
#define MX_NA_INT INT_MIN                       // R NA_INTEGER / NA_LOGICAL
#define MX_NA_REAL_BITS 0x7FF00000000007A2ULL   // R NA_real_ (NaN, low word 1954)

namespace mx {

// thread-local error text behind mx_last_error()
int set_error(const char *fmt, ...);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define MX_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return mx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                  \
    } while (0)

#define MX_LAUNCH_CHECK()                                                              \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess)                                                          \
            return mx::set_error("kernel launch failed: %s (%s:%d)",                   \
                                 hipGetErrorString(_e), __FILE__, __LINE__);           \
    } while (0)

#define MX_REQUIRE(cond, ...)                                                          \
    do { if (!(cond)) return mx::set_error(__VA_ARGS__); } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The one host round trip of the count -> fill operations (a size, a flag, the matrix profile): `bytes` <= 256 from device memory into
// `host_dst`, through a pinned landing zone and an event of the calling thread and current device (a hipMemcpyAsync into
// pageable memory + hipStreamSynchronize costs ~100 us of runtime staging per call; this is ~10 us).  Returns when the
// value is there: everything enqueued on `st` before it has completed.  (scan.hip)
int read_back_small(void *host_dst, const void *dev_src, size_t bytes, hipStream_t st);

// A result word that a kernel stores straight into pinned host memory (system-scope store of `generation << 32 | value`)
// and the host spins on: what `read_back_small` costs — a D2H copy packet, an event, their completion signals: ~10 us —
// shrinks to the PCIe write (~1-2 us after the kernel's last workgroup).  One word per thread and device, one stream per
// thread and device at a time.  host_signal_next: the word's device-visible address and the generation the kernel must
// write; host_signal_wait: spins, then falls back to synchronising the stream (other work queued before the kernel), and
// returns the 32-bit value.  (scan.hip)
struct HostSignal { unsigned long long *word; unsigned gen; };
int host_signal_next(HostSignal *s);
int host_signal_wait(const HostSignal &s, unsigned *value, hipStream_t st);

// The long-rows path of the row-split kernel for ONE product: the piece length (0 = off, -1 = not decided: from the profile in
// scope) and what the profile knows about how many rows / pieces there will be (-1 = unknown: sized by what nnz allows).
// Chosen once for the WHOLE product (spmm_auto_family) and carried BY VALUE into every block of it, so that neither the
// piece length — a regrouping of a long row's sum — nor anything else about the last bits depends on how the export
// pipeline or the device list cut the product.  The counts only size the scratch: a kernel checks on the device that the
// launch's long rows fit (longrows_fit_kernel) and leaves them to the product kernels when they do not.
struct LongHint { int piece = 0; long long rows = -1, pieces = -1; };
// tile_cv: the row-length cv of the WHOLE product's matrix (0: unknown) — the tile kernel deals a block's rows by length from it
struct SpmmFamily { int family = 0, segments = 0, panels = 0; LongHint lh = {-1, -1, -1}; float tile_cv = 0.0f; };
inline int canonical_long_piece(double mean_row)         // ~6 mean rows per piece, a power of two in [128, 1024]
{
    int piece = 128;
    while (piece < 1024 && piece < 6.0 * mean_row) piece <<= 1;
    return piece;
}

// Grow-only device scratch of the calling thread and current device, one buffer per `slot` (kernels' internal tables: the
// SpMV slice table, per-workgroup partial counts ...).  Like AUTO's SpMM plan it assumes one stream per thread and device
// at a time.  nullptr when the allocation fails.  (scan.hip)
enum { MX_SCRATCH_SPMV_SLICES = 0, MX_SCRATCH_PARTIALS = 1, MX_SCRATCH_EXPORT_B = 2, MX_SCRATCH_EXPORT_C = 3, MX_SCRATCH_ROWSPLIT = 4,
       MX_SCRATCH_AUTO_PLAN = 5,      // no buffer of its own: orders the users of AUTO's per-thread plan (spmm_plan.hip) across streams
       MX_SCRATCH_PACKED_B = 6,       // likewise for the per-thread slab-major copy of B (spmm_slab.hip slab_pack_workspace)
       MX_SCRATCH_TILE_FLAGS = 7,     // per-row "not sorted by column" flags of the tile kernel (spmm_tile.hip)
       MX_SCRATCH_PROFILE = 8,        // counters of the one-shot AUTO's profile pass (profile.hip)
       MX_SCRATCH_LONGROWS = 9,       // list + pieces' partial sums of the row-split kernel's long rows (spmm_rowsplit.hip)
       MX_SCRATCH_MERGE_LONG = 10,    // list of the very long row pairs of a CSR (+) CSR launch (merge.hip)
       MX_SCRATCH_TILE_PERM = 11,     // slot -> row map of the tile kernel for rows of uneven length (spmm_tile.hip)
       MX_SCRATCH_TILE_X = 12,        // partial sums of the parts of cut rows (spmm_tile.hip)
       MX_SCRATCH_SLOTS = 13 };
void *scratch_buffer(int slot, size_t bytes);
unsigned scratch_generation(int slot);           // changes whenever the slot's buffer is (re)allocated — also after scratch_release
void *scratch_buffer_zeroed(int slot, size_t bytes, hipStream_t st, bool *fresh);   // zero-filled when (re)allocated
void scratch_release();
void scratch_acquire(int slot, hipStream_t st);     // before queueing work that uses the slot: waits for the previous user when the stream changed
void scratch_done(int slot, hipStream_t st);        // after queueing it

// Device blocks that are freed and allocated again call after call (export-level operands and results, the plans kept on
// cache entries): pool_free keeps a block (total capped: MXGPU_POOL_MB, default min(32 GiB, 1/8 of the device)) for the
// next pool_malloc of about its size on the same device.  pool_free waits for the device first, like hipFree; a pointer
// that did not come from pool_malloc is simply hipFree'd.  A pointer from pool_malloc must go back through pool_free.
// (pool.hip)
hipError_t pool_malloc(void **out, size_t bytes);
void pool_free(void *p);
void pool_trim();
void pool_stats(long long *idle_bytes, long long *idle_blocks, long long *hits, long long *misses);
void pool_live(long long *live_bytes, long long *live_blocks);       // handed out by pool_malloc, not yet back

// lanes-per-row for the sub-wave ("group") kernels: smallest power of two
// >= avg row length, clamped to [lo, 64]
inline int pick_group(double avg_len, int lo = 4)
{
    int g = lo;
    while (g < 64 && (double)g < avg_len) g <<= 1;
    return g;
}

#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return threadIdx.x & (MX_WAVE - 1); }

// wave-uniform broadcast of a value known to be uniform (moves it to an SGPR)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

__device__ __forceinline__ double na_real()
{
    return __longlong_as_double((long long)MX_NA_REAL_BITS);
}

// R's three-valued logic, src/operators.cpp:17-25
__device__ __forceinline__ int r_logical_or(int x, int y)
{
    if (x == MX_NA_INT) return (y == MX_NA_INT) ? MX_NA_INT : (y ? 1 : MX_NA_INT);
    if (y == MX_NA_INT) return x ? 1 : MX_NA_INT;
    return (x != 0) || (y != 0);
}
__device__ __forceinline__ int r_logical_and(int x, int y)
{
    if (x == MX_NA_INT) return (y == MX_NA_INT) ? MX_NA_INT : (y ? MX_NA_INT : 0);
    if (y == MX_NA_INT) return x ? MX_NA_INT : 0;
    return (x != 0) && (y != 0);
}
__device__ __forceinline__ int r_logical_xor(int x, int y)
{
    if (x == MX_NA_INT || y == MX_NA_INT) return MX_NA_INT;
    return (x != 0) != (y != 0);
}

// Decoupled look-back over per-tile totals (one 64-bit word per tile: 2 flag bits + value; relaxed agent-scope loads /
// stores — the value travels in the same word as its flag, so nothing else needs ordering).  Called by ONE whole
// wavefront of the tile's workgroup; tiles must be numbered in starting order (atomic ticket) so that every predecessor
// is running or done.  Publishes this tile's total, returns the sum of the totals of all earlier tiles (wave-uniform)
// and publishes the inclusive prefix.  LOOK windows of 64 predecessors are read per round trip: a tile that is still in
// flight only offers its own total, so the nearest known prefix is about as many tiles back as are resident.
constexpr unsigned long long LB_FLAG_AGG = 1ULL << 62, LB_FLAG_PRE = 2ULL << 62, LB_VALUE = (1ULL << 62) - 1;
template <int LOOK>
__device__ __forceinline__ long long lookback_exclusive(unsigned long long *__restrict__ tile_state, int tile, long long tile_total)
{
    const int lane = lane_id();
    long long excl = 0;
    if (tile == 0) {
        if (lane == 0)
            __hip_atomic_store(&tile_state[0], LB_FLAG_PRE | (unsigned long long)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0)
        __hip_atomic_store(&tile_state[tile], LB_FLAG_AGG | (unsigned long long)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int look = tile - 1;                                              // window u, lane i: tile look - 64 u - i
    for (;;) {
        unsigned long long st[LOOK];
#pragma unroll
        for (int u = 0; u < LOOK; u++) {
            const int t = look - 64 * u - lane;
            st[u] = LB_FLAG_PRE;                                      // before tile 0: an empty prefix
            if (t >= 0) st[u] = __hip_atomic_load(&tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool done = false, stalled = false;
#pragma unroll
        for (int u = 0; u < LOOK; u++) {
            if (done || stalled) continue;                            // uniform
            const unsigned long long empty = __ballot((st[u] >> 62) == 0);
            const unsigned long long pre = __ballot((st[u] >> 62) == 2);
            // usable part of the window: the lanes before the first empty one, cut after the first prefix
            const int first_empty = empty ? __builtin_ctzll(empty) : 64;
            const int first_pre = pre ? __builtin_ctzll(pre) : 64;
            const int upto = first_pre < first_empty ? first_pre + 1 : first_empty;
            long long v = lane < upto ? (long long)(st[u] & LB_VALUE) : 0;
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2, 64);
            excl += v;
            look -= upto;
            if (first_pre < first_empty) done = true;                 // reached a tile that knows its prefix
            else if (upto < 64) stalled = true;                       // a tile that has not published yet: read again from there
        }
        if (done) break;
        if (stalled) __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0)
        __hip_atomic_store(&tile_state[tile], LB_FLAG_PRE | (unsigned long long)(excl + tile_total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// The same look-back over state words that carry a GENERATION: [63:62] flag, [61:32] generation (30 bits), [31:0] value
// (saturating at 2^32 - 1).  A word of another generation reads as "not published yet", so the array is never cleared
// between launches — the caller passes a generation it has not used on this array for the last 2^30 launches (the array
// starts zeroed; generation 0 is never used).  Values are per-tile entry counts: results beyond R's int32 range saturate
// and are reported as "too many" by the caller.
constexpr unsigned long long LBG_VALUE = 0xFFFFFFFFULL;
__device__ __forceinline__ unsigned long long lbg_pack(unsigned flag, unsigned gen, long long v)
{
    const unsigned long long sat = v > (long long)LBG_VALUE ? LBG_VALUE : (unsigned long long)v;
    return ((unsigned long long)flag << 62) | ((unsigned long long)(gen & 0x3FFFFFFFu) << 32) | sat;
}
template <int LOOK>
__device__ __forceinline__ long long lookback_exclusive_gen(unsigned long long *__restrict__ tile_state, int tile, long long tile_total,
                                                            unsigned gen)
{
    const int lane = lane_id();
    long long excl = 0;
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&tile_state[0], lbg_pack(2, gen, tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&tile_state[tile], lbg_pack(1, gen, tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int look = tile - 1;
    for (;;) {
        unsigned long long st[LOOK];
#pragma unroll
        for (int u = 0; u < LOOK; u++) {
            const int t = look - 64 * u - lane;
            st[u] = lbg_pack(2, gen, 0);                              // before tile 0: an empty prefix
            if (t >= 0) {
                st[u] = __hip_atomic_load(&tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((st[u] >> 32) & 0x3FFFFFFFu) != (gen & 0x3FFFFFFFu)) st[u] = 0;   // another launch's word
            }
        }
        bool done = false, stalled = false;
#pragma unroll
        for (int u = 0; u < LOOK; u++) {
            if (done || stalled) continue;                            // uniform
            const unsigned long long empty = __ballot((st[u] >> 62) == 0);
            const unsigned long long pre = __ballot((st[u] >> 62) == 2);
            const int first_empty = empty ? __builtin_ctzll(empty) : 64;
            const int first_pre = pre ? __builtin_ctzll(pre) : 64;
            const int upto = first_pre < first_empty ? first_pre + 1 : first_empty;
            long long v = lane < upto ? (long long)(st[u] & LBG_VALUE) : 0;
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2, 64);
            excl += v;
            look -= upto;
            if (first_pre < first_empty) done = true;
            else if (upto < 64) stalled = true;
        }
        if (done) break;
        if (stalled) __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0)
        __hip_atomic_store(&tile_state[tile], lbg_pack(2, gen, excl + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// first position in [first, first+count) whose value is >= key
__device__ __forceinline__ int lower_bound_dev(const int32_t *__restrict__ first, int count, int key)
{
    int lo = 0;
    while (count > 0) {
        const int step = count >> 1;
        if (first[lo + step] < key) { lo += step + 1; count -= step + 1; }
        else count = step;
    }
    return lo;
}
#endif  // __HIPCC__

}  // namespace mx
