// Device blocks that come back: the export level allocates and frees the same few sizes call after call (operands,
// results, the plan hanging off a cache entry), and on this runtime a large hipMalloc / hipFree pair is not only the two
// driver calls (~0.1-1 ms each above a few MB): freed VRAM is scrubbed by the copy engines afterwards, and that scrub
// competes with the next call's downloads — measured on cfg2's cold export: 38.2 ms when the previous call's plan and CSR
// (~1 GB) had just been freed, 31.6 ms with a 100 ms pause after the free, 31 ms when nothing was freed
// (tools/cold_forms_probe.py).  Blocks freed through pool_free are kept (up to a cap) and handed to the next pool_malloc
// of about their size.
//
// Like hipFree, pool_free first waits for the device: a block goes back only when nothing queued can still touch it.
#include "mx_common.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace mx {

namespace {
struct Block { void *p; size_t bytes; int dev; unsigned long long stamp; };
struct Pool {
    std::mutex mu;
    std::unordered_map<void *, Block> live;                       // blocks handed out by pool_malloc
    std::vector<Block> idle;
    size_t idle_bytes = 0, cap_bytes = 0;
    unsigned long long clock = 0, hits = 0, misses = 0;
    bool cap_known = false;
};
Pool &pool() { static Pool *p = new Pool(); return *p; }         // (never destroyed: frees may come after static teardown)
constexpr size_t POOL_MAX_BLOCKS = 160;

size_t pool_cap_locked(Pool &P)
{
    if (!P.cap_known) {
        const char *e = getenv("MXGPU_POOL_MB");
        if (e) P.cap_bytes = (size_t)std::max(0LL, atoll(e)) << 20;
        else {
            // (round 4: 32 GiB / an eighth of the device; rounds 2-3 kept 8 GiB / a sixteenth, which a sharded export of
            // configs[4] — 8 shards x 2 GB of operands and result — overran on every call: half of its blocks were
            // hipFree'd and hipMalloc'ed again, 190 ms of driver work at the head of the next call)
            size_t fr = 0, tot = 0;
            P.cap_bytes = (size_t)32 << 30;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess) P.cap_bytes = std::min(P.cap_bytes, tot / 8);
            else (void)hipGetLastError();
        }
        P.cap_known = true;
    }
    return P.cap_bytes;
}
inline size_t slack_of(size_t n) { return std::max<size_t>(n / 8, (size_t)64 << 10); }
// MXGPU_POOL_POISON=1 (test runs): every block handed out — new or reused — is first filled with 0xA5 bytes, so that a
// kernel which counts on fresh, zeroed memory shows up as a wrong result instead of working by accident
bool poison_on() { static const bool on = [] { const char *e = getenv("MXGPU_POOL_POISON"); return e && atoi(e) != 0; }(); return on; }
hipError_t hand_out(void *q, size_t bytes)
{
    if (!poison_on()) return hipSuccess;
    hipError_t e = hipMemset(q, 0xA5, bytes);
    return e == hipSuccess ? hipDeviceSynchronize() : e;
}
}  // namespace

hipError_t pool_malloc(void **out, size_t n)
{
    if (n == 0) n = 16;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    Pool &P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        int best = -1;
        for (int i = 0; i < (int)P.idle.size(); i++) {
            const Block &b = P.idle[i];
            if (b.dev != dev || b.bytes < n || b.bytes - n > slack_of(n)) continue;
            if (best < 0 || b.bytes < P.idle[best].bytes) best = i;
        }
        if (best >= 0) {
            Block b = P.idle[best];
            P.idle.erase(P.idle.begin() + best);
            P.idle_bytes -= b.bytes;
            P.live[b.p] = b;
            P.hits++;
            *out = b.p;
            return hand_out(b.p, b.bytes);
        }
        P.misses++;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n);
    if (e != hipSuccess) {                                          // give the kept blocks back and try once more
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&q, n);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[q] = Block{q, n, dev, 0};
    *out = q;
    return hand_out(q, n);
}

void pool_free(void *p)
{
    if (!p) return;
    Pool &P = pool();
    Block b{};
    bool known = false;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) { b = it->second; P.live.erase(it); known = true; }
    }
    int dev = -1;
    if (!known || hipGetDevice(&dev) != hipSuccess || dev != b.dev) { (void)hipFree(p); return; }   // (not ours / another device current)
    // the block may stay: what hipFree would have waited for is waited for here, BEFORE another thread can be handed it
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return; }
    std::vector<Block> out;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        const size_t cap = pool_cap_locked(P);
        if (b.bytes > cap) out.push_back(b);
        else {
            b.stamp = ++P.clock;
            P.idle.push_back(b);
            P.idle_bytes += b.bytes;
            while (P.idle_bytes > cap || P.idle.size() > POOL_MAX_BLOCKS) {     // oldest first
                size_t o = 0;
                for (size_t i = 1; i < P.idle.size(); i++) if (P.idle[i].stamp < P.idle[o].stamp) o = i;
                out.push_back(P.idle[o]);
                P.idle_bytes -= P.idle[o].bytes;
                P.idle.erase(P.idle.begin() + (long)o);
            }
        }
    }
    for (const Block &o : out) (void)hipFree(o.p);
}

void pool_trim()
{
    Pool &P = pool();
    std::vector<Block> out;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        out.swap(P.idle);
        P.idle_bytes = 0;
    }
    for (const Block &o : out) (void)hipFree(o.p);
}

void pool_live(long long *live_bytes, long long *live_blocks)
{
    Pool &P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    long long b = 0;
    for (const auto &kv : P.live) b += (long long)kv.second.bytes;
    if (live_bytes) *live_bytes = b;
    if (live_blocks) *live_blocks = (long long)P.live.size();
}

void pool_stats(long long *idle_bytes, long long *idle_blocks, long long *hits, long long *misses)
{
    Pool &P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (idle_bytes) *idle_bytes = (long long)P.idle_bytes;
    if (idle_blocks) *idle_blocks = (long long)P.idle.size();
    if (hits) *hits = (long long)P.hits;
    if (misses) *misses = (long long)P.misses;
}

}  // namespace mx
