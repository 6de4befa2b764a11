// Host <-> device transfers for the export-level calls (the R boundary hands over ordinary pageable vectors and
// expects ordinary vectors back: INTEGRATION.md §5).
//
// Measured on the MI355X boxes (tools/host_xfer_probe.py): hipHostRegister of memory whose pages EXIST costs ~2 ms per
// GB (0.9 ms for the 388 MB of the headline CSR) and a copy to / from registered memory is one DMA at the PCIe rate with
// no CPU copy; registering FRESH memory (the result matrix R has just allocated and never touched) costs 43 ms per GB —
// the page faults, taken by one thread.  So large transfers
//   * H2D: register the caller's buffer, one direct DMA, unregister;
//   * D2H: first-touch the destination with a pool of host threads (prefault_begin: started by the export as soon as it
//     knows the result's address, i.e. it runs under the uploads and the kernels), then register + direct DMA.
// Should registration fail (locked-memory limit, exotic mappings) the previous scheme takes over: a pipeline over three
// pinned 8 MiB slots in which the DMA engine moves one slot while the pool copies the previous one to / from the
// caller's buffer.  MXGPU_XFER=0: plain hipMemcpy; MXGPU_XFER=2: always the staged pipeline.  Synchronous at return.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>
#include <sys/mman.h>

#include "host_pool.h"
#include "mx_common.h"

namespace mx {

namespace {

using Pool = HostPool;

constexpr int XF_SLOTS = 3;
constexpr size_t XF_CHUNK = (size_t)8 << 20;
constexpr size_t XF_MIN = (size_t)16 << 20;           // below this: no team first-touch, no registration
// Caller memory of XF_STAGE_MIN .. XF_REG_MIN bytes never reaches the runtime as a copy operand: a plain hipMemcpy of
// pageable memory of that size makes the runtime pin the caller's pages on the fly (and remember the mapping), and for
// memory of the brk heap that ended in "Memory access fault by GPU" at a heap address some calls later — again with
// tools/fuzz_export_tiles.py, whose cases allocate and free 1-13 MB index-pointer arrays between gigabyte results.  Such
// buffers go through the engine's own pinned slots (host copy by the team, DMA out of / into the slot); smaller ones are
// staged by the runtime itself.
constexpr size_t XF_STAGE_MIN = (size_t)64 << 10;
// Caller memory is only REGISTERED when it is a mapping of its own: glibc never serves more than 32 MiB
// (DEFAULT_MMAP_THRESHOLD_MAX) from the shared brk heap, and its threshold climbs to that value as large blocks are freed —
// a 16-32 MiB vector can therefore sit in the heap, whose pages are trimmed and re-grown behind the runtime's back.
// Round 3's longer fuzz loops (more and larger host arrays per case) ended twice in "Memory access fault by GPU" at a
// brk-heap address, some calls after such a vector had been registered and unregistered; every operation alone ran clean
// for 10-20 k cases.  Buffers of 16-32 MiB take the staged pipeline (pinned slots + host copy team) instead.
constexpr size_t XF_REG_MIN = ((size_t)32 << 20) + 4096;
// downloads from this size on are touched, registered and copied piece by piece (pieces of >= XF_PIECE: each well above
// XF_REG_MIN, inside one allocation that is a mapping of its own)
constexpr size_t XF_PIECE = (size_t)64 << 20, XF_PIECES_MIN = (size_t)192 << 20;
constexpr int XF_MAX_PIECES = 16;

int xfer_mode()
{
    static const int m = [] { const char *e = getenv("MXGPU_XFER"); return e ? atoi(e) : 1; }();
    return m;                                          // 0 plain hipMemcpy, 1 register + direct DMA (default), 2 staged pipeline
}

struct Engine {
    std::mutex mu;                                    // one staged transfer at a time (the exports are synchronous anyway)
    Pool *pool = nullptr;
    void *slot[XF_SLOTS] = {};
    hipEvent_t ev[XF_SLOTS] = {};
    hipStream_t st = nullptr;
    int dev = -1;
    bool ok = false;

    Pool *team()
    {
        if (!pool) {
            unsigned hw = std::thread::hardware_concurrency();
            int n = hw >= 32 ? 16 : (hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1));   // (the GPU boxes grant 16 CPUs' worth of time)
            if (const char *e = getenv("MXGPU_COPY_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = v; }
            pool = new Pool(n);
        }
        return pool;
    }
    bool init()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return false;
        if (ok && d == dev) return true;
        release_device();
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return false; }
        for (int i = 0; i < XF_SLOTS; i++) {
            if (hipHostMalloc(&slot[i], XF_CHUNK, hipHostMallocDefault) != hipSuccess) { slot[i] = nullptr; release_device(); return false; }
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { ev[i] = nullptr; release_device(); return false; }
        }
        team();
        dev = d;
        ok = true;
        return true;
    }
    void release_device()
    {
        for (int i = 0; i < XF_SLOTS; i++) {
            if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
            if (slot[i]) { (void)hipHostFree(slot[i]); slot[i] = nullptr; }
        }
        if (st) { (void)hipStreamDestroy(st); st = nullptr; }
        ok = false;
    }
};

Engine &engine()
{
    static Engine *e = new Engine();                  // intentionally leaked: no HIP calls from static destructors
    return *e;
}

// (smaller chunks for mid-sized transfers — bytes / 4 or / 8, so that DMA and host copy overlap there too — were measured and
// are slower, also with a team that polls before it sleeps: what a chunk costs is its copy call, event and wait, ~30 us;
// 8 MiB down: 0.26 ms in one chunk, 0.39 ms in eight; a plain hipMemcpy — the runtime pins the caller's pages — 0.16 ms)
inline size_t chunk_for(size_t) { return XF_CHUNK; }

int staged_d2h(Engine &e, void *dst_host, const void *src_dev, size_t bytes)
{
    MX_HIP(hipStreamSynchronize(nullptr));
    const size_t CH = chunk_for(bytes);
    const size_t nch = (bytes + CH - 1) / CH;
    auto len_of = [&](size_t c) { return c + 1 < nch ? CH : bytes - c * CH; };
    auto issue = [&](size_t c) -> int {
        const int s = (int)(c % XF_SLOTS);
        MX_HIP(hipMemcpyAsync(e.slot[s], (const char *)src_dev + c * CH, len_of(c), hipMemcpyDeviceToHost, e.st));
        MX_HIP(hipEventRecord(e.ev[s], e.st));
        return 0;
    };
    for (size_t c = 0; c < nch && c < (size_t)XF_SLOTS - 1; c++)
        if (issue(c)) return 1;
    for (size_t c = 0; c < nch; c++) {
        if (c + XF_SLOTS - 1 < nch && issue(c + XF_SLOTS - 1)) return 1;      // its slot was drained one iteration ago
        MX_HIP(hipEventSynchronize(e.ev[c % XF_SLOTS]));
        e.pool->copy((char *)dst_host + c * CH, e.slot[c % XF_SLOTS], len_of(c));
    }
    return 0;
}

int staged_h2d(Engine &e, void *dst_dev, const void *src_host, size_t bytes)
{
    const size_t CH = chunk_for(bytes);
    const size_t nch = (bytes + CH - 1) / CH;
    for (size_t c = 0; c < nch; c++) {
        const int s = (int)(c % XF_SLOTS);
        const size_t len = c + 1 < nch ? CH : bytes - c * CH;
        if (c >= (size_t)XF_SLOTS) MX_HIP(hipEventSynchronize(e.ev[s]));      // the DMA out of this slot has finished
        e.pool->copy(e.slot[s], (const char *)src_host + c * CH, len);
        MX_HIP(hipMemcpyAsync((char *)dst_dev + c * CH, e.slot[s], len, hipMemcpyHostToDevice, e.st));
        MX_HIP(hipEventRecord(e.ev[s], e.st));
    }
    MX_HIP(hipStreamSynchronize(e.st));
    return 0;
}

}  // namespace

// ---- caller memory: first-touch, registration ----------------------------------------------------------------
// Starts write-touching every page of [p, p + bytes) on the host team and returns; prefault_wait() joins.  For the
// destination of a large D2H copy that the caller has just allocated: call it as early as the address is known.
// The destination of a large download is memory the caller has only just allocated.  R (and any plain malloc) gets it
// from mmap without huge-page advice, and on a machine whose THP mode is "madvise" (the MI355X boxes) that means 4-KiB
// pages: first-touching 1 GB of them from 16 threads took 313 ms (mmap-lock contention), registering them 37 ms more —
// the cold cfg2 call cost 370-600 ms instead of 28 (tools/cold_export_probe.py malloc).  So the whole 2-MiB-aligned
// interior is advised MADV_HUGEPAGE before the first touch: 28 ms for malloc'ed results as well.  Advice only: contents,
// protection and ownership are untouched.  numpy already advises its large arrays; there the extra call costs 0-3 ms.
// MXGPU_HUGEPAGE=0 switches it off.
static void advise_huge(void *p, size_t bytes)
{
    static const int huge = [] { const char *e = getenv("MXGPU_HUGEPAGE"); return e ? atoi(e) : 1; }();
    if (!huge) return;
    const uintptr_t two = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)p + two - 1) & ~(two - 1), b = ((uintptr_t)p + bytes) & ~(two - 1);
    if (b > a) (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
}

void prefault_begin(void *p, size_t bytes)
{
    if (!p || bytes < XF_MIN || xfer_mode() != 1) return;
    advise_huge(p, bytes);
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    e.team()->touch(p, bytes);
}
// The pieces of one buffer [base, base + total), touched one after the other without the caller in between: returns the
// team's size — piece g is touched when arrived[g] has reached it — or 0 when nothing was started (small buffer, another
// transfer mode: the pieces then count as touched).  prefault_wait() joins; `arrived` must live until then.
int prefault_begin_pieces(void *base, size_t total, const std::vector<std::pair<void *, size_t>> &pieces, std::atomic<int> *arrived)
{
    if (!base || total < XF_MIN || xfer_mode() != 1 || pieces.empty()) return 0;
    advise_huge(base, total);
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    Pool *t = e.team();
    t->touch_pieces(pieces, arrived);
    return t->threads();
}
// content hash of caller memory on the host team (the CSR cache's fingerprint)
uint64_t host_hash(const void *p, size_t bytes, uint64_t seed)
{
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    return e.team()->hash(p, bytes, seed);
}
void prefault_wait()
{
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    if (e.pool) e.pool->wait();
}
// hipHostRegister / hipHostUnregister of caller memory; false (and no error state left behind) when it cannot be pinned
bool pin_host(const void *p, size_t bytes, bool all_devices = false)
{
    if (!p || bytes == 0 || xfer_mode() != 1) return false;
    if (hipHostRegister(const_cast<void *>(p), bytes, all_devices ? hipHostRegisterPortable : hipHostRegisterDefault) == hipSuccess)
        return true;
    (void)hipGetLastError();
    return false;
}
void unpin_host(const void *p)
{
    if (p && hipHostUnregister(const_cast<void *>(p)) != hipSuccess) (void)hipGetLastError();
}

// Register only the whole pages INSIDE [p, p + bytes): the first and last partial page may be shared with a neighbouring
// heap object (another operand of the same call, registered and unregistered on its own schedule), and a page that is
// pinned and unpinned through two overlapping registrations has been seen to leave the GPU with a stale mapping (a GPU
// memory-access fault some calls later).  Returns the registered interior; the fragments are copied separately.
struct Interior { char *p = nullptr; size_t bytes = 0; };
static Interior pin_interior(const void *ptr, size_t bytes)
{
    Interior in;
    if (bytes < XF_REG_MIN) return in;                             // possibly a piece of the brk heap: never registered
    const uintptr_t a = ((uintptr_t)ptr + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)ptr + bytes) & ~(uintptr_t)4095;
    if (b <= a || b - a < XF_MIN / 2) return in;
    if (!pin_host((const void *)a, b - a)) return in;
    in.p = (char *)a;
    in.bytes = b - a;
    return in;
}

// Everything previously enqueued on the null stream (kernels of the export that produced `src`) is complete before
// the first byte moves, as with hipMemcpy.
int xfer_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return 0;
    if (bytes < XF_STAGE_MIN || xfer_mode() == 0) { MX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); return 0; }
    Engine &e = engine();
    if (bytes < XF_MIN) {
        std::lock_guard<std::mutex> lk(e.mu);
        if (!e.init()) { MX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); return 0; }
        return staged_d2h(e, dst_host, src_dev, bytes);
    }
    if (xfer_mode() == 1 && bytes >= XF_PIECES_MIN) {
        // Large destinations piece by piece: the team walks through the pieces on its own (touch_pieces), this thread
        // registers piece g when the last worker is through it and queues its DMA — the copy of piece g runs under the
        // first touch of the pieces behind it (1 GB: 7.5 ms touch + 2 ms registration + 18 ms DMA in series before).
        // Pieces are cut at 2-MiB marks of the interior (whole huge pages); the edge fragments go by plain copies.
        const uintptr_t a = ((uintptr_t)dst_host + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)dst_host + bytes) & ~(uintptr_t)4095;
        const int np = (int)std::min<size_t>(XF_MAX_PIECES, (b - a) / XF_PIECE);
        std::vector<std::pair<void *, size_t>> pieces;
        for (int g = 0; g < np; g++) {
            const uintptr_t two = (uintptr_t)2 << 20;
            const uintptr_t lo = g == 0 ? a : (a + (b - a) / (uintptr_t)np * (uintptr_t)g) & ~(two - 1);
            const uintptr_t hi = g + 1 == np ? b : (a + (b - a) / (uintptr_t)np * (uintptr_t)(g + 1)) & ~(two - 1);
            pieces.emplace_back((void *)lo, (size_t)(hi - lo));
        }
        std::atomic<int> arrived[XF_MAX_PIECES];
        for (auto &c : arrived) c.store(0);
        int team = 0;
        {
            advise_huge(dst_host, bytes);
            std::lock_guard<std::mutex> lk(e.mu);
            Pool *t = e.team();
            t->touch_pieces(pieces, arrived);
            team = t->threads();
        }
        int pinned = 0;
        hipError_t rc = hipSuccess;
        for (int g = 0; g < np && rc == hipSuccess; g++) {
            while (arrived[g].load(std::memory_order_acquire) < team) std::this_thread::yield();
            if (!pin_host(pieces[g].first, pieces[g].second)) break;
            pinned++;
            rc = hipMemcpyAsync(pieces[g].first, (const char *)src_dev + ((uintptr_t)pieces[g].first - (uintptr_t)dst_host),
                                pieces[g].second, hipMemcpyDeviceToHost, nullptr);
        }
        const hipError_t rs = hipStreamSynchronize(nullptr);       // nothing may touch the pages once they are unpinned
        for (int g = 0; g < pinned; g++) unpin_host(pieces[g].first);
        {
            std::lock_guard<std::mutex> lk(e.mu);
            e.pool->wait();                                         // (the counters above live on this stack)
        }
        MX_HIP(rc);
        MX_HIP(rs);
        if (pinned == np) {
            const size_t head = (size_t)(a - (uintptr_t)dst_host), tail = bytes - head - (size_t)(b - a);
            if (head) MX_HIP(hipMemcpy(dst_host, src_dev, head, hipMemcpyDeviceToHost));
            if (tail) MX_HIP(hipMemcpy((char *)dst_host + head + (b - a), (const char *)src_dev + head + (b - a), tail, hipMemcpyDeviceToHost));
            return 0;
        }
        // a piece could not be registered: everything once more through the pinned slots below
    } else if (xfer_mode() == 1) {
        {
            advise_huge(dst_host, bytes);
            std::lock_guard<std::mutex> lk(e.mu);
            e.team()->touch(dst_host, bytes);          // (a no-op pass when prefault_begin already did it)
            e.pool->wait();
        }
        const Interior in = pin_interior(dst_host, bytes);
        if (in.p) {
            const size_t head = (size_t)(in.p - (char *)dst_host), tail = bytes - head - in.bytes;
            hipError_t rc = hipMemcpyAsync(in.p, (const char *)src_dev + head, in.bytes, hipMemcpyDeviceToHost, nullptr);
            const hipError_t rs = hipStreamSynchronize(nullptr);   // nothing may touch the pages once they are unpinned
            if (rc == hipSuccess) rc = rs;
            unpin_host(in.p);
            MX_HIP(rc);
            if (head) MX_HIP(hipMemcpy(dst_host, src_dev, head, hipMemcpyDeviceToHost));
            if (tail) MX_HIP(hipMemcpy((char *)dst_host + head + in.bytes, (const char *)src_dev + head + in.bytes, tail, hipMemcpyDeviceToHost));
            return 0;
        }
    }
    std::lock_guard<std::mutex> lk(e.mu);
    if (!e.init()) { MX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); return 0; }
    return staged_d2h(e, dst_host, src_dev, bytes);
}

int xfer_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return 0;
    if (bytes < XF_STAGE_MIN || xfer_mode() == 0) { MX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice)); return 0; }
    if (bytes < XF_MIN) {
        Engine &e = engine();
        std::lock_guard<std::mutex> lk(e.mu);
        if (!e.init()) { MX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice)); return 0; }
        return staged_h2d(e, dst_dev, src_host, bytes);
    }
    if (xfer_mode() == 1) {   // (piece by piece like the downloads: measured, no gain — registering touched memory is ~0.3 ms per GB)
        const Interior in = pin_interior(src_host, bytes);
        if (in.p) {
            const size_t head = (size_t)(in.p - (const char *)src_host), tail = bytes - head - in.bytes;
            hipError_t rc = hipMemcpyAsync((char *)dst_dev + head, in.p, in.bytes, hipMemcpyHostToDevice, nullptr);
            const hipError_t rs = hipStreamSynchronize(nullptr);   // nothing may touch the pages once they are unpinned
            if (rc == hipSuccess) rc = rs;
            unpin_host(in.p);
            MX_HIP(rc);
            if (head) MX_HIP(hipMemcpy(dst_dev, src_host, head, hipMemcpyHostToDevice));
            if (tail) MX_HIP(hipMemcpy((char *)dst_dev + head + in.bytes, (const char *)src_host + head + in.bytes, tail, hipMemcpyHostToDevice));
            return 0;
        }
    }
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    if (!e.init()) { MX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice)); return 0; }
    return staged_h2d(e, dst_dev, src_host, bytes);
}

}  // namespace mx
