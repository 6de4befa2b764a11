// Host <-> device transfers for the export-level calls (the R boundary hands over ordinary pageable vectors and
// expects ordinary vectors back: INTEGRATION.md §5).
//
// hipMemcpy on pageable memory stages through the runtime's own bounce buffer with one CPU thread doing the
// host-side copy, and the result matrix R (or numpy) just allocated has never been touched, so that thread also
// takes a page fault per 4 KiB: 1 GB of C came back at ~10-14 GB/s.  Here large transfers run as a pipeline over
// three pinned 8 MiB slots: the DMA engine moves slot k+1 while a small pool of host threads copies slot k to /
// from the caller's buffer (page faults and memcpy spread over the pool).  Synchronous at return, like hipMemcpy.
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "mx_common.h"

namespace mx {

namespace {

class CopyPool {
public:
    explicit CopyPool(int n) : n_(n)
    {
        for (int i = 0; i < n_; i++) th_.emplace_back([this, i] { loop(i); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // dst[0..n) = src[0..n), split over the pool in page-aligned pieces; returns when all pieces are done
    void copy(void *dst, const void *src, size_t n)
    {
        if (n < ((size_t)1 << 20) || n_ <= 1) { memcpy(dst, src, n); return; }
        std::unique_lock<std::mutex> lk(mu_);
        dst_ = (char *)dst; src_ = (const char *)src; bytes_ = n;
        piece_ = ((n + n_ - 1) / n_ + 4095) & ~(size_t)4095;
        pending_ = n_;
        gen_++;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
    }
    int threads() const { return n_; }

private:
    void loop(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            char *d; const char *s; size_t off, len;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                off = piece_ * (size_t)id;
                len = off < bytes_ ? (bytes_ - off < piece_ ? bytes_ - off : piece_) : 0;
                d = dst_ + off; s = src_ + off;
            }
            if (len) memcpy(d, s, len);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    bool stop_ = false;
    unsigned long gen_ = 0;
    int pending_ = 0;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0, piece_ = 0;
};

constexpr int XF_SLOTS = 3;
constexpr size_t XF_CHUNK = (size_t)8 << 20;
constexpr size_t XF_MIN = (size_t)16 << 20;           // below this a plain hipMemcpy is as good

struct Engine {
    std::mutex mu;                                    // one transfer at a time (the exports are synchronous anyway)
    CopyPool *pool = nullptr;
    void *slot[XF_SLOTS] = {};
    hipEvent_t ev[XF_SLOTS] = {};
    hipStream_t st = nullptr;
    int dev = -1;
    bool ok = false;

    bool init()
    {
        static const bool disabled = [] { const char *e = getenv("MXGPU_XFER"); return e && atoi(e) == 0; }();
        if (disabled) return false;                    // MXGPU_XFER=0: plain hipMemcpy (for comparison)
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return false;
        if (ok && d == dev) return true;
        release_device();
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return false; }
        for (int i = 0; i < XF_SLOTS; i++) {
            if (hipHostMalloc(&slot[i], XF_CHUNK, hipHostMallocDefault) != hipSuccess) { slot[i] = nullptr; release_device(); return false; }
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { ev[i] = nullptr; release_device(); return false; }
        }
        if (!pool) {
            unsigned hw = std::thread::hardware_concurrency();
            int n = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
            if (const char *e = getenv("MXGPU_COPY_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = v; }
            pool = new CopyPool(n);
        }
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

}  // namespace

// Everything previously enqueued on the null stream (kernels of the export that produced `src`) is complete before
// the first chunk moves: callers synchronise the null stream, as hipMemcpy would.
int xfer_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return 0;
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    if (bytes < XF_MIN || !e.init()) { MX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); return 0; }
    MX_HIP(hipStreamSynchronize(nullptr));
    const size_t nch = (bytes + XF_CHUNK - 1) / XF_CHUNK;
    auto len_of = [&](size_t c) { return c + 1 < nch ? XF_CHUNK : bytes - c * XF_CHUNK; };
    auto issue = [&](size_t c) -> int {
        const int s = (int)(c % XF_SLOTS);
        MX_HIP(hipMemcpyAsync(e.slot[s], (const char *)src_dev + c * XF_CHUNK, len_of(c), hipMemcpyDeviceToHost, e.st));
        MX_HIP(hipEventRecord(e.ev[s], e.st));
        return 0;
    };
    for (size_t c = 0; c < nch && c < (size_t)XF_SLOTS - 1; c++)
        if (issue(c)) return 1;
    for (size_t c = 0; c < nch; c++) {
        if (c + XF_SLOTS - 1 < nch && issue(c + XF_SLOTS - 1)) return 1;      // its slot was drained one iteration ago
        MX_HIP(hipEventSynchronize(e.ev[c % XF_SLOTS]));
        e.pool->copy((char *)dst_host + c * XF_CHUNK, e.slot[c % XF_SLOTS], len_of(c));
    }
    return 0;
}

int xfer_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return 0;
    Engine &e = engine();
    std::lock_guard<std::mutex> lk(e.mu);
    if (bytes < XF_MIN || !e.init()) { MX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice)); return 0; }
    const size_t nch = (bytes + XF_CHUNK - 1) / XF_CHUNK;
    for (size_t c = 0; c < nch; c++) {
        const int s = (int)(c % XF_SLOTS);
        const size_t len = c + 1 < nch ? XF_CHUNK : bytes - c * XF_CHUNK;
        if (c >= (size_t)XF_SLOTS) MX_HIP(hipEventSynchronize(e.ev[s]));      // the DMA out of this slot has finished
        e.pool->copy(e.slot[s], (const char *)src_host + c * XF_CHUNK, len);
        MX_HIP(hipMemcpyAsync((char *)dst_dev + c * XF_CHUNK, e.slot[s], len, hipMemcpyHostToDevice, e.st));
        MX_HIP(hipEventRecord(e.ev[s], e.st));
    }
    MX_HIP(hipStreamSynchronize(e.st));
    return 0;
}

}  // namespace mx
