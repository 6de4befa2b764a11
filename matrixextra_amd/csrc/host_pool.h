// host_pool.h — the team of host threads behind the transfer engine (xfer.hip): parallel memcpy between pinned slots
// and caller memory, and the parallel first-touch of freshly allocated result pages.  Plain C++ (no HIP): also built
// stand-alone under ThreadSanitizer / AddressSanitizer by tools/sanitize.sh.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace mx {

// a fixed team of host threads; run(job) hands every worker (id, n) and returns at once, wait() joins the job
class HostPool {
public:
    explicit HostPool(int n) : n_(n)
    {
        for (int i = 0; i < n_; i++) th_.emplace_back([this, i] { loop(i); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(std::function<void(int, int)> job)
    {
        wait();
        std::lock_guard<std::mutex> lk(mu_);
        job_ = std::move(job);
        pending_ = n_;
        gen_++;
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }
    // dst[0..n) = src[0..n), split over the team in page-aligned pieces
    void copy(void *dst, const void *src, size_t n)
    {
        if (n < ((size_t)1 << 20) || n_ <= 1) { wait(); memcpy(dst, src, n); return; }
        const size_t piece = ((n + n_ - 1) / n_ + 4095) & ~(size_t)4095;
        run([=](int id, int) {
            const size_t off = piece * (size_t)id;
            if (off < n) memcpy((char *)dst + off, (const char *)src + off, n - off < piece ? n - off : piece);
        });
        wait();
    }
    // write-touch every page of [p, p + n) (contents kept): the first-touch faults of a fresh allocation, spread over
    // the team.  Returns at once.
    void touch(void *p, size_t n)
    {
        const size_t piece = ((n + n_ - 1) / n_ + 4095) & ~(size_t)4095;
        run([=](int id, int) {
            const size_t off = piece * (size_t)id;
            if (off >= n) return;
            const size_t len = n - off < piece ? n - off : piece;
            volatile char *q = (volatile char *)p + off;
            for (size_t i = 0; i < len; i += 4096) q[i] = q[i];
            q[len - 1] = q[len - 1];
        });
    }
    // The same over several pieces, one after the other, every worker walking through all of them without meeting the
    // others: arrived[g] counts the workers that are through piece g — the piece is touched when it reaches threads().
    // Returns at once; `arrived` must stay alive until wait().
    void touch_pieces(std::vector<std::pair<void *, size_t>> pieces, std::atomic<int> *arrived)
    {
        run([pieces, arrived](int id, int team) {
            for (size_t g = 0; g < pieces.size(); g++) {
                const size_t n = pieces[g].second;
                const size_t piece = ((n + (size_t)team - 1) / (size_t)team + 4095) & ~(size_t)4095;
                const size_t off = piece * (size_t)id;
                if (off < n) {
                    const size_t len = n - off < piece ? n - off : piece;
                    volatile char *q = (volatile char *)pieces[g].first + off;
                    for (size_t i = 0; i < len; i += 4096) q[i] = q[i];
                    q[len - 1] = q[len - 1];
                }
                arrived[g].fetch_add(1, std::memory_order_release);
            }
        });
    }
    // 64-bit content hash of [p, p + n): 1 MiB chunks over the team, four multiply-xor lanes of 64-bit words per chunk,
    // chunk hashes folded in order.  Every step h -> (h ^ w) * odd is a bijection in h, so two buffers that differ in
    // ONE word (an element overwritten in place) can never hash alike; unrelated contents collide with probability 2^-64.
    uint64_t hash(const void *p, size_t n, uint64_t seed)
    {
        constexpr size_t CH = (size_t)1 << 20;
        const size_t nch = (n + CH - 1) / CH;
        auto one = [=](size_t c) -> uint64_t {
            const unsigned char *q = (const unsigned char *)p + c * CH;
            const size_t len = c + 1 < nch ? CH : n - c * CH;
            const uint64_t P = 0x100000001b3ULL;
            uint64_t h0 = seed ^ (0x9e3779b97f4a7c15ULL * (c + 1)), h1 = h0 ^ 0xbf58476d1ce4e5b9ULL,
                     h2 = h0 ^ 0x94d049bb133111ebULL, h3 = h0 ^ 0xd6e8feb86659fd93ULL;
            size_t i = 0;
            for (; i + 32 <= len; i += 32) {
                uint64_t w[4];
                memcpy(w, q + i, 32);
                h0 = (h0 ^ w[0]) * P; h1 = (h1 ^ w[1]) * P; h2 = (h2 ^ w[2]) * P; h3 = (h3 ^ w[3]) * P;
            }
            for (; i < len; i++) h0 = (h0 ^ q[i]) * P;
            return (((h0 * P ^ h1) * P ^ h2) * P ^ h3) * P ^ (uint64_t)len;
        };
        std::vector<uint64_t> part(nch);
        if (nch <= 2 || n_ <= 1) {
            wait();
            for (size_t c = 0; c < nch; c++) part[c] = one(c);
        } else {
            uint64_t *out = part.data();
            const int team = n_;
            run([=](int id, int) { for (size_t c = (size_t)id; c < nch; c += (size_t)team) out[c] = one(c); });
            wait();
        }
        uint64_t h = seed ^ (uint64_t)n;
        for (size_t c = 0; c < nch; c++) h = (h ^ part[c]) * 0x100000001b3ULL;
        return h;
    }
    int threads() const { return n_; }

private:
    void loop(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            std::function<void(int, int)> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = job_;
            }
            job(id, n_);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_all();
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
    std::function<void(int, int)> job_;
};

}  // namespace mx
