// host_pool.h — the team of host threads behind the transfer engine (xfer.hip): parallel memcpy between pinned slots
// and caller memory, and the parallel first-touch of freshly allocated result pages.  Plain C++ (no HIP): also built
// stand-alone under ThreadSanitizer / AddressSanitizer by tools/sanitize.sh.
#pragma once
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
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
