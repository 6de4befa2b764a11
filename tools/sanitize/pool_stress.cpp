// Stress of the host thread team (matrixextra_amd/csrc/host_pool.h) for the sanitizer builds of tools/sanitize.sh:
// interleaved asynchronous first-touch jobs, parallel copies of odd sizes and waits, checked byte for byte.
#include "../../matrixextra_amd/csrc/host_pool.h"
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

int main()
{
    mx::HostPool pool(6);
    std::vector<unsigned char> src((size_t)37 << 20), dst(src.size() + 4096);
    for (size_t i = 0; i < src.size(); i++) src[i] = (unsigned char)(i * 2654435761u >> 24);
    for (int round = 0; round < 20; round++) {
        const size_t n = src.size() - (size_t)round * 123457, off = (size_t)round * 13;
        std::vector<unsigned char> fresh(n);                     // first-touch job running while the copy is prepared
        pool.touch(fresh.data(), n);
        pool.copy(dst.data() + off, src.data(), n);              // run() joins the touch job first
        pool.wait();
        if (memcmp(dst.data() + off, src.data(), n) != 0) { printf("copy mismatch in round %d\n", round); return 1; }
        pool.copy(fresh.data(), dst.data() + off, n);
        if (memcmp(fresh.data(), src.data(), n) != 0) { printf("second copy mismatch in round %d\n", round); return 1; }
        pool.touch(fresh.data(), n);                             // touching keeps the contents
        pool.wait();
        if (memcmp(fresh.data(), src.data(), n) != 0) { printf("touch changed the data in round %d\n", round); return 1; }
    }
    // pieces touched one after the other without the caller in between: the counters reach the team size in order,
    // a reader that has seen piece g complete may write to it while later pieces are still being touched
    for (int round = 0; round < 10; round++) {
        std::vector<unsigned char> buf(src.begin(), src.begin() + ((size_t)30 << 20) + 4096 * 3 + 17 * (size_t)round);
        std::vector<std::pair<void *, size_t>> pieces;
        const size_t np = 7, step = (buf.size() / np) & ~(size_t)4095;
        for (size_t g = 0; g < np; g++)
            pieces.emplace_back((void *)(buf.data() + g * step), g + 1 < np ? step : buf.size() - g * step);
        std::atomic<int> arrived[7];
        for (auto &a : arrived) a.store(0);
        pool.touch_pieces(pieces, arrived);
        for (size_t g = 0; g < np; g++) {
            while (arrived[g].load(std::memory_order_acquire) < pool.threads()) std::this_thread::yield();
            for (size_t h = 0; h < g; h++)
                if (arrived[h].load() != pool.threads()) { printf("piece %zu complete before piece %zu\n", g, h); return 1; }
            memset(pieces[g].first, 0xEE, pieces[g].second);      // (the consumer's write: the DMA in the real thing)
        }
        pool.wait();
        for (size_t i = 0; i < buf.size(); i++)
            if (buf[i] != 0xEE) { printf("touch_pieces wrote over the consumer at %zu\n", i); return 1; }
    }
    // content hash: the same for the same bytes whatever the team size, different after ONE byte changes anywhere
    {
        mx::HostPool one(1);
        for (size_t n : {(size_t)0, (size_t)5, (size_t)4096, ((size_t)1 << 20) + 3, src.size() - 77}) {
            const uint64_t h = pool.hash(src.data(), n, 42), h1 = one.hash(src.data(), n, 42);
            if (h != h1 || h != pool.hash(src.data(), n, 42)) { printf("hash not reproducible at n = %zu\n", n); return 1; }
            if (n == 0) continue;
            for (size_t at : {(size_t)0, n / 3, n - 1}) {
                src[at] ^= 1;
                const bool same = pool.hash(src.data(), n, 42) == h;
                src[at] ^= 1;
                if (same) { printf("hash blind to a change at %zu of %zu\n", at, n); return 1; }
            }
            if (pool.hash(src.data(), n, 43) == h) { printf("hash ignores its seed\n"); return 1; }
        }
    }
    pool.copy(dst.data(), src.data(), 1000);                     // below the parallel threshold: inline
    printf("pool stress ok\n");
    return 0;
}
