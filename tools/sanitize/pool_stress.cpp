// Stress of the host thread team (matrixextra_amd/csrc/host_pool.h) for the sanitizer builds of tools/sanitize.sh:
// interleaved asynchronous first-touch jobs, parallel copies of odd sizes and waits, checked byte for byte.
#include "../../matrixextra_amd/csrc/host_pool.h"
#include <cstdio>
#include <cstdlib>
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
