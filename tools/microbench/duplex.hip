// PCIe both ways at once: H2D alone, D2H alone, both together on two streams (pinned host memory, 2 GB each way).
// hipcc --offload-arch=gfx950 -O2 tools/microbench/duplex.hip -o /tmp/duplex && /tmp/duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const size_t n = (size_t)2 << 30;
    void *hu, *hd, *du, *dd;
    CK(hipHostMalloc(&hu, n, hipHostMallocDefault));
    CK(hipHostMalloc(&hd, n, hipHostMallocDefault));
    memset(hu, 1, n); memset(hd, 2, n);
    CK(hipMalloc(&du, n)); CK(hipMalloc(&dd, n));
    CK(hipMemset(dd, 3, n));
    hipStream_t up, down;
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    auto run = [&](bool u, bool d, int chunks) -> double {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        const size_t c = n / chunks;
        for (int i = 0; i < chunks; i++) {
            if (u) hipMemcpyAsync((char *)du + i * c, (char *)hu + i * c, c, hipMemcpyHostToDevice, up);
            if (d) hipMemcpyAsync((char *)hd + i * c, (char *)dd + i * c, c, hipMemcpyDeviceToHost, down);
        }
        hipStreamSynchronize(up); hipStreamSynchronize(down);
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    run(true, true, 4);
    for (int chunks : {1, 16}) {
        const double tu = run(true, false, chunks), td = run(false, true, chunks), tb = run(true, true, chunks);
        printf("chunks %2d: H2D %.1f GB/s  D2H %.1f GB/s  both: %.1f ms for 2+2 GB = %.1f GB/s each way (alone: %.1f + %.1f ms)\n", chunks,
               n / tu / 1e9, n / td / 1e9, tb * 1e3, n / tb / 1e9, tu * 1e3, td * 1e3);
    }
    return 0;
}
