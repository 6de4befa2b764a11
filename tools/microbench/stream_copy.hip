// stream_copy.hip — which float4 copy shape reaches the box's HBM rate (MI355X_MICROARCH.md: 6.29 TB/s measured for a float4
// copy)?  Variants: loads in flight per lane (U), plain / nontemporal stores, plain / nontemporal loads, grid size.  Prints GB/s of
// read + write bytes for a 1 GiB and a 4 GiB copy.  Calibration for csrc/stream.hip (bench.py's stream_copy probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT_ST, bool NT_LD>
__global__ __launch_bounds__(256) void copy_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, long long n16)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT_LD ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT_ST) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// contiguous chunk per workgroup instead of a grid stride
template <int U, bool NT_ST>
__global__ __launch_bounds__(256) void copy_chunk_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, long long n16)
{
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long long b = (long long)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
    long long i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT_ST) __builtin_nontemporal_store(v[u], &dst[i + u * 256]); else dst[i + u * 256] = v[u]; }
    }
    for (; i < e; i += 256) dst[i] = src[i];
}

template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    for (size_t bytes : {(size_t)1 << 30, (size_t)4 << 30}) {
        f4 *a, *b;
        hipMalloc(&a, bytes); hipMalloc(&b, bytes);
        hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
        const long long n16 = (long long)(bytes >> 4);
        printf("copy of %zu MiB (GB/s of read + write bytes)\n", bytes >> 20);
        for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
#define RUN(NAME, K)                                                                                  \
            { const double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 10); \
              printf("  grid %6d  %-28s %8.1f\n", grid, NAME, 2.0 * bytes / ms / 1e6); }
            RUN("U4 plain", (copy_kernel<4, false, false>))
            RUN("U4 nt-store", (copy_kernel<4, true, false>))
            RUN("U8 plain", (copy_kernel<8, false, false>))
            RUN("U8 nt-store", (copy_kernel<8, true, false>))
            RUN("U8 nt-store nt-load", (copy_kernel<8, true, true>))
            RUN("U2 plain", (copy_kernel<2, false, false>))
            RUN("chunk U4 plain", (copy_chunk_kernel<4, false>))
            RUN("chunk U8 nt-store", (copy_chunk_kernel<8, true>))
#undef RUN
        }
        {   // the runtime's own device-to-device copy
            const double ms = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
            printf("  hipMemcpyAsync D2D              %8.1f\n", 2.0 * bytes / ms / 1e6);
        }
        hipFree(a); hipFree(b);
    }
    return 0;
}
