// stream.hip — the on-box STREAM-copy probe: one float4 (16 B per lane) copy kernel, what "achievable HBM bandwidth" means
// on the GPU a benchmark actually runs on (report the nominal-8-TB/s fraction AND the STREAM fraction;
// MI355X_MICROARCH.md measures 6.29 TB/s for this kernel shape).  bench.py times it in the same run as the kernels it
// prices.
#include "mx_common.h"

namespace mx {

typedef float f4 __attribute__((ext_vector_type(4)));
// One contiguous chunk per workgroup, 8 loads in flight per lane, nontemporal stores (the destination is not read again).
// Calibrated with tools/microbench/stream_copy.hip on the MI355X boxes of this pool (1 GiB / 4 GiB copies): contiguous
// chunks 5.6-5.7 TB/s at 4096 workgroups, grid-stride loops of the same loads 4.5-5.0, the runtime's own hipMemcpyAsync
// D2D 5.1-5.5 (the guide's figure for a float4 copy is 6.29 TB/s).
constexpr int SC_U = 8;
__global__ __launch_bounds__(256)
void stream_copy_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, long long n16)
{
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long long b = (long long)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
    long long i = b + threadIdx.x;
    for (; i + (SC_U - 1) * 256 < e; i += SC_U * 256) {
        f4 v[SC_U];
#pragma unroll
        for (int u = 0; u < SC_U; u++) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < SC_U; u++) __builtin_nontemporal_store(v[u], &dst[i + u * 256]);
    }
    for (; i < e; i += 256) __builtin_nontemporal_store(src[i], &dst[i]);
}

}  // namespace mx

extern "C" int mxd_stream_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    MX_REQUIRE(dst && src, "mxd_stream_copy: null pointer");
    MX_REQUIRE(((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0 && (bytes & 15) == 0,
               "mxd_stream_copy: pointers and size must be multiples of 16 bytes");
    if (bytes == 0) return 0;
    const long long n16 = (long long)(bytes >> 4);
    const long long want = mx::ceil_div(n16, 256 * mx::SC_U);
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);              // 16 workgroups per CU
    hipLaunchKernelGGL(mx::stream_copy_kernel, dim3(grid), dim3(256), 0, mx::as_stream(stream), (const mx::f4 *)src,
                       (mx::f4 *)dst, n16);
    MX_LAUNCH_CHECK();
    return 0;
}
