// stream.hip — the on-box STREAM-copy probe: one float4 (16 B per lane) copy kernel, what "achievable HBM bandwidth" means
// on the GPU a benchmark actually runs on (report the nominal-8-TB/s fraction AND the STREAM fraction;
// MI355X_MICROARCH.md measures 6.29 TB/s for this kernel shape).  bench.py times it in the same run as the kernels it
// prices.
#include "mx_common.h"

namespace mx {

typedef float f4 __attribute__((ext_vector_type(4)));
// grid-stride over 16-byte pieces, 4 loads in flight per lane; nontemporal stores (the destination is not read again)
__global__ __launch_bounds__(256)
void stream_copy_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, long long n16)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        __builtin_nontemporal_store(a, &dst[i]);
        __builtin_nontemporal_store(b, &dst[i + stride]);
        __builtin_nontemporal_store(c, &dst[i + 2 * stride]);
        __builtin_nontemporal_store(d, &dst[i + 3 * stride]);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(src[i], &dst[i]);
}

}  // namespace mx

extern "C" int mxd_stream_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    MX_REQUIRE(dst && src, "mxd_stream_copy: null pointer");
    MX_REQUIRE(((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0 && (bytes & 15) == 0,
               "mxd_stream_copy: pointers and size must be multiples of 16 bytes");
    if (bytes == 0) return 0;
    const long long n16 = (long long)(bytes >> 4);
    const long long want = mx::ceil_div(n16, 256 * 4);
    const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);      // <= 32 workgroups per CU
    hipLaunchKernelGGL(mx::stream_copy_kernel, dim3(grid), dim3(256), 0, mx::as_stream(stream), (const mx::f4 *)src,
                       (mx::f4 *)dst, n16);
    MX_LAUNCH_CHECK();
    return 0;
}
