// dpp_rate.hip — issue cost of the tile kernel's inner-loop instructions on gfx950: wave-cycles per instruction for
// v_fmac_f64_dpp (row_newbcast), plain v_fma_f64, v_add_u32_dpp, v_mov_b32_dpp, v_fmac_f32_dpp, ds_read_b128, at 1 .. 4
// wavefronts per SIMD.  One workgroup per CU; s_memtime around an unrolled loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(unsigned long long *out, int iters, double seed)
{
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63;
    double acc0 = seed * lane, acc1 = seed + lane, acc2 = seed - lane, acc3 = seed * 2 + lane, a = seed + 1.0, b = seed * 0.5;
    float f0 = (float)acc0, f1 = (float)acc1, f2 = (float)acc2, f3 = (float)acc3, fa = (float)a, fb = (float)b;
    unsigned x0 = lane, x1 = lane * 3, off = lane * 16;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0}, r3 = {0, 0};
    for (int i = threadIdx.x; i < 65536 / 8; i += blockDim.x) reinterpret_cast<double *>(smem)[i] = seed;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (KIND == 0) {          // 4 independent v_fmac_f64_dpp
                asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b));
            } else if constexpr (KIND == 1) {   // 4 independent v_fma_f64
                asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b));
            } else if constexpr (KIND == 2) {   // 4 v_add_u32_dpp
                asm volatile("v_add_u32_dpp %0, %2, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_u32_dpp %1, %2, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_u32_dpp %0, %2, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_u32_dpp %1, %2, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1) : "v"(off));
            } else if constexpr (KIND == 3) {   // 4 v_fmac_f32_dpp
                asm volatile("v_fmac_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));
            } else if constexpr (KIND == 4) {   // 4 ds_read_b128, conflict-free rows
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                             "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(off) : "memory");
            } else if constexpr (KIND == 5) {   // the tile kernel's step x 4: add_dpp, ds_read_b128, 2 fmac_dpp (reads 4 ahead)
                unsigned a0, a1, a2, a3;
                asm volatile("v_add_u32_dpp %4, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_u32_dpp %5, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_u32_dpp %6, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_u32_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(x0 & 0xFF00u), "v"(off) : "memory");
                asm volatile("s_waitcnt lgkmcnt(3)\n\t"
                             "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %2, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                             "s_waitcnt lgkmcnt(2)\n\t"
                             "v_fmac_f64_dpp %0, %2, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %2, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                             "s_waitcnt lgkmcnt(1)\n\t"
                             "v_fmac_f64_dpp %0, %2, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "v_fmac_f64_dpp %0, %2, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %2, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                             : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(r0.x), "v"(r0.y), "v"(r1.x), "v"(r1.y), "v"(r2.x), "v"(r2.y), "v"(r3.x), "v"(r3.y));
            } else if constexpr (KIND == 6) {   // broadcast by two v_mov_b32_dpp, then 2 plain FMAs (x2)
                double ab;
                asm volatile("v_mov_b64_dpp %0, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b64_dpp %0, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_fma_f64 %1, %0, %4, %1\n\tv_fma_f64 %2, %0, %4, %2"
                             : "=&v"(ab), "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b));
            } else if constexpr (KIND == 7) {   // 2 dependent v_fmac_f64_dpp chains of the kernel (acc0, acc1 only)
                asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %0, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                             : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (acc0 + acc1 + acc2 + acc3 + f0 + f1 + f2 + f3 + x0 + x1 + r0.x + r1.x + r2.x + r3.x == 12345.678) out[0] = 0;
}

template <int KIND> static void run(const char *name, int per_iter)
{
    unsigned long long *d;
    hipMalloc(&d, 256 * 16 * 8);
    const int iters = 2000;
    for (int waves : {1, 4, 8, 12, 16}) {
        hipMemset(d, 0, 256 * 16 * 8);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(waves * 64), 0, 0, d, iters, 1.0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(waves * 64), 0, 0, d, iters, 1.0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) worst = worst > (double)h[b * 16 + w] ? worst : (double)h[b * 16 + w];
        // s_memtime ticks at 100 MHz on gfx9; report ns per instruction per wave and per SIMD
        const double ns = worst * 10.0;
        printf("%-28s waves/CU %2d: %.2f ns per instruction per wave, %.3f ns per instruction per SIMD\n", name, waves,
               ns / (iters * 8.0 * per_iter), ns / (iters * 8.0 * per_iter) / ((waves + 3) / 4));
    }
    hipFree(d);
}

int main()
{
    run<1>("v_fma_f64", 4);
    run<0>("v_fmac_f64_dpp", 4);
    run<7>("v_fmac_f64_dpp 2 chains", 4);
    run<2>("v_add_u32_dpp", 4);
    run<3>("v_fmac_f32_dpp", 4);
    run<4>("ds_read_b128 + wait", 4);
    run<5>("tile step (1+1+2) x4", 16);
    run<6>("2 mov_b64_dpp + 2 fma_f64", 4);
    return 0;
}
