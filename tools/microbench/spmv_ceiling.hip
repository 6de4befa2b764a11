// spmv_ceiling.hip — what bounds a one-shot SpMV on MI355X: the (j, a) stream or the v[j] gather?
// Four kernels over the SAME synthetic CSR shape (m rows x K columns, k entries per row, uniform random columns),
// all with 16 B per lane loads of the index stream and 64 consecutive entries per wavefront:
//   stream    read indices + values, no gather                       -> the HBM streaming floor of the (j, a) arrays
//   gather    read indices, gather v[j] (8 B) from the K-vector      -> the L2 -> L1 line traffic of the gather alone
//   gather_l1 as gather, but j & 1023 (an 8 KB window: L1-resident)  -> the same instructions without the L2 traffic
//   both      indices + values + gather (an SpMV without the row reduction)
// Build: make -C tools/microbench ; run on the GPU box: tools/microbench/build/spmv_ceiling [m K k]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef int i4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0 stream, 1 gather, 2 gather_l1, 3 both
__global__ __launch_bounds__(256) void k(long long nnz, const int *__restrict__ j, const double *__restrict__ a,
                                         const double *__restrict__ v, double *__restrict__ out)
{
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e + 4 > nnz) return;
    const i4 c = *reinterpret_cast<const i4 *>(j + e);
    double s = 0.0;
    if (MODE == 0 || MODE == 3) {
        const d2 a01 = *reinterpret_cast<const d2 *>(a + e), a23 = *reinterpret_cast<const d2 *>(a + e + 2);
        if (MODE == 0) s = a01[0] + a01[1] + a23[0] + a23[1] + (double)(c[0] ^ c[1] ^ c[2] ^ c[3]);
        else s = a01[0] * v[c[0]] + a01[1] * v[c[1]] + a23[0] * v[c[2]] + a23[1] * v[c[3]];
    } else if (MODE == 1) {
        s = v[c[0]] + v[c[1]] + v[c[2]] + v[c[3]];
    } else {
        s = v[c[0] & 1023] + v[c[1] & 1023] + v[c[2] & 1023] + v[c[3] & 1023];
    }
    // one store per wavefront keeps the loads alive without a write stream of its own
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) out[((long long)blockIdx.x * 256 + threadIdx.x) >> 6] = s;
}

int main(int argc, char **argv)
{
    const long long m = argc > 1 ? atoll(argv[1]) : 1000000, K = argc > 2 ? atoll(argv[2]) : 100000, kk = argc > 3 ? atoll(argv[3]) : 32;
    const long long nnz = m * kk;
    std::vector<int> hj(nnz);
    std::mt19937_64 rng(1);
    for (long long i = 0; i < nnz; i++) hj[i] = (int)(rng() % K);
    int *j; double *a, *v, *out;
    CK(hipMalloc(&j, nnz * 4)); CK(hipMalloc(&a, nnz * 8)); CK(hipMalloc(&v, K * 8)); CK(hipMalloc(&out, nnz / 64 * 8 + 64));
    CK(hipMemcpy(j, hj.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemset(a, 0, nnz * 8)); CK(hipMemset(v, 0, K * 8));
    const unsigned grid = (unsigned)((nnz / 4 + 255) / 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"stream", "gather", "gather_l1", "both"};
    printf("{\"shape\": \"%lld x %lld, %lld/row, nnz %lld\", \"results\": {", m, K, kk, nnz);
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 12; rep++) {
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, nnz, j, a, v, out); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, nnz, j, a, v, out); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, nnz, j, a, v, out); break;
                default: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, nnz, j, a, v, out); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
        }
        const double stream_bytes = (mode == 0 || mode == 3) ? 12.0 * nnz : 4.0 * nnz;
        const double lines = (mode == 1 || mode == 3) ? (double)nnz : 0.0;
        printf("%s\"%s\": {\"us\": %.1f, \"stream_GBps\": %.0f, \"gathered_lines_per_s_G\": %.1f, \"gather_line_traffic_GBps\": %.0f}",
               mode ? ", " : "", names[mode], best * 1e3, stream_bytes / best / 1e6, lines / best / 1e6, lines * 128 / best / 1e6);
    }
    printf("}}\n");
    return 0;
}
