// spmm_gather_ceiling.hip — the B-row gather of CSR x dense SpMM with nothing around it.
// For every nonzero of a synthetic CSR (m rows, k entries per row, uniform random columns in [0, K)) one wavefront loads
// the n-column row B[j, :] (n = 128 doubles = 1 KiB = one 16-byte load per lane), 8 loads in flight, and ORs it into
// registers: no FMA, no LDS, no (j, a) stream to speak of (the column ids are generated in registers), one store per
// wavefront at the end.  Run with K = 100000 (B = 102 MB: the headline operand, served from the Infinity Cache / L2) and
// K = 4096 (B = 4 MB: everything L2-resident): what remains is the L2 -> L1 line traffic that any SpMM kernel has to move
// (nnz * n * 8 bytes), i.e. the gather ceiling the planned kernel is compared with in DESIGN.md §4.1.
// Build: make -C tools/microbench ; run: tools/microbench/build/spmm_gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// U loads in flight per wavefront; BLOCK threads per workgroup (1024 with one workgroup per CU = the planned kernel's 16
// wavefronts per CU)
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gather(long long nrows, int k, unsigned K, const double *__restrict__ B, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * BLOCK) >> 6;
    d2 acc = {0.0, 0.0};
    for (long long row = wave; row < nrows; row += nw) {
        for (int e0 = 0; e0 < k; e0 += U) {
            d2 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned j = mix((unsigned)(row * 64 + e0 + u)) % K;               // wave-uniform column id
                v[u] = *reinterpret_cast<const d2 *>(B + (size_t)j * 128 + lane * 2);
            }
#pragma unroll
            for (int u = 0; u < U; u++) { acc[0] += v[u][0]; acc[1] += v[u][1]; }
        }
    }
    if (acc[0] + acc[1] == 123.456) out[wave] = acc[0];                                  // keeps the loads alive, never true
}

int main()
{
    const long long m = 1000000; const int k = 32;
    double *B, *out;
    CK(hipMalloc(&B, (size_t)100000 * 128 * 8)); CK(hipMemset(B, 0, (size_t)100000 * 128 * 8)); CK(hipMalloc(&out, 1 << 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"shape\": \"1M rows x 32 entries, rows of B = 128 f64 (1 KiB)\", \"line_bytes_moved\": %lld, \"results\": {", m * k * 1024LL);
    const unsigned Ks[2] = {100000u, 4096u};
    const char *cfgs[4] = {"32 waves/CU x 8 loads", "16 waves/CU x 8 loads (the planned kernel's concurrency)", "16 waves/CU x 16 loads", "16 waves/CU x 4 loads"};
    bool first = true;
    for (int t = 0; t < 2; t++)
        for (int c = 0; c < 4; c++) {
            float best = 1e30f;
            for (int rep = 0; rep < 8; rep++) {
                CK(hipEventRecord(e0));
                switch (c) {
                    case 0: hipLaunchKernelGGL((gather<8, 256>), dim3(256 * 8), dim3(256), 0, 0, m, k, Ks[t], B, out); break;
                    case 1: hipLaunchKernelGGL((gather<8, 1024>), dim3(256), dim3(1024), 0, 0, m, k, Ks[t], B, out); break;
                    case 2: hipLaunchKernelGGL((gather<16, 1024>), dim3(256), dim3(1024), 0, 0, m, k, Ks[t], B, out); break;
                    default: hipLaunchKernelGGL((gather<4, 1024>), dim3(256), dim3(1024), 0, 0, m, k, Ks[t], B, out); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("%s\"B_%s, %s\": {\"ms\": %.3f, \"l2_to_l1_TBps\": %.1f}", first ? "" : ", ", t ? "4MB" : "102MB", cfgs[c], best,
                   m * k * 1024.0 / best / 1e9);
            first = false;
        }
    printf("}}\n");
    return 0;
}
