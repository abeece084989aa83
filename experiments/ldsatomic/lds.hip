// What a returning LDS atomic costs on gfx950 when a workgroup ranks keys in 256 bins (the ranking of the first and
// second partition level): 16 atomics per thread on random bins, 256 threads, three workgroups per CU.  Variants:
// shared bins / one set of bins per wave / non-returning.  Standalone experiment, not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lds lds.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>      // 0 returning shared, 1 returning per-wave bins, 2 non-returning shared, 3 no atomics (baseline)
__global__ __launch_bounds__(256, 3) void rank_kernel(uint32_t* out, uint32_t rounds)
{
    __shared__ uint32_t bins[4][288];
    __shared__ uint32_t pad[11000];          // brings the workgroup to ~52 KB of LDS: three per CU, as in the product
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    for (uint32_t i = tid; i < 4 * 288; i += 256) (&bins[0][0])[i] = 0;
    if (tid == 0) pad[0] = 0;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rounds; ++r)
    {
        uint32_t d[16], rk[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = mix((blockIdx.x * 256 + tid) * 16 + i + r * 0x9E3779B9u) & 255u;
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            if (MODE == 0) rk[i] = atomicAdd(&bins[0][d[i]], 1u);
            else if (MODE == 1) rk[i] = atomicAdd(&bins[w][d[i]], 1u);
            else if (MODE == 2) { atomicAdd(&bins[0][d[i]], 1u); rk[i] = d[i]; }
            else rk[i] = d[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += rk[i];
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc + pad[0];
}

int main()
{
    const uint32_t grid = 768, rounds = 4096;
    uint32_t* out; CHECK(hipMalloc(&out, grid * 256 * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const char* names[4] = {"returning, shared bins", "returning, bins per wave", "non-returning, shared bins", "no atomics"};
    for (int m = 0; m < 4; ++m)
    {
        float ms = 0;
        for (int it = 0; it < 2; ++it)
        {
            CHECK(hipEventRecord(a));
            if (m == 0) hipLaunchKernelGGL(rank_kernel<0>, dim3(grid), dim3(256), 0, 0, out, rounds);
            if (m == 1) hipLaunchKernelGGL(rank_kernel<1>, dim3(grid), dim3(256), 0, 0, out, rounds);
            if (m == 2) hipLaunchKernelGGL(rank_kernel<2>, dim3(grid), dim3(256), 0, 0, out, rounds);
            if (m == 3) hipLaunchKernelGGL(rank_kernel<3>, dim3(grid), dim3(256), 0, 0, out, rounds);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
        }
        // per CU: 3 workgroups x 4 waves x 16 atomics x rounds wave-instructions
        const double instr_per_cu = 3.0 * 4 * 16 * rounds;
        std::printf("%-28s %7.3f ms  = %.1f cycles of a CU per wave-instruction (2.4 GHz)\n", names[m], ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
    return 0;
}
