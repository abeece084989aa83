// What random LDS addresses cost in bank conflicts on gfx950, measured: the access shapes of the partition kernels
// (8-byte stores to random slots, 32-bit returning atomics on 256 random bins, 8-byte loads of random table entries)
// against the same instructions at linear addresses.  Standalone experiment, not part of the product.  Run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- ./scatter
// and divide the counters per kernel (profiles/r03/SUMMARY.md has the figures and the balls-in-bins expectation).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scatter scatter.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

constexpr int kSlots = 4096;
template <int SHAPE, bool RANDOM>      // 0: 8-byte store, 1: 32-bit returning atomic on 256 bins, 2: 8-byte load of 256 entries
__global__ __launch_bounds__(256, 3) void lds_kernel(uint64_t* out, uint32_t rounds)
{
    __shared__ uint64_t slots[kSlots];
    __shared__ uint32_t bins[256];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < kSlots; i += 256) slots[i] = i;
    bins[tid] = 0;
    __syncthreads();
    uint64_t acc = 0;
    for (uint32_t r = 0; r < rounds; ++r)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            const uint32_t h = mix((blockIdx.x * 256 + tid) * 16 + i + r * 0x9E3779B9u);
            if (SHAPE == 0) slots[RANDOM ? (h & (kSlots - 1)) : ((tid + 256 * i) & (kSlots - 1))] = h;
            else if (SHAPE == 1) acc += atomicAdd(&bins[RANDOM ? (h & 255u) : tid], 1u);
            else acc += slots[RANDOM ? (h & 255u) : tid];
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc + slots[tid];
}

int main()
{
    const uint32_t grid = 768, rounds = 2048;
    uint64_t* out; CHECK(hipMalloc(&out, grid * 256 * 8));
#define RUN(S, R) hipLaunchKernelGGL(HIP_KERNEL_NAME(lds_kernel<S, R>), dim3(grid), dim3(256), 0, 0, out, rounds)
    RUN(0, true); RUN(0, false); RUN(1, true); RUN(1, false); RUN(2, true); RUN(2, false);
    CHECK(hipDeviceSynchronize());
    std::printf("done\n");
    return 0;
}
