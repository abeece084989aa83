// What the first partition level's store pattern costs on MI355X, by itself: a persistent grid of 768 workgroups
// (three per CU), each appending to private blocks in `NB` bucket regions -- per "tile" every workgroup writes
// 32 KB as runs of `RUN` bytes, one run per bucket visited, consecutive tiles appending behind each other (the
// extraction kernel: 256 buckets, ~128 bytes per bucket and tile, 64-byte granules).  Nothing else is done: no
// loads, no LDS, no ranking.  Result (MI355X): 4.0 TB/s in the kernel's pattern against the kernel's 2.14: the pattern is
// not what bounds it.  Blocks that start at odd multiples of 16 bytes: 2.4 TB/s.  Standalone experiment, not part of
// the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store store.hip ; run: ./store
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// RUN bytes per bucket and tile (64 .. 2048), tile = 32 KB per workgroup = 4096 8-byte keys
template <int RUN, int POLICY>
__global__ __launch_bounds__(256, 3) void store_kernel(uint64_t* out, uint64_t region_keys, uint32_t tiles, uint32_t nb_log2)
{
    constexpr uint32_t kRunKeys = RUN / 8;                      // keys per run
    constexpr uint32_t kRunsPerTile = 4096 / kRunKeys;          // runs a tile writes
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = 1u << nb_log2;
    const uint64_t share = (region_keys / gridDim.x) & ~15ULL;  // this workgroup's private part of every region: whole 128-byte lines
    uint64_t* mine = out + (uint64_t)blockIdx.x * share;
    for (uint32_t t = 0; t < tiles; ++t)
    {
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u)
        {
            const uint32_t i = tid + 256 * u;
            const uint32_t run = i / kRunKeys, in = i % kRunKeys;
            // runs of a tile go to buckets (run + 7 t) mod nb; visit v of a bucket appends at v * kRunKeys
            const uint32_t b = (run + 7u * t) & (nb - 1);
            const uint32_t visits = kRunsPerTile >= nb ? t * (kRunsPerTile >> nb_log2) + (run >> nb_log2) : (t * kRunsPerTile + run) >> nb_log2;
            uint64_t* at = mine + (uint64_t)b * region_keys + visits * kRunKeys + in;
            const uint64_t v = ((uint64_t)t << 32) | i;
            if (POLICY == 1) __builtin_nontemporal_store(v, at); else *at = v;
        }
    }
}

template <int RUN, int POLICY = 0>
static void run(uint64_t* d, uint64_t total_keys, uint32_t nb)
{
    uint32_t nb_log2 = 0;
    while ((1u << nb_log2) < nb) ++nb_log2;
    const uint32_t grid = 768;
    const uint64_t region_keys = total_keys / nb;
    const uint32_t tiles = (uint32_t)(total_keys / grid / 4096 / 2);          // half of the room: no wrap
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel<RUN, POLICY>), dim3(grid), dim3(256), 0, 0, d, region_keys, tiles, nb_log2);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)grid * tiles * 32768.0;
        if (rep) std::printf("%s run %4d B  buckets %4u  %.1f GB in %.2f ms = %.2f TB/s\n", POLICY ? "nt   " : "plain", RUN, nb, bytes / 1e9, ms, bytes / ms / 1e9);
    }
}

// the same with TB threads per workgroup and tiles of KPT keys per thread (a tile = TB * KPT keys): what a larger tile
// on fewer, larger workgroups would write
template <int RUN, int TB, int KPT, int OCC, int MIS>
__global__ __launch_bounds__(TB, OCC) void store_kernel_t(uint64_t* out, uint64_t region_keys, uint32_t tiles, uint32_t nb_log2)
{
    constexpr uint32_t kRunKeys = RUN / 8;
    constexpr uint32_t kTile = TB * KPT;
    constexpr uint32_t kRunsPerTile = kTile / kRunKeys;
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = 1u << nb_log2;
    const uint64_t share = (region_keys / gridDim.x) & ~15ULL;
    uint64_t* mine = out + (uint64_t)blockIdx.x * share + (MIS ? 8 : 0);
    for (uint32_t t = 0; t < tiles; ++t)
    {
#pragma unroll
        for (uint32_t u = 0; u < (uint32_t)KPT; ++u)
        {
            const uint32_t i = tid + TB * u;
            const uint32_t run = i / kRunKeys, in = i % kRunKeys;
            const uint32_t b = (run + 7u * t) & (nb - 1);
            const uint32_t visits = (t * kRunsPerTile + run) >> nb_log2;
            mine[(uint64_t)b * region_keys + visits * kRunKeys + in] = ((uint64_t)t << 32) | i;
        }
    }
}

template <int RUN, int TB, int KPT, int OCC, int MIS = 0>
static void run_t(uint64_t* d, uint64_t total_keys, uint32_t nb)
{
    uint32_t nb_log2 = 0;
    while ((1u << nb_log2) < nb) ++nb_log2;
    const uint32_t grid = 256 * OCC;
    const uint64_t region_keys = total_keys / nb;
    const uint32_t tiles = (uint32_t)(total_keys / grid / (TB * KPT) / 2);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel_t<RUN, TB, KPT, OCC, MIS>), dim3(grid), dim3(TB), 0, 0, d, region_keys, tiles, nb_log2);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)grid * tiles * (TB * KPT) * 8.0;
        if (rep == 2) std::printf("%d x %d threads, tile %5d keys, run %4d B%s, buckets %4u: %.1f GB in %.2f ms = %.2f TB/s\n", grid, TB, TB * KPT, RUN, MIS ? " straddling two lines" : "", nb, bytes / 1e9, ms, bytes / ms / 1e9);
    }
}

// 16 bytes per lane (two keys): does the store path take lanes or bytes?
template <int RUN, int OCC>
__global__ __launch_bounds__(256, OCC) void store_kernel_x4(uint4* out, uint64_t region_pairs, uint32_t tiles, uint32_t nb_log2)
{
    constexpr uint32_t kRunPairs = RUN / 16;
    constexpr uint32_t kTile = 2048;                           // pairs per tile = 4096 keys
    constexpr uint32_t kRunsPerTile = kTile / kRunPairs;
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = 1u << nb_log2;
    const uint64_t share = (region_pairs / gridDim.x) & ~7ULL;
    uint4* mine = out + (uint64_t)blockIdx.x * share;
    for (uint32_t t = 0; t < tiles; ++t)
    {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
        {
            const uint32_t i = tid + 256 * u;
            const uint32_t run = i / kRunPairs, in = i % kRunPairs;
            const uint32_t b = (run + 7u * t) & (nb - 1);
            const uint32_t visits = (t * kRunsPerTile + run) >> nb_log2;
            mine[(uint64_t)b * region_pairs + visits * kRunPairs + in] = make_uint4(t, i, t, i);
        }
    }
}

template <int RUN, int OCC>
static void run_x4(uint64_t* d, uint64_t total_keys, uint32_t nb)
{
    uint32_t nb_log2 = 0;
    while ((1u << nb_log2) < nb) ++nb_log2;
    const uint32_t grid = 256 * OCC;
    const uint64_t region_pairs = total_keys / 2 / nb;
    const uint32_t tiles = (uint32_t)(total_keys / grid / 4096 / 2);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel_x4<RUN, OCC>), dim3(grid), dim3(256), 0, 0, (uint4*)d, region_pairs, tiles, nb_log2);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)grid * tiles * 32768.0;
        if (rep == 2) std::printf("%d x 256 threads, 16 bytes per lane, tile 4096 keys, run %4d B, buckets %4u: %.1f GB in %.2f ms = %.2f TB/s\n", grid, RUN, nb, bytes / 1e9, ms, bytes / ms / 1e9);
    }
}

int main()
{
    const uint64_t total_keys = 5ULL << 30;            // 40 GB of room
    uint64_t* d = nullptr;
    CHECK(hipMalloc(&d, total_keys * 8));
    CHECK(hipMemset(d, 0, total_keys * 8));
    run<64>(d, total_keys, 512);
    run<64>(d, total_keys, 256);
    run<128>(d, total_keys, 256);
    run<256>(d, total_keys, 256);
    run<256>(d, total_keys, 128);
    run<512>(d, total_keys, 64);
    run<2048>(d, total_keys, 16);
    run<64, 1>(d, total_keys, 256);
    run<128, 1>(d, total_keys, 256);
    run<256, 1>(d, total_keys, 256);
    run_t<64, 256, 16, 3>(d, total_keys, 512);
    run_t<64, 256, 16, 3>(d, total_keys, 256);            // the kernel's pattern: a bucket's two granules by two store instructions
    run_t<128, 256, 16, 3>(d, total_keys, 256);
    run_t<128, 256, 16, 3, 1>(d, total_keys, 256);
    run_t<256, 256, 16, 3>(d, total_keys, 256);
    run_t<128, 256, 8, 3>(d, total_keys, 256);
    run_t<128, 256, 16, 2>(d, total_keys, 256);
    run_t<256, 512, 16, 2>(d, total_keys, 256);
    run_t<256, 768, 16, 1>(d, total_keys, 256);
    run_x4<64, 3>(d, total_keys, 256);
    run_x4<128, 3>(d, total_keys, 256);
    run_x4<64, 1>(d, total_keys, 256);
    CHECK(hipFree(d));
    return 0;
}
