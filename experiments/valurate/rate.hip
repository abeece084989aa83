// Issue cost of a few integer vector instructions on gfx950 (MI355X): each kernel runs a loop of 8 independent chains
// of one instruction per lane, 8 waves per SIMD, every CU busy -- the time per wave-instruction and SIMD in cycles
// (2.4 GHz assumed) is what the instruction costs when the counting / extraction kernels issue it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rate rate.hip ; run: ./rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t iters, uint32_t m)
{
    uint32_t a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 977u + j * 131u + blockIdx.x;
    for (uint32_t it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[j]) : "v"(m));
                if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[j]) : "v"(m));
                if (OP == 6) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[j]));
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(m));
                if (OP == 8) asm volatile("v_cmp_eq_u32 vcc, %0, %1" ::"v"(a[j]), "v"(m) : "vcc");
                if (OP == 9) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 10) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 11) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[j]) : "v"(m));
                if (OP == 12) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*reinterpret_cast<uint64_t*>(&a[j & 6])));
                if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(m));
            }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s ^= a[j];
    if (s == 0x12345678u) out[0] = s;
}

template <int OP>
static void run(const char* name, uint32_t* d)
{
    const uint32_t iters = 2000, grid = 256 * 8;            // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, d, iters, 0x9E3779B1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double winstr = (double)grid * 4 * iters * 64;          // wave-instructions
    const double per_simd = winstr / 1024.0;
    std::printf("%-18s %.3f ms  = %.2f cycles per wave-instruction and SIMD at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    uint32_t* d = nullptr;
    CHECK(hipMalloc(&d, 4096));
    run<0>("v_xor_b32", d);
    run<1>("v_mul_lo_u32", d);
    run<2>("v_mul_u32_u24", d);
    run<3>("v_mad_u32_u24", d);
    run<4>("v_lshl_add_u32", d);
    run<5>("v_alignbit_b32", d);
    run<6>("v_bfe_u32", d);
    run<7>("v_cndmask_b32", d);
    run<8>("v_cmp_eq_u32", d);
    run<9>("v_mul_hi_u32", d);
    run<10>("v_perm_b32", d);
    run<11>("v_bfi_b32", d);
    run<12>("v_lshlrev_b64", d);
    run<13>("v_add3_u32", d);
    CHECK(hipFree(d));
    return 0;
}
