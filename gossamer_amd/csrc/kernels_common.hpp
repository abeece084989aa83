// kernels_common.hpp -- wave / block primitives, scans, small helpers shared by all kernels.
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"

namespace goss {

constexpr int kTB = 256;           // threads per workgroup
constexpr int kWaves = kTB / 64;

// Volatile views of __shared__ arrays keep their address space: through a generic volatile
// pointer the compiler emits FLAT loads and stores instead of ds_read / ds_write.
typedef volatile __attribute__((address_space(3))) uint32_t* lds_vu32;
typedef volatile __attribute__((address_space(3))) unsigned long long* lds_vu64;

// --------------------------------------------------------------------------------------
// wave / block primitives
// --------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }
// Kernels with one workgroup per segment are launched on a (x, y) grid (unit_grid in goss_gpu.hip):
// HIP refuses a launch whose gridDim.x * blockDim.x reaches 2^32, which 2^24 segments of 256
// threads do.  The workgroup's unit number:
__device__ __forceinline__ uint32_t unit_block() { return blockIdx.y * gridDim.x + blockIdx.x; }

template <class T>
__device__ __forceinline__ T wave_incl_scan(T v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        T o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

// Exclusive scan of one value per thread across the 256-thread block; `sh` holds kWaves+1
// elements of scratch.  Returns the exclusive prefix; *total = block sum.
template <class T>
__device__ __forceinline__ T block_excl_scan(T v, T* sh, T* total)
{
    T inc = wave_incl_scan(v);
    if (lane_id() == 63) sh[wave_id()] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
    {
        T s = sh[w];
        if ((int)wave_id() > w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// The same without the closing barrier: for a caller that passes another barrier before `sh` is written again.
template <class T>
__device__ __forceinline__ T block_excl_scan_open(T v, T* sh, T* total)
{
    T inc = wave_incl_scan(v);
    if (lane_id() == 63) sh[wave_id()] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
    {
        T s = sh[w];
        if ((int)wave_id() > w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// Inclusive scan of one 32-bit value per lane with data-parallel-primitive moves: four shifts inside every row of 16
// lanes, then the last lane of a row broadcast to the next row, then lane 31 to the upper half -- six vector
// instructions and no trip through the LDS crossbar (__shfl_up is a ds_bpermute: six dependent round trips).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);       // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);       // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);       // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);       // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t block_excl_scan_open_u32(uint32_t v, uint32_t* sh, uint32_t* total)
{
    const uint32_t inc = wave_incl_scan_u32(v);
    if (lane_id() == 63) sh[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
    {
        const uint32_t s = sh[w];
        if ((int)wave_id() > w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// ... and with the closing barrier (`sh` may be written again right behind it).
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* sh, uint32_t* total)
{
    const uint32_t r = block_excl_scan_open_u32(v, sh, total);
    __syncthreads();
    return r;
}

// The same for a workgroup of NW waves.
template <class T, int NW>
__device__ __forceinline__ T block_excl_scan_n(T v, T* sh, T* total)
{
    T inc = wave_incl_scan(v);
    if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
    {
        T s = sh[w];
        if ((int)(threadIdx.x >> 6) > w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// --------------------------------------------------------------------------------------
// device-wide exclusive scan of a u64 array (in place): reduce / scan partials / apply
// --------------------------------------------------------------------------------------

constexpr int kScanItems = 16;
constexpr int kScanChunk = kTB * kScanItems;

__global__ __launch_bounds__(kTB) void scan_reduce_kernel(const uint64_t* __restrict__ a, uint64_t n,
                                                          uint64_t* __restrict__ partial)
{
    __shared__ uint64_t sh[kWaves + 1];
    uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
    {
        uint64_t i = base + (uint64_t)j * kTB + threadIdx.x;
        if (i < n) s += a[i];
    }
    uint64_t tot;
    block_excl_scan<uint64_t>(s, sh, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// Exclusive scan of one chunk, adding partial[blockIdx] (already exclusive-scanned) as offset.
__global__ __launch_bounds__(kTB) void scan_apply_kernel(uint64_t* __restrict__ a, uint64_t n,
                                                         const uint64_t* __restrict__ partial)
{
    __shared__ uint64_t sh[kWaves + 1];
    uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
    {
        uint64_t i = base + j;
        v[j] = i < n ? a[i] : 0;
        s += v[j];
    }
    uint64_t tot;
    uint64_t off = block_excl_scan<uint64_t>(s, sh, &tot) + (partial ? partial[blockIdx.x] : 0);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
    {
        uint64_t i = base + j;
        if (i < n) a[i] = off;
        off += v[j];
    }
}

// Four input bytes -> their 2-bit codes, one per byte (A 0, C 1, G 2, T 3: bits 1-2 of the lower-cased letter, G and T
// put in order), and `bad` = 0x80 in every byte that is not one of ACGTacgt: the letter the code stands for is
// looked up by a byte permute and compared -- one non-zero test instead of one per letter.
// (GossReadBaseString::getBase, GossReadBaseString.hh:52-103: A/a C/c G/g T/t, anything else ends the windows.)
__device__ __forceinline__ uint32_t base_codes(uint32_t w, uint32_t& bad)
{
    const uint32_t l = w | 0x20202020u;
    uint32_t x = (l >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    const uint32_t v = l ^ __builtin_amdgcn_perm(0u, 0x74676361u, x);        // byte i of the second operand = "acgt"[code i]
    bad = (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
    return x;
}
// the codes of four bytes -> 8 bits, their flags -> 4 bits: a multiplication gathers one field per byte in the top byte
__device__ __forceinline__ uint32_t pack_codes(uint32_t x) { return (x * 0x01041040u) >> 24; }
__device__ __forceinline__ uint32_t pack_flags(uint32_t bad) { return ((bad >> 7) * 0x01020408u) >> 24; }

// Sixteen positions of the input at offset `byte0` (a multiple of 16) of a 16-byte aligned string -> 32 bits of 2-bit codes
// and 16 non-base flags; positions at or beyond `limit` are no bases.  The string is bytes (one letter per position,
// anything but ACGTacgt a non-base), or -- PACKED, what the host's parser threads hand over (goss_gpu_push_packed_host*)
// and the staging buffer keeps as it is -- one u32 of codes at src[4 g] and one u16 of flags at pbad[g] per group g of
// sixteen: the role of GossReadBaseString's encoder (GossReadBaseString.hh:133-188) is the host packer's, the device
// reads 3 bits per base and encodes nothing.
template <bool PACKED>
__device__ __forceinline__ void load_group16(const uint8_t* __restrict__ src, const uint16_t* __restrict__ pbad, uint64_t byte0, uint64_t limit,
                                             uint32_t& codes, uint32_t& bads)
{
    if constexpr (PACKED)
    {
        codes = 0; bads = 0xFFFFu;
        if (byte0 < limit)
        {
            const uint64_t g = byte0 >> 4;
            codes = reinterpret_cast<const uint32_t*>(src)[g];
            bads = pbad[g];
            if (byte0 + 16 > limit) bads = (bads | (0xFFFFu << (uint32_t)(limit - byte0))) & 0xFFFFu;
        }
    }
    else
    {
        uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
        if (byte0 + 16 <= limit)
        {
            const uint4 q = *reinterpret_cast<const uint4*>(src + byte0);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        }
        else if (byte0 < limit)
        {
            for (int j = 0; j < 16; ++j)
            {
                const uint64_t b = byte0 + j;
                const uint32_t c = b < limit ? src[b] : 0x0Au;
                w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
            }
        }
        codes = 0; bads = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            uint32_t bad;
            const uint32_t x = base_codes(w[i], bad);
            codes |= pack_codes(x) << (8 * i);
            bads |= pack_flags(bad) << (4 * i);
        }
    }
}

}  // namespace goss
