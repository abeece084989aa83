// goss_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X): key extraction,
// LSD radix sort, run compaction, Elias-Fano (SparseArray) + DenseSelect image build.
// All integer / byte work, HBM-bound; no MFMA by design.
//
// Wave size is 64 everywhere; workgroups are 256 threads (4 waves, one per SIMD).
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

// --------------------------------------------------------------------------------------
// K2: rolling / canonical k-mer extraction straight from ASCII bases
// --------------------------------------------------------------------------------------
//
// One workgroup handles a tile of T = 256*P window starts.  Phase A loads T+80 bytes with
// 16-byte vector loads and writes one code byte (0..3, 4 = not a base) per position to LDS.
// Phase B1: every thread derives the validity mask of its P windows; block scan gives the
// compacted slot of each thread.  Phase B2: threads roll the forward and reverse-complement
// key together, hash both (FNV-1a, in registers) for valid windows only and store the
// canonical key (or both strands) into an LDS staging buffer at the compacted slot.
// Phase C: one atomicAdd per tile reserves dense output space; staged keys are written with
// fully coalesced stores.
//
// MODE 0: canonical key per window.  MODE 1: forward key and its reverse complement.

struct ExtractCounters {
    unsigned long long keys_out;   // dense output cursor (keys)
    unsigned long long windows;    // valid windows
    unsigned long long hist[512];  // partition digit histograms (extract1_kernel)
};

// Bytes that are not one of ACGTacgt in `nslices` slices of `slice` bytes (a multiple of 16),
// `stride` bytes apart, of a 16-byte aligned string: out[0] += such bytes, out[1] += bytes looked
// at.  The host sizes the key buffers of a chunk from it (a non-base removes at most `len` windows).
__global__ __launch_bounds__(kTB) void nonbase_sample_kernel(const uint8_t* __restrict__ aligned, uint64_t nslices,
                                                             uint64_t stride, uint32_t slice,
                                                             unsigned long long* __restrict__ out)
{
    unsigned long long bad = 0, seen = 0;
    auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
    for (uint64_t s = blockIdx.x; s < nslices; s += gridDim.x)
    {
        const uint4* p = reinterpret_cast<const uint4*>(aligned + s * stride);
        for (uint32_t v = threadIdx.x; v < slice / 16; v += kTB)
        {
            const uint4 q = p[v];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                const uint32_t l = w[i] | 0x20202020u;
                bad += __popc(nz(l ^ 0x61616161u) & nz(l ^ 0x63636363u) & nz(l ^ 0x67676767u) & nz(l ^ 0x74747474u));
            }
            seen += 16;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_down(bad, o, 64); seen += __shfl_down(seen, o, 64); }
    if ((threadIdx.x & 63u) == 0 && seen) { atomicAdd(&out[0], bad); atomicAdd(&out[1], seen); }
}

// Strand representative of a k-mer for COUNTING: of {x, rc(x)} the one whose bits, rotated left
// by len (the central bases first), are smaller.  It is a function of the unordered pair, so both
// strands of a k-mer count as one key; the rotation makes the choice depend on the central bases,
// which leaves the leading bases -- the partition digits -- uniform; and it costs a handful of
// integer operations where gossamer's canonical form (the smaller FNV-1a hash, RankSelect.hh:126-140)
// costs two chains of 64-bit multiplies per window.  The distinct representatives are mapped to
// that canonical form once, after counting (canonical_map_kernel): 126 x fewer hashes on 150 x
// coverage.  x == rc(x) is the only tie.
__device__ __forceinline__ uint64_t rot_half(uint64_t v, uint32_t len, uint64_t lmask)
{
    return ((v & lmask) << len) | (v >> len);
}
__device__ __forceinline__ Key1 strand_rep(const Key1& f, const Key1& rc, uint32_t len, uint64_t lmask)
{
    return rot_half(rc.lo, len, lmask) < rot_half(f.lo, len, lmask) ? rc : f;
}

struct Rem96 { uint32_t r0, r1, r2; };           // the low 96 bits of a two-word key, packed (12-byte records)
__device__ __forceinline__ bool is_pad_key(const Key1& k) { return k.lo == ~0ULL; }
__device__ __forceinline__ bool is_pad_key(const Key2& k) { return (k.lo & k.hi) == ~0ULL; }

template <class K> struct KeyOps;
template <> struct KeyOps<Key1> {
    static __device__ __forceinline__ Key1 zero() { return Key1{0}; }
    static __device__ __forceinline__ void push(Key1& f, Key1& r, uint32_t c, uint64_t mask_lo, uint64_t, uint32_t topshift)
    {
        f.lo = ((f.lo << 2) | c) & mask_lo;
        r.lo = (r.lo >> 2) | ((uint64_t)(3u - c) << topshift);
    }
};
template <> struct KeyOps<Key2> {
    static __device__ __forceinline__ Key2 zero() { return Key2{0, 0}; }
    static __device__ __forceinline__ void push(Key2& f, Key2& r, uint32_t c, uint64_t mask_lo, uint64_t mask_hi, uint32_t topshift)
    {
        f.hi = ((f.hi << 2) | (f.lo >> 62)) & mask_hi;
        f.lo = ((f.lo << 2) | c) & mask_lo;
        r.lo = (r.lo >> 2) | (r.hi << 62);
        r.hi >>= 2;
        uint64_t cc = (uint64_t)(3u - c);
        if (topshift >= 64) r.hi |= cc << (topshift - 64);
        else r.lo |= cc << topshift;
    }
};

template <class K, int MODE, int P>
__global__ __launch_bounds__(kTB) void extract_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                      uint64_t nstarts, uint64_t navail, uint32_t len,
                                                      K* __restrict__ out, ExtractCounters* __restrict__ ctr)
{
    constexpr int T = kTB * P;
    constexpr int NVEC = T / 16 + 5;
    constexpr int S = MODE == 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) uint8_t code[NVEC * 16];
    __shared__ K stage[T * S];
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ unsigned long long sh_base;

    const uint64_t tile_base = (uint64_t)blockIdx.x * T;   // first window start of the tile
    const uint32_t tid = threadIdx.x;

    // ---- phase A: ASCII -> code bytes -------------------------------------------------
    // LDS index a corresponds to byte (tile_base + a) of the aligned stream, i.e. window
    // position (tile_base + a - mis).  Bytes whose position is >= navail are invalid.
    for (uint32_t v = tid; v < NVEC; v += kTB)
    {
        uint64_t byte0 = tile_base + (uint64_t)v * 16;            // aligned-stream offset
        uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
        // positions byte0-mis .. byte0-mis+15 ; fully in range?
        if (byte0 + 16 <= navail + mis)
        {
            uint4 q = *reinterpret_cast<const uint4*>(bases_aligned + byte0);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        }
        else if (byte0 < navail + mis)
        {
            for (int j = 0; j < 16; ++j)
            {
                uint64_t b = byte0 + j;
                uint32_t c = b < navail + mis ? bases_aligned[b] : 0x0Au;
                w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            // SWAR over 4 bytes: lower-case, 2-bit code, validity.
            uint32_t l = w[i] | 0x20202020u;
            uint32_t x = (l >> 1) & 0x03030303u;
            x ^= (x >> 1) & 0x01010101u;
            // nz(v): 0x80 in every byte of v that is non-zero
            auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
            uint32_t bad = nz(l ^ 0x61616161u) & nz(l ^ 0x63636363u) & nz(l ^ 0x67676767u) & nz(l ^ 0x74747474u);
            // bad byte -> code 4
            uint32_t badm = (bad >> 7) * 0xFFu;      // 0xFF in bad bytes
            o[i] = (x & ~badm) | ((bad >> 5) & 0x04040404u);
        }
        *reinterpret_cast<uint4*>(&code[v * 16]) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // ---- phase B1: validity of this thread's P windows --------------------------------
    const uint32_t q0 = tid * P + mis;               // LDS index of the first base of window 0
    const uint64_t p0 = tile_base + (uint64_t)tid * P;   // global window start
    uint32_t vmask = 0;
    {
        uint32_t run = 0;
        const uint32_t steps = P + len - 1;
        for (uint32_t j = 0; j < steps; ++j)
        {
            uint32_t c = code[q0 + j];
            run = c < 4 ? run + 1 : 0;
            if (j + 1 >= len && run >= len) vmask |= 1u << (j + 1 - len);
        }
        // windows starting at or beyond nstarts do not belong to this launch
        if (p0 + P > nstarts)
        {
            uint32_t keep = p0 >= nstarts ? 0u : (uint32_t)(nstarts - p0);
            vmask &= keep >= 32 ? 0xFFFFFFFFu : ((1u << keep) - 1u);
        }
    }
    uint32_t cnt = __popc(vmask);
    uint32_t tile_cnt;
    uint32_t slot = block_excl_scan<uint32_t>(cnt, sh_scan, &tile_cnt);

    // ---- phase B2: roll keys, canonicalise valid windows ------------------------------
    if (cnt)
    {
        const uint32_t bits = 2 * len;
        uint64_t mask_lo, mask_hi;
        if (bits >= 128) { mask_lo = ~0ULL; mask_hi = ~0ULL; }
        else if (bits >= 64) { mask_lo = ~0ULL; mask_hi = bits == 64 ? 0 : ((1ULL << (bits - 64)) - 1); }
        else { mask_lo = (1ULL << bits) - 1; mask_hi = 0; }
        const uint32_t topshift = bits - 2;
        K f = KeyOps<K>::zero(), r = KeyOps<K>::zero();
        const uint32_t steps = P + len - 1;
        uint32_t s = slot * S;
        for (uint32_t j = 0; j < steps; ++j)
        {
            uint32_t c = code[q0 + j] & 3u;
            KeyOps<K>::push(f, r, c, mask_lo, mask_hi, topshift);
            if (j + 1 >= len && ((vmask >> (j + 1 - len)) & 1u))
            {
                if (MODE == 0) stage[s++] = canonical(f, r);
                else { stage[s++] = f; stage[s++] = r; }
            }
        }
    }
    if (tid == 0)
    {
        unsigned long long b = 0;
        if (tile_cnt)
        {
            b = atomicAdd(&ctr->keys_out, (unsigned long long)tile_cnt * S);
            atomicAdd(&ctr->windows, (unsigned long long)tile_cnt);
        }
        sh_base = b;
    }
    __syncthreads();

    // ---- phase C: coalesced dense store -----------------------------------------------
    const uint64_t ob = sh_base;
    const uint32_t total = tile_cnt * S;
    for (uint32_t i = tid; i < total; i += kTB) out[ob + i] = stage[i];
}

// --------------------------------------------------------------------------------------
// K2, one-word keys: windows cut out of packed registers
// --------------------------------------------------------------------------------------
//
// Same contract as extract_kernel<Key1,...>.  Phase A packs every 16 loaded bytes into a
// 32-bit word of 2-bit codes (base j at bits 2j) and a 16-bit mask of non-bases.  A thread then
// holds the 128 code bits + 64 mask bits that cover its P windows in registers: the window
// starting at base i is the field E_i = bits [2i, 2i+2len), its reverse complement is simply
// ~E_i (complement of every 2-bit code; little-endian packing already reverses the order), its
// forward value is rolled, and it is valid iff the mask bits [i, i+len) are all zero.  No LDS
// access and no per-base loop remains in the window loop.

// NB = number of significant key bytes, ceil(2*len / 8): the FNV rounds of the zero bytes above
// them fold into one multiplication (goss_key.hpp, key_hash_short).
// REP: MODE 0 stores the strand representative (strand_rep) instead of the canonical form -- the
// key space extract1_part_kernel counts in; its sample must be drawn from the same space.
template <int MODE, int P, int G, int NB, bool REP = false>
__global__ __launch_bounds__(kTB) void extract1_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                       uint64_t nstarts, uint64_t navail, uint32_t len,
                                                       Key1* __restrict__ out, ExtractCounters* __restrict__ ctr,
                                                       uint32_t hist_shift, uint64_t nsuper,
                                                       uint64_t slice_tiles = 0, uint64_t slice_stride = 0)
{
    // Persistent grid: a workgroup loops over super-tiles (blockIdx.x, +gridDim.x, ...).
    // A super-tile is G consecutive sub-tiles of T = 256*P window starts and reserves the
    // output space of all of them with ONE atomicAdd: a single cursor word serves only ~88 M
    // returning atomics per second chip-wide, which bounded the one-reservation-per-tile form.
    constexpr int T = kTB * P;
    constexpr int NVEC = G * T / 16 + 4;
    constexpr int S = MODE == 1 ? 2 : 1;
    static_assert(P <= 16, "window mask is 16 bits");
    __shared__ uint32_t pk[NVEC];
    __shared__ uint32_t iv[NVEC];
    __shared__ Key1 stage[T * S];
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ unsigned long long sh_base;
    // histograms of the two partition digits (bits hist_shift.. and hist_shift+8..) of every key
    // this workgroup emits: saves the sort's separate histogram read of all keys
    __shared__ uint32_t lh[512];

    const uint32_t tid = threadIdx.x;
    const bool do_hist = hist_shift != 0xFFFFFFFFu;
    lh[tid] = 0; lh[tid + 256] = 0;

    for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x)
    {
    // sampling mode (slice_tiles != 0): super-tile st is the (st % slice_tiles)-th of slice
    // st / slice_tiles, slices lie slice_stride window starts apart (a multiple of 16)
    const uint64_t tile_base = slice_tiles ? (st / slice_tiles) * slice_stride + (st % slice_tiles) * (uint64_t)(G * T)
                                           : st * (uint64_t)(G * T);

    // ---- phase A: ASCII -> packed 2-bit codes + non-base mask, all G sub-tiles -------------
    for (uint32_t v = tid; v < NVEC; v += kTB)
    {
        uint64_t byte0 = tile_base + (uint64_t)v * 16;
        uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
        if (byte0 + 16 <= navail + mis)
        {
            uint4 q = *reinterpret_cast<const uint4*>(bases_aligned + byte0);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        }
        else if (byte0 < navail + mis)
        {
            for (int j = 0; j < 16; ++j)
            {
                uint64_t b = byte0 + j;
                uint32_t c = b < navail + mis ? bases_aligned[b] : 0x0Au;
                w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
            }
        }
        uint32_t codes = 0, bads = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            uint32_t l = w[i] | 0x20202020u;
            uint32_t x = (l >> 1) & 0x03030303u;
            x ^= (x >> 1) & 0x01010101u;
            auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
            uint32_t bad = nz(l ^ 0x61616161u) & nz(l ^ 0x63636363u) & nz(l ^ 0x67676767u) & nz(l ^ 0x74747474u);
            // four code bytes -> 8 bits, four bad flags -> 4 bits
            uint32_t c8 = (x & 0x3u) | ((x >> 6) & 0xCu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xC0u);
            uint32_t b1 = bad >> 7;
            uint32_t b4 = (b1 | (b1 >> 7) | (b1 >> 14) | (b1 >> 21)) & 0xFu;
            codes |= c8 << (8 * i);
            bads |= b4 << (4 * i);
        }
        pk[v] = codes;
        iv[v] = bads;
    }
    __syncthreads();

    const uint32_t bits = 2 * len;
    const uint64_t kmask = (1ULL << bits) - 1;               // len <= 31
    const uint64_t lmask = (1ULL << len) - 1;

    // ---- phase B: validity masks and compacted slots of every sub-tile ---------------------
    uint32_t vmask[G], slot[G], sub_cnt[G];
    uint32_t total = 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
        const uint32_t q0 = (g * kTB + tid) * P + mis;
        const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
        const uint64_t p0 = tile_base + (uint64_t)(g * kTB + tid) * P;
        uint64_t i0 = iv[v0], i1 = iv[v0 + 1], i2 = iv[v0 + 2], i3 = iv[v0 + 3];
        const uint64_t inv = (i0 | (i1 << 16) | (i2 << 32) | (i3 << 48)) >> sh;
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < P; ++i)
        {
            bool ok = ((inv >> i) & lmask) == 0 && (p0 + i < nstarts);
            m |= ok ? (1u << i) : 0u;
        }
        vmask[g] = m;
        uint32_t tc;
        slot[g] = block_excl_scan<uint32_t>(__popc(m), sh_scan, &tc);
        sub_cnt[g] = tc;
        total += tc;
    }
    if (tid == 0)
    {
        unsigned long long b = 0;
        if (total) b = atomicAdd(&ctr->keys_out, (unsigned long long)total * S);
        sh_base = b;
    }
    __syncthreads();
    uint64_t ob = sh_base;

    // ---- phase C: per sub-tile, cut the windows out of registers, stage, store --------------
#pragma unroll 1
    for (int g = 0; g < G; ++g)
    {
        const uint32_t vm = vmask[g];
        if (vm)
        {
            const uint32_t q0 = (g * kTB + tid) * P + mis;
            const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
            uint64_t w0 = pk[v0], w1 = pk[v0 + 1], w2 = pk[v0 + 2], w3 = pk[v0 + 3];
            uint64_t lo = w0 | (w1 << 32), hi = w2 | (w3 << 32);
            const uint32_t s2 = 2 * sh;
            const uint64_t blo = s2 ? ((lo >> s2) | (hi << (64 - s2))) : lo;
            const uint64_t bhi = hi >> s2;
            uint32_t s = slot[g] * S;
            // forward value of window 0: base-4 reversal of its field
            uint64_t f = rev64(blo & kmask) >> (64 - bits);
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                // field of window i: bits [2i, 2i + 2len) of the 128-bit buffer
                uint64_t e = i ? ((blo >> (2 * i)) | (bhi << (64 - 2 * i))) : blo;
                e &= kmask;
                if (i)
                {
                    uint32_t pos = 2 * (i + len - 1);      // new last base of the window
                    uint64_t nb = (pos < 64 ? (blo >> pos) : (bhi >> (pos - 64))) & 3u;
                    f = ((f << 2) | nb) & kmask;
                }
                if ((vm >> i) & 1u)
                {
                    Key1 fk{f}, rk{(~e) & kmask};
                    if (MODE == 0) stage[s++] = REP ? strand_rep(fk, rk, len, lmask) : canonical_short<NB>(fk, rk);
                    else { stage[s++] = fk; stage[s++] = rk; }
                }
            }
        }
        __syncthreads();
        const uint32_t nk = sub_cnt[g] * S;
        for (uint32_t i = tid; i < nk; i += kTB)
        {
            const Key1 k = stage[i];
            out[ob + i] = k;
            if (do_hist)
            {
                atomicAdd(&lh[(uint32_t)(k.lo >> hist_shift) & 0xFFu], 1u);
                atomicAdd(&lh[256u + ((uint32_t)(k.lo >> (hist_shift + 8)) & 0xFFu)], 1u);
            }
        }
        ob += nk;
        __syncthreads();
    }
    }   // super-tile loop
    if (do_hist)
    {
        if (lh[tid]) atomicAdd(&ctr->hist[tid], (unsigned long long)lh[tid]);
        if (lh[tid + 256]) atomicAdd(&ctr->hist[tid + 256], (unsigned long long)lh[tid + 256]);
    }
}

// --------------------------------------------------------------------------------------
// K2, two-word keys (32 <= len <= 63): the same windows-out-of-registers scheme with a 192-bit
// buffer of 2-bit codes per thread (96 bases >= 15 + P - 1 + 63)
// --------------------------------------------------------------------------------------
template <int MODE, int P, int G, int NBH = 8>
__global__ __launch_bounds__(kTB) void extract2_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                       uint64_t nstarts, uint64_t navail, uint32_t len,
                                                       Key2* __restrict__ out, ExtractCounters* __restrict__ ctr, uint64_t nsuper,
                                                       uint64_t slice_tiles = 0, uint64_t slice_stride = 0)
{
    constexpr int T = kTB * P;
    constexpr int NVEC = G * T / 16 + 6;
    constexpr int S = MODE == 1 ? 2 : 1;
    static_assert(P <= 16, "96 bases per thread");
    __shared__ uint32_t pk[NVEC];
    __shared__ uint32_t iv[NVEC];
    __shared__ Key2 stage[T * S];
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ unsigned long long sh_base;

    const uint32_t tid = threadIdx.x;
    const uint32_t bits = 2 * len;                                       // 64..126
    const uint64_t mask_hi = bits == 128 ? ~0ULL : ((1ULL << (bits - 64)) - 1);
    const uint64_t lmask = (1ULL << len) - 1;                            // len <= 63

    for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x)
    {
        // sampling mode (slice_tiles != 0): as in extract1_kernel
        const uint64_t tile_base = slice_tiles ? (st / slice_tiles) * slice_stride + (st % slice_tiles) * (uint64_t)(G * T)
                                               : st * (uint64_t)(G * T);
        // ---- phase A: ASCII -> packed 2-bit codes + non-base mask (as extract1_kernel) ----------
        for (uint32_t v = tid; v < NVEC; v += kTB)
        {
            uint64_t byte0 = tile_base + (uint64_t)v * 16;
            uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
            if (byte0 + 16 <= navail + mis)
            {
                uint4 q = *reinterpret_cast<const uint4*>(bases_aligned + byte0);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            }
            else if (byte0 < navail + mis)
            {
                for (int j = 0; j < 16; ++j)
                {
                    uint64_t b = byte0 + j;
                    uint32_t c = b < navail + mis ? bases_aligned[b] : 0x0Au;
                    w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
                }
            }
            uint32_t codes = 0, bads = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                uint32_t l = w[i] | 0x20202020u;
                uint32_t x = (l >> 1) & 0x03030303u;
                x ^= (x >> 1) & 0x01010101u;
                auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
                uint32_t bad = nz(l ^ 0x61616161u) & nz(l ^ 0x63636363u) & nz(l ^ 0x67676767u) & nz(l ^ 0x74747474u);
                uint32_t c8 = (x & 0x3u) | ((x >> 6) & 0xCu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xC0u);
                uint32_t b1 = bad >> 7;
                uint32_t b4 = (b1 | (b1 >> 7) | (b1 >> 14) | (b1 >> 21)) & 0xFu;
                codes |= c8 << (8 * i);
                bads |= b4 << (4 * i);
            }
            pk[v] = codes;
            iv[v] = bads;
        }
        __syncthreads();

        // ---- phase B: validity masks and compacted slots of every sub-tile ---------------------
        uint32_t vmask[G], slot[G], sub_cnt[G];
        uint32_t total = 0;
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            const uint32_t q0 = (g * kTB + tid) * P + mis;
            const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
            const uint64_t p0 = tile_base + (uint64_t)(g * kTB + tid) * P;
            const uint64_t inv_lo = (uint64_t)iv[v0] | ((uint64_t)iv[v0 + 1] << 16) | ((uint64_t)iv[v0 + 2] << 32) | ((uint64_t)iv[v0 + 3] << 48);
            const uint64_t inv_hi = (uint64_t)iv[v0 + 4] | ((uint64_t)iv[v0 + 5] << 16);
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                const uint32_t t = sh + i;                               // 0..30
                const uint64_t win = t ? ((inv_lo >> t) | (inv_hi << (64 - t))) : inv_lo;
                bool ok = (win & lmask) == 0 && (p0 + i < nstarts);
                m |= ok ? (1u << i) : 0u;
            }
            vmask[g] = m;
            uint32_t tc;
            slot[g] = block_excl_scan<uint32_t>(__popc(m), sh_scan, &tc);
            sub_cnt[g] = tc;
            total += tc;
        }
        if (tid == 0)
        {
            unsigned long long b = 0;
            if (total) b = atomicAdd(&ctr->keys_out, (unsigned long long)total * S);
            sh_base = b;
        }
        __syncthreads();
        uint64_t ob = sh_base;

        // ---- phase C: per sub-tile, cut the windows out of registers, stage, store --------------
#pragma unroll 1
        for (int g = 0; g < G; ++g)
        {
            const uint32_t vm = vmask[g];
            if (vm)
            {
                const uint32_t q0 = (g * kTB + tid) * P + mis;
                const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
                const uint64_t w0 = (uint64_t)pk[v0] | ((uint64_t)pk[v0 + 1] << 32);
                const uint64_t w1 = (uint64_t)pk[v0 + 2] | ((uint64_t)pk[v0 + 3] << 32);
                const uint64_t w2 = (uint64_t)pk[v0 + 4] | ((uint64_t)pk[v0 + 5] << 32);
                uint32_t s = slot[g] * S;
                Key2 f{0, 0};
#pragma unroll
                for (int i = 0; i < P; ++i)
                {
                    // field of window i: bits [2(sh+i), 2(sh+i) + 2len) of the 192-bit buffer
                    const uint32_t t2 = 2 * (sh + i);                    // 0..60
                    Key2 e;
                    e.lo = t2 ? ((w0 >> t2) | (w1 << (64 - t2))) : w0;
                    e.hi = (t2 ? ((w1 >> t2) | (w2 << (64 - t2))) : w1) & mask_hi;
                    if (i == 0)
                    {
                        // forward value of window 0: base-4 reversal of its field
                        const uint64_t rlo = rev64(e.hi), rhi = rev64(e.lo);   // reversed 128 bits = {rhi:rlo}
                        const uint32_t sft = 128 - bits;                        // 2..64
                        if (sft == 64) { f.lo = rhi; f.hi = 0; }
                        else { f.lo = (rlo >> sft) | (rhi << (64 - sft)); f.hi = rhi >> sft; }
                    }
                    else
                    {
                        const uint32_t pos = 2 * (sh + i + len - 1);     // new last base, bit position in the buffer
                        const uint64_t nb = (pos < 64 ? (w0 >> pos) : pos < 128 ? (w1 >> (pos - 64)) : (w2 >> (pos - 128))) & 3u;
                        f.hi = ((f.hi << 2) | (f.lo >> 62)) & mask_hi;
                        f.lo = (f.lo << 2) | nb;
                    }
                    if ((vm >> i) & 1u)
                    {
                        const Key2 rk{~e.lo, (~e.hi) & mask_hi};
                        if (MODE == 0) stage[s++] = canonical_tail<NBH>(f, rk);
                        else { stage[s++] = f; stage[s++] = rk; }
                    }
                }
            }
            __syncthreads();
            const uint32_t nk = sub_cnt[g] * S;
            for (uint32_t i = tid; i < nk; i += kTB) out[ob + i] = stage[i];
            ob += nk;
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------------------
// K2+K4 fused, one-word canonical keys: extraction that writes its keys already partitioned on
// the first partition digit
// --------------------------------------------------------------------------------------
//
// The first partition pass of the segment path may place tiles inside a bucket in any order, so
// it needs no scan over tiles -- only room in every bucket.  This kernel therefore partitions
// the keys of a super-tile (G*256*P window starts, at most 8192 keys) while they are still in
// registers: rank by LDS atomics on the digit at bit `shift`, reserve the tile's share of every
// bucket region with one atomic per digit, sort through LDS, store coalesced bucket runs.  That
// removes one write and one read of every key (16 of the 48 bytes per k-mer the unfused pipeline
// moves).  Bucket regions are sized by the host from a sample of the input (GapTable); a region
// that turns out too small raises `overflow` and the host redoes the chunk with the unfused
// kernels.  The histograms of the next two digits are accumulated for the passes that follow.

constexpr int kCursorStride = 32;                // u64 words between bucket cursors (256 B)

struct GapTable {
    unsigned long long reg_start[256];   // first key slot of bucket d
    unsigned long long reg_cap[256];     // slots reserved for bucket d
    unsigned long long cnt[256];         // keys actually in bucket d (filled by the host after extraction)
    unsigned long long tile_first[257];  // first tile of bucket d when its keys are cut into sort tiles
};

struct PartCounters {
    unsigned long long keys_out, windows, overflow, pad;
    unsigned long long hist[512];                        // digits at shift+8 and shift+16
    unsigned long long cursors[256 * kCursorStride];     // keys placed in bucket d so far
};

// Second level of the same idea (used when exactly two partition digits are needed): the fused
// kernel partitions on the HIGH digit, and the next pass places the keys of region b by their
// LOW digit into sub-regions (b, d) of the second key buffer, again by atomic cursors -- no
// look-back chain, no digit histograms.  Sub-region (b, d) IS segment b*256+d of the counting
// kernel.  Capacities come from the joint histogram of a larger sample; an overflow anywhere
// makes the host redo the chunk with the exact (look-back) sequence.
constexpr int kSubCursorStride = 4;              // u64 words between sub-region cursors (32 B)
struct SubTable {
    unsigned long long start[65536];     // first slot of sub-region (b, d), index b*256+d
    unsigned long long cap[65536];
};

constexpr uint64_t kPadKey = ~0ULL;              // no key: one-word keys use at most 62 bits

// Extraction fused with the first partition level, third form.  What bounded the second form
// (one returning atomic per tile and bucket on 256 cursor words, bucket runs of ~128 bytes landing
// on partial 64-byte granules of HBM, two FNV chains per window) is designed out:
//   * a workgroup owns a private BLOCK of B key slots in every bucket region and appends to it;
//     a bucket cursor is touched only when a block is used up (B = 256: 16 x fewer atomics);
//   * stores reach HBM as they are issued (nothing merges two partial writes of a 64-byte granule
//     on the way: 1.4-1.5 x the bytes when runs start anywhere), so a tile stores only whole
//     granules: per bucket the keys beyond a multiple of 8 wait in the registers of the thread that
//     owns the bucket (<= 7 keys) and go in front of the next tile's keys of that bucket; in LDS the
//     stored parts of all buckets lie back to back, each a multiple of 8 keys, so that 8 aligned
//     lanes of ONE store instruction cover one aligned granule;
//   * MODE 0 stores the strand representative (strand_rep) instead of the canonical form.
// The unused tail of every workgroup's last block is filled with kPadKey, which the next pass
// skips; pc->cursors[d] = slots handed out in bucket d (whole blocks), pc->keys_out = keys.
// The pk/iv arrays of phase A live in the memory of `sorted` (dead until the scatter).
template <int MODE, int NH, bool ODD>
__global__ __launch_bounds__(kTB, 3) void extract1_part_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                            uint64_t nstarts, uint64_t navail, uint32_t len,
                                                            Key1* __restrict__ out, PartCounters* __restrict__ pc,
                                                            const GapTable* __restrict__ gt, uint32_t shift, uint64_t nsuper,
                                                            uint32_t blk_log2)
{
    // MODE 0: one key per window, 16 windows per thread.  MODE 1 (graph): forward key and
    // reverse complement of every window, 8 windows per thread -- 16 keys per thread either way.
    constexpr int S = MODE == 1 ? 2 : 1;
    constexpr int P = 16 / S;
    constexpr int T = kTB * P;                   // window starts per tile
    constexpr int NVEC = T / 16 + 4;
    constexpr int NK = P * S;                    // keys per thread
    constexpr int kCarry = 7;
    // the tile's keys and the keys carried in: first the parts stored now, bucket after bucket
    // (each a multiple of 8), then the parts carried out
    // (+ 64 slots nobody reads: LDS writes that do not apply go there instead of under a branch, whose
    // exec-mask bookkeeping costs scalar issue slots; reads past the live part land there too)
    constexpr uint32_t kSpare = T * S + 256 * kCarry;
    __shared__ __attribute__((aligned(64))) Key1 sorted[T * S + 256 * kCarry + 64];
    __shared__ uint32_t dh[256 + 32];            // new keys of this tile per digit (rank counter); 32 spare ones for windows that are not valid
    // per bucket: x = first slot in `sorted` of the stored part | its length << 16,
    //             y = first slot of the part carried out | keys carried in << 13 | stored keys that fit the current block << 16
    __shared__ uint2 t_lay[256];
    // per bucket: slot / 8 of the current block's write position (x) and of the new block(s) (y)
    __shared__ uint2 t_base[256];
    __shared__ uint32_t sh_ovf;
    __shared__ uint32_t lh[NH ? 256 * NH : 1];   // histograms of the next NH digits
    __shared__ uint32_t sh_scan[kWaves + 1];
    uint32_t* pk = reinterpret_cast<uint32_t*>(sorted);
    uint32_t* iv = pk + NVEC;

    const uint32_t tid = threadIdx.x;
    if (NH > 0) lh[tid] = 0;
    if (NH > 1) lh[tid + 256] = 0;
    dh[tid] = 0;
    if (tid == 0) sh_ovf = 0;
    const uint64_t my_start = gt->reg_start[tid], my_cap = gt->reg_cap[tid];
    const uint32_t B = 1u << blk_log2;
    const uint32_t bits = 2 * len;
    const uint64_t kmask = (1ULL << bits) - 1;               // len <= 31
    const uint64_t lmask = (1ULL << len) - 1;
    unsigned long long nvalid = 0;
    uint64_t wpos = 0;                           // next slot of bucket tid's open block (a block boundary = none open)
    uint32_t ccnt = 0;                           // keys of bucket tid carried over from the previous tile
    Key1 kc[kCarry];                             // ... and the keys themselves
#pragma unroll
    for (int j = 0; j < kCarry; ++j) kc[j].lo = 0;

    // 16 bytes of the input at `byte0` -> 32 bits of 2-bit codes + 16 non-base flags
    auto fetch = [&](uint64_t byte0, uint4& q) -> bool {
        if (byte0 + 16 <= navail + mis) { q = *reinterpret_cast<const uint4*>(bases_aligned + byte0); return true; }
        q = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        if (byte0 < navail + mis)
        {
            uint32_t w[4] = {q.x, q.y, q.z, q.w};
            for (int j = 0; j < 16; ++j)
            {
                uint64_t b = byte0 + j;
                uint32_t c = b < navail + mis ? bases_aligned[b] : 0x0Au;
                w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
            }
            q = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return true;
    };
    auto encode = [](const uint4& q, uint32_t& codes, uint32_t& bads) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        codes = 0; bads = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            uint32_t l = w[i] | 0x20202020u;
            uint32_t x = (l >> 1) & 0x03030303u;
            x ^= (x >> 1) & 0x01010101u;
            auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
            uint32_t bad = nz(l ^ 0x61616161u) & nz(l ^ 0x63636363u) & nz(l ^ 0x67676767u) & nz(l ^ 0x74747474u);
            uint32_t c8 = (x & 0x3u) | ((x >> 6) & 0xCu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xC0u);
            uint32_t b1 = bad >> 7;
            uint32_t b4 = (b1 | (b1 >> 7) | (b1 >> 14) | (b1 >> 21)) & 0xFu;
            codes |= c8 << (8 * i);
            bads |= b4 << (4 * i);
        }
    };
    // The bytes of a tile are fetched one tile ahead and wait, encoded, in registers: the load's
    // latency passes behind the previous tile's ranking and sorting, and the stores of a tile have
    // half a tile's time to drain before anything waits on this wave's memory counter again.
    constexpr uint32_t NV0 = T / 16;             // thread tid < NV0 encodes vector tid, threads 0..3 also vector NV0 + tid
    static_assert(NVEC == NV0 + 4 && NV0 <= kTB, "one vector per thread and four more");
    uint32_t c0 = 0, b0 = 0, c1 = 0, b1 = 0;
    if (blockIdx.x < nsuper)
    {
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        const uint64_t tb = (uint64_t)blockIdx.x * T;
        if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
        if (tid < 4) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        encode(q0, c0, b0);
        if (tid < 4) encode(q1, c1, b1);
    }

    for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x)
    {
        const uint64_t tile_base = st * (uint64_t)T;

        // ---- phase A: this tile's codes from registers to LDS, the next tile's bytes on their way ----
        if (tid < NV0) { pk[tid] = c0; iv[tid] = b0; }
        if (tid < 4) { pk[NV0 + tid] = c1; iv[NV0 + tid] = b1; }
        __syncthreads();
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        const bool more = st + gridDim.x < nsuper;
        if (more)
        {
            const uint64_t tb = (st + gridDim.x) * (uint64_t)T;
            if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
            if (tid < 4) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        }

        // ---- phase B: windows out of registers, keys, rank inside their digit --------------------
        // Written without branches around the LDS operations: a window that is not valid still gets a
        // (meaningless) key and ranks itself in a spare counter, so that the sixteen returning atomics
        // of a thread are issued back to back and waited for once, not one round trip after the other.
        Key1 kreg[NK];
        uint32_t rk[NK];
        uint32_t vm;
        {
            const uint32_t q0 = tid * P + mis;
            const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
            const uint64_t p0 = tile_base + (uint64_t)tid * P;
            uint64_t i0 = iv[v0], i1 = iv[v0 + 1], i2 = iv[v0 + 2], i3 = iv[v0 + 3];
            uint64_t w0 = pk[v0], w1 = pk[v0 + 1], w2 = pk[v0 + 2], w3 = pk[v0 + 3];
            const uint64_t inv = (i0 | (i1 << 16) | (i2 << 32) | (i3 << 48)) >> sh;
            // window i is valid iff bits [i, i + len) of `inv` are zero.  All P windows at once: runs of good bases
            // of length 1, 2, 4, .. by doubling, and the AND of the runs that make up len (its binary digits) at
            // their offsets -- six steps of a few 64-bit operations instead of a shift, mask and compare per window
            uint32_t m;
            {
                uint64_t run = ~inv, acc = ~0ULL;
                uint32_t covered = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j)               // len <= 31
                {
                    if ((len >> j) & 1u) { acc &= run >> covered; covered += 1u << j; }
                    run &= run >> (1u << j);
                }
                // (bits of `inv` above the 64 that were read count as good: they belong to windows beyond P anyway)
                const uint64_t left = nstarts > p0 ? nstarts - p0 : 0;
                const uint32_t lim = left >= (uint64_t)P ? (uint32_t)((1ULL << P) - 1ULL) : ((1u << (uint32_t)left) - 1u);
                m = (uint32_t)acc & lim;
            }
            vm = m;
            nvalid += __popc(m);
            const uint64_t lo = w0 | (w1 << 32), hi = w2 | (w3 << 32);
            const uint32_t s2 = 2 * sh;
            const uint64_t blo = s2 ? ((lo >> s2) | (hi << (64 - s2))) : lo;
            const uint64_t bhi = hi >> s2;
            // forward key f and reverse complement r of window 0, then one base rolled in per window
            uint64_t f = rev64(blo & kmask) >> (64 - bits);
            uint64_t r = (~blo) & kmask;
            const uint32_t top = bits - 2;
            const uint32_t spare = 256u + (tid & 31u);          // counters nobody reads
            uint32_t bin[NK];
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                if (i)
                {
                    const uint32_t pos = 2 * (i + len - 1);
                    const uint32_t nb = (uint32_t)(pos < 64 ? (blo >> pos) : (bhi >> (pos - 64))) & 3u;
                    f = ((f << 2) | nb) & kmask;
                    r = (r >> 2) | ((uint64_t)(nb ^ 3u) << top);
                }
                const bool ok = (m >> i) & 1u;
                const Key1 fk{f}, rck{r};
                if (MODE == 0)
                {
                    // odd length: the central base decides (its low bit differs between the strands)
                    const Key1 k = ODD ? (((f >> (len - 1)) & 1ULL) ? rck : fk) : strand_rep(fk, rck, len, lmask);
                    kreg[i] = k;
                    bin[i] = ok ? ((uint32_t)(k.lo >> shift) & 0xFFu) : spare;
                }
                else
                {
                    kreg[i * 2] = fk;
                    bin[i * 2] = ok ? ((uint32_t)(fk.lo >> shift) & 0xFFu) : spare;
                    kreg[i * 2 + 1] = rck;
                    bin[i * 2 + 1] = ok ? ((uint32_t)(rck.lo >> shift) & 0xFFu) : spare;
                }
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) rk[i] = atomicAdd(&dh[bin[i]], 1u);
        }
        __syncthreads();

        // ---- phase C: bookkeeping of bucket tid: what is stored now, where, what is carried out ----
        uint32_t total_store;
        {
            const uint32_t cnt = dh[tid];
            const uint32_t tot = ccnt + cnt;           // the bucket's stream: carried keys, then the new ones by rank
            const uint32_t fl = tot & ~7u, rem = tot & 7u;
            uint32_t sums;
            const uint32_t pre = block_excl_scan<uint32_t>(fl | (rem << 16), sh_scan, &sums);
            total_store = sums & 0xFFFFu;
            const uint32_t f_at = pre & 0xFFFFu, l_at = total_store + (pre >> 16);
            dh[tid] = 0;                               // ready for the next tile (its ranking starts behind two barriers)
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            uint32_t thr = fl;
            uint2 tb = make_uint2((uint32_t)(wpos >> 3), 0u);
            if (fl > room)
            {
                thr = room;
                const uint32_t need = fl - room;
                const uint64_t want = ((uint64_t)(need + B - 1) >> blk_log2) << blk_log2;
                const unsigned long long at = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)want);
                // a region that is too small: nothing of this tile is stored, the host redoes the chunk
                if (at + want > my_cap) { atomicOr(&pc->overflow, 1ULL); sh_ovf = 1; }
                tb.y = (uint32_t)((my_start + at) >> 3);
                wpos = my_start + at + need;
            }
            else wpos += fl;
            t_base[tid] = tb;
            t_lay[tid] = make_uint2(f_at | (fl << 16), l_at | (ccnt << 13) | (thr << 16));
            // the keys carried in go first (phase A's arrays in `sorted` are dead: every thread is past phase B)
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
                sorted[(uint32_t)j < ccnt ? ((uint32_t)j < fl ? f_at + j : l_at + j) : kSpare + (tid & 63u)] = kc[j];
            ccnt = rem;
        }
        __syncthreads();
        // new keys to their place: position ccnt_in + rank of the bucket's stream (the table reads of
        // all sixteen keys first, then the writes: no round trip per key)
        {
            uint2 tl[NK];
#pragma unroll
            for (int i = 0; i < NK; ++i) tl[i] = t_lay[(uint32_t)(kreg[i].lo >> shift) & 0xFFu];
#pragma unroll
            for (int i = 0; i < NK; ++i)
            {
                const bool ok = (vm >> (i / S)) & 1u;
                const Key1 k = kreg[i];
                const uint32_t p = ((tl[i].y >> 13) & 7u) + rk[i];
                const uint32_t fl = tl[i].x >> 16;
                const uint32_t at = p < fl ? (tl[i].x & 0xFFFFu) + p : (tl[i].y & 0x1FFFu) + (p - fl);
                sorted[ok ? at : kSpare + (tid & 63u)] = k;
                if (NH > 0) atomicAdd(&lh[(uint32_t)(k.lo >> (shift + 8)) & 0xFFu], ok ? 1u : 0u);
                if (NH > 1) atomicAdd(&lh[256u + ((uint32_t)(key_shr64(k, shift + 16)) & 0xFFu)], ok ? 1u : 0u);
            }
        }
        if (more)
        {
            encode(q0, c0, b0);
            if (tid < 4) encode(q1, c1, b1);
        }
        __syncthreads();

        // ---- phase D: whole granules to the bucket blocks; every 8 aligned lanes store one -------
        if (sh_ovf == 0)
            for (uint32_t i0 = tid; i0 < total_store; i0 += 4 * kTB)
            {
                // four keys at a time: their LDS reads, then their table reads, then their stores
                Key1 kk[4];
                uint2 tl[4], tb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kk[u] = sorted[min(i0 + u * kTB, kSpare)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t d = (uint32_t)(kk[u].lo >> shift) & 0xFFu;
                    tl[u] = t_lay[d]; tb[u] = t_base[d];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t i = i0 + u * kTB;
                    if (i < total_store)
                    {
                        const uint32_t p = i - (tl[u].x & 0xFFFFu);
                        const uint32_t thr = tl[u].y >> 16;
                        const uint64_t o = p < thr ? ((uint64_t)tb[u].x << 3) + p : ((uint64_t)tb[u].y << 3) + (p - thr);
                        out[o] = kk[u];
                    }
                }
            }
        // what bucket tid carries out, back into registers
        {
            const uint32_t l_at = t_lay[tid].y & 0x1FFFu;
#pragma unroll
            for (int j = 0; j < kCarry; ++j) kc[j] = sorted[l_at + j];       // (those beyond ccnt are never used)
        }
        __syncthreads();
    }

    // ---- the end: carried keys and the unused tail of every open block ---------------------------
    if (sh_ovf == 0)
    {
        if (ccnt)
        {
            // one more granule: the carried keys, padding behind them
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            if (room == 0)
            {
                const unsigned long long at = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)B);
                if (at + B > my_cap) { atomicOr(&pc->overflow, 1ULL); ccnt = 0; wpos = 0; }
                else wpos = my_start + at;
            }
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
                if ((uint32_t)j < ccnt) out[wpos + j] = kc[j];
            wpos += ccnt;
        }
        const uint32_t tail = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
        // pad [wpos, end of block) of every bucket: all threads share the work through LDS
        __syncthreads();
        t_base[tid].x = tail;
        reinterpret_cast<uint64_t*>(sorted)[tid] = wpos;
        __syncthreads();
        for (uint32_t d = 0; d < 256; ++d)
        {
            const uint32_t n = t_base[d].x;
            const uint64_t from = reinterpret_cast<const uint64_t*>(sorted)[d];
            for (uint32_t j = tid; j < n; j += kTB) out[from + j] = Key1{kPadKey};
        }
    }
    if (NH > 0) { if (lh[tid]) atomicAdd(&pc->hist[tid], (unsigned long long)lh[tid]); }
    if (NH > 1) { if (lh[tid + 256]) atomicAdd(&pc->hist[tid + 256], (unsigned long long)lh[tid + 256]); }
    // valid windows of this workgroup
    for (int o = 32; o > 0; o >>= 1) nvalid += __shfl_down(nvalid, o, 64);
    if (lane_id() == 0 && nvalid) { atomicAdd(&pc->keys_out, nvalid * S); atomicAdd(&pc->windows, nvalid); }
}

// Strand representatives -> gossamer's canonical form (position_type::normalize, RankSelect.hh:126-140),
// for the distinct keys only; the result is no longer sorted.
template <class K>
__global__ __launch_bounds__(kTB) void canonical_map_kernel(const K* in, K* outk, uint64_t m, uint32_t len)       // in == outk is fine
{
    const uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= m) return;
    const K x = in[i];
    outk[i] = canonical(x, revcomp(x, len));
}

// The same fusion for two-word keys (32 <= len <= 63): windows out of a 192-bit register buffer
// (extract2_kernel), NKEYS keys per thread, a tile of 256*NKEYS keys partitioned on the digit at `shift`,
// in the form of extract1_part_kernel: private blocks of B slots per workgroup and bucket (a cursor is
// touched once per block), whole 64-byte granules (4 keys) stored from 4 aligned lanes, the remainder of a
// bucket (<= 3 keys) carried in the registers of the thread that owns it, kPadKey pairs behind the last
// keys of every last block, the next tile's bytes fetched one tile ahead.  MODE 0 still computes gossamer's
// canonical form in the kernel (two FNV chains over 16 bytes).
template <int MODE, int NH, int NKEYS, int NBH = 8>
__global__ __launch_bounds__(kTB, 2) void extract2_part_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                               uint64_t nstarts, uint64_t navail, uint32_t len,
                                                               Key2* __restrict__ out, PartCounters* __restrict__ pc,
                                                               const GapTable* __restrict__ gt, uint32_t shift, uint64_t nsuper,
                                                               uint32_t blk_log2)
{
    constexpr int S = MODE == 1 ? 2 : 1;
    constexpr int P = NKEYS / S;                 // windows per thread; NKEYS keys per thread
    constexpr int T = kTB * P;
    constexpr int NVEC = T / 16 + 6;
    constexpr int NK = P * S;
    constexpr int kCarry = 3;                    // keys of a bucket below a granule of 4
    constexpr uint32_t kSpare = T * S + 256 * kCarry;
    __shared__ __attribute__((aligned(64))) Key2 sorted[T * S + 256 * kCarry + 64];
    __shared__ uint32_t dh[256 + 32];
    __shared__ uint2 t_lay[256];                 // x = first slot of the stored part | its length << 16; y = first slot of the carried part | keys carried in << 13 | stored keys that fit the current block << 16
    __shared__ uint2 t_base[256];                // slot / 4 of the current block's write position (x) and of the new block(s) (y)
    __shared__ uint32_t lh[NH ? 256 * NH : 1];
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ uint32_t sh_ovf;
    uint32_t* pk = reinterpret_cast<uint32_t*>(sorted);
    uint32_t* iv = pk + NVEC;

    const uint32_t tid = threadIdx.x;
    if (NH > 0) lh[tid] = 0;
    if (NH > 1) lh[tid + 256] = 0;
    dh[tid] = 0;
    if (tid < 32) dh[256 + tid] = 0;
    if (tid == 0) sh_ovf = 0;
    const uint64_t my_start = gt->reg_start[tid], my_cap = gt->reg_cap[tid];
    const uint32_t B = 1u << blk_log2;
    const uint32_t bits = 2 * len;                                       // 64..126
    const uint64_t mask_hi = bits == 128 ? ~0ULL : ((1ULL << (bits - 64)) - 1);
    const uint64_t lmask = (1ULL << len) - 1;
    unsigned long long nvalid = 0;
    uint64_t wpos = 0;
    uint32_t ccnt = 0;
    Key2 kc[kCarry];
#pragma unroll
    for (int j = 0; j < kCarry; ++j) kc[j] = Key2{0, 0};

    auto fetch = [&](uint64_t byte0, uint4& q) {
        if (byte0 + 16 <= navail + mis) { q = *reinterpret_cast<const uint4*>(bases_aligned + byte0); return; }
        q = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        if (byte0 < navail + mis)
        {
            uint32_t w[4] = {q.x, q.y, q.z, q.w};
            for (int j = 0; j < 16; ++j)
            {
                uint64_t b = byte0 + j;
                uint32_t c = b < navail + mis ? bases_aligned[b] : 0x0Au;
                w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
            }
            q = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };
    auto encode = [](const uint4& q, uint32_t& codes, uint32_t& bads) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        codes = 0; bads = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            uint32_t l = w[i] | 0x20202020u;
            uint32_t x = (l >> 1) & 0x03030303u;
            x ^= (x >> 1) & 0x01010101u;
            auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
            uint32_t bad = nz(l ^ 0x61616161u) & nz(l ^ 0x63636363u) & nz(l ^ 0x67676767u) & nz(l ^ 0x74747474u);
            uint32_t c8 = (x & 0x3u) | ((x >> 6) & 0xCu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xC0u);
            uint32_t b1 = bad >> 7;
            uint32_t b4 = (b1 | (b1 >> 7) | (b1 >> 14) | (b1 >> 21)) & 0xFu;
            codes |= c8 << (8 * i);
            bads |= b4 << (4 * i);
        }
    };
    constexpr uint32_t NV0 = T / 16;             // thread tid < NV0 encodes vector tid, threads 0..5 also vector NV0 + tid
    static_assert(NVEC == NV0 + 6 && NV0 <= kTB, "one vector per thread and six more");
    uint32_t c0 = 0, b0 = 0, c1 = 0, b1 = 0;
    if (blockIdx.x < nsuper)
    {
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        const uint64_t tb = (uint64_t)blockIdx.x * T;
        if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
        if (tid < 6) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        encode(q0, c0, b0);
        if (tid < 6) encode(q1, c1, b1);
    }

    for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x)
    {
        const uint64_t tile_base = st * (uint64_t)T;
        if (tid < NV0) { pk[tid] = c0; iv[tid] = b0; }
        if (tid < 6) { pk[NV0 + tid] = c1; iv[NV0 + tid] = b1; }
        __syncthreads();
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        const bool more = st + gridDim.x < nsuper;
        if (more)
        {
            const uint64_t tb = (st + gridDim.x) * (uint64_t)T;
            if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
            if (tid < 6) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        }

        // ---- windows out of registers, keys, rank inside their digit (no branch around the LDS atomics) ----
        Key2 kreg[NK];
        uint32_t rk[NK];
        uint32_t vm = 0;
        {
            const uint32_t q0i = tid * P + mis;
            const uint32_t v0 = q0i >> 4, sh = q0i & 15u;
            const uint64_t p0 = tile_base + (uint64_t)tid * P;
            const uint64_t inv_lo = (uint64_t)iv[v0] | ((uint64_t)iv[v0 + 1] << 16) | ((uint64_t)iv[v0 + 2] << 32) | ((uint64_t)iv[v0 + 3] << 48);
            const uint64_t inv_hi = (uint64_t)iv[v0 + 4] | ((uint64_t)iv[v0 + 5] << 16);
            const uint64_t w0 = (uint64_t)pk[v0] | ((uint64_t)pk[v0 + 1] << 32);
            const uint64_t w1 = (uint64_t)pk[v0 + 2] | ((uint64_t)pk[v0 + 3] << 32);
            const uint64_t w2 = (uint64_t)pk[v0 + 4] | ((uint64_t)pk[v0 + 5] << 32);
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                const uint32_t t = sh + i;
                const uint64_t win = t ? ((inv_lo >> t) | (inv_hi << (64 - t))) : inv_lo;
                bool ok = (win & lmask) == 0 && (p0 + i < nstarts);
                vm |= ok ? (1u << i) : 0u;
            }
            nvalid += __popc(vm);
            const uint32_t spare = 256u + (tid & 31u);
            uint32_t bin[NK];
            // forward key f and reverse complement r of window 0 from the register buffer, then one base
            // rolled into both per window
            Key2 f{0, 0}, r{0, 0};
            const uint32_t top = bits - 2;                  // position of a key's first base (>= 62)
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                if (i == 0)
                {
                    const uint32_t t2 = 2 * sh;
                    Key2 e;
                    e.lo = t2 ? ((w0 >> t2) | (w1 << (64 - t2))) : w0;
                    e.hi = (t2 ? ((w1 >> t2) | (w2 << (64 - t2))) : w1) & mask_hi;
                    const uint64_t rlo = rev64(e.hi), rhi = rev64(e.lo);
                    const uint32_t sft = 128 - bits;
                    if (sft == 64) { f.lo = rhi; f.hi = 0; }
                    else { f.lo = (rlo >> sft) | (rhi << (64 - sft)); f.hi = rhi >> sft; }
                    r.lo = ~e.lo; r.hi = (~e.hi) & mask_hi;
                }
                else
                {
                    const uint32_t pos = 2 * (sh + i + len - 1);
                    const uint64_t nb = (pos < 64 ? (w0 >> pos) : pos < 128 ? (w1 >> (pos - 64)) : (w2 >> (pos - 128))) & 3u;
                    f.hi = ((f.hi << 2) | (f.lo >> 62)) & mask_hi;
                    f.lo = (f.lo << 2) | nb;
                    const uint64_t cb = nb ^ 3u;
                    r.lo = (r.lo >> 2) | (r.hi << 62);
                    r.hi = (r.hi >> 2) | (top >= 64 ? cb << (top - 64) : 0ULL);
                    if (top < 64) r.lo |= cb << top;
                }
                const bool ok = (vm >> i) & 1u;
                const Key2 rck = r;
                if (MODE == 0)
                {
                    const Key2 k = canonical_tail<NBH>(f, rck);
                    kreg[i] = k;
                    bin[i] = ok ? key_digit(k, shift) : spare;
                }
                else
                {
                    kreg[2 * i] = f;
                    bin[2 * i] = ok ? key_digit(f, shift) : spare;
                    kreg[2 * i + 1] = rck;
                    bin[2 * i + 1] = ok ? key_digit(rck, shift) : spare;
                }
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) rk[i] = atomicAdd(&dh[bin[i]], 1u);
        }
        __syncthreads();

        // ---- bookkeeping of bucket tid: what is stored now, where, what is carried out ----
        uint32_t total_store;
        {
            const uint32_t cnt = dh[tid];
            const uint32_t tot = ccnt + cnt;
            const uint32_t fl = tot & ~3u, rem = tot & 3u;
            uint32_t sums;
            const uint32_t pre = block_excl_scan<uint32_t>(fl | (rem << 16), sh_scan, &sums);
            total_store = sums & 0xFFFFu;
            const uint32_t f_at = pre & 0xFFFFu, l_at = total_store + (pre >> 16);
            dh[tid] = 0;
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            uint32_t thr = fl;
            uint2 tb = make_uint2((uint32_t)(wpos >> 2), 0u);
            if (fl > room)
            {
                thr = room;
                const uint32_t need = fl - room;
                const uint64_t want = ((uint64_t)(need + B - 1) >> blk_log2) << blk_log2;
                const unsigned long long at = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)want);
                if (at + want > my_cap) { atomicOr(&pc->overflow, 1ULL); sh_ovf = 1; }
                tb.y = (uint32_t)((my_start + at) >> 2);
                wpos = my_start + at + need;
            }
            else wpos += fl;
            t_base[tid] = tb;
            t_lay[tid] = make_uint2(f_at | (fl << 16), l_at | (ccnt << 13) | (thr << 16));
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
                sorted[(uint32_t)j < ccnt ? ((uint32_t)j < fl ? f_at + j : l_at + j) : kSpare + (tid & 63u)] = kc[j];
            ccnt = rem;
        }
        __syncthreads();
        {
            uint2 tl[NK];
#pragma unroll
            for (int i = 0; i < NK; ++i) tl[i] = t_lay[key_digit(kreg[i], shift)];
#pragma unroll
            for (int i = 0; i < NK; ++i)
            {
                const bool ok = (vm >> (i / S)) & 1u;
                const Key2 k = kreg[i];
                const uint32_t p = ((tl[i].y >> 13) & 7u) + rk[i];
                const uint32_t fl = tl[i].x >> 16;
                const uint32_t at = p < fl ? (tl[i].x & 0xFFFFu) + p : (tl[i].y & 0x1FFFu) + (p - fl);
                sorted[ok ? at : kSpare + (tid & 63u)] = k;
                if (NH > 0) atomicAdd(&lh[key_digit(k, shift + 8)], ok ? 1u : 0u);
                if (NH > 1) atomicAdd(&lh[256u + key_digit(k, shift + 16)], ok ? 1u : 0u);
            }
        }
        if (more)
        {
            encode(q0, c0, b0);
            if (tid < 6) encode(q1, c1, b1);
        }
        __syncthreads();

        // ---- whole granules to the bucket blocks; every 4 aligned lanes store one ----
        if (sh_ovf == 0)
            for (uint32_t i0 = tid; i0 < total_store; i0 += 2 * kTB)
            {
                Key2 kk[2];
                uint2 tl[2], tb[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) kk[u] = sorted[min(i0 + u * kTB, kSpare)];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const uint32_t d = key_digit(kk[u], shift);
                    tl[u] = t_lay[d]; tb[u] = t_base[d];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const uint32_t i = i0 + u * kTB;
                    if (i < total_store)
                    {
                        const uint32_t p = i - (tl[u].x & 0xFFFFu);
                        const uint32_t thr = tl[u].y >> 16;
                        const uint64_t o = p < thr ? ((uint64_t)tb[u].x << 2) + p : ((uint64_t)tb[u].y << 2) + (p - thr);
                        out[o] = kk[u];
                    }
                }
            }
        {
            const uint32_t l_at = t_lay[tid].y & 0x1FFFu;
#pragma unroll
            for (int j = 0; j < kCarry; ++j) kc[j] = sorted[l_at + j];
        }
        __syncthreads();
    }

    // ---- the end: carried keys and the unused tail of every open block ----
    if (sh_ovf == 0)
    {
        if (ccnt)
        {
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            if (room == 0)
            {
                const unsigned long long at = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)B);
                if (at + B > my_cap) { atomicOr(&pc->overflow, 1ULL); ccnt = 0; wpos = 0; }
                else wpos = my_start + at;
            }
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
                if ((uint32_t)j < ccnt) out[wpos + j] = kc[j];
            wpos += ccnt;
        }
        const uint32_t tail = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
        __syncthreads();
        t_base[tid].x = tail;
        reinterpret_cast<uint64_t*>(sorted)[tid] = wpos;
        __syncthreads();
        for (uint32_t d = 0; d < 256; ++d)
        {
            const uint32_t n = t_base[d].x;
            const uint64_t from = reinterpret_cast<const uint64_t*>(sorted)[d];
            for (uint32_t j = tid; j < n; j += kTB) out[from + j] = Key2{~0ULL, ~0ULL};
        }
    }
    if (NH > 0) { if (lh[tid]) atomicAdd(&pc->hist[tid], (unsigned long long)lh[tid]); }
    if (NH > 1) { if (lh[tid + 256]) atomicAdd(&pc->hist[tid + 256], (unsigned long long)lh[tid + 256]); }
    for (int o = 32; o > 0; o >>= 1) nvalid += __shfl_down(nvalid, o, 64);
    if (lane_id() == 0 && nvalid) { atomicAdd(&pc->keys_out, nvalid * S); atomicAdd(&pc->windows, nvalid); }
}

// --------------------------------------------------------------------------------------
// K4: LSD radix sort, 8-bit digits: per-tile histogram, scan (above), stable scatter
// --------------------------------------------------------------------------------------

#ifndef GOSS_SORT_ITEMS1
#define GOSS_SORT_ITEMS1 32
#endif
#ifndef GOSS_SORT_ITEMS2
#define GOSS_SORT_ITEMS2 16
#endif
#ifndef GOSS_LB_BATCH
#define GOSS_LB_BATCH 1
#endif
// Tile of the second level in sub-region mode: 5632 (u64) / 2816 (u128) keys -- 52 KB of LDS, three workgroups
// per CU.  With the tiles dealt out by XCD the shorter runs cost nothing and the third workgroup hides the LDS
// phases of the other two (C2: 41.0 -> 37.5 ms; the chained passes keep the large tile, their bound is the chain).
template <class K> struct SubCfg {
    static constexpr int kItems = sizeof(K) == 8 ? 22 : 11;
    static constexpr int kTile = 256 * kItems;
};

template <class K, bool HAS_VAL = false> struct SortCfg {
    static constexpr int kItems = sizeof(K) == 8 ? (HAS_VAL ? 16 : GOSS_SORT_ITEMS1) : (HAS_VAL ? 8 : GOSS_SORT_ITEMS2);   // keys per thread
    static constexpr int kTile = kTB * kItems;                 // 4096 (u64) / 2048 (u128) keys
};

// table layout: table[digit * ntiles + tile]
template <class K, bool HAS_VAL>
__global__ __launch_bounds__(kTB) void radix_hist_kernel(const K* __restrict__ keys, uint64_t n, uint32_t digit,
                                                         uint64_t ntiles, uint64_t* __restrict__ table)
{
    constexpr int kSortItems = SortCfg<K, HAS_VAL>::kItems;
    constexpr int kSortTile = SortCfg<K, HAS_VAL>::kTile;
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
#pragma unroll 4
    for (int j = 0; j < kSortItems; ++j)
    {
        uint64_t i = base + (uint64_t)j * kTB + threadIdx.x;
        if (i < n) atomicAdd(&hist[key_digit(keys[i], digit)], 1u);
    }
    __syncthreads();
    table[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

// Peers of this lane = lanes of the wave whose (valid) item has the same 8-bit digit.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b)
    {
        bool bit = (d >> b) & 1u;
        uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

template <class K, bool HAS_VAL>
__global__ __launch_bounds__(kTB) void radix_scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            uint64_t n, uint32_t digit, uint64_t ntiles,
                                                            const uint64_t* __restrict__ table)
{
    constexpr int kSortItems = SortCfg<K, HAS_VAL>::kItems;
    constexpr int kSortTile = SortCfg<K, HAS_VAL>::kTile;
    __shared__ uint32_t wave_hist[kWaves][256];
    __shared__ uint32_t digit_start[256];
    __shared__ uint64_t global_base[256];
    __shared__ K stage[kSortTile];
    __shared__ uint32_t vstage[HAS_VAL ? kSortTile : 1];
    __shared__ uint32_t sh_scan[kWaves + 1];

    const uint32_t tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t tile_base = (uint64_t)blockIdx.x * kSortTile;
    const uint32_t tile_n = (uint32_t)(n - tile_base < (uint64_t)kSortTile ? n - tile_base : (uint64_t)kSortTile);

#pragma unroll
    for (int i = 0; i < kWaves; ++i) wave_hist[i][tid] = 0;
    __syncthreads();

    K key[kSortItems];
    uint32_t val[HAS_VAL ? kSortItems : 1];
    uint16_t rank[kSortItems];
    const uint32_t wbase = w * 64 * kSortItems;
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;

#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
    {
        uint32_t li = wbase + r * 64 + lane;
        bool valid = li < tile_n;
        if (valid)
        {
            key[r] = keys_in[tile_base + li];
            if (HAS_VAL) val[r] = vals_in[tile_base + li];
        }
    }
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
    {
        uint32_t li = wbase + r * 64 + lane;
        bool valid = li < tile_n;
        uint32_t d = valid ? key_digit(key[r], digit) : 0u;
        uint64_t peers = match_digit(d, valid);
        uint32_t before = __popcll(peers & lt_mask);
        uint32_t base = 0;
        lds_vu32 wh = (lds_vu32)wave_hist[w];
        if (valid) base = wh[d];
        // all reads of this round happen before the leader's update (same wave, in order)
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) wh[d] = base + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[r] = (uint16_t)(base + before);
    }
    __syncthreads();

    // per digit: exclusive prefix over waves, tile totals, exclusive scan over digits
    {
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < kWaves; ++i)
        {
            uint32_t c = wave_hist[i][tid];
            wave_hist[i][tid] = tot;
            tot += c;
        }
        uint32_t tile_total;
        uint32_t start = block_excl_scan<uint32_t>(tot, sh_scan, &tile_total);
        digit_start[tid] = start;
        global_base[tid] = table[(uint64_t)tid * ntiles + blockIdx.x] - start;
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
    {
        uint32_t li = wbase + r * 64 + lane;
        if (li < tile_n)
        {
            uint32_t d = key_digit(key[r], digit);
            uint32_t pos = digit_start[d] + wave_hist[w][d] + rank[r];
            stage[pos] = key[r];
            if (HAS_VAL) vstage[pos] = val[r];
        }
    }
    __syncthreads();

    for (uint32_t i = tid; i < tile_n; i += kTB)
    {
        K k = stage[i];
        uint64_t o = global_base[key_digit(k, digit)] + i;
        keys_out[o] = k;
        if (HAS_VAL) vals_out[o] = vstage[i];
    }
}

// --------------------------------------------------------------------------------------
// K4 single-pass form: global digit histograms once, then a scatter whose tile offsets come
// from a chained scan with decoupled look-back (no per-tile histogram table, no second read
// of the keys).
// --------------------------------------------------------------------------------------

// hist[p * 256 + d] += number of keys whose digit at bit (first_shift + 8p) is d, for
// p < npass (npass <= 16).  Persistent grid: every workgroup accumulates in LDS over many
// tiles and flushes once.
template <class K>
__global__ __launch_bounds__(kTB) void global_hist_kernel(const K* __restrict__ keys, uint64_t n, uint32_t first_shift,
                                                          uint32_t npass, unsigned long long* __restrict__ hist)
{
    __shared__ uint32_t lh[16 * 256];
    for (uint32_t i = threadIdx.x; i < npass * 256; i += kTB) lh[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kTB;
    uint32_t since_flush = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += stride)
    {
        K k = keys[i];
        for (uint32_t p = 0; p < npass; ++p) atomicAdd(&lh[p * 256 + key_digit(k, first_shift + 8 * p)], 1u);
        // a 32-bit LDS bin cannot overflow before 2^32 keys have gone through this workgroup
        (void)since_flush;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < npass * 256; i += kTB)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// In-place exclusive scan of each 256-entry row (one workgroup per row).
__global__ __launch_bounds__(kTB) void scan_rows256_kernel(unsigned long long* __restrict__ hist)
{
    __shared__ uint64_t sh[kWaves + 1];
    uint64_t v = hist[blockIdx.x * 256 + threadIdx.x];
    uint64_t tot;
    uint64_t ex = block_excl_scan<uint64_t>(v, sh, &tot);
    hist[blockIdx.x * 256 + threadIdx.x] = ex;
}

struct LookbackCtl {
    uint32_t ticket;       // next tile number
    uint32_t error;        // a look-back spin gave up (never expected)
    // diagnostics (GOSS_LB_STATS builds only): per-tile sums recorded by digit 0's thread
    unsigned long long walk_steps, spin_polls, max_depth, tiles;
};

constexpr uint64_t kLbFlagAgg = 1ULL << 62;      // tile's own count is published
constexpr uint64_t kLbFlagPrefix = 2ULL << 62;   // inclusive prefix up to this tile is published
constexpr uint64_t kLbValueMask = (1ULL << 62) - 1;

// GAPPED: the input is the output of extract1_part_kernel -- 256 bucket regions with unused
// slots between them (GapTable); tile t is the (t - tile_first[b])-th tile of bucket b.  Every
// tile then lies inside one bucket of the previous digit, so no tile needs a stable rank.
template <class K, bool HAS_VAL, bool ORDERED, bool GAPPED = false, int ITEMS = SortCfg<K, HAS_VAL>::kItems>
__global__ __launch_bounds__(kTB) void radix_onesweep_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                             K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                             uint64_t n, uint32_t digit, uint32_t sorted_lo,
                                                             const unsigned long long* __restrict__ bucket_base,
                                                             unsigned long long* __restrict__ status,
                                                             LookbackCtl* __restrict__ ctl,
                                                             unsigned long long* __restrict__ cursors,
                                                             const GapTable* __restrict__ gt = nullptr,
                                                             const SubTable* __restrict__ sub = nullptr,
                                                             uint32_t rem_out = 0)
{
    // rem_out (two-word keys, sub-region mode): slot o of keys_out is a 12-byte record holding the key's low
    // `digit` bits -- what is left below the 16-bit segment prefix, which the slot's sub-region implies; the
    // counting kernel of 96-bit remainders reads those (a quarter less to write here and to read there)
    constexpr int kSortItems = ITEMS;
    constexpr int kSortTile = kTB * ITEMS;
    __shared__ uint32_t wave_hist[kWaves][256];
    __shared__ uint32_t digit_start[256];
    __shared__ uint64_t global_base[256];
    __shared__ K stage[kSortTile];
    __shared__ uint32_t vstage[HAS_VAL ? kSortTile : 1];
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ uint32_t sh_tile;
    __shared__ uint32_t sh_bucket;
    __shared__ uint32_t sh_skip;
    __shared__ uint32_t sh_total;

    const uint32_t tid = threadIdx.x, lane = lane_id(), w = wave_id();
    // Tile number.  ORDERED: a ticket (one returning atomic per tile: every lower-numbered tile
    // has then started, so the chain cannot stall, but a single word serves only ~88 M
    // tickets/s chip-wide).  Otherwise blockIdx.x: the dispatcher starts workgroups in
    // blockIdx order in practice; HIP does not promise it, so the look-back spin is bounded and
    // a give-up makes the host redo the pass with the histogram-table kernels.
    if (ORDERED) { if (tid == 0) sh_tile = atomicAdd(&ctl->ticket, 1u); }
    if (tid == 0) sh_skip = 0;
    // Sub-region mode (no chain, any tile order): workgroups go round the eight XCDs in blockIdx order, and each
    // XCD has its own L2.  Tile = (blockIdx % 8) * tiles/8 + blockIdx / 8 gives every XCD a contiguous range of
    // tiles, i.e. its own bucket regions: the runs that consecutive tiles append to a sub-region then pass through
    // ONE L2, where the partial 64-byte granules at their seams can meet.  (The grid is rounded up to a multiple of 8.)
    uint32_t my_tile = blockIdx.x;
    if (GAPPED && !ORDERED && sub)
    {
        const uint32_t total = (uint32_t)gt->tile_first[256];
#ifndef GOSS_K2_NO_XCD
        const uint32_t per = (total + 7u) / 8u;
        my_tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
#endif
        if (my_tile >= total) return;
    }
    if (GAPPED && !ORDERED)
    {
        const unsigned long long t = my_tile;
        if (gt->tile_first[tid] <= t && t < gt->tile_first[tid + 1]) sh_bucket = tid;
    }
#pragma unroll
    for (int i = 0; i < kWaves; ++i) wave_hist[i][tid] = 0;
    __syncthreads();
    const uint32_t tile = ORDERED ? sh_tile : my_tile;
    uint64_t tile_base = (uint64_t)tile * kSortTile;
    uint32_t tile_n = 0;
    if (GAPPED)
    {
        const uint32_t b = sh_bucket;
        const uint64_t j = (uint64_t)tile - gt->tile_first[b];
        const uint64_t left = gt->cnt[b] - j * kSortTile;
        tile_base = gt->reg_start[b] + j * kSortTile;
        tile_n = (uint32_t)(left < (uint64_t)kSortTile ? left : (uint64_t)kSortTile);
    }
    else tile_n = (uint32_t)(n - tile_base < (uint64_t)kSortTile ? n - tile_base : (uint64_t)kSortTile);

    K key[kSortItems];
    uint32_t val[HAS_VAL ? kSortItems : 1];
    uint16_t rank[kSortItems];
    const uint32_t wbase = w * 64 * kSortItems;
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;
    // GAPPED: slots of a bucket region that hold no key (the padding of extract1_part_kernel's
    // last blocks) are skipped: bit r of `have` = item r of this thread is a key
    uint32_t have = 0;
    static_assert(kSortItems <= 32, "one validity bit per item");

#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
    {
        uint32_t li = wbase + r * 64 + lane;
        if (li < tile_n)
        {
            key[r] = keys_in[tile_base + li];
            if (HAS_VAL) val[r] = vals_in[tile_base + li];
            have |= 1u << r;
        }
    }
    // (a second loop: looking at a key inside the load loop would wait for every load in turn)
    if (GAPPED)
    {
#pragma unroll
        for (int r = 0; r < kSortItems; ++r)
            if (((have >> r) & 1u) && is_pad_key(key[r])) have &= ~(1u << r);
    }
    // Does this tile need a STABLE rank?  Stability only matters when the tile holds keys
    // that differ in the bits the previous passes sorted (bits [sorted_lo, digit)): a tile whose
    // keys all share them -- almost every tile of the second partition pass -- may be ranked in
    // any order, which costs one LDS atomic per key instead of eight ballots.
    bool stable = false;
    if (!GAPPED && digit > sorted_lo)
    {
        const uint32_t nb = digit - sorted_lo;
        if (nb > 56) stable = true;
        else
        {
            const uint64_t fmask = (1ULL << nb) - 1;
            const uint64_t first = key_shr64(keys_in[tile_base], sorted_lo) & fmask;
            uint64_t diff = 0;
#pragma unroll
            for (int r = 0; r < kSortItems; ++r)
            {
                uint32_t li = wbase + r * 64 + lane;
                if (li < tile_n) diff |= (key_shr64(key[r], sorted_lo) & fmask) ^ first;
            }
            stable = __syncthreads_or(diff != 0);
        }
    }
    if (stable)
    {
#pragma unroll
        for (int r = 0; r < kSortItems; ++r)
        {
            bool valid = (have >> r) & 1u;
            uint32_t d = valid ? key_digit(key[r], digit) : 0u;
            uint64_t peers = match_digit(d, valid);
            uint32_t before = __popcll(peers & lt_mask);
            uint32_t base = 0;
            lds_vu32 wh = (lds_vu32)wave_hist[w];
            if (valid) base = wh[d];
            __builtin_amdgcn_wave_barrier();
            if (valid && before == 0) wh[d] = base + __popcll(peers);
            __builtin_amdgcn_wave_barrier();
            rank[r] = (uint16_t)(base + before);
        }
    }
    else
    {
#pragma unroll
        for (int r = 0; r < kSortItems; ++r)
            if ((have >> r) & 1u) rank[r] = (uint16_t)atomicAdd(&wave_hist[0][key_digit(key[r], digit)], 1u);
    }
    __syncthreads();

    {
        // thread tid owns digit tid
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < kWaves; ++i)
        {
            uint32_t c = wave_hist[i][tid];
            wave_hist[i][tid] = stable ? tot : 0u;     // unstable ranks are tile-wide already
            tot += c;
        }
        unsigned long long* mine = status + (uint64_t)tile * 256 + tid;
        uint64_t excl = 0;
        const bool chain = !cursors && tile != 0;
        uint64_t sub_start = 0;
        if (GAPPED && sub)
        {
            // sub-region mode: the tile's keys of low digit tid go to sub-region (bucket, tid)
            const uint32_t sidx = sh_bucket * 256u + tid;
            excl = tot ? atomicAdd(&cursors[(uint64_t)sidx * kSubCursorStride], (unsigned long long)tot) : 0ULL;
            sub_start = sub->start[sidx];
            // too small a sub-region: nothing of this tile is stored, the host redoes the chunk
            if (tot && excl + tot > sub->cap[sidx]) { atomicOr(&ctl->error, 2u); sh_skip = 1; }
        }
        else if (cursors)
        {
            // first pass of a sort: the order of tiles inside a bucket is irrelevant, so the
            // tile just reserves its share of every bucket with one atomic per digit (cursor
            // words 256 B apart: separate lines and channels) -- no chain, no waiting
            excl = tot ? atomicAdd(&cursors[tid * kCursorStride], (unsigned long long)tot) : 0ULL;
        }
        else if (tile == 0)
            __hip_atomic_store(mine, kLbFlagPrefix | (uint64_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            __hip_atomic_store(mine, kLbFlagAgg | (uint64_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t tile_total;
        const uint32_t start = block_excl_scan<uint32_t>(tot, sh_scan, &tile_total);
        digit_start[tid] = start;
        if (tid == 0) sh_total = tile_total;
        __syncthreads();

        // the keys go to their sorted place in LDS before the look-back: that work needs only the
        // tile's own counts, and the predecessors get time to publish theirs
#pragma unroll
        for (int r = 0; r < kSortItems; ++r)
        {
            if ((have >> r) & 1u)
            {
                uint32_t d = key_digit(key[r], digit);
                // unstable ranks are tile-wide already: no per-wave offset to read
                uint32_t pos = digit_start[d] + rank[r] + (stable ? wave_hist[w][d] : 0u);
                stage[pos] = key[r];
                if (HAS_VAL) vstage[pos] = val[r];
            }
        }

        if (chain)
        {
            // Walk back over the predecessors kLbBatch tiles at a time: the loads of one batch
            // are independent, so a deep walk costs one memory latency per batch instead of
            // one per tile.
            constexpr int kLbBatch = GOSS_LB_BATCH;
            int64_t t = (int64_t)tile - 1;
            uint32_t spins = 0;
            bool found = false;
            while (!found)
            {
                unsigned long long v[kLbBatch];
#pragma unroll
                for (int j = 0; j < kLbBatch; ++j)
                {
                    int64_t tj = t - j;
                    v[j] = tj >= 0 ? __hip_atomic_load(status + (uint64_t)tj * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : kLbFlagPrefix;       // before tile 0: an empty prefix
                }
                int used = 0;
#pragma unroll
                for (int j = 0; j < kLbBatch; ++j)
                {
                    if (found || used != j) continue;     // stopped at an unpublished tile
                    uint64_t f = v[j] & ~kLbValueMask;
                    if (f == 0) continue;
                    excl += v[j] & kLbValueMask;
                    used = j + 1;
                    if (f == kLbFlagPrefix) found = true;
                }
                t -= used;
#if defined(GOSS_LB_STATS)
                if (tid == 0) { atomicAdd(&ctl->walk_steps, (unsigned long long)used); }
#endif
                if (!found && used < kLbBatch)
                {
                    if (++spins > (1u << 20)) { atomicOr(&ctl->error, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#if defined(GOSS_LB_STATS)
            if (tid == 0)
            {
                atomicAdd(&ctl->spin_polls, (unsigned long long)spins);
                atomicMax(&ctl->max_depth, (unsigned long long)((int64_t)tile - 1 - t));
                atomicAdd(&ctl->tiles, 1ULL);
            }
#endif
            __hip_atomic_store(mine, kLbFlagPrefix | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        global_base[tid] = ((GAPPED && sub) ? sub_start : bucket_base[tid]) + excl - start;
    }
    __syncthreads();

    if (GAPPED && sh_skip) return;
    const uint32_t tile_keys = GAPPED ? sh_total : tile_n;      // padding slots hold no key
    if constexpr (sizeof(K) == 16)
    {
        if (rem_out)
        {
            const uint32_t hb = digit > 64 ? digit - 64 : 0;
            const uint32_t hmask = hb >= 32 ? 0xFFFFFFFFu : ((1u << hb) - 1u);
            const uint64_t lmask64 = digit >= 64 ? ~0ULL : ((1ULL << digit) - 1ULL);
            Rem96* out96 = reinterpret_cast<Rem96*>(keys_out);
            for (uint32_t i = tid; i < tile_keys; i += kTB)
            {
                const K k = stage[i];
                const uint64_t o = global_base[key_digit(k, digit)] + i;
                const uint64_t lo = key_lo_word(k) & lmask64;
                out96[o] = Rem96{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)key_hi_word(k) & hmask};
            }
            return;
        }
    }
    for (uint32_t i = tid; i < tile_keys; i += kTB)
    {
        K k = stage[i];
        uint64_t o = global_base[key_digit(k, digit)] + i;
        keys_out[o] = k;
        if (HAS_VAL) vals_out[o] = vstage[i];
    }
}

// out[q] = first index of the sorted array whose key is >= query[q] (one thread per query).
template <class K>
__global__ void lower_bound_keys_kernel(const K* __restrict__ keys, uint64_t n, const K* __restrict__ query, uint32_t nq,
                                        uint64_t* __restrict__ out)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const K x = query[q];
    uint64_t a = 0, b = n;
    while (a < b)
    {
        const uint64_t mid = a + ((b - a) >> 1);
        if (keys[mid] < x) a = mid + 1; else b = mid;
    }
    out[q] = a;
}

// Segment bounds of the sub-region layout: segment s holds cursors[s] keys from start[s].
__global__ void sub_bounds_kernel(const SubTable* __restrict__ sub, const unsigned long long* __restrict__ cursors,
                                  uint64_t* __restrict__ seg_beg, uint64_t* __restrict__ seg_end)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 65536u) return;
    seg_beg[s] = sub->start[s];
    seg_end[s] = sub->start[s] + cursors[(uint64_t)s * kSubCursorStride];
}

// --------------------------------------------------------------------------------------
// K5: run compaction (merge equal adjacent keys)
// --------------------------------------------------------------------------------------

constexpr int kRedItems = 16;
constexpr int kRedTile = kTB * kRedItems;

template <class K>
__global__ __launch_bounds__(kTB) void heads_count_kernel(const K* __restrict__ keys, uint64_t n,
                                                          uint64_t* __restrict__ tile_counts)
{
    __shared__ uint32_t sh[kWaves + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kRedTile;
    uint32_t c = 0;
#pragma unroll 4
    for (int j = 0; j < kRedItems; ++j)
    {
        uint64_t i = base + (uint64_t)j * kTB + threadIdx.x;
        if (i < n) c += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    }
    uint32_t tot;
    block_excl_scan<uint32_t>(c, sh, &tot);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// ---- multiplicity spectrum of a slice of the key space (distinct-key estimate) -----------------------
// Keys whose mixed bits are 0 under qmask: ALL copies of a key are kept or dropped together, so the kept
// keys are an unbiased 1 / (qmask + 1) slice of the key space with its multiplicities intact.
__device__ __forceinline__ uint32_t slice_mix(const Key1& k) { return (uint32_t)((k.lo * 0x9E3779B97F4A7C15ULL) >> 40); }
__device__ __forceinline__ uint32_t slice_mix(const Key2& k) { return (uint32_t)(((k.lo ^ (k.hi * 0xC2B2AE3D27D4EB4FULL)) * 0x9E3779B97F4A7C15ULL) >> 40); }

template <class K>
__global__ __launch_bounds__(kTB) void slice_filter_kernel(const K* __restrict__ keys, uint64_t n, uint32_t qmask, K* __restrict__ out,
                                                           unsigned long long* __restrict__ counter, uint64_t cap)
{
    // kept keys are collected in LDS and leave in batches: ONE global atomic per ~800 kept keys (a returning
    // atomic per wave on the one counter word ran at 88 M/s: 31 ms for a 196 M-key sample)
    constexpr uint32_t kBuf = 1024;
    __shared__ K buf[kBuf];
    __shared__ uint32_t fill;
    __shared__ unsigned long long gbase;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) fill = 0;
    __syncthreads();
    auto flush = [&]() {
        const uint32_t cnt = fill;
        if (tid == 0) gbase = atomicAdd(counter, (unsigned long long)cnt);
        __syncthreads();
        const unsigned long long g = gbase;
        for (uint32_t j = tid; j < cnt; j += kTB)
            if (g + j < cap) out[g + j] = buf[j];
        __syncthreads();
        if (tid == 0) fill = 0;
        __syncthreads();
    };
    const uint64_t stride = (uint64_t)gridDim.x * kTB;
    const uint64_t rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r)
    {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * kTB + tid;
        K k{};
        bool keep = false;
        if (i < n) { k = keys[i]; keep = (slice_mix(k) & qmask) == 0u; }
        const uint64_t m = __ballot(keep);
        uint32_t wbase = 0;
        if (m != 0)
        {
            if (lane_id() == 0) wbase = atomicAdd(&fill, (uint32_t)__popcll(m));
            wbase = __shfl(wbase, 0, 64);
        }
        if (keep) buf[wbase + (uint32_t)__popcll(m & ((1ULL << lane_id()) - 1ULL))] = k;
        __syncthreads();
        if (fill > kBuf - kTB) flush();             // (the same value for every thread: read behind the barrier)
    }
    flush();
}

// sorted keys -> f[0] = distinct keys, f[1..3] = keys that occur exactly once / twice / three times
template <class K>
__global__ __launch_bounds__(kTB) void spectrum_kernel(const K* __restrict__ keys, uint64_t n, unsigned long long* __restrict__ f)
{
    __shared__ uint32_t sh[4];
    if (threadIdx.x < 4) sh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kTB;
    for (uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += stride)
    {
        if (i != 0 && keys[i] == keys[i - 1]) continue;
        uint32_t len = 1;
        while (len < 4 && i + len < n && keys[i + len] == keys[i]) ++len;
        atomicAdd(&sh[0], 1u);
        if (len < 4) atomicAdd(&sh[len], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 4 && sh[threadIdx.x]) atomicAdd(&f[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

// Writes distinct keys and the index at which each run starts.  tile_offsets = exclusive
// scan of tile_counts.  Row j of the tile is the 256 consecutive keys base + j*256 + tid
// (coalesced); output order is (row, wave, lane) = index order.
template <class K>
__global__ __launch_bounds__(kTB) void heads_write_kernel(const K* __restrict__ keys, uint64_t n,
                                                          const uint64_t* __restrict__ tile_offsets,
                                                          K* __restrict__ out_keys, uint64_t* __restrict__ starts)
{
    __shared__ uint32_t cnt[kRedItems * kWaves];
    const uint64_t base = (uint64_t)blockIdx.x * kRedTile;
    const uint32_t lane = lane_id(), w = wave_id();
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;
    K key[kRedItems];
    uint32_t flags = 0;
#pragma unroll
    for (int j = 0; j < kRedItems; ++j)
    {
        uint64_t i = base + (uint64_t)j * kTB + threadIdx.x;
        bool head = false;
        if (i < n)
        {
            key[j] = keys[i];
            head = (i == 0) || (key[j] != keys[i - 1]);
        }
        if (head) flags |= 1u << j;
        uint64_t bal = __ballot(head);
        if (lane == 0) cnt[j * kWaves + w] = __popcll(bal);
    }
    __syncthreads();
    if (threadIdx.x < 64)
    {
        uint32_t c = cnt[threadIdx.x];
        uint32_t inc = wave_incl_scan(c);
        cnt[threadIdx.x] = inc - c;
    }
    __syncthreads();
    const uint64_t tile_off = tile_offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kRedItems; ++j)
    {
        bool head = (flags >> j) & 1u;
        uint64_t bal = __ballot(head);
        if (head)
        {
            uint64_t o = tile_off + cnt[j * kWaves + w] + __popcll(bal & lt_mask);
            out_keys[o] = key[j];
            starts[o] = base + (uint64_t)j * kTB + threadIdx.x;
        }
    }
}

// counts[j] = starts[j+1] - starts[j] (run length), last run ends at n.
__global__ void run_lengths_kernel(const uint64_t* __restrict__ starts, uint64_t m, uint64_t n,
                                   uint32_t* __restrict__ counts, uint32_t* __restrict__ overflow)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    uint64_t e = j + 1 < m ? starts[j + 1] : n;
    uint64_t c = e - starts[j];
    if (c >= 0xFFFFFFFFULL) { atomicOr(overflow, 1u); c = 0xFFFFFFFFULL; }       // 0xFFFFFFFF is the marker of a count kept elsewhere
    counts[j] = (uint32_t)c;
}

// weighted form: counts[j] = sum of vals over the run (runs are short: <= number of merged
// sorted runs), used when merging (key,count) runs.
__global__ void run_sums_kernel(const uint64_t* __restrict__ starts, uint64_t m, uint64_t n,
                                const uint32_t* __restrict__ vals, uint32_t* __restrict__ counts,
                                uint32_t* __restrict__ overflow)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    uint64_t e = j + 1 < m ? starts[j + 1] : n;
    uint64_t s = 0;
    for (uint64_t i = starts[j]; i < e; ++i) s += vals[i];
    if (s >= 0xFFFFFFFFULL) { atomicOr(overflow, 1u); s = 0xFFFFFFFFULL; }       // (also a single marker entry: its exact count moves on)
    counts[j] = (uint32_t)s;
}

// Counts that do not fit 32 bits (graph mode keeps them: the reference's histogram key is the u64 count,
// Graph.hh:101-106).  A saturated count is the marker 0xFFFFFFFF; the host resolves the few keys that
// carry it with these two kernels: where they are, and what their entries in the merged inputs add up to.
__global__ void find_saturated_kernel(const uint32_t* __restrict__ counts, uint64_t m, unsigned long long* __restrict__ out, uint32_t cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || counts[i] != 0xFFFFFFFFu) return;
    const unsigned long long at = atomicAdd(&out[0], 1ULL);
    if (at < cap) out[1 + at] = i;
}
// For query q: over every input run r (entries [run_off[r], run_off[r+1]) sorted by key, equal keys
// adjacent) the sum of the values of the entries equal to it -- their number when vals is NULL (raw
// keys) -- leaving out marker values, which are counted in markers[q] instead.
template <class K>
__global__ void sum_equal_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ vals, const uint64_t* __restrict__ run_off,
                                 uint32_t nruns, const K* __restrict__ queries, uint32_t nq,
                                 unsigned long long* __restrict__ sums, unsigned long long* __restrict__ markers)
{
    const uint32_t q = blockIdx.x, r = threadIdx.x;
    if (q >= nq || r >= nruns) return;
    const K key = queries[q];
    uint64_t a = run_off[r], b = run_off[r + 1];
    const uint64_t end = b;
    while (a < b) { const uint64_t mid = a + ((b - a) >> 1); if (keys[mid] < key) a = mid + 1; else b = mid; }
    unsigned long long s = 0, mk = 0;
    uint64_t hi = a, top = end;
    // upper bound
    while (hi < top) { const uint64_t mid = hi + ((top - hi) >> 1); if (key < keys[mid]) top = mid; else hi = mid + 1; }
    if (!vals) s = hi - a;
    else
        for (uint64_t i = a; i < hi; ++i)
        {
            const uint32_t v = vals[i];
            if (v == 0xFFFFFFFFu) ++mk; else s += v;
        }
    if (s) atomicAdd(&sums[q], s);
    if (mk) atomicAdd(&markers[q], mk);
}
__global__ void patch_counts_kernel(uint32_t* __restrict__ counts, const unsigned long long* __restrict__ idx,
                                    const uint32_t* __restrict__ values, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[idx[i]] = values[i];
}

// graph-to-kmer-set (GossCmdGraphToKmerSet.cc:40-55): an edge stays when it is its own canonical
// form -- Gossamer::edge_type::isNormal, RankSelect.hh:117-124, the same hash order and tie rule
// as normalize.  Writes 1 / 0 over the item's count; the count filter below does the rest.
template <class K>
__global__ __launch_bounds__(kTB) void mark_normal_kernel(const K* __restrict__ keys, uint64_t n, uint32_t len,
                                                          uint32_t* __restrict__ counts)
{
    const uint64_t i = (uint64_t)unit_block() * kTB + threadIdx.x;
    if (i >= n) return;
    const K x = keys[i];
    const K rc = revcomp(x, len);
    counts[i] = canonical<K>(x, rc) == x ? 1u : 0u;
}

// Selection by count: keeps the (key,count) items with lo <= count <= hi, order preserved.
// The set algebra of intersect-kmer-sets / subtract-kmer-set is a merge of weighted runs
// followed by this filter (GossCmdIntersectKmerSets.cc:29-79, GossCmdSubtractKmerSet.cc:47-66).
__global__ __launch_bounds__(kTB) void select_count_kernel(const uint32_t* __restrict__ counts, uint64_t n,
                                                           uint32_t lo, uint32_t hi, uint64_t* __restrict__ tile_counts)
{
    __shared__ uint32_t sh[kWaves + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kRedTile;
    uint32_t c = 0;
#pragma unroll 4
    for (int j = 0; j < kRedItems; ++j)
    {
        uint64_t i = base + (uint64_t)j * kTB + threadIdx.x;
        if (i < n) { uint32_t v = counts[i]; c += (v >= lo && v <= hi) ? 1u : 0u; }
    }
    uint32_t tot;
    block_excl_scan<uint32_t>(c, sh, &tot);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

template <class K>
__global__ __launch_bounds__(kTB) void select_write_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ counts,
                                                           uint64_t n, uint32_t lo, uint32_t hi,
                                                           const uint64_t* __restrict__ tile_offsets,
                                                           K* __restrict__ out_keys, uint32_t* __restrict__ out_counts)
{
    __shared__ uint32_t cnt[kRedItems * kWaves];
    const uint64_t base = (uint64_t)blockIdx.x * kRedTile;
    const uint32_t lane = lane_id(), w = wave_id();
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;
    K key[kRedItems];
    uint32_t val[kRedItems];
    uint32_t flags = 0;
#pragma unroll
    for (int j = 0; j < kRedItems; ++j)
    {
        uint64_t i = base + (uint64_t)j * kTB + threadIdx.x;
        bool keep = false;
        if (i < n)
        {
            key[j] = keys[i];
            val[j] = counts[i];
            keep = val[j] >= lo && val[j] <= hi;
        }
        if (keep) flags |= 1u << j;
        uint64_t bal = __ballot(keep);
        if (lane == 0) cnt[j * kWaves + w] = __popcll(bal);
    }
    __syncthreads();
    if (threadIdx.x < 64)
    {
        uint32_t c = cnt[threadIdx.x];
        uint32_t inc = wave_incl_scan(c);
        cnt[threadIdx.x] = inc - c;
    }
    __syncthreads();
    const uint64_t tile_off = tile_offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kRedItems; ++j)
    {
        bool keep = (flags >> j) & 1u;
        uint64_t bal = __ballot(keep);
        if (keep)
        {
            uint64_t o = tile_off + cnt[j * kWaves + w] + __popcll(bal & lt_mask);
            out_keys[o] = key[j];
            out_counts[o] = val[j];
        }
    }
}

// One bit per item: bit i = (counts[i] & mask) != 0, WordyBitVector word layout (bit b of word w
// = position 64w+b).  One wave per 64 words: lane l ballots item (word*64 + l).
__global__ __launch_bounds__(256) void count_bits_kernel(const uint32_t* __restrict__ counts, uint64_t n, uint32_t mask,
                                                         uint64_t* __restrict__ words, uint64_t nwords)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t w0 = wave * 64;
    if (w0 >= nwords) return;
    uint64_t mine = 0;
    for (uint32_t j = 0; j < 64; ++j)
    {
        uint64_t w = w0 + j;
        if (w >= nwords) break;                      // uniform across the wave
        uint64_t i = w * 64 + lane;
        bool bit = i < n && (counts[i] & mask) != 0;
        uint64_t bal = __ballot(bit);
        if (lane == j) mine = bal;
    }
    if (w0 + lane < nwords) words[w0 + lane] = mine;
}

// --------------------------------------------------------------------------------------
// Text form of an object (dump-kmer-set / dump-graph) and the graph self-check (lint-graph)
// --------------------------------------------------------------------------------------

// kmerToString (RankSelect.hh:299-308): base j of a len-mer, first base = most significant.
template <class K>
__device__ inline uint8_t key_base_char(const K& k, uint32_t len, uint32_t j)
{
    const uint32_t code = (uint32_t)key_shr64(k, 2u * (len - 1u - j)) & 3u;
    return (uint8_t)((0x54474341u >> (8u * code)) & 0xFFu);          // "ACGT"
}

// One line per k-mer: len bases + '\n' (GossCmdDumpKmerSet.cc:47-53).  One thread per byte.
template <class K>
__global__ void dump_kmers_kernel(const K* __restrict__ keys, uint64_t m, uint32_t len, uint8_t* __restrict__ out)
{
    const uint64_t stride = len + 1u, total = m * stride;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t i = idx / stride;
        const uint32_t j = (uint32_t)(idx - i * stride);
        out[idx] = j == len ? (uint8_t)'\n' : key_base_char(keys[i], len, j);
    }
}

__device__ inline uint32_t dec_digits(uint32_t v)
{
    uint32_t d = 1;
    while (v >= 10u) { v /= 10u; ++d; }
    return d;
}

// Bytes of the line "<len bases>\t<count>\n" (GossCmdDumpGraph.cc:53-61).
__global__ void dump_line_len_kernel(const uint32_t* __restrict__ counts, uint64_t m, uint32_t len, uint64_t* __restrict__ lens)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) lens[i] = (uint64_t)len + 2u + dec_digits(counts[i]);
}

template <class K>
__global__ void dump_edges_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ counts,
                                  const uint64_t* __restrict__ offsets, uint64_t m, uint32_t len, uint8_t* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const K k = keys[i];
    uint8_t* p = out + offsets[i];
    for (uint32_t j = 0; j < len; ++j) p[j] = key_base_char(k, len, j);
    p[len] = (uint8_t)'\t';
    uint32_t v = counts[i];
    const uint32_t nd = dec_digits(v);
    for (uint32_t d = nd; d-- > 0;) { p[len + 1u + d] = (uint8_t)('0' + v % 10u); v /= 10u; }
    p[len + 1u + nd] = (uint8_t)'\n';
}

// lint-graph pass 1 (GossCmdLintGraph.cc:131-199) over the decoded edge list: every edge must
// have its reverse complement in the graph (accessAndRank = binary search here), with the same
// multiplicity (or, in an asymmetric graph, not both zero); multiplicities must be positive;
// and the list itself must be strictly increasing (what pass 2's iterator/rank agreement rests on).
struct LintReport {
    unsigned long long missing_rc, count_mismatch, zero_count, order_violation;
    uint32_t nexamples, pad;
    unsigned long long ex_index[32], ex_other[32];
    uint32_t ex_kind[32];
};

template <class K>
__global__ void lint_edges_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ counts, uint64_t m, uint32_t len,
                                  int asymmetric, LintReport* __restrict__ rep)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const K e = keys[i];
    const uint32_t mult = counts[i];
    uint32_t kind = 0;
    uint64_t other = ~0ULL;
    if (i > 0 && !(keys[i - 1] < e)) { atomicAdd(&rep->order_violation, 1ULL); kind = 4; }
    const K rc = revcomp(e, len);
    uint64_t lo = 0, hi = m;
    while (lo < hi)
    {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < rc) lo = mid + 1; else hi = mid;
    }
    if (lo >= m || keys[lo] != rc) { atomicAdd(&rep->missing_rc, 1ULL); kind = 1; }
    else
    {
        other = lo;
        const uint32_t mp = counts[lo];
        if (asymmetric) { if (mult == 0 && mp == 0) { atomicAdd(&rep->count_mismatch, 1ULL); kind = 2; } }
        else
        {
            if (mult != mp) { atomicAdd(&rep->count_mismatch, 1ULL); kind = 2; }
            if (mult == 0) { atomicAdd(&rep->zero_count, 1ULL); if (!kind) kind = 3; }
        }
    }
    if (kind)
    {
        uint32_t slot = atomicAdd(&rep->nexamples, 1u);
        if (slot < 32u) { rep->ex_index[slot] = i; rep->ex_other[slot] = other; rep->ex_kind[slot] = kind; }
    }
}

// --------------------------------------------------------------------------------------
// K3/K5 fast path: per-segment counting in an LDS hash table
// --------------------------------------------------------------------------------------
//
// After two partition passes on the top 16 key bits the keys of one segment (equal top 16
// bits) are contiguous.  One workgroup streams a segment through an open-addressing table
// held in LDS (64-bit CAS on the key, 32-bit add on the count), then sorts the table with a
// bitonic network and appends (key,count) pairs to a staging area; segment order is restored
// by a gather.  This replaces the remaining radix passes whenever a segment has at most
// kSegLimit distinct keys -- the high-coverage regime of read sets.  A segment that exceeds
// the limit raises a flag and the caller falls back to the full LSD sort.

constexpr int kSegBits = 16;                 // default number of partition bits
constexpr int kSegBitsMax = 24;
constexpr int kSegSlots = 4096;              // one-word keys: 48 KB of LDS per workgroup
constexpr int kSegLimit = 3072;
constexpr int kSegSlots2 = 2048;             // two-word keys: 40 KB
constexpr int kSegLimit2 = 1536;

// seg_off[s] = first index whose top-`segbits` value is >= s (s = 0..nseg).
template <class K>
__global__ void seg_bounds_kernel(const K* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t nseg,
                                  uint64_t* __restrict__ seg_off)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseg) return;
    uint64_t a = 0, b = n;
    while (a < b)
    {
        uint64_t mid = a + ((b - a) >> 1);
        if (key_shr64(keys[mid], shift) < (uint64_t)s) a = mid + 1; else b = mid;
    }
    seg_off[s] = a;
}

struct SegOut {
    unsigned long long cursor;     // staging cursor (entries)
    uint32_t overflow;             // some segment had more distinct keys than the LDS table holds,
                                   // or the staging area is full
    uint32_t count_overflow;
    unsigned long long stage_cap;  // entries the staging area can take
};

// NT threads per workgroup, a table of SLOTS slots (a power of two) taking SLOTS * 3 / 4 distinct keys.
template <int NT, int SLOTS, bool FILTER = false>
__device__ __forceinline__ void seg_hash_reduce_body(const Key1* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                     const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so, uint64_t* __restrict__ seg_pos,
                                                     uint64_t* __restrict__ seg_cnt,
                                                     Key1* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                     uint32_t rem_bits_all, uint32_t round_bits)
{
    // round_bits > 0: workgroup v = (segment, r) streams the whole segment but counts only the keys
    // whose next round_bits bits equal r -- 2^round_bits workgroups share the reading of a segment
    // that holds more distinct keys than one table takes; (segment, r) pairs are the units of the
    // staging area and of the gather, in key order
    constexpr int kLimit = SLOTS / 4 * 3;
    constexpr int kBucketBits = SLOTS == 4096 ? 11 : SLOTS == 8192 ? 12 : SLOTS == 2048 ? 10 : -1;   // log2(SLOTS / 2)
    static_assert(kBucketBits > 0 && SLOTS % NT == 0, "table size");
    __shared__ __attribute__((aligned(16))) unsigned long long tab[SLOTS];
    __shared__ uint32_t cnt[SLOTS];
    __shared__ uint32_t ndist;
    __shared__ uint32_t ovf;
    __shared__ unsigned long long sh_base;
    const uint32_t s = unit_block(), tid = threadIdx.x;        // unit (segment, round)
    const uint32_t sseg = s >> round_bits, rnd = s & ((1u << round_bits) - 1u);
    const uint32_t rem_bits = rem_bits_all - round_bits;       // key bits below the unit's prefix
    const uint64_t b = seg_off[sseg], e = seg_end[sseg];
    if (b == e)
    {
        if (tid == 0) { seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    if (e - b > 0xFFFFFFFFULL)
    {
        // a 32-bit slot count could wrap: leave this chunk to the full sort, whose run lengths
        // saturate and report the overflow
        if (tid == 0) { atomicOr(&so->overflow, 2u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    constexpr unsigned long long kEmpty = ~0ULL;
    auto mine_only = [&](unsigned long long x) {
        return (round_bits && (((uint32_t)(x >> rem_bits)) & ((1u << round_bits) - 1u)) != rnd) ? kEmpty : x;
    };
    for (uint32_t i = tid; i < SLOTS; i += NT) { tab[i] = kEmpty; cnt[i] = 0; }
    if (tid == 0) { ndist = 0; ovf = 0; }
    __syncthreads();

    // A key has TWO buckets of two slots, both given by one 32-bit mix of its words (the top kBucketBits bits and the
    // kBucketBits below them; equal -> the neighbour).  It lives in the first that had room when it came, else in the
    // second, else in the buckets behind the second: insertion and search walk the same sequence, and a bucket that is
    // full stays full, so a key is never behind an empty slot of its sequence.  With the neighbour as second bucket
    // 1.2 % of C2's keys sat further out and took the wave-wide slow path at each of their ~126 occurrences; an
    // independent second bucket leaves a third of that.
    auto key_mix = [](unsigned long long k) -> uint32_t {
        return ((uint32_t)k ^ __builtin_rotateleft32((uint32_t)(k >> 32), 15)) * 0x9E3779B1u;
    };
    auto second_bucket = [](uint32_t f, uint32_t b1) -> uint32_t {
        const uint32_t b = (f >> (32 - 2 * kBucketBits)) & (uint32_t)(SLOTS / 2 - 1);
        return b == b1 ? ((b1 + 1u) & (uint32_t)(SLOTS / 2 - 1)) : b;
    };
    lds_vu32 vovf = (lds_vu32)&ovf;
    // kSegUnroll independent coalesced loads are issued before the first insert so that
    // enough bytes are in flight per CU to cover the HBM latency
#ifndef GOSS_SEG_UNROLL
#define GOSS_SEG_UNROLL 16
#endif
    constexpr int kSegUnroll = GOSS_SEG_UNROLL;
    if constexpr (FILTER)
    {
        // Shared segments: three of four (one of two) keys this workgroup streams belong to another workgroup.
        // Probing them as empty keys costs as many issue slots as counting them, so every wave first COMPACTS its
        // own keys: a ballot per batch row, the owners write their key to the wave's ring in LDS (no barrier: a
        // wave's LDS accesses execute in order), and whenever the ring holds kG keys per lane the wave takes them
        // out, dense, and counts them with the same two-step insert as below.
        constexpr int kG = 2, kQ = 256;
        static_assert(kQ >= 64 * kG + 128, "ring: a drain's leftover + two batch rows");
        __shared__ unsigned long long wq_all[NT / 64][kQ];
        typedef volatile __attribute__((address_space(3))) unsigned long long* lds_vu64;
        const lds_vu64 wq = (lds_vu64)wq_all[tid >> 6];
        const uint32_t lane = tid & 63u;
        const uint32_t rmask = (1u << round_bits) - 1u;
        uint32_t head = 0, tail = 0;                    // wave-uniform ring positions (free-running)
        typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
        typedef const volatile __attribute__((address_space(3))) ull2* lds_bucket_ptr;
        typedef const __attribute__((address_space(3))) ull2* lds_bucket_plain;
        const lds_bucket_ptr vt2 = (lds_bucket_ptr)tab;
        const lds_bucket_plain pt2 = (lds_bucket_plain)tab;
        auto insert_g = [&](const unsigned long long (&kq)[kG]) {
            uint32_t bkt[kG], b2[kG];
            ull2 q[kG], q2[kG];
            uint32_t pend = 0, stm = 0;
#pragma unroll
            for (int j = 0; j < kG; ++j)
            {
                const uint32_t f = key_mix(kq[j]);
                bkt[j] = f >> (32 - kBucketBits);
                b2[j] = second_bucket(f, bkt[j]);
                q[j] = pt2[bkt[j]];
                q2[j] = pt2[b2[j]];
            }
#pragma unroll
            for (int j = 0; j < kG; ++j)
            {
                const unsigned long long s0 = q[j].x, s1 = q[j].y, s2 = q2[j].x, s3 = q2[j].y;
                const uint32_t live = kq[j] != kEmpty ? 1u : 0u;
                const uint32_t h0 = s0 == kq[j] ? 1u : 0u, h1 = s1 == kq[j] ? 1u : 0u;
                const uint32_t h2 = s2 == kq[j] ? 1u : 0u, h3 = s3 == kq[j] ? 1u : 0u;
                const uint32_t hit = (h0 | h1 | h2 | h3) & live;
                const uint32_t second = h2 | h3;
                atomicAdd(&cnt[2 * (second ? b2[j] : bkt[j]) + (h1 | h3)], hit);
                const uint32_t miss = live & (hit ^ 1u);
                pend |= miss << j;
                const uint32_t full = (s0 != kEmpty ? 1u : 0u) & (s1 != kEmpty ? 1u : 0u) & miss;
                bkt[j] = full ? b2[j] : bkt[j];
                stm |= full << j;
            }
            unsigned long long key = kEmpty;
            uint32_t bk = 0, st = 0;
            for (;;)
            {
                if (key == kEmpty && pend)
                {
                    const uint32_t u = __ffs(pend) - 1;
                    pend &= pend - 1;
                    st = (stm >> u) & 1u;
#pragma unroll
                    for (int uu = 0; uu < kG; ++uu)
                        if (u == (uint32_t)uu) { key = kq[uu]; bk = bkt[uu]; }
                }
                if (!__ballot(key != kEmpty)) break;
                if (key != kEmpty)
                {
                    const ull2 q01 = vt2[bk];
                    const unsigned long long s0 = q01.x, s1 = q01.y;
                    uint32_t hit = ~0u;
                    if (s0 == key) hit = 2 * bk;
                    else if (s1 == key) hit = 2 * bk + 1;
                    else if (s0 == kEmpty || s1 == kEmpty)
                    {
                        const uint32_t slot = 2 * bk + (s0 == kEmpty ? 0u : 1u);
                        const unsigned long long old = atomicCAS(&tab[slot], kEmpty, key);
                        if (old == kEmpty)
                        {
                            uint32_t nd = atomicAdd(&ndist, 1u);
                            if (nd + 1 > kLimit) *vovf = 1;
                            hit = slot;
                        }
                        else if (old == key) hit = slot;
                    }
                    else if (st == 0) { bk = second_bucket(key_mix(key), bk); st = 1; }
                    else bk = (bk + 1) & (SLOTS / 2 - 1);
                    if (hit != ~0u) { atomicAdd(&cnt[hit], 1u); key = kEmpty; }
                }
                if (*vovf) break;
            }
        };
        auto drain = [&](bool all) {
            while (tail - head >= (all ? 1u : 64u * kG))
            {
                const uint32_t fill = tail - head;
                unsigned long long kq[kG];
#pragma unroll
                for (int g = 0; g < kG; ++g)
                {
                    const uint32_t o = (uint32_t)g * 64u + lane;
                    const unsigned long long v = wq[(head + o) & (kQ - 1)];
                    kq[g] = o < fill ? v : kEmpty;
                }
                head += fill < 64u * kG ? fill : 64u * kG;
                insert_g(kq);
                if (*vovf) { head = tail; break; }
            }
        };
        unsigned long long nxt[kSegUnroll];
#pragma unroll
        for (int u = 0; u < kSegUnroll; ++u)
        {
            const uint64_t i = b + (uint64_t)u * NT + tid;
            const unsigned long long v = __builtin_nontemporal_load(&keys[i < e ? i : e - 1].lo);
            nxt[u] = i < e ? v : kEmpty;
        }
        for (uint64_t i0 = b; i0 < e; i0 += (uint64_t)NT * kSegUnroll)
        {
            unsigned long long kv[kSegUnroll];
#pragma unroll
            for (int u = 0; u < kSegUnroll; ++u) kv[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < kSegUnroll; ++u)
            {
                const uint64_t i = i0 + (uint64_t)(kSegUnroll + u) * NT + tid;
                const unsigned long long v = __builtin_nontemporal_load(&keys[i < e ? i : e - 1].lo);
                nxt[u] = i < e ? v : kEmpty;
            }
#pragma unroll
            for (int u = 0; u < kSegUnroll; ++u)
            {
                const uint32_t own = (kv[u] != kEmpty ? 1u : 0u) & ((((uint32_t)(kv[u] >> rem_bits)) & rmask) == rnd ? 1u : 0u);
                const uint64_t m = __ballot(own != 0);
                const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (own) wq[(tail + before) & (kQ - 1)] = kv[u];
                tail += (uint32_t)__popcll(m);
                if (u & 1) drain(false);
            }
            if (*vovf) break;
        }
        drain(true);
    }
    else
    {
    unsigned long long nxt[kSegUnroll];
#pragma unroll
    for (int u = 0; u < kSegUnroll; ++u)
    {
        // (clamped index and a select instead of a branch around the load)
        const uint64_t i = b + (uint64_t)u * NT + tid;
        const unsigned long long v = __builtin_nontemporal_load(&keys[i < e ? i : e - 1].lo);
        nxt[u] = i < e ? v : kEmpty;
    }
    for (uint64_t i0 = b; i0 < e; i0 += (uint64_t)NT * kSegUnroll)
    {
        unsigned long long kv[kSegUnroll];
#pragma unroll
        for (int u = 0; u < kSegUnroll; ++u) kv[u] = mine_only(nxt[u]);
        // software pipeline: the next batch's loads are in flight while this one is inserted
#pragma unroll
        for (int u = 0; u < kSegUnroll; ++u)
        {
            const uint64_t i = i0 + (uint64_t)(kSegUnroll + u) * NT + tid;
            const unsigned long long v = __builtin_nontemporal_load(&keys[i < e ? i : e - 1].lo);
            nxt[u] = i < e ? v : kEmpty;
        }
        // The table is probed in buckets of two adjacent slots (one 16-byte LDS read): at a load
        // of ~0.4 a present key is almost always in its home bucket.
        // fast path: home buckets of all keys of the batch at once (independent LDS reads); a
        // key that is already there only needs its count bumped
        // (an LDS-typed pointer to a 16-byte vector: one ds_read_b128; through a generic volatile
        // pointer the compiler emits two 8-byte FLAT loads)
        typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
        typedef const volatile __attribute__((address_space(3))) ull2* lds_bucket_ptr;
        const lds_bucket_ptr vt2 = (lds_bucket_ptr)tab;
        uint32_t bkt[kSegUnroll];
        uint32_t pend = 0, stm = 0;
        // The probes of half a batch are issued together (plain LDS loads: a stale miss only sends
        // the key to the slow path, which reads the bucket again through the volatile view; keys are
        // never removed, so there is no stale hit) and then consumed -- as volatile loads the
        // compiler kept them in program order, one LDS round trip per key.
        typedef const __attribute__((address_space(3))) ull2* lds_bucket_plain;
        const lds_bucket_plain pt2 = (lds_bucket_plain)tab;
        constexpr int kHalf = kSegUnroll / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            // (the home bucket AND the one behind it: at a load of 0.37 about 6 % of the keys were pushed out of
            // a full home bucket, and such a key would take the slow path -- a wave-wide loop -- every one of
            // the ~100 times it occurs; two buckets leave about 0.5 %)
            ull2 q[kHalf], q2[kHalf];
            uint32_t b2[kHalf];
#pragma unroll
            for (int j = 0; j < kHalf; ++j)
            {
                const int u = h * kHalf + j;
                // one 32-bit multiply (a 64-bit one is three quarter-rate instructions): the high
                // word, rotated, folded into the low one, times the golden ratio
                const uint32_t f = key_mix(kv[u]);
                bkt[u] = f >> (32 - kBucketBits);
                b2[j] = second_bucket(f, bkt[u]);
                q[j] = pt2[bkt[u]];
                q2[j] = pt2[b2[j]];
            }
#pragma unroll
            for (int j = 0; j < kHalf; ++j)
            {
                const int u = h * kHalf + j;
                const unsigned long long s0 = q[j].x, s1 = q[j].y, s2 = q2[j].x, s3 = q2[j].y;
                // No branch per key: the count of the slot that holds the key (or of slot 0 of the bucket,
                // by 0) is bumped unconditionally, a miss sets a bit.  Written with && / if-else chains the
                // compiler emits a branch per term, and the scalar exec-mask bookkeeping then costs more
                // issue slots than the vector work.
                const uint32_t live = kv[u] != kEmpty ? 1u : 0u;
                const uint32_t h0 = s0 == kv[u] ? 1u : 0u, h1 = s1 == kv[u] ? 1u : 0u;
                const uint32_t h2 = s2 == kv[u] ? 1u : 0u, h3 = s3 == kv[u] ? 1u : 0u;
                const uint32_t hit = (h0 | h1 | h2 | h3) & live;
                const uint32_t second = h2 | h3;
                const uint32_t slot = 2 * (second ? b2[j] : bkt[u]) + (h1 | h3);
                atomicAdd(&cnt[slot], hit);
                const uint32_t miss = live & (hit ^ 1u);
                pend |= miss << u;
                // a full first bucket cannot take the key: the slow path starts at the second one
                const uint32_t full = (s0 != kEmpty ? 1u : 0u) & (s1 != kEmpty ? 1u : 0u) & miss;
                bkt[u] = full ? b2[j] : bkt[u];
                stm |= full << u;
            }
        }
        // slow path (key absent from its home bucket): every lane walks its OWN queue of
        // leftover keys, one probe per wave iteration, so the wave iterates max-over-lanes of the
        // lane totals instead of the sum over the eight keys of per-key maxima
        unsigned long long key = kEmpty;
        uint32_t bk = 0, st = 0;                          // st: 0 = at the first bucket, 1 = at the second or beyond
        for (;;)
        {
            if (key == kEmpty && pend)
            {
                const uint32_t u = __ffs(pend) - 1;
                pend &= pend - 1;
                st = (stm >> u) & 1u;
#pragma unroll
                for (int uu = 0; uu < kSegUnroll; ++uu)
                    if (u == (uint32_t)uu) { key = kv[uu]; bk = bkt[uu]; }
            }
            if (!__ballot(key != kEmpty)) break;
            if (key != kEmpty)
            {
                const ull2 q01 = vt2[bk];
                const unsigned long long s0 = q01.x, s1 = q01.y;
                uint32_t hit = ~0u;                       // slot that holds (or now holds) the key
                if (s0 == key) hit = 2 * bk;
                else if (s1 == key) hit = 2 * bk + 1;
                else if (s0 == kEmpty || s1 == kEmpty)
                {
                    const uint32_t slot = 2 * bk + (s0 == kEmpty ? 0u : 1u);
                    const unsigned long long old = atomicCAS(&tab[slot], kEmpty, key);
                    if (old == kEmpty)
                    {
                        uint32_t nd = atomicAdd(&ndist, 1u);
                        if (nd + 1 > kLimit) *vovf = 1;
                        hit = slot;
                    }
                    else if (old == key) hit = slot;
                    // else: somebody else took the slot; look at this bucket again
                }
                else if (st == 0) { bk = second_bucket(key_mix(key), bk); st = 1; }      // full: on to the second bucket,
                else bk = (bk + 1) & (SLOTS / 2 - 1);                                     // then to the ones behind it
                if (hit != ~0u) { atomicAdd(&cnt[hit], 1u); key = kEmpty; }
            }
            if (*vovf) break;
        }
        if (*vovf) break;
    }
    }
    __syncthreads();
    if (ovf)
    {
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }

    // Order the occupied slots.  Every thread takes its 16 slots into registers (all reads happen
    // before any write), then the entries are bucket-sorted in place on the 10 bits below the
    // segment bits: rank inside the bucket by an LDS atomic, exclusive scan of the 1024 bucket
    // sizes, scatter, and an insertion sort of every bucket (1.5 keys on average at kLimit/2).
    // Five barriers instead of the 66 of a bitonic network over 2048 slots; a bucket with more
    // than 24 keys (skewed low bits) falls back to the bitonic sort of the compacted entries.
    constexpr int kPer = SLOTS / NT;
    constexpr int kBins = SLOTS / 4, kBinsPer = kBins / NT, kBinBits = kBucketBits - 1;
    __shared__ uint32_t bins[kBins];
    __shared__ uint32_t sh_scan2[NT / 64 + 1];
    __shared__ uint32_t big;
    unsigned long long ck[kPer];
    uint32_t cc[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)
    {
        ck[j] = tab[tid * kPer + j];
        cc[j] = cnt[tid * kPer + j];
    }
    for (uint32_t i = tid; i < kBins; i += NT) bins[i] = 0;
    if (tid == 0) big = 0;
    __syncthreads();
    const uint32_t bsh = rem_bits > (uint32_t)kBinBits ? rem_bits - kBinBits : 0;
    uint32_t rnk[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (ck[j] != kEmpty) rnk[j] = atomicAdd(&bins[(uint32_t)(ck[j] >> bsh) & (kBins - 1)], 1u);
    __syncthreads();
    uint32_t bn[kBinsPer], bs[kBinsPer], mine = 0;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q) { bn[q] = bins[tid * kBinsPer + q]; mine += bn[q]; }
    uint32_t tot_occ;
    uint32_t at = block_excl_scan_n<uint32_t, NT / 64>(mine, sh_scan2, &tot_occ);
    lds_vu32 vbig = (lds_vu32)&big;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q)
    {
        bs[q] = at; bins[tid * kBinsPer + q] = at; at += bn[q];
        if (bn[q] > 24) *vbig = 1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (ck[j] != kEmpty)
        {
            const uint32_t pos = bins[(uint32_t)(ck[j] >> bsh) & (kBins - 1)] + rnk[j];
            tab[pos] = ck[j]; cnt[pos] = cc[j];
        }
    __syncthreads();
    if (!big)
    {
#pragma unroll
        for (int q = 0; q < kBinsPer; ++q)
            for (uint32_t i = 1; i < bn[q]; ++i)
            {
                const unsigned long long kk = tab[bs[q] + i];
                const uint32_t vv = cnt[bs[q] + i];
                uint32_t j = i;
                while (j > 0 && tab[bs[q] + j - 1] > kk)
                {
                    tab[bs[q] + j] = tab[bs[q] + j - 1]; cnt[bs[q] + j] = cnt[bs[q] + j - 1];
                    --j;
                }
                tab[bs[q] + j] = kk; cnt[bs[q] + j] = vv;
            }
        __syncthreads();
    }
    else
    {
        uint32_t nsort = 512;
        while (nsort < tot_occ) nsort <<= 1;
        for (uint32_t i = tot_occ + tid; i < nsort; i += NT) tab[i] = kEmpty;
        __syncthreads();
        // bitonic sort of the first nsort (key,count) slots by key; empty slots (all ones) sort last
        for (uint32_t k2 = 2; k2 <= nsort; k2 <<= 1)
        {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
            {
                for (uint32_t t = tid; t < nsort / 2; t += NT)
                {
                    uint32_t i = 2 * t - (t & (j - 1));       // element with bit j clear
                    uint32_t p = i + j;
                    bool up = (i & k2) == 0;
                    unsigned long long a = tab[i], c = tab[p];
                    if ((a > c) == up)
                    {
                        tab[i] = c; tab[p] = a;
                        uint32_t ca = cnt[i]; cnt[i] = cnt[p]; cnt[p] = ca;
                    }
                }
                __syncthreads();
            }
        }
    }
    uint32_t d = ndist;
    if (tid == 0)
    {
        sh_base = atomicAdd(&so->cursor, (unsigned long long)d);
        if (sh_base + d > so->stage_cap) { atomicOr(&so->overflow, 2u); sh_base = ~0ULL; }
        seg_pos[s] = sh_base;
        seg_cnt[s] = d;
    }
    __syncthreads();
    const uint64_t ob = sh_base;
    if (ob == ~0ULL) return;
    for (uint32_t i = tid; i < d; i += NT)
    {
        stage_keys[ob + i].lo = tab[i];
        stage_counts[ob + i] = cnt[i];
    }
}


__global__ __launch_bounds__(kTB) void seg_hash_reduce_kernel(const Key1* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                              const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so, uint64_t* __restrict__ seg_pos,
                                                              uint64_t* __restrict__ seg_cnt,
                                                              Key1* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                              uint32_t rem_bits)
{
    seg_hash_reduce_body<kTB, kSegSlots>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits, 0u);
}

// The same with one 1024-thread workgroup per CU and a table of 8192 slots (100 KB of LDS): segments
// of up to 6144 distinct keys, i.e. 65 536 segments still do where the table above would need a
// third partition digit (1.5e8 to 3e8 distinct keys in a chunk).
constexpr int kSegBigThreads = 1024;
constexpr int kSegBigSlots = 8192;
constexpr int kSegBigLimit = kSegBigSlots / 4 * 3;
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_reduce_big_kernel(const Key1* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                                             const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                             uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                             Key1* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                             uint32_t rem_bits, uint32_t round_bits)
{
    (void)round_bits;
    seg_hash_reduce_body<kSegBigThreads, kSegBigSlots, false>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits, 0u);
}

// 2^round_bits workgroups per segment, each counting the keys of one value of the next round_bits
// key bits: 16-bit segments of up to 4 x 4600 distinct keys (1.2e9 distinct keys in a chunk) without
// a third partition digit, at the price of streaming every key 2^round_bits times.
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_reduce_shared_kernel(const Key1* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                                                const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                                uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                                Key1* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                                uint32_t rem_bits, uint32_t round_bits)
{
    seg_hash_reduce_body<kSegBigThreads, kSegBigSlots, true>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits, round_bits);
}

// Two-word keys.  LDS has no 128-bit compare-and-swap, so a slot is claimed through its state
// word: 0 = empty, kSegLock = being written, otherwise the count of a published key.  The
// insert loop is a per-lane state machine with exactly one probe per wave iteration and no
// wait inside an iteration: a lane that meets a locked slot simply looks again next iteration,
// by which time the owner (which needs no other lane to make progress) has published.
constexpr uint32_t kSegLock = 0x80000000u;

template <int NT, int SLOTS>
__device__ __forceinline__ void seg_hash_reduce2_body(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                      const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so, uint64_t* __restrict__ seg_pos,
                                                      uint64_t* __restrict__ seg_cnt,
                                                      Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                      uint32_t rem_bits_all, uint32_t round_bits)
{
    // round_bits > 0: unit (segment, r) as in seg_hash_reduce_body -- 2^round_bits workgroups stream
    // the segment, each counting the keys whose next round_bits bits equal r
    constexpr int kLimit = SLOTS / 4 * 3;
    // SLOTS is a power of two, or 6144 (the largest table of two-word keys that fits a CU's LDS:
    // slot = high half of hash * SLOTS instead of the hash's top bits, wrap-around by comparison)
    constexpr bool kPow2 = (SLOTS & (SLOTS - 1)) == 0;
    constexpr int kSlotBits = SLOTS == 2048 ? 11 : SLOTS == 4096 ? 12 : SLOTS == 6144 ? 13 : -1;
    static_assert(kSlotBits > 0 && SLOTS % NT == 0, "table size");
    auto next_slot = [](uint32_t a) { return kPow2 ? ((a + 1) & (uint32_t)(SLOTS - 1)) : (a + 1 == (uint32_t)SLOTS ? 0u : a + 1); };
    __shared__ unsigned long long tlo[SLOTS];
    __shared__ unsigned long long thi[SLOTS];
    __shared__ uint32_t st[SLOTS];
    __shared__ uint32_t ndist;
    __shared__ uint32_t ovf;
    __shared__ unsigned long long sh_base;
    const uint32_t s = unit_block(), tid = threadIdx.x;
    const uint32_t sseg = s >> round_bits, rnd = s & ((1u << round_bits) - 1u);
    const uint32_t rsh = rem_bits_all - round_bits;            // position of the round bits in the key
    const uint64_t b = seg_off[sseg], e = seg_end[sseg];
    if (b == e)
    {
        if (tid == 0) { seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    if (e - b > 0xFFFFFFFFULL)
    {
        // a 32-bit slot count could wrap: leave this chunk to the full sort, whose run lengths
        // saturate and report the overflow
        if (tid == 0) { atomicOr(&so->overflow, 2u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    for (uint32_t i = tid; i < SLOTS; i += NT) st[i] = 0;
    if (tid == 0) { ndist = 0; ovf = 0; }
    __syncthreads();

    lds_vu32 vovf = (lds_vu32)&ovf;
    lds_vu32 vst = (lds_vu32)st;
    lds_vu64 vlo = (lds_vu64)tlo;
    lds_vu64 vhi = (lds_vu64)thi;
#ifndef GOSS_SEG_UNROLL2
#define GOSS_SEG_UNROLL2 8
#endif
    constexpr int kU = GOSS_SEG_UNROLL2;
    // software pipeline: the next batch's loads are in flight while this one is inserted
    // (hi = all ones marks "no key": 2*len <= 126 bits)
    Key2 nxt[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u)
    {
        uint64_t i = b + (uint64_t)u * NT + tid;
        nxt[u] = i < e ? keys[i] : Key2{~0ULL, ~0ULL};
    }
    for (uint64_t i0 = b; i0 < e; i0 += (uint64_t)NT * kU)
    {
        Key2 kv[kU];
        uint32_t slots[kU];
        uint32_t pend = 0;
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            kv[u] = nxt[u];
            if (round_bits && ((uint32_t)key_shr64(kv[u], rsh) & ((1u << round_bits) - 1u)) != rnd) kv[u].hi = ~0ULL;   // another workgroup's key
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            uint64_t i = i0 + (uint64_t)(kU + u) * NT + tid;
            nxt[u] = i < e ? keys[i] : Key2{~0ULL, ~0ULL};
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            slots[u] = 0;
            if (kv[u].hi != ~0ULL)
            {
                // one 32-bit multiply: the four words rotated against each other and folded (two
                // 64-bit multiplies are seven quarter-rate instructions)
                const uint32_t w0 = (uint32_t)kv[u].lo, w1 = (uint32_t)(kv[u].lo >> 32);
                const uint32_t w2 = (uint32_t)kv[u].hi, w3 = (uint32_t)(kv[u].hi >> 32);
                const uint32_t f = w0 ^ __builtin_rotateleft32(w1, 15) ^ __builtin_rotateleft32(w2, 7) ^ __builtin_rotateleft32(w3, 23);
                slots[u] = kPow2 ? (f * 0x9E3779B1u) >> (32 - kSlotBits) : __umulhi(f * 0x9E3779B1u, (uint32_t)SLOTS);
            }
        }
        // fast path: the home slots of the whole batch are read together (plain LDS loads, the
        // states before the keys: a slot whose state shows a count was published with its key, and
        // keys never change afterwards); a key found there only needs its count bumped.  Everything
        // else -- empty, locked, another key, or a state that was not there yet -- goes through the
        // state machine below, which reads through the volatile views.
        {
            typedef const __attribute__((address_space(3))) uint32_t* lds_u32_plain;
            typedef const __attribute__((address_space(3))) unsigned long long* lds_u64_plain;
            const lds_u32_plain pst = (lds_u32_plain)st;
            const lds_u64_plain plo = (lds_u64_plain)tlo, phi = (lds_u64_plain)thi;
            // home slot and its neighbour (a key displaced once sits there: at a load of 0.37 that
            // leaves ~7 % instead of ~20 % of the keys to the state machine), four keys at a time
            constexpr int kQ = 4;
            static_assert(kU % kQ == 0, "quarter batches");
#pragma unroll
            for (int h = 0; h < kU / kQ; ++h)
            {
                uint32_t fs[kQ], gs[kQ];
                unsigned long long fl[kQ], fh[kQ], gl[kQ], gh[kQ];
#pragma unroll
                for (int j = 0; j < kQ; ++j)
                {
                    const uint32_t a = slots[h * kQ + j], b2 = next_slot(a);
                    fs[j] = pst[a]; gs[j] = pst[b2];
                }
                asm volatile("" ::: "memory");       // the compiler keeps the states ahead of the keys; the LDS runs a wave's operations in order
#pragma unroll
                for (int j = 0; j < kQ; ++j)
                {
                    const uint32_t a = slots[h * kQ + j], b2 = next_slot(a);
                    fl[j] = plo[a]; fh[j] = phi[a]; gl[j] = plo[b2]; gh[j] = phi[b2];
                }
#pragma unroll
                for (int j = 0; j < kQ; ++j)
                {
                    const int u = h * kQ + j;
                    if (kv[u].hi == ~0ULL) continue;
                    const bool at0 = fs[j] != 0u && fs[j] != kSegLock && fl[j] == kv[u].lo && fh[j] == kv[u].hi;
                    const bool at1 = gs[j] != 0u && gs[j] != kSegLock && gl[j] == kv[u].lo && gh[j] == kv[u].hi;
                    if (at0) atomicAdd(&st[slots[u]], 1u);
                    else if (at1) atomicAdd(&st[next_slot(slots[u])], 1u);
                    else pend |= 1u << u;
                }
            }
        }
        Key2 key{0, 0};
        uint32_t slot = 0;
        bool have = false;
        for (;;)
        {
            if (!have && pend)
            {
                const uint32_t u = __ffs(pend) - 1;
                pend &= pend - 1;
#pragma unroll
                for (int uu = 0; uu < kU; ++uu)
                    if (u == (uint32_t)uu) { key = kv[uu]; slot = slots[uu]; }
                have = true;
            }
            if (!__ballot(have)) break;
            if (have)
            {
                uint32_t state = vst[slot];
                if (state == 0)
                {
                    uint32_t old = atomicCAS(&st[slot], 0u, kSegLock);
                    if (old == 0)
                    {
                        vlo[slot] = key.lo;
                        vhi[slot] = key.hi;
                        vst[slot] = 1u;                      // publish (LDS ops of a lane are in order)
                        uint32_t nd = atomicAdd(&ndist, 1u);
                        if (nd + 1 > kLimit) *vovf = 1;
                        have = false;
                    }
                    // else: look at this slot again next iteration
                }
                else if (state != kSegLock)
                {
                    if (vlo[slot] == key.lo && vhi[slot] == key.hi) { atomicAdd(&st[slot], 1u); have = false; }
                    else slot = next_slot(slot);
                }
            }
            if (*vovf) break;
        }
        if (*vovf) break;
    }
    __syncthreads();
    if (ovf)
    {
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    // Order the occupied slots as the one-word kernel does: every thread takes its slots into
    // registers (all reads before any write), the entries are bucket-sorted in place on the key bits
    // just below the unit's prefix -- rank inside the bucket by an LDS atomic, exclusive scan of the
    // bucket sizes, scatter, insertion sort of every bucket -- and only a bucket above 24 keys (skewed
    // bits) sends the compacted entries through the bitonic network (66 barriers for 2048 of them).
    constexpr int kPer2 = SLOTS / NT;
    constexpr int kBins = kPow2 ? SLOTS / 4 : 2048, kBinsPer = kBins / NT, kBinBits = kPow2 ? kSlotBits - 2 : 11;
    static_assert(kBins % NT == 0, "bins per thread");
    __shared__ uint32_t bins[kBins];
    __shared__ uint32_t sh_scan2[NT / 64 + 1];
    __shared__ uint32_t big;
    unsigned long long cl[kPer2], ch[kPer2];
    uint32_t cs[kPer2];
#pragma unroll
    for (int j = 0; j < kPer2; ++j)
    {
        cl[j] = tlo[tid * kPer2 + j]; ch[j] = thi[tid * kPer2 + j]; cs[j] = st[tid * kPer2 + j];
    }
    for (uint32_t i = tid; i < kBins; i += NT) bins[i] = 0;
    if (tid == 0) big = 0;
    __syncthreads();
    const uint32_t rem_unit = rem_bits_all - round_bits;
    const uint32_t bsh = rem_unit > (uint32_t)kBinBits ? rem_unit - kBinBits : 0;
    uint32_t rnk[kPer2], bin[kPer2];
#pragma unroll
    for (int j = 0; j < kPer2; ++j)
        if (cs[j] != 0)
        {
            bin[j] = (uint32_t)key_shr64(Key2{cl[j], ch[j]}, bsh) & (kBins - 1);
            rnk[j] = atomicAdd(&bins[bin[j]], 1u);
        }
    __syncthreads();
    uint32_t bn[kBinsPer], bs[kBinsPer], mine = 0;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q) { bn[q] = bins[tid * kBinsPer + q]; mine += bn[q]; }
    uint32_t tot_occ;
    uint32_t at = block_excl_scan_n<uint32_t, NT / 64>(mine, sh_scan2, &tot_occ);
    lds_vu32 vbig = (lds_vu32)&big;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q)
    {
        bs[q] = at; bins[tid * kBinsPer + q] = at; at += bn[q];
        if (bn[q] > 24) *vbig = 1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer2; ++j)
        if (cs[j] != 0)
        {
            const uint32_t pos = bins[bin[j]] + rnk[j];
            tlo[pos] = cl[j]; thi[pos] = ch[j]; st[pos] = cs[j];
        }
    __syncthreads();
    if (!big)
    {
#pragma unroll
        for (int q = 0; q < kBinsPer; ++q)
            for (uint32_t i = 1; i < bn[q]; ++i)
            {
                const unsigned long long kl = tlo[bs[q] + i], kh = thi[bs[q] + i];
                const uint32_t vv = st[bs[q] + i];
                uint32_t j = i;
                while (j > 0 && (thi[bs[q] + j - 1] > kh || (thi[bs[q] + j - 1] == kh && tlo[bs[q] + j - 1] > kl)))
                {
                    tlo[bs[q] + j] = tlo[bs[q] + j - 1]; thi[bs[q] + j] = thi[bs[q] + j - 1]; st[bs[q] + j] = st[bs[q] + j - 1];
                    --j;
                }
                tlo[bs[q] + j] = kl; thi[bs[q] + j] = kh; st[bs[q] + j] = vv;
            }
        __syncthreads();
    }
    else
    {
        uint32_t nsort = 64;
        while (nsort < tot_occ) nsort <<= 1;
        if (nsort > (uint32_t)SLOTS)
        {
            // (6144-slot table only) more entries than the largest network the arrays hold: the host retries
            // with more partition bits
            if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = 0; }
            return;
        }
        for (uint32_t i = tot_occ + tid; i < nsort; i += NT) { thi[i] = ~0ULL; tlo[i] = ~0ULL; st[i] = 0; }
        __syncthreads();
        // empty slots sort last: hi = all ones is never a key (2*len <= 126 bits)
        for (uint32_t k2 = 2; k2 <= nsort; k2 <<= 1)
        {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
            {
                for (uint32_t t = tid; t < nsort / 2; t += NT)
                {
                    uint32_t i = 2 * t - (t & (j - 1));
                    uint32_t p = i + j;
                    bool up = (i & k2) == 0;
                    unsigned long long ah = thi[i], al = tlo[i], bh = thi[p], bl = tlo[p];
                    bool gt = ah > bh || (ah == bh && al > bl);
                    if (gt == up)
                    {
                        thi[i] = bh; tlo[i] = bl; thi[p] = ah; tlo[p] = al;
                        uint32_t ca = st[i]; st[i] = st[p]; st[p] = ca;
                    }
                }
                __syncthreads();
            }
        }
    }
    uint32_t d = ndist;
    if (tid == 0)
    {
        sh_base = atomicAdd(&so->cursor, (unsigned long long)d);
        if (sh_base + d > so->stage_cap) { atomicOr(&so->overflow, 2u); sh_base = ~0ULL; }
        seg_pos[s] = sh_base;
        seg_cnt[s] = d;
    }
    __syncthreads();
    const uint64_t ob = sh_base;
    if (ob == ~0ULL) return;
    for (uint32_t i = tid; i < d; i += NT)
    {
        stage_keys[ob + i] = Key2{tlo[i], thi[i]};
        stage_counts[ob + i] = st[i];
    }
}

// The same for two-word keys whose bits below the segment prefix fit 96 bits (2*len <= 112 with 16-bit
// segments: every build-graph k <= 55, every k-mer set k <= 56).  Inside a segment all keys share the
// prefix, so a slot holds the 96-bit REMAINDER and the count in 16 bytes: a probe is one ds_read_b128
// instead of three reads from three arrays, and 8192 slots fit a CU's LDS (128 KB) -- up to 6144
// distinct keys per segment counted by one workgroup.  Slot word w: 0 = empty, kSegLock = being
// written, otherwise the count of a published key (the protocol of seg_hash_reduce2_body).
struct __attribute__((aligned(16))) Slot96 { uint32_t r0, r1, r2, w; };
__device__ __forceinline__ uint4 tbl4(const Slot96* t, uint32_t i) { return reinterpret_cast<const uint4*>(t)[i]; }

// MERGE: the input is not one slice of raw keys but the segment's slice of each of `nruns` sorted (key,count)
// runs (run r = entries [run_off[r], run_off[r+1]) of keys / vals, its segment bounds in bounds[r * 65537 ..]):
// every entry adds its count.  That is the k-way merge of the chunk runs of a large build -- the runs of a
// high-coverage input all hold the same keys, so the table stays small -- done as hash inserts instead of
// ordering networks and binary searches.  A count that would reach 2^31 (the lock bit of the slot word)
// makes the kernel give up; the host then merges the general way.
// PACKED: `keys` is an array of 12-byte Rem96 records (the second partition level wrote remainders).
template <int NT, int SLOTS, bool MERGE = false, bool PACKED = false>
__device__ __forceinline__ void seg_hash_reduce96_body(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                       const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so, uint64_t* __restrict__ seg_pos,
                                                       uint64_t* __restrict__ seg_cnt,
                                                       Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                       uint32_t rem_bits, const uint32_t* __restrict__ vals = nullptr,
                                                       const uint64_t* __restrict__ run_off = nullptr, uint32_t nruns = 1)
{
    constexpr int kLimit = SLOTS / 4 * 3;
    constexpr int kSlotBits = SLOTS == 4096 ? 12 : SLOTS == 8192 ? 13 : -1;
    static_assert(kSlotBits > 0 && SLOTS % NT == 0, "table size");
    __shared__ Slot96 tbl[SLOTS];
    __shared__ uint32_t ndist;
    __shared__ uint32_t ovf;
    __shared__ unsigned long long sh_base;
    const uint32_t s = unit_block(), tid = threadIdx.x;
    if (!MERGE)
    {
        const uint64_t b = seg_off[s], e = seg_end[s];
        if (b == e)
        {
            if (tid == 0) { seg_pos[s] = 0; seg_cnt[s] = 0; }
            return;
        }
        if (e - b > 0xFFFFFFFFULL)
        {
            if (tid == 0) { atomicOr(&so->overflow, 2u); seg_pos[s] = 0; seg_cnt[s] = 0; }
            return;
        }
    }
    uint32_t* tw = reinterpret_cast<uint32_t*>(tbl);
    for (uint32_t i = tid; i < SLOTS; i += NT) tw[4 * i + 3] = 0;
    if (tid == 0) { ndist = 0; ovf = 0; }
    __syncthreads();
    // remainder of a key: its low rem_bits bits (64 <= rem_bits <= 96, or fewer: then r2 = 0)
    const uint32_t hbits = rem_bits > 64 ? rem_bits - 64 : 0;
    const uint32_t hmask = hbits >= 32 ? 0xFFFFFFFFu : ((1u << hbits) - 1u);
    const uint64_t lmask64 = rem_bits >= 64 ? ~0ULL : ((1ULL << rem_bits) - 1ULL);

    lds_vu32 vovf = (lds_vu32)&ovf;
    lds_vu32 vt = (lds_vu32)tw;
    constexpr int kU = 8;
    for (uint32_t run = 0; run < (MERGE ? nruns : 1u); ++run)
    {
    // this run's slice of the segment (MERGE), or the segment itself
    const uint64_t b = MERGE ? run_off[run] + seg_off[(uint64_t)run * 65537u + s] : seg_off[s];
    const uint64_t e = MERGE ? run_off[run] + seg_off[(uint64_t)run * 65537u + s + 1] : seg_end[s];
    if (b >= e) continue;
    Key2 nxt[kU];
    uint32_t nwt[kU];
    const Rem96* packed = reinterpret_cast<const Rem96*>(keys);
    // key i of the input as (lo, hi) with hi = all ones for "no key" (an index beyond the slice)
    auto load = [&](uint64_t i) -> Key2 {
        Key2 v;
        if (PACKED)
        {
            const Rem96 r = packed[i < e ? i : e - 1];
            v.lo = (uint64_t)r.r0 | ((uint64_t)r.r1 << 32); v.hi = r.r2;
        }
        else v = keys[i < e ? i : e - 1];
        v.hi = i < e ? v.hi : ~0ULL;
        return v;
    };
#pragma unroll
    for (int u = 0; u < kU; ++u)
    {
        // (clamped index and a select instead of a branch around the load)
        const uint64_t i = b + (uint64_t)u * NT + tid;
        nxt[u] = load(i);
        nwt[u] = MERGE ? vals[i < e ? i : e - 1] : 1u;
    }
    for (uint64_t i0 = b; i0 < e; i0 += (uint64_t)NT * kU)
    {
        uint32_t r0[kU], r1[kU], r2[kU], slots[kU], wt[kU];
        uint32_t pend = 0, live = 0;
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            const Key2 kv = nxt[u];
            if (kv.hi != ~0ULL) live |= 1u << u;
            const uint64_t lo = kv.lo & lmask64;
            r0[u] = (uint32_t)lo; r1[u] = (uint32_t)(lo >> 32); r2[u] = (uint32_t)kv.hi & hmask;
            wt[u] = nwt[u];
            // a weight that alone reaches the lock bit (or is the marker of a count kept elsewhere): not here
            if (MERGE && (live >> u & 1u) && wt[u] >= kSegLock) *vovf = 1;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            const uint64_t i = i0 + (uint64_t)(kU + u) * NT + tid;
            nxt[u] = load(i);
            nwt[u] = MERGE ? vals[i < e ? i : e - 1] : 1u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            const uint32_t f = r0[u] ^ __builtin_rotateleft32(r1[u], 15) ^ __builtin_rotateleft32(r2[u], 7);
            slots[u] = (f * 0x9E3779B1u) >> (32 - kSlotBits);
        }
        // fast path: the home slot and the three behind it, two keys at a time (plain 16-byte LDS loads: a
        // slot whose word shows a count was published with its key, and keys never change).  Once a segment's
        // keys are in the table -- after its first few batches -- nearly every key is found here (at a load
        // of 0.37 about 1 % sit further from home); what is not goes through the state machine below, whose
        // wave-wide loop costs every lane of the wave its iterations.
        {
            constexpr int kQ = 2;
            static_assert(kU % kQ == 0, "pairs");
#pragma unroll
            for (int h = 0; h < kU / kQ; ++h)
            {
                uint4 f[kQ][4];
#pragma unroll
                for (int j = 0; j < kQ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) f[j][q] = tbl4(tbl, (slots[h * kQ + j] + q) & (SLOTS - 1));
#pragma unroll
                for (int j = 0; j < kQ; ++j)
                {
                    const int u = h * kQ + j;
                    // (integer arithmetic, no && chains: the compiler turns those into a branch per term, and
                    // the scalar instructions of exec-mask bookkeeping then outnumber the vector ones)
                    uint32_t off = 4;
#pragma unroll
                    for (int q = 3; q >= 0; --q)
                    {
                        const uint32_t diff = (f[j][q].x ^ r0[u]) | (f[j][q].y ^ r1[u]) | (f[j][q].z ^ r2[u]);
                        // a published slot: 1 <= w < kSegLock
                        const uint32_t bad = diff | (uint32_t)((f[j][q].w - 1u) >= (kSegLock - 1u));
                        off = bad ? off : (uint32_t)q;
                    }
                    const uint32_t is_live = (live >> u) & 1u;
                    const uint32_t hit = (off < 4u ? 1u : 0u) & is_live;
                    if (hit)
                    {
                        const uint32_t old = atomicAdd(&tw[4 * ((slots[u] + off) & (SLOTS - 1)) + 3], wt[u]);
                        if (MERGE && old + wt[u] >= kSegLock) *vovf = 1;
                    }
                    pend |= (is_live & (hit ^ 1u)) << u;
                }
            }
        }
        uint32_t k0 = 0, k1 = 0, k2 = 0, slot = 0, kw = 1;
        bool have = false;
        for (;;)
        {
            if (!have && pend)
            {
                const uint32_t u = __ffs(pend) - 1;
                pend &= pend - 1;
#pragma unroll
                for (int uu = 0; uu < kU; ++uu)
                    if (u == (uint32_t)uu) { k0 = r0[uu]; k1 = r1[uu]; k2 = r2[uu]; slot = slots[uu]; kw = wt[uu]; }
                have = true;
            }
            if (!__ballot(have)) break;
            if (have)
            {
                const uint32_t state = vt[4 * slot + 3];
                if (state == 0)
                {
                    const uint32_t old = atomicCAS(&tw[4 * slot + 3], 0u, kSegLock);
                    if (old == 0)
                    {
                        vt[4 * slot] = k0; vt[4 * slot + 1] = k1; vt[4 * slot + 2] = k2;
                        vt[4 * slot + 3] = kw;               // publish (LDS ops of a lane are in order)
                        const uint32_t nd = atomicAdd(&ndist, 1u);
                        if (nd + 1 > kLimit) *vovf = 1;
                        have = false;
                    }
                }
                else if (state != kSegLock)
                {
                    if (vt[4 * slot] == k0 && vt[4 * slot + 1] == k1 && vt[4 * slot + 2] == k2)
                    {
                        const uint32_t old = atomicAdd(&tw[4 * slot + 3], kw);
                        if (MERGE && old + kw >= kSegLock) *vovf = 1;
                        have = false;
                    }
                    else slot = (slot + 1) & (SLOTS - 1);
                }
            }
            if (*vovf) break;
        }
        if (*vovf) break;
    }
    if (*vovf) break;
    }   // runs
    __syncthreads();
    if (ovf)
    {
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    // order the occupied slots by remainder (= by key: the prefix is common): registers, bucket sort on
    // the top remainder bits, insertion sort inside the buckets; bitonic network only for skewed bits
    constexpr int kPer = SLOTS / NT;
    constexpr int kBins = SLOTS / 4, kBinsPer = kBins / NT, kBinBits = kSlotBits - 2;
    static_assert(kBins % NT == 0, "bins per thread");
    __shared__ uint32_t bins[kBins];
    __shared__ uint32_t sh_scan2[NT / 64 + 1];
    __shared__ uint32_t big;
    uint4 c[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) c[j] = tbl4(tbl, tid * kPer + j);
    for (uint32_t i = tid; i < kBins; i += NT) bins[i] = 0;
    if (tid == 0) big = 0;
    __syncthreads();
    const uint32_t bsh = rem_bits > (uint32_t)kBinBits ? rem_bits - kBinBits : 0;
    auto rem_shr = [](const uint4& v, uint32_t sh) -> uint32_t {      // bits [sh, sh + 32) of the 96-bit remainder
        const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        if (sh == 0) return (uint32_t)lo;
        if (sh < 64) return (uint32_t)((lo >> sh) | ((uint64_t)v.z << (64 - sh)));
        return sh >= 96 ? 0u : (v.z >> (sh - 64));
    };
    auto rem_less = [](const uint4& a, const uint4& b2) { return a.z < b2.z || (a.z == b2.z && (a.y < b2.y || (a.y == b2.y && a.x < b2.x))); };
    uint32_t rnk[kPer], bin[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (c[j].w != 0)
        {
            bin[j] = rem_shr(c[j], bsh) & (kBins - 1);
            rnk[j] = atomicAdd(&bins[bin[j]], 1u);
        }
    __syncthreads();
    uint32_t bn[kBinsPer], bs[kBinsPer], mine = 0;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q) { bn[q] = bins[tid * kBinsPer + q]; mine += bn[q]; }
    uint32_t tot_occ;
    uint32_t at = block_excl_scan_n<uint32_t, NT / 64>(mine, sh_scan2, &tot_occ);
    lds_vu32 vbig = (lds_vu32)&big;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q)
    {
        bs[q] = at; bins[tid * kBinsPer + q] = at; at += bn[q];
        if (bn[q] > 24) *vbig = 1;
    }
    __syncthreads();
    uint4* t4 = reinterpret_cast<uint4*>(tbl);
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (c[j].w != 0) t4[bins[bin[j]] + rnk[j]] = c[j];
    __syncthreads();
    if (!big)
    {
#pragma unroll
        for (int q = 0; q < kBinsPer; ++q)
            for (uint32_t i = 1; i < bn[q]; ++i)
            {
                const uint4 v = t4[bs[q] + i];
                uint32_t j = i;
                while (j > 0 && rem_less(v, t4[bs[q] + j - 1])) { t4[bs[q] + j] = t4[bs[q] + j - 1]; --j; }
                t4[bs[q] + j] = v;
            }
        __syncthreads();
    }
    else
    {
        uint32_t nsort = 64;
        while (nsort < tot_occ) nsort <<= 1;
        for (uint32_t i = tot_occ + tid; i < nsort; i += NT) t4[i] = make_uint4(~0u, ~0u, ~0u, 0u);     // sorts last
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= nsort; k2 <<= 1)
        {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
            {
                for (uint32_t t = tid; t < nsort / 2; t += NT)
                {
                    const uint32_t i = 2 * t - (t & (j - 1));
                    const uint32_t p = i + j;
                    const bool up = (i & k2) == 0;
                    const uint4 a = t4[i], b2 = t4[p];
                    const bool gt = rem_less(b2, a);
                    if (gt == up) { t4[i] = b2; t4[p] = a; }
                }
                __syncthreads();
            }
        }
    }
    const uint32_t d = ndist;
    if (tid == 0)
    {
        sh_base = atomicAdd(&so->cursor, (unsigned long long)d);
        if (sh_base + d > so->stage_cap) { atomicOr(&so->overflow, 2u); sh_base = ~0ULL; }
        seg_pos[s] = sh_base;
        seg_cnt[s] = d;
    }
    __syncthreads();
    const uint64_t ob = sh_base;
    if (ob == ~0ULL) return;
    // the full key again: remainder | segment number << rem_bits
    const unsigned __int128 prefix = (unsigned __int128)s << rem_bits;
    for (uint32_t i = tid; i < d; i += NT)
    {
        const uint4 v = t4[i];
        const unsigned __int128 full = prefix | ((unsigned __int128)v.z << 64) | ((uint64_t)v.x | ((uint64_t)v.y << 32));
        stage_keys[ob + i] = Key2{(uint64_t)full, (uint64_t)(full >> 64)};
        stage_counts[ob + i] = v.w;
    }
}

constexpr int kSeg96Slots = 8192;
constexpr int kSeg96Limit = kSeg96Slots / 4 * 3;
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_reduce96_kernel(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                                           const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                           uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                           Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                           uint32_t rem_bits)
{
    seg_hash_reduce96_body<kSegBigThreads, kSeg96Slots>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits);
}
// the same reading 12-byte remainder records (the second level's rem_out form)
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_reduce96p_kernel(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                                            const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                            uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                            Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                            uint32_t rem_bits)
{
    seg_hash_reduce96_body<kSegBigThreads, kSeg96Slots, false, true>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits);
}
// merge of sorted (key,count) runs by 16-bit segments through the same table (bounds: [nruns][65537] from seg_bounds_kernel)
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_merge96_kernel(const Key2* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                          const uint64_t* __restrict__ run_off, const uint64_t* __restrict__ bounds,
                                                                          uint32_t nruns, SegOut* __restrict__ so,
                                                                          uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                          Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                          uint32_t rem_bits)
{
    seg_hash_reduce96_body<kSegBigThreads, kSeg96Slots, true>(keys, bounds, nullptr, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits,
                                                              vals, run_off, nruns);
}

__global__ __launch_bounds__(kTB) void seg_hash_reduce2_kernel(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                               const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so, uint64_t* __restrict__ seg_pos,
                                                               uint64_t* __restrict__ seg_cnt,
                                                               Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                               uint32_t rem_bits)
{
    seg_hash_reduce2_body<kTB, kSegSlots2>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits, 0u);
}

// One 1024-thread workgroup per CU and 4096 slots (80 KB of LDS): 16-bit segments of up to 2 300
// distinct two-word keys keep the two-level form (see seg_hash_reduce_big_kernel).
constexpr int kSegBigSlots2 = 4096;
constexpr int kSegBigLimit2 = kSegBigSlots2 / 4 * 3;
// ... and with the largest table a CU's LDS holds for two-word keys (6144 slots of 20 bytes + the sort's
// bins = 128 KB): up to 4608 distinct keys per segment counted by ONE workgroup, where the 4096-slot table
// needs two that each read the whole segment
constexpr int kSegWideSlots2 = 6144;
constexpr int kSegWideLimit2 = kSegWideSlots2 / 4 * 3;
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_reduce2_wide_kernel(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                                               const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                               uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                               Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                               uint32_t rem_bits)
{
    seg_hash_reduce2_body<kSegBigThreads, kSegWideSlots2>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits, 0u);
}
__global__ __launch_bounds__(kSegBigThreads) void seg_hash_reduce2_big_kernel(const Key2* __restrict__ keys, const uint64_t* __restrict__ seg_off,
                                                                              const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                              uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                              Key2* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                              uint32_t rem_bits, uint32_t round_bits)
{
    seg_hash_reduce2_body<kSegBigThreads, kSegBigSlots2>(keys, seg_off, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rem_bits, round_bits);
}

// --------------------------------------------------------------------------------------
// Merge of sorted (key,count) runs by segments
// --------------------------------------------------------------------------------------
//
// The runs of the chunks (or of the ranks of a multi-GPU exchange, or of the inputs of merge-*)
// are each sorted and distinct.  Instead of sorting their concatenation again (7 to 14 radix
// passes), the key space is cut into segments small enough for LDS: the bounds of every segment
// inside every run come from binary searches (seg_bounds_kernel), and one workgroup per segment
// loads its at most kMergeCap entries from all runs, orders them with a bitonic network, adds up
// the counts of equal keys and appends the result to a staging area -- every entry is read once
// and written once.
constexpr int kMergeCap = 2048;
constexpr int kMergeRuns = 64;

// total[s] = sum over runs of the segment's length; *maxv = largest total.
__global__ void seg_totals_kernel(const uint64_t* __restrict__ bounds, uint32_t nruns, uint32_t nseg,
                                  unsigned long long* __restrict__ maxv)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    uint64_t t = 0;
    for (uint32_t r = 0; r < nruns; ++r)
    {
        const uint64_t* b = bounds + (uint64_t)r * (nseg + 1);
        t += b[s + 1] - b[s];
    }
    atomicMax(maxv, (unsigned long long)t);
}

template <class K> __device__ inline K key_max();
template <> __device__ inline Key1 key_max<Key1>() { return Key1{~0ULL}; }
template <> __device__ inline Key2 key_max<Key2>() { return Key2{~0ULL, ~0ULL}; }

template <class K>
__global__ __launch_bounds__(kTB) void seg_merge_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                        const uint64_t* __restrict__ run_off, const uint64_t* __restrict__ bounds,
                                                        uint32_t nruns, uint32_t nseg, SegOut* __restrict__ so,
                                                        uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                        K* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                        uint32_t* __restrict__ count_overflow)
{
    __shared__ K lk[kMergeCap];
    __shared__ uint32_t lc[kMergeCap];
    __shared__ uint64_t rsrc[kMergeRuns];        // first source index of the segment in run r
    __shared__ uint32_t rpre[kMergeRuns + 1];    // entries of runs < r
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ unsigned long long sh_base;
    const uint32_t s = unit_block(), tid = threadIdx.x;
    if (tid == 0)
    {
        uint32_t n0 = 0;
        for (uint32_t r = 0; r < nruns; ++r)
        {
            const uint64_t* b = bounds + (uint64_t)r * (nseg + 1);
            rsrc[r] = run_off[r] + b[s];
            rpre[r] = n0;
            n0 += (uint32_t)(b[s + 1] - b[s]);       // the host checked total <= kMergeCap
        }
        rpre[nruns] = n0;
    }
    __syncthreads();
    const uint32_t n = rpre[nruns];
    if (n == 0)
    {
        if (tid == 0) { seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    // load: entry i of the segment's concatenation comes from run r(i)
    for (uint32_t i = tid; i < n; i += kTB)
    {
        uint32_t r = 0;
        while (i >= rpre[r + 1]) ++r;
        const uint64_t src = rsrc[r] + (i - rpre[r]);
        lk[i] = keys[src];
        lc[i] = vals[src];
    }
    __syncthreads();
    // merge by ranks: every sub-run is sorted, so the final place of an entry is its index in
    // its own run plus, for every other run, the number of that run's entries that go before it
    // (ties go to the lower run) -- binary searches in LDS, no barriers in between
    constexpr int kPerT = kMergeCap / kTB;
    K mk[kPerT];
    uint32_t mc[kPerT], mp[kPerT];
#pragma unroll
    for (int j = 0; j < kPerT; ++j)
    {
        const uint32_t i = tid + j * kTB;
        if (i < n)
        {
            uint32_t r = 0;
            while (i >= rpre[r + 1]) ++r;
            const K k = lk[i];
            uint32_t pos = i - rpre[r];
            for (uint32_t q = 0; q < nruns; ++q)
            {
                if (q == r) continue;
                uint32_t lo = rpre[q], hi = rpre[q + 1];
                const uint32_t base = lo;
                if (q < r) { while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (lk[m] < k || lk[m] == k) lo = m + 1; else hi = m; } }
                else       { while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (lk[m] < k) lo = m + 1; else hi = m; } }
                pos += lo - base;
            }
            mk[j] = k; mc[j] = lc[i]; mp[j] = pos;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPerT; ++j)
        if (tid + j * kTB < n) { lk[mp[j]] = mk[j]; lc[mp[j]] = mc[j]; }
    __syncthreads();
    // distinct keys of the segment, then their places in the staging area
    uint32_t heads = 0;
    for (uint32_t i = tid; i < n; i += kTB) heads += (i == 0 || lk[i] != lk[i - 1]) ? 1u : 0u;
    uint32_t d;
    block_excl_scan<uint32_t>(heads, sh_scan, &d);
    if (tid == 0)
    {
        sh_base = atomicAdd(&so->cursor, (unsigned long long)d);
        if (sh_base + d > so->stage_cap) { atomicOr(&so->overflow, 2u); sh_base = ~0ULL; }
        seg_pos[s] = sh_base;
        seg_cnt[s] = d;
    }
    __syncthreads();
    const uint64_t ob = sh_base;
    if (ob == ~0ULL) return;
    uint32_t done = 0;                               // heads in the chunks before this one
    for (uint32_t c0 = 0; c0 < n; c0 += kTB)
    {
        const uint32_t i = c0 + tid;
        const bool head = i < n && (i == 0 || lk[i] != lk[i - 1]);
        uint32_t tot;
        const uint32_t before = block_excl_scan<uint32_t>(head ? 1u : 0u, sh_scan, &tot);
        if (head)
        {
            uint64_t sum = 0;
            for (uint32_t j = i; j < n && lk[j] == lk[i]; ++j) sum += lc[j];
            if (sum >= 0xFFFFFFFFULL) { atomicOr(count_overflow, 1u); sum = 0xFFFFFFFFULL; }
            stage_keys[ob + done + before] = lk[i];
            stage_counts[ob + done + before] = (uint32_t)sum;
        }
        done += tot;
    }
}

// Order the (key,count) pairs of every segment in place: the pairs are already grouped by their top
// bits (segment s = [seg_off[s], seg_off[s+1])), at most kSortCap per segment.  One workgroup per
// segment: pairs into registers, bucket sort through LDS on the 10 bits below the segment prefix (rank
// by LDS atomic, scan, scatter, insertion sort of the ~1.5-pair buckets), coalesced write-back.  A
// segment above kSortCap raises *fallback: the host orders the array by a full radix sort instead; a bucket
// above 24 pairs (clustered keys) sends that segment through a bitonic network.  Used by canonicalize_run after two radix passes on the top 16 bits.
constexpr int kSortCap = 4096;
__global__ __launch_bounds__(kTB) void seg_sort_pairs_kernel(Key1* __restrict__ keys, uint32_t* __restrict__ vals,
                                                             const uint64_t* __restrict__ seg_off, uint32_t rem_bits,
                                                             uint32_t* __restrict__ fallback)
{
    constexpr int kPer = kSortCap / kTB;          // 16
    constexpr int kBins = 1024, kBinsPer = kBins / kTB;
    __shared__ unsigned long long tab[kSortCap];
    __shared__ uint32_t cnt[kSortCap];
    __shared__ uint32_t bins[kBins];
    __shared__ uint32_t sh_scan[kWaves + 1];
    const uint32_t s = unit_block(), tid = threadIdx.x;
    const uint64_t b = seg_off[s], e = seg_off[s + 1];
    const uint32_t n = (uint32_t)(e - b);
    if (e - b > (uint64_t)kSortCap) { if (tid == 0) atomicOr(fallback, 1u); return; }
    if (n < 2) return;
    unsigned long long ck[kPer];
    uint32_t cc[kPer], rnk[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)
    {
        const uint32_t i = tid + j * kTB;
        ck[j] = i < n ? keys[b + i].lo : ~0ULL;
        cc[j] = i < n ? vals[b + i] : 0u;
    }
    for (uint32_t i = tid; i < kBins; i += kTB) bins[i] = 0;
    __syncthreads();
    const uint32_t bsh = rem_bits > 10 ? rem_bits - 10 : 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (tid + j * kTB < n) rnk[j] = atomicAdd(&bins[(uint32_t)(ck[j] >> bsh) & (kBins - 1)], 1u);
    __syncthreads();
    uint32_t bn[kBinsPer], bs[kBinsPer], mine = 0;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q) { bn[q] = bins[tid * kBinsPer + q]; mine += bn[q]; }
    uint32_t tot;
    uint32_t at = block_excl_scan<uint32_t>(mine, sh_scan, &tot);
    bool big = false;
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q)
    {
        bs[q] = at; bins[tid * kBinsPer + q] = at; at += bn[q];
        big |= bn[q] > 24;
    }
    if (__syncthreads_or(big))
    {
        // clustered keys (the variants of a k-mer that differ in their last bases share a bin): this segment is
        // ordered by a bitonic network over its pairs instead -- a local matter, the other segments keep the fast way
        uint32_t nsort = 64;
        while (nsort < n) nsort <<= 1;
#pragma unroll
        for (int j = 0; j < kPer; ++j)
        {
            const uint32_t i = tid + j * kTB;
            if (i < nsort) { tab[i] = ck[j]; cnt[i] = cc[j]; }        // beyond n: all ones, sorts last
        }
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= nsort; k2 <<= 1)
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
            {
                for (uint32_t t = tid; t < nsort / 2; t += kTB)
                {
                    const uint32_t i = 2 * t - (t & (j - 1));
                    const uint32_t p = i + j;
                    const bool up = (i & k2) == 0;
                    const unsigned long long a = tab[i], c2 = tab[p];
                    if ((a > c2) == up)
                    {
                        tab[i] = c2; tab[p] = a;
                        const uint32_t ca = cnt[i]; cnt[i] = cnt[p]; cnt[p] = ca;
                    }
                }
                __syncthreads();
            }
        for (uint32_t i = tid; i < n; i += kTB) { keys[b + i].lo = tab[i]; vals[b + i] = cnt[i]; }
        return;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (tid + j * kTB < n)
        {
            const uint32_t pos = bins[(uint32_t)(ck[j] >> bsh) & (kBins - 1)] + rnk[j];
            tab[pos] = ck[j]; cnt[pos] = cc[j];
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kBinsPer; ++q)
        for (uint32_t i = 1; i < bn[q]; ++i)
        {
            const unsigned long long kk = tab[bs[q] + i];
            const uint32_t vv = cnt[bs[q] + i];
            uint32_t j = i;
            while (j > 0 && tab[bs[q] + j - 1] > kk)
            {
                tab[bs[q] + j] = tab[bs[q] + j - 1]; cnt[bs[q] + j] = cnt[bs[q] + j - 1];
                --j;
            }
            tab[bs[q] + j] = kk; cnt[bs[q] + j] = vv;
        }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kTB) { keys[b + i].lo = tab[i]; vals[b + i] = cnt[i]; }
}

// Restore segment order: out[seg_dst[s] + i] = stage[seg_pos[s] + i].
template <class K>
__global__ __launch_bounds__(kTB) void seg_gather_kernel(const K* __restrict__ stage_keys, const uint32_t* __restrict__ stage_counts,
                                                         const uint64_t* __restrict__ seg_pos, const uint64_t* __restrict__ seg_dst,
                                                         const uint64_t* __restrict__ seg_cnt_unscanned,
                                                         K* __restrict__ out_keys, uint32_t* __restrict__ out_counts)
{
    const uint32_t s = unit_block();
    const uint64_t d = seg_cnt_unscanned[s];
    const uint64_t src = seg_pos[s], dst = seg_dst[s];
    for (uint64_t i = threadIdx.x; i < d; i += kTB)
    {
        out_keys[dst + i] = stage_keys[src + i];
        out_counts[dst + i] = stage_counts[src + i];
    }
}

// --------------------------------------------------------------------------------------
// K7: Elias-Fano split (SparseArray::Builder::push_back, SparseArray.hh:87-118)
// --------------------------------------------------------------------------------------

struct EfColumn { uint8_t* dst; uint32_t bytes; uint32_t shift; };
struct EfColumns { EfColumn c[4]; uint32_t n; };

// bits [shift, shift+64) of (key & (2^D - 1))
template <class K>
__device__ __forceinline__ uint64_t masked_bits(const K& k, uint32_t D, uint32_t shift)
{
    uint64_t lo = key_lo_word(k), hi = key_hi_word(k);
    if (D < 64) { lo &= (1ULL << D) - 1; hi = 0; }
    else if (D < 128) { hi &= D == 64 ? 0 : ((1ULL << (D - 64)) - 1); }
    if (shift == 0) return lo;
    if (shift < 64) return (lo >> shift) | (hi << (64 - shift));
    return shift >= 128 ? 0 : (hi >> (shift - 64));
}

template <class K>
__global__ void ef_low_bits_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D, EfColumns cols)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    K k = keys[i];
    for (uint32_t c = 0; c < cols.n; ++c)
    {
        uint64_t v = masked_bits(k, D, cols.c[c].shift);
        uint8_t* p = cols.c[c].dst + i * cols.c[c].bytes;
        switch (cols.c[c].bytes)
        {
            case 1: *p = (uint8_t)v; break;
            case 2: *reinterpret_cast<uint16_t*>(p) = (uint16_t)v; break;
            case 4: *reinterpret_cast<uint32_t*>(p) = (uint32_t)v; break;
            default: *reinterpret_cast<uint64_t*>(p) = v; break;
        }
    }
}

// high part of key i: (key >> D) as u64 (D >= 128 -> 0)
template <class K>
__device__ __forceinline__ uint64_t ef_hi(const K* keys, uint64_t i, uint32_t D)
{
    return D >= 128 ? 0 : key_shr64(keys[i], D);
}

template <class K>
__global__ void ef_check_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D, uint32_t* __restrict__ err)
{
    // the largest key decides whether every high part fits 64 bits
    if (blockIdx.x == 0 && threadIdx.x == 0 && m)
    {
        if (D < 128 && key_shr_overflows(keys[m - 1], D)) atomicOr(err, 1u);
    }
}

// One thread per 64-bit word of the high-bits bitmap.  Position of one i is
// h_i = (key_i >> D) + i, strictly increasing, so the ones of word w are found by a binary
// search for the first h_i >= 64w and a short forward walk.
template <class K>
__global__ void ef_high_bits_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D,
                                    uint64_t nwords, uint64_t* __restrict__ words)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const uint64_t lo_pos = w * 64;
    uint64_t a = 0, b = m;
    while (a < b)
    {
        uint64_t mid = a + ((b - a) >> 1);
        if (ef_hi(keys, mid, D) + mid < lo_pos) a = mid + 1; else b = mid;
    }
    uint64_t bits = 0;
    for (uint64_t i = a; i < m; ++i)
    {
        uint64_t h = ef_hi(keys, i, D) + i;
        if (h >= lo_pos + 64) break;
        bits |= 1ULL << (h - lo_pos);
    }
    words[w] = bits;
}

// --------------------------------------------------------------------------------------
// K8: DenseSelect image (DenseSelect::Builder, DenseArray.cc:446-694)
// --------------------------------------------------------------------------------------
//
// The indexed positions are never materialised: for sense 1 (ones) position i is h_i; for
// sense 0 (zeros) the j-th zero sits at j + #{i : (key_i >> D) <= j}.

template <class K>
__device__ __forceinline__ uint64_t ds_pos(const K* keys, uint64_t m, uint32_t D, int invert, uint64_t idx)
{
    if (!invert) return ef_hi(keys, idx, D) + idx;
    uint64_t a = 0, b = m;                  // upper_bound of idx among the high parts
    while (a < b)
    {
        uint64_t mid = a + ((b - a) >> 1);
        if (ef_hi(keys, mid, D) <= idx) a = mid + 1; else b = mid;
    }
    return idx + a;
}

enum : uint32_t { kDsSmall = 0, kDsSpill64 = 1, kDsSpill32 = 2, kDsSpill16 = 3, kDsSpill8 = 4, kDsIntermediate = 5 };

// Pass 1: one thread per block of 8192 indexed positions: block type and byte size (already
// padded to 8).  count = number of indexed positions.
template <class K>
__global__ void ds_classify_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D, int invert,
                                   uint64_t count, uint64_t nblocks,
                                   uint32_t* __restrict__ btype, uint64_t* __restrict__ bbytes,
                                   uint64_t* __restrict__ brank)
{
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint64_t first = b << 13;
    uint64_t cnt = count - first < 8192 ? count - first : 8192;
    uint64_t pp = ds_pos(keys, m, D, invert, first);
    uint64_t p = ds_pos(keys, m, D, invert, first + cnt - 1);
    uint64_t span = p - pp;
    uint32_t t;
    uint64_t bytes;
    if (span >= (1ULL << 24) || cnt < 8192)
    {
        if (span < (1ULL << 32)) { t = kDsSpill32; bytes = cnt * 4; }
        else { t = kDsSpill64; bytes = cnt * 8; }
    }
    else if (span >= (1ULL << 16))
    {
        t = kDsIntermediate;
        bytes = 128 * 6;
        for (uint32_t s = 0; s < 128; ++s)
        {
            uint64_t r = ds_pos(keys, m, D, invert, first + s * 64 + 63) - ds_pos(keys, m, D, invert, first + s * 64);
            if (r <= 128) {}
            else if (r < 256) bytes += 64;
            else if (r < 65536) bytes += 128;
            else bytes += 256;
        }
    }
    else { t = kDsSmall; bytes = 256; }
    btype[b] = t;
    bbytes[b] = (bytes + 7) & ~7ULL;
    brank[b] = pp;
}

// Pass 2: one workgroup of 128 threads per block writes the block body at boff[b] and the
// master index entry.  The image was zero-filled, so alignment padding is already there.
template <class K>
__global__ __launch_bounds__(128) void ds_fill_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D, int invert,
                                                      uint64_t count, const uint32_t* __restrict__ btype,
                                                      const uint64_t* __restrict__ boff, const uint64_t* __restrict__ brank,
                                                      uint8_t* __restrict__ image, uint64_t* __restrict__ index)
{
    __shared__ uint32_t sh_sub[128];
    const uint64_t b = blockIdx.x;
    const uint32_t s = threadIdx.x;
    const uint64_t first = b << 13;
    const uint64_t cnt = count - first < 8192 ? count - first : 8192;
    const uint32_t t = btype[b];
    const uint64_t off = boff[b];
    const uint64_t pp = brank[b];
    uint8_t* blk = image + off;
    if (s == 0) index[b] = off | t;
    if (t == kDsSmall)
    {
        uint16_t v = (uint16_t)(ds_pos(keys, m, D, invert, first + (uint64_t)s * 64) - pp);
        reinterpret_cast<uint16_t*>(blk)[s] = v;
    }
    else if (t == kDsSpill32)
    {
        for (uint64_t i = s; i < cnt; i += 128)
            reinterpret_cast<uint32_t*>(blk)[i] = (uint32_t)(ds_pos(keys, m, D, invert, first + i) - pp);
    }
    else if (t == kDsSpill64)
    {
        for (uint64_t i = s; i < cnt; i += 128)
            reinterpret_cast<uint64_t*>(blk)[i] = ds_pos(keys, m, D, invert, first + i);
    }
    else
    {
        // intermediate: 128 x u32 sample offsets, 128 x u16 internal pointers, sub-blocks
        uint64_t p0 = ds_pos(keys, m, D, invert, first + (uint64_t)s * 64);
        uint64_t p1 = ds_pos(keys, m, D, invert, first + (uint64_t)s * 64 + 63);
        uint64_t r = p1 - p0;
        reinterpret_cast<uint32_t*>(blk)[s] = (uint32_t)(p0 - pp);
        uint32_t sz = r <= 128 ? 0u : r < 256 ? 64u : r < 65536 ? 128u : 256u;
        uint32_t ty = r <= 128 ? 0u : r < 256 ? kDsSpill8 : r < 65536 ? kDsSpill16 : kDsSpill32;
        sh_sub[s] = sz;
        __syncthreads();
        uint32_t base = 768;
        for (uint32_t i = 0; i < s; ++i) base += sh_sub[i];
        uint16_t ip = sz ? (uint16_t)(base | ty) : (uint16_t)0;
        reinterpret_cast<uint16_t*>(blk + 512)[s] = ip;
        if (sz)
        {
            for (uint32_t j = 0; j < 64; ++j)
            {
                uint64_t d = ds_pos(keys, m, D, invert, first + (uint64_t)s * 64 + j) - p0;
                if (ty == kDsSpill8) blk[base + j] = (uint8_t)d;
                else if (ty == kDsSpill16) reinterpret_cast<uint16_t*>(blk + base)[j] = (uint16_t)d;
                else reinterpret_cast<uint32_t*>(blk + base)[j] = (uint32_t)d;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// K9: counts -> VariableByteArray pieces (VariableByteArray.hh:81-103)
// --------------------------------------------------------------------------------------

__global__ void vba_ord0_kernel(const uint32_t* __restrict__ counts, uint64_t m, uint8_t* __restrict__ ord0,
                                uint64_t* __restrict__ flag1)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t c = counts[i];
    ord0[i] = (uint8_t)(c & 0xFF);
    flag1[i] = (c >> 8) ? 1 : 0;
}

// after an exclusive scan of flag1 -> slot: gather the items with count > 255
__global__ void vba_ord1_kernel(const uint32_t* __restrict__ counts, uint64_t m, const uint64_t* __restrict__ slot,
                                uint64_t n1, uint64_t* __restrict__ pos1, uint8_t* __restrict__ ord1,
                                uint32_t* __restrict__ hi16)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t c = counts[i];
    if (c >> 8)
    {
        uint64_t s = slot[i];
        (void)n1;
        pos1[s] = i;
        ord1[s] = (uint8_t)((c >> 8) & 0xFF);
        hi16[s] = c >> 16;
    }
}

__global__ void vba_flag2_kernel(const uint32_t* __restrict__ hi16, uint64_t n1, uint64_t* __restrict__ flag2)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    flag2[i] = hi16[i] ? 1 : 0;
}

__global__ void vba_ord2_kernel(const uint32_t* __restrict__ hi16, uint64_t n1, const uint64_t* __restrict__ slot,
                                uint64_t* __restrict__ pos2, uint16_t* __restrict__ ord2)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    uint32_t h = hi16[i];
    if (h)
    {
        uint64_t s = slot[i];
        pos2[s] = i;
        ord2[s] = (uint16_t)h;
    }
}

// Distributed emission: the high part key >> D of every key of a range, as u32 or u64.
template <class K, class T>
__global__ void ef_high_part_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D, T* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = (T)(D >= 128 ? 0 : key_shr64(keys[i], D));
}

// ... and the entries with count > 255 as (global index, count) pairs, after a scan of vba_ord0_kernel's flags
struct BigCount { unsigned long long index; uint32_t count, pad; };
__global__ void vba_big_kernel(const uint32_t* __restrict__ counts, uint64_t m, const uint64_t* __restrict__ slot,
                               uint64_t first_index, BigCount* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t c = counts[i];
    if (c >> 8) out[slot[i]] = BigCount{first_index + i, c, 0u};
}

__global__ void widen_counts_kernel(const uint32_t* __restrict__ counts, uint64_t m, Key1* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i].lo = counts[i];
}

// VariableByteArray read side (VariableByteArray::operator[] / GeneralIterator, VariableByteArray.hh:120-247) as
// three passes over the whole array: byte 0 of every value from ord0; the items listed in the ord1p presence
// array get bits 8..15 from ord1; the entries of that list which ord2p names get bits 16..31 from ord2.
__global__ void vba_read0_kernel(const uint8_t* __restrict__ ord0, uint64_t m, uint32_t* __restrict__ counts)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) counts[i] = ord0[i];
}
__global__ void vba_read1_kernel(const Key1* __restrict__ pos1, const uint8_t* __restrict__ ord1, uint64_t n1, uint64_t m,
                                 uint32_t* __restrict__ counts)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n1 && pos1[j].lo < m) counts[pos1[j].lo] |= (uint32_t)ord1[j] << 8;
}
__global__ void vba_read2_kernel(const Key1* __restrict__ pos2, const uint16_t* __restrict__ ord2, uint64_t n2,
                                 const Key1* __restrict__ pos1, uint64_t n1, uint64_t m, uint32_t* __restrict__ counts)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n2 || pos2[t].lo >= n1) return;
    const uint64_t i = pos1[pos2[t].lo].lo;
    if (i < m) counts[i] |= (uint32_t)ord2[t] << 16;
}

// --------------------------------------------------------------------------------------
// SparseArray decode (SparseArray::LazyIterator, SparseArray.hh:185-224): the i-th one of the
// high-bits bitmap at position p gives the key ((p - i) << D) + low[i].  Used to read existing
// KmerSet / Graph objects back as sorted runs (merge-kmer-sets, merge-graphs).
// --------------------------------------------------------------------------------------

__global__ void popc_words_kernel(const uint64_t* __restrict__ words, uint64_t nwords, uint64_t* __restrict__ counts)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) counts[w] = (uint64_t)__popcll(words[w]);
}

struct EfColumnsIn { const uint8_t* src[4]; uint32_t bytes[4]; uint32_t shift[4]; uint32_t n; };

template <class K>
__global__ void ef_decode_kernel(const uint64_t* __restrict__ words, uint64_t nwords, const uint64_t* __restrict__ prefix,
                                 uint32_t D, EfColumnsIn cols, uint64_t count, K* __restrict__ out)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint64_t x = words[w];
    uint64_t r = prefix[w];
    while (x)
    {
        const uint32_t b = (uint32_t)__ffsll((unsigned long long)x) - 1;
        x &= x - 1;
        if (r >= count) break;
        unsigned __int128 low = 0;
        for (uint32_t c = 0; c < cols.n; ++c)
        {
            uint64_t v = 0;
            const uint8_t* p = cols.src[c] + r * cols.bytes[c];
            switch (cols.bytes[c])
            {
                case 1: v = *p; break;
                case 2: v = *reinterpret_cast<const uint16_t*>(p); break;
                case 4: v = *reinterpret_cast<const uint32_t*>(p); break;
                default: v = *reinterpret_cast<const uint64_t*>(p); break;
            }
            low |= (unsigned __int128)v << cols.shift[c];
        }
        unsigned __int128 pos = (unsigned __int128)(w * 64 + b - r);
        pos = D >= 128 ? 0 : (pos << D);
        pos += low;
        K k;
        k.lo = (uint64_t)pos;
        if (K::kWords == 2) reinterpret_cast<uint64_t*>(&k)[K::kWords - 1] = (uint64_t)(pos >> 64);
        out[r] = k;
        ++r;
    }
}

__global__ void fill_u32_kernel(uint32_t* __restrict__ a, uint64_t n, uint32_t v)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

// --------------------------------------------------------------------------------------
// synthetic reads
// --------------------------------------------------------------------------------------

__global__ void synth_reads_kernel(uint8_t* __restrict__ out, uint64_t nreads, uint32_t read_len,
                                   uint64_t genome_len, uint64_t seed, uint64_t first_read)
{
    const uint64_t stride = read_len + 1;
    uint64_t total = nreads * stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t r = i / stride;
        uint32_t j = (uint32_t)(i - r * stride);
        out[i] = (uint8_t)synth_read_byte(seed, genome_len, read_len, first_read + r, j);
    }
}

}  // namespace goss
