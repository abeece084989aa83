// kernels_extract.hpp -- k-mer extraction: plain kernels and the extraction fused with the first partition level.
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_common.hpp"

namespace goss {

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
// (PACKED: `aligned` is the string's first group of sixteen positions in the flag array of a packed string -- one u16 a group)
template <bool PACKED = false>
__global__ __launch_bounds__(kTB) void nonbase_sample_kernel(const uint8_t* __restrict__ aligned, uint64_t nslices,
                                                             uint64_t stride, uint32_t slice,
                                                             unsigned long long* __restrict__ out)
{
    unsigned long long bad = 0, seen = 0;
    for (uint64_t s = blockIdx.x; s < nslices; s += gridDim.x)
    {
        if constexpr (PACKED)
        {
            const uint16_t* f = reinterpret_cast<const uint16_t*>(aligned) + (s * stride >> 4);
            for (uint32_t v = threadIdx.x; v < slice / 16; v += kTB) { bad += __popc((uint32_t)f[v]); seen += 16; }
            continue;
        }
        const uint4* p = reinterpret_cast<const uint4*>(aligned + s * stride);
        for (uint32_t v = threadIdx.x; v < slice / 16; v += kTB)
        {
            const uint4 q = p[v];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                uint32_t b;
                (void)base_codes(w[i], b);
                bad += __popc(b);
            }
            seen += 16;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_down(bad, o, 64); seen += __shfl_down(seen, o, 64); }
    if ((threadIdx.x & 63u) == 0 && seen) { atomicAdd(&out[0], bad); atomicAdd(&out[1], seen); }
}

// 2-bit packed bases (goss_gpu_push_packed_host*: one u32 of codes and one u16 of non-base flags per 16 positions,
// packed by the host's parser threads so that 3 bits per base cross PCIe instead of 8) -> the byte string the
// extraction kernels read: "ACGT"[code], or a newline where the flag is set.  One thread per group of 16; `out` is
// 16-byte aligned.  The role of GossReadBaseString's per-base encoder (GossReadBaseString.hh:133-188), inverted:
// the device keeps ONE input form, and unpacking costs one write and one read of a byte per base in HBM.
__global__ __launch_bounds__(kTB) void unpack_bases_kernel(const uint32_t* __restrict__ codes, const uint16_t* __restrict__ nonbase,
                                                           uint64_t ngroups, uint64_t npos, uint8_t* __restrict__ out)
{
    // groups 0 .. ngroups-1 hold positions; positions at or beyond npos, and one more whole group behind the last
    // (written by the thread ngroups), are separators: the string a push leaves behind ends on a 16-byte boundary
    // with a separator, and the host issues no fill calls of its own
    const uint64_t g = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (g > ngroups) return;
    uint32_t c = 0, b = 0xFFFFu;
    if (g < ngroups)
    {
        c = codes[g];
        b = nonbase[g];
        const uint64_t left = npos - g * 16;
        if (left < 16) b |= 0xFFFFu << (uint32_t)left;
    }
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const uint32_t code = (c >> (2 * (4 * i + j))) & 3u;
            // A C G T = 0x41 0x43 0x47 0x54
            uint32_t ch = code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u;
            if ((b >> (4 * i + j)) & 1u) ch = 0x0Au;
            x |= ch << (8 * j);
        }
        w[i] = x;
    }
    reinterpret_cast<uint4*>(out)[g] = make_uint4(w[0], w[1], w[2], w[3]);
}

// The byte form of a base string -> the packed form (goss_gpu_pack_bases_device): group g = positions 16 g .. 16 g + 15
// of the string that starts `mis` bytes into a 16-byte aligned address; one u32 of codes and one u16 of flags per
// group, positions at or beyond npos flagged.  GossReadBaseString's per-base encoder (GossReadBaseString.hh:133-188) as
// a kernel of its own: what the fused kernels do with every 16-byte vector they fetch, for a caller that wants the
// reads resident in 3 bits per base.  1 byte read, 3/8 byte written per position.
__global__ __launch_bounds__(kTB) void pack_bases_kernel(const uint8_t* __restrict__ aligned, uint32_t mis, uint64_t npos,
                                                         uint32_t* __restrict__ codes, uint16_t* __restrict__ nonbase, uint64_t ngroups)
{
    const uint64_t g = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (g >= ngroups) return;
    const uint64_t limit = (uint64_t)mis + npos;                 // bytes of the aligned string that exist
    // the two aligned vectors the group's 16 bytes lie in (one when the string itself is aligned)
    uint32_t w[8];
#pragma unroll
    for (int v = 0; v < 2; ++v)
    {
        const uint64_t byte0 = (g + v) * 16;
        uint32_t x[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
        if (v == 0 || mis)
        {
            if (byte0 + 16 <= limit)
            {
                const uint4 q = *reinterpret_cast<const uint4*>(aligned + byte0);
                x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
            }
            else if (byte0 < limit)
                for (int j = 0; j < 16; ++j)
                {
                    const uint64_t b = byte0 + j;
                    const uint32_t c = b < limit ? aligned[b] : 0x0Au;
                    x[j >> 2] = (x[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) w[4 * v + i] = x[i];
    }
    // bytes mis .. mis + 15 of the 32: a byte funnel over neighbouring words (mis = 4 a + b)
    const uint32_t a = mis >> 2, b = mis & 3u;
    uint32_t c = 0, bads = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        uint32_t lo = w[i], hi = w[i + 1];
        if (a == 1) { lo = w[i + 1]; hi = w[i + 2]; }
        else if (a == 2) { lo = w[i + 2]; hi = w[i + 3]; }
        else if (a == 3) { lo = w[i + 3]; hi = w[i + 4]; }
        const uint32_t word = __builtin_amdgcn_alignbyte(hi, lo, b);
        uint32_t bad;
        const uint32_t x = base_codes(word, bad);
        c |= pack_codes(x) << (8 * i);
        bads |= pack_flags(bad) << (4 * i);
    }
    const uint64_t left = npos - g * 16;                         // (>= 1: g < ngroups)
    if (left < 16) bads = (bads | (0xFFFFu << (uint32_t)left)) & 0xFFFFu;
    codes[g] = c;
    nonbase[g] = (uint16_t)bads;
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
// The same for two-word keys (32 <= len <= 63), used where BOTH strands of every window are wanted in the end (graph
// mode): the windows are counted as one representative per strand pair and the pairs are expanded after counting
// (graph_expand_kernel).  Odd len: the strand whose middle base has a clear low bit (bit len - 1).  Even len: the
// strand whose halves (len bits each), swapped, give the smaller value -- low half first, then high half.
__device__ __forceinline__ Key2 strand_rep2(const Key2& f, const Key2& rc, uint32_t len, uint64_t lmask)
{
    // The strand is picked WORD BY WORD from values pinned in registers: `cond ? rc : f` on the two references -- and
    // equally two selects between their fields, which the compiler folds back into one -- selects an ADDRESS, which puts
    // both keys into scratch memory and every window through a store, a dependent load and a wait for the memory
    // counter: what two-word extraction spent most of its time on until round 4.
    uint64_t flo = f.lo, fhi = f.hi, rlo = rc.lo, rhi = rc.hi;
    asm volatile("" : "+v"(flo), "+v"(fhi), "+v"(rlo), "+v"(rhi));
    if (len & 1u)
    {
        const uint32_t b = len - 1;              // 32 .. 62
        const bool take = (flo >> b) & 1ULL;
        return Key2{take ? rlo : flo, take ? rhi : fhi};
    }
    const uint64_t lf = flo & lmask, lr = rlo & lmask;
    bool take_rc = lr < lf;
    // (the high halves decide a tie of the low ones -- one window in 2^len: looked at only when a lane of the wave has one)
    if (__builtin_amdgcn_ballot_w64(lr == lf))
    {
        const uint64_t hf = ((flo >> len) | (fhi << (64 - len))) & lmask, hr = ((rlo >> len) | (rhi << (64 - len))) & lmask;
        if (lr == lf) take_rc = hr < hf;
    }
    return Key2{take_rc ? rlo : flo, take_rc ? rhi : fhi};
}

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

template <class K, int MODE, int P, bool PACKED = false>
__global__ __launch_bounds__(kTB) void extract_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                      uint64_t nstarts, uint64_t navail, uint32_t len,
                                                      K* __restrict__ out, ExtractCounters* __restrict__ ctr,
                                                      const uint16_t* __restrict__ pbad = nullptr)
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
        if constexpr (PACKED)
        {
            // (a packed string: the group's codes and flags spread out to the code bytes this kernel works on)
            uint32_t cd, bd, o[4];
            load_group16<true>(bases_aligned, pbad, byte0, navail + mis, cd, bd);
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    x |= (((bd >> (4 * i + j)) & 1u) ? 4u : ((cd >> (2 * (4 * i + j))) & 3u)) << (8 * j);
                o[i] = x;
            }
            *reinterpret_cast<uint4*>(&code[v * 16]) = make_uint4(o[0], o[1], o[2], o[3]);
            continue;
        }
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
            // SWAR over 4 bytes: 2-bit code per byte, 0x80 where the byte is no base
            uint32_t bad;
            const uint32_t x = base_codes(w[i], bad);
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
template <int MODE, int P, int G, int NB, bool REP = false, bool PACKED = false>
__global__ __launch_bounds__(kTB) void extract1_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                       uint64_t nstarts, uint64_t navail, uint32_t len,
                                                       Key1* __restrict__ out, ExtractCounters* __restrict__ ctr,
                                                       uint32_t hist_shift, uint64_t nsuper,
                                                       uint64_t slice_tiles = 0, uint64_t slice_stride = 0,
                                                       const uint16_t* __restrict__ pbad = nullptr)
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
        uint32_t codes, bads;
        load_group16<PACKED>(bases_aligned, pbad, tile_base + (uint64_t)v * 16, navail + mis, codes, bads);
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
template <int MODE, int P, int G, int NBH = 8, bool PACKED = false>
__global__ __launch_bounds__(kTB) void extract2_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                       uint64_t nstarts, uint64_t navail, uint32_t len,
                                                       Key2* __restrict__ out, ExtractCounters* __restrict__ ctr, uint64_t nsuper,
                                                       uint64_t slice_tiles = 0, uint64_t slice_stride = 0,
                                                       const uint16_t* __restrict__ pbad = nullptr)
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
            uint32_t codes, bads;
            load_group16<PACKED>(bases_aligned, pbad, tile_base + (uint64_t)v * 16, navail + mis, codes, bads);
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
                        if (MODE == 0) { if constexpr (NBH == 0) stage[s++] = strand_rep2(f, rk, len, lmask); else stage[s++] = canonical_tail<NBH>(f, rk); }
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

// Extraction fused with the first partition level, fourth form.  What bounded the second form
// (one returning atomic per tile and bucket on 256 cursor words, bucket runs of ~128 bytes landing
// on partial 64-byte granules of HBM, two FNV chains per window) is designed out:
//   * a workgroup owns a private BLOCK of B key slots in every bucket region and appends to it;
//     a bucket cursor is touched only when a block is used up (B = 256: 16 x fewer atomics);
//   * stores reach HBM as they are issued (nothing merges two partial writes of a 64-byte granule
//     on the way: 1.4-1.5 x the bytes when runs start anywhere), so a tile stores only whole
//     granules: per bucket the keys beyond a multiple of 8 wait in the registers of the thread that
//     owns the bucket (<= 7 keys) for the next tile's keys of that bucket;
//   * MODE 0 stores the strand representative (strand_rep) instead of the canonical form.
// The fourth form (round 4) takes the arithmetic out of the three places that handled every key:
//   * in LDS every bucket's NEW keys of the tile lie in one piece, by rank, that starts on a granule: a key's place
//     is tab[digit] + rank -- one 4-byte table read and one add where the third form laid a stored and a carried
//     part of every bucket apart (an 8-byte table read, a compare and two selects per key).  Order inside a bucket
//     does not matter, so the carried keys need not go in front: the bucket's last, partial granule is TOPPED UP from
//     them when together they fill it (the rest stay in their registers), else its keys join them after the scatter;
//   * the thread that owns a bucket writes the HBM address of every whole granule of its keys into `gaddr` (`skip`
//     for a partial one), so the store loop is: 16 bytes of keys, one 4-byte address, one store -- no digit, no
//     table of block positions, no 64-bit selects (17 instructions per key before, 3 now);
//   * a bucket's next block is reserved a tile AHEAD of the one that opens it: the returning atomic on the bucket's
//     cursor (1-3 us under load) is off the critical path of phase C;
//   * windows come out of the register buffer by constant funnel shifts (the reverse complement is a bit field of
//     the complemented bases; the bases rolled into the forward strand and the strand-deciding middle bits are
//     pre-shifted once per thread), in 32-bit halves: 27 -> 16 instructions per window.
// The unused tail of every workgroup's last block is filled with kPadKey, which the next pass
// skips; pc->cursors[d] = slots handed out in bucket d (whole blocks), pc->keys_out = keys.
// The pk/iv arrays of phase A live in the memory of `sorted` (dead until the scatter).
// REC: the input is not bases but super-k-mer records (kernels_route.hpp: 12 bytes = 1..16 windows and their bases);
// bases_aligned = the records, navail = their number.  A workgroup walks its own contiguous share of the records: per
// tile it stages up to 512 of them in LDS, takes as many whole records as hold at most T windows (prefix sums of
// their window counts), and thread t extracts windows 16 t .. 16 t + 15 of the tile's window sequence -- wherever the
// record boundaries fall -- so every key register holds a valid window whatever the records' lengths are.
// windows of a record from its third word: bits 28..31 = windows - 1; a PAD (word 1 << 27: a single window's bases end
// below bit 64, so no record of windows has that bit without bits 28..31) holds none
constexpr uint32_t kSkPadWord2 = 1u << 27;
__host__ __device__ inline uint32_t rec_windows(uint32_t w2) { return (w2 >> 27) == 1u ? 0u : (w2 >> 28) + 1u; }
// keys per thread / workgroups per CU of extract1_part_kernel.  Measured per 40 M reads (third form): 16 / 3 (52 KB of LDS, 168 VGPRs)
// 18.5 ms (17.0 since the stores take two keys per lane); 12 / 3 20.0; 8 / 4 (36 KB, 128 VGPRs) 20.9; 8 / 3 23.0 -- what a tile costs beside its keys (carried keys,
// scans, barriers) weighs more than the fourth workgroup brings; and 16 / 3 with 928 bytes more LDS runs two
// workgroups per CU: 22.5; 512 threads of 8 keys (the same tile, two workgroups per CU, four waves per SIMD): 27.7.  The
// record form needs 16.
#ifndef GOSS_E1_NK
#define GOSS_E1_NK 16
#endif
#ifndef GOSS_E1_OCC
#define GOSS_E1_OCC 3
#endif
#ifndef GOSS_E1_NCH
#define GOSS_E1_NCH 4          // (NARROW) chunks of three keys per lane and round of the store loop
#endif
// Wave priorities by phase (s_setprio; the SIMD's issue arbiter takes priority before age).  The three workgroups of a
// CU are in different phases; a wave in the store phase issues few instructions, each of which starts a long-latency
// operation (LDS read, global store) that the tile's end waits for, a wave in the ranking phase issues hundreds of
// vector instructions nothing waits for yet.  Stores first, ranking (and the next tile's byte encoder) last:
// 36.9 -> 33.7 ms on C2.  Measured (bookkeeping, scatter, encoder, stores): (0,0,0,0) 36.3, (3,3,3,3) 35.5, (3,3,0,3) 34.9,
// (0,3,0,3) 34.3, (0,0,0,3) 33.9, (1,2,0,3) 33.7, (3,0,0,0) 36.7.
// (round 5, the narrow form: the encoder at 1 -- (1,2,1,3) 29.04 ms against (1,2,0,3) 29.26-29.41, (1,3,0,3) 30.23, (0,1,0,2) 29.77)
#ifndef GOSS_E1_PRIO_C
#define GOSS_E1_PRIO_C 1
#define GOSS_E1_PRIO_S 2
#define GOSS_E1_PRIO_E 1
#define GOSS_E1_PRIO_D 3
#endif
// REPK (MODE 0): which strand of a k-mer is stored -- 0: strand_rep (even k), 1: the strand whose middle base has a clear
// low bit (odd k), 2: gossamer's canonical form (the smaller FNV-1a hash: two hash chains per window; chunks with many
// distinct keys, whose re-ordering after counting would cost more)
// (v_bfi_b32 written out: the compiler turns the and-or form of a select by an all-ones / all-zeros mask into
// compares and conditional moves -- three or four instructions where this is one)
__device__ __forceinline__ uint32_t bit_select(uint32_t mask, uint32_t ones, uint32_t zeros)
{
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(ones), "v"(zeros));
    return d;
}
// FAST: the 32-bit forms of the window arithmetic and of the digit -- keys of 32 bits or more whose digit lies at bit 34
// or above (the headline's: len >= 21, digit = the key's top eight bits); the host picks the instantiation.
// NARROW (round 5): what leaves is not the 8-byte key but what the second level of the 32-bit-remainder form needs of
// it -- the 32-bit remainder (rem32_pack, kernels_partition.hpp) and the second-level digit (<= 10 bits) -- TWELVE keys
// to a 64-byte granule instead of eight: a granule is four 16-byte chunks {rem, rem, rem, D}, D = the three digits
// at bits 0, 10, 20 and, at bits 30-31, how many of the chunk's three keys are keys (pads: an all-zero chunk holds
// none); slot s = 4 m + c of a granule is field m of chunk c.  5.33 bytes per key written here and read by
// subpart32_kernel<.., NARROW> instead of 8 (profiles/r05: with two granules of three stored the first level takes
// 28.3 ms instead of 33.3, the second level is bound by the bytes it moves).  In LDS the remainders and the digits are
// staged in two arrays by slot; a bucket's piece starts on a granule and begins with the <= 11 keys it carries, which
// the bucket's thread holds in registers AS A GRANULE (48 + 24 bytes) between tiles and copies to the head of the
// next piece with six wide LDS writes -- no key is moved one by one and the new keys simply follow (rank + carried).
// Cursors, blocks and regions keep counting 8-byte slots: a granule is eight of them whatever it holds.
template <int MODE, int NH, int REPK, bool REC = false, bool FAST = false, bool NARROW = false, bool PACKED = false>
__global__ __launch_bounds__(kTB, GOSS_E1_OCC) void extract1_part_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                            uint64_t nstarts, uint64_t navail, uint32_t len,
                                                            Key1* __restrict__ out, PartCounters* __restrict__ pc,
                                                            const GapTable* __restrict__ gt, uint32_t shift, uint64_t nsuper,
                                                            uint32_t blk_log2, uint32_t nr_rbits = 0, uint32_t nr_sqbit = 0, uint32_t nr_dmask = 0, uint32_t nr_capg = 0,
                                                            const uint16_t* __restrict__ pbad = nullptr)
{
    // PACKED: the input is a packed string (load_group16, kernels_common.hpp) -- bases_aligned its codes, pbad its flags; the
    // tile's vectors are fetched as they are and nothing is encoded
    static_assert(!(PACKED && REC), "records are not bases");
    // (nr_capg: granules of the LDS layout a tile may take before its carried granules are sent off short -- the layout's
    // 664, or down to the 576 a tile without carried keys needs at most: tests make the rare path common with it)
    static_assert(!NARROW || NH == 0, "the narrow form is the two-level form's");
    // MODE 0: one key per window, 16 windows per thread.  MODE 1 (graph): forward key and
    // reverse complement of every window, 8 windows per thread -- 16 keys per thread either way.
    constexpr int S = MODE == 1 ? 2 : 1;
    constexpr int P = GOSS_E1_NK / S;
#if !defined(GOSS_E1_EXPERIMENT)
    static_assert(!REC || GOSS_E1_NK == 16, "a record of 16 windows holds at most one thread's first window");
#endif
    constexpr int T = kTB * P;                   // window starts per tile
    constexpr int NVEC = T / 16 + 4;
    constexpr int NK = P * S;                    // keys per thread
    constexpr int kCarry = NARROW ? 11 : 7;
    constexpr uint32_t kGran = NARROW ? 12 : 8;                      // keys per 64-byte granule
    // the NEW keys of the 256 buckets, each bucket's rounded up to whole granules: at most T * S keys and 7 (11) slots per bucket
    // NARROW: a bucket's piece holds its carried keys as well -- at most T * S + 2 * 256 * 11 slots, but 27 per bucket on
    // average (16 new, 5.5 carried, 5.5 of rounding) with a deviation of ~80 for the whole tile: 656 granules are what
    // three workgroups per CU leave room for and 12 deviations above the mean; a tile that needs more (keys dealt out
    // on purpose) first sends every carried granule to its bucket as it is, short, and then holds T * S + 256 * 11 at most
    constexpr uint32_t kSlots = NARROW ? 656 * 12 : T * S + 256 * kCarry;
    static_assert(kSlots % kGran == 0 && kSlots >= T * S + 256 * kCarry, "whole granules, and room for a tile without carried keys");
    // + 128 slots nobody reads: the keys of windows that are not valid go there (they rank themselves in one of 32
    // spare counters, 8 threads each: at most 128 per counter and tile) instead of under a branch, whose exec-mask
    // bookkeeping costs scalar issue slots; reads past the live part land there too
    constexpr uint32_t kGarb = kSlots;
    constexpr int kPairs = GOSS_E1_NK >= 16 ? 5 : 3;                 // pairs of keys per lane and round of the store loop
    constexpr uint32_t kStep = 2 * kPairs * kTB;                     // slots per round: two rounds cover the 4 096 + 3.5 x 256 slots an average tile takes
    // granules of the layout; entry kSlots / kGran is always kSkip.  (NARROW: the store loop reads its table entries, remainders
    // and digits at fixed strides from the lane's first chunk, past the layout's end in its last round -- up to chunk
    // 3 071 = granule 767, slot 9 220: all inside this allocation, and nothing of it is stored)
    constexpr uint32_t kNG = NARROW ? 772 : kSlots / kGran + 4;
    constexpr uint32_t kSkip = 0xFFFFFFFFu;
    // NARROW: remainders (4 bytes a slot), then digits (2 bytes a slot)
    constexpr uint32_t kSortedBytes = NARROW ? (kSlots + 128) * 6 : (kSlots + 128) * 8;
    constexpr uint32_t kDigBase = (kSlots + 128) * 4;
    static_assert(kDigBase % 16 == 0, "the digit array starts on a 16-byte boundary");
    __shared__ __attribute__((aligned(64))) unsigned char lds_all[kSortedBytes + (288 + 288 + kNG + 8 + (NH ? 256 * NH : 0)) * 4];
    Key1* const sorted = reinterpret_cast<Key1*>(lds_all);
    uint32_t* const dh = reinterpret_cast<uint32_t*>(lds_all + kSortedBytes);       // new keys of this tile per digit (rank counter); 32 spare ones for windows that are not valid
    uint32_t* const tab = dh + 288;              // per bucket: the byte offset in `sorted` of its first NEW key (stream start + keys carried in); spare ones: kGarb
    uint32_t* const gaddr = tab + 288;           // [kNG] per granule of `sorted`: slot / 8 in `out` it is stored to, or kSkip
    uint32_t* const sh_scan = gaddr + kNG;       // [kWaves + 1]
    uint32_t& sh_ovf = sh_scan[6];
    uint32_t* const lh = sh_scan + 8;            // [256 NH] histograms of the next NH digits
    uint32_t* pk = reinterpret_cast<uint32_t*>(sorted);
    uint32_t* iv = pk + NVEC;

    const uint32_t tid = threadIdx.x;
    if (NH > 0) lh[tid] = 0;
    if (NH > 1) lh[tid + 256] = 0;
    dh[tid] = 0;
    if (tid < 32) { dh[256 + tid] = 0; tab[256 + tid] = kGarb << (NARROW ? 2 : 3); }
    if (tid == 0) { sh_ovf = 0; gaddr[kSlots / kGran] = kSkip; }
    const uint64_t my_start = gt->reg_start[tid], my_cap = gt->reg_cap[tid];
    const uint32_t B = 1u << blk_log2;
    const uint32_t bits = 2 * len;
    const uint64_t kmask = (1ULL << bits) - 1;               // len <= 31
    const uint64_t lmask = (1ULL << len) - 1;
    unsigned long long nvalid = 0;
    uint64_t wpos = 0;                           // next slot of bucket tid's open block (a block boundary = none open)
    uint32_t ccnt = 0;                           // keys of bucket tid carried over from the previous tile
    Key1 kc[NARROW ? 1 : kCarry];                // ... and the keys themselves
#pragma unroll
    for (int j = 0; j < (NARROW ? 1 : kCarry); ++j) kc[j].lo = 0;
    // NARROW: the carried keys as the granule they are part of -- twelve remainders, twelve 16-bit digits
    uint4 clo0 = make_uint4(0, 0, 0, 0), clo1 = clo0, clo2 = clo0;
    uint64_t cdig0 = 0, cdig1 = 0, cdig2 = 0;
    // the carried granule as it stands to `at`, every chunk saying how many of its slots c, c + 4, c + 8 hold a key
    [[maybe_unused]] auto store_short_granule = [&](Key1* at) {
        const uint32_t lw[12] = {clo0.x, clo0.y, clo0.z, clo0.w, clo1.x, clo1.y, clo1.z, clo1.w, clo2.x, clo2.y, clo2.z, clo2.w};
        const uint64_t dw[3] = {cdig0, cdig1, cdig2};
        uint4* const g = reinterpret_cast<uint4*>(at);
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            const uint32_t nv = ccnt > (uint32_t)c + 8 ? 3u : ccnt > (uint32_t)c + 4 ? 2u : ccnt > (uint32_t)c ? 1u : 0u;
            const uint32_t da = (uint32_t)(dw[c / 4] >> (16 * (c % 4))) & 0x3FFu;
            const uint32_t db = (uint32_t)(dw[(c + 4) / 4] >> (16 * ((c + 4) % 4))) & 0x3FFu;
            const uint32_t dc = (uint32_t)(dw[(c + 8) / 4] >> (16 * ((c + 8) % 4))) & 0x3FFu;
            g[c] = make_uint4(lw[c], lw[c + 4], lw[c + 8], da | (db << 10) | (dc << 20) | (nv << 30));
        }
    };

    // 16 bytes of the input at `byte0` -> 32 bits of 2-bit codes + 16 non-base flags
    auto fetch = [&](uint64_t byte0, uint4& q) -> bool {
        if constexpr (PACKED)
        {
            // (codes and flags as they lie; positions beyond the string are flagged where the vector is "encoded")
            q = make_uint4(0u, 0xFFFFu, 0u, 0u);
            if (byte0 < navail + mis)
            {
                const uint64_t g = byte0 >> 4;
                q.x = reinterpret_cast<const uint32_t*>(bases_aligned)[g];
                q.y = pbad[g];
                const uint64_t left = navail + mis - byte0;
                q.z = left < 16 ? (0xFFFFu << (uint32_t)left) & 0xFFFFu : 0u;
            }
            return true;
        }
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
        if constexpr (PACKED) { codes = q.x; bads = q.y | q.z; return; }
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        codes = 0; bads = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            uint32_t bad;
            const uint32_t x = base_codes(w[i], bad);
            codes |= pack_codes(x) << (8 * i);
            bads |= pack_flags(bad) << (4 * i);
        }
    };
    // The bytes of a tile are fetched one tile ahead and wait, encoded, in registers: the load's
    // latency passes behind the previous tile's ranking and sorting, and the stores of a tile have
    // half a tile's time to drain before anything waits on this wave's memory counter again.
    constexpr uint32_t NV0 = T / 16;             // thread tid < NV0 encodes vector tid, threads 0..3 also vector NV0 + tid
    static_assert(NVEC == NV0 + 4 && NV0 <= kTB, "one vector per thread and four more");
    uint32_t c0 = 0, b0 = 0, c1 = 0, b1 = 0;
    // REC: this workgroup's records [rc_next, rc_end), two per thread and tile staged (three words each)
    const uint32_t* recw = reinterpret_cast<const uint32_t*>(bases_aligned);
    const uint64_t nrec = navail;                            // (REC: the number of records travels in `navail`)
    uint64_t rc_next = 0, rc_end = 0;
    uint32_t ra0 = 0, ra1 = 0, ra2 = kSkPadWord2, rb0 = 0, rb1 = 0, rb2 = kSkPadWord2;      // the records the thread stages next (a pad: none)
    // Staged records, 32 bytes each (phase A's share of `sorted`): words 0..2 the record's bases COMPLEMENTED (the reverse
    // complement of window w is the field at bit 2 w), words 3..5 the bases in reverse order, base j at bit 2 (len + 14 - j)
    // (the forward form of window w is the field at bit 30 - 2 w), word 6 the windows staged in front of the record, word 7
    // twice its windows: a record is turned ONCE, by the thread that stages it, and every window is two funnel shifts
    // of each image -- cut out of the bases and reversed window by window, phase B took 8 400 of a tile's 15 700 cycles.
    uint32_t* rbuf = pk;                                     // [512][8]
    uint32_t* rmark = pk + 512 * 8;                          // [256] the staged record that holds thread g's first window; [256] = records taken, [257] = their windows
    auto stage_rec = [&](uint32_t at, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t before) {
        const uint32_t x2m = x2 & 0x0FFFFFFFu, nw = (x2 >> 28) + 1u;
        if constexpr (!(MODE == 0 && REPK == 1))
        {
            // (the forms that read the bases only: see kTurned in phase B)
            uint4* const q = reinterpret_cast<uint4*>(rbuf + 8u * at);
            q[0] = make_uint4(~x0, ~x1, ~x2m, 0u);
            q[1] = make_uint4(0u, 0u, before, 2u * nw);
            return;
        }
        auto rev4 = [](uint32_t w) { const uint32_t r = __brev(w); return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1); };
        // the 96 bits in reverse base order: base j at bit 2 (47 - j); moved down by 2 (33 - len) bits
        uint32_t y0 = rev4(x2m), y1 = rev4(x1), y2 = rev4(x0);
        uint32_t sh = 2u * (33u - len);
        if (sh >= 32u) { y0 = y1; y1 = y2; y2 = 0u; sh -= 32u; }          // (uniform; twice for len = 1)
        if (sh >= 32u) { y0 = y1; y1 = y2; y2 = 0u; sh -= 32u; }
        const uint32_t v0 = __builtin_amdgcn_alignbit(y1, y0, sh), v1 = __builtin_amdgcn_alignbit(y2, y1, sh), v2 = y2 >> sh;
        uint4* const q = reinterpret_cast<uint4*>(rbuf + 8u * at);
        q[0] = make_uint4(~x0, ~x1, ~x2m, v0);
        q[1] = make_uint4(v1, v2, before, 2u * nw);
    };
    auto fetch_recs = [&](uint64_t base) {
        const uint64_t ia = base + 2 * (uint64_t)tid, ib = ia + 1;
        // (loads only: nothing here looks at what they return -- the window counts are taken where the records are
        // staged, a tile later; taken here they put a wait for memory behind each load)
        ra0 = ra1 = rb0 = rb1 = 0; ra2 = rb2 = kSkPadWord2;
        if (ia < rc_end) { ra0 = recw[3 * ia]; ra1 = recw[3 * ia + 1]; ra2 = recw[3 * ia + 2]; }
        if (ib < rc_end) { rb0 = recw[3 * ib]; rb1 = recw[3 * ib + 1]; rb2 = recw[3 * ib + 2]; }
    };
    if constexpr (REC)
    {
        const uint64_t per = ((nrec + gridDim.x - 1) / gridDim.x + 1) & ~1ULL;
        rc_next = (uint64_t)blockIdx.x * per;
        rc_end = rc_next + per < nrec ? rc_next + per : nrec;
        if (rc_next < rc_end) fetch_recs(rc_next);
    }
    else if (blockIdx.x < nsuper)
    {
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        const uint64_t tb = (uint64_t)blockIdx.x * T;
        if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
        if (tid < 4) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        encode(q0, c0, b0);
        if (tid < 4) encode(q1, c1, b1);
    }

    // A bucket's next block is reserved AHEAD of the tile that opens it (phase A, when the open block has less than
    // kAhead slots left -- more than a tile ever asked for in 10^4 tiles): a returning atomic on a cursor takes 1-3 us
    // under load, a third of a tile, and in phase C every wave has a lane that waits for one.
    // (The read of the reservation in phase C carries an `s_waitcnt vmcnt(0)`, which also waits for the previous tile's
    // stores.  A form that hides the atomic from the compiler -- inline assembly, issued a tile earlier still, covered by
    // the wait for the next tile's bytes -- took that wait out of phase C (wave 0's bookkeeping 2 676 -> 1 597 cycles per
    // tile, GOSS_STAMPS build) and left the kernel's time where it was, 33.6 ms: the waiting moved into the store
    // phase.  Not kept.)
    constexpr uint32_t kAhead = 40;
    unsigned long long resv = 0;
    bool has_resv = false;
#if defined(GOSS_E1_STAGGER)
    // (timing experiment: the workgroups that share a CU start a third of a tile apart)
    for (uint32_t z = 0; z < (blockIdx.x >> 8) * GOSS_E1_STAGGER; ++z) __builtin_amdgcn_s_sleep(100);
#endif
    // (the bucket's region, loaded at the top, is looked at here once: the compiler's wait for it would otherwise sit at
    // its first use inside the loop -- `s_waitcnt vmcnt(0)` in phase C of every tile, i.e. a wait for the previous
    // tile's stores)
    asm volatile("" ::"v"(my_start), "v"(my_cap));
#if defined(GOSS_STAMPS)
    // (timing build: cycles of wave 0 per phase, summed over its tiles, into the unused histogram words)
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define GOSS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define GOSS_STAMP(i)
#endif
    for (uint64_t st = blockIdx.x;; st += gridDim.x)
    {
        if constexpr (REC) { if (rc_next >= rc_end) break; }
        else { if (st >= nsuper) break; }
        const uint64_t tile_base = st * (uint64_t)T;
#if defined(GOSS_STAMPS)
        st_acc[5] += 1;
#endif

        // ---- phase A: this tile's codes from registers to LDS, the next tile's bytes on their way ----
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        bool more;
        uint32_t wt = 0;                          // (REC) windows of this tile
        if constexpr (REC)
        {
            // The records that hold windows are staged back to back (pads -- the unused ends of the routing kernel's
            // blocks -- and what lies beyond the share's end hold none and take no place), with the running sum of
            // their windows: one scan carries both sums (windows <= 8 192 in the low half, staged records above).
            if (tid == 0) { rmark[kTB] = 512; rmark[kTB + 1] = 0xFFFFFFFFu; }     // (records that fit: all, unless a thread finds the one that does not)
            const uint32_t rna = rec_windows(ra2), rnb = rec_windows(rb2);
            const uint32_t pa = rna ? 1u : 0u, pb = rnb ? 1u : 0u;
            uint32_t tot2;
            const uint32_t sc = block_excl_scan_u32((rna + rnb) | ((pa + pb) << 16), sh_scan, &tot2);
            const uint32_t ex = sc & 0xFFFFu, ca = sc >> 16, cb = ca + pa;
            const uint32_t ea = ex + rna, eb = ea + rnb;          // ends of the thread's two records in the window sequence
            if (pa) stage_rec(ca, ra0, ra1, ra2, ex);
            if (pb) stage_rec(cb, rb0, rb1, rb2, ea);
            // the first record that does not fit ends the tile: records taken from the share, and their windows
            if (ex <= (uint32_t)T && eb > (uint32_t)T)
            {
                const uint32_t one = ea <= (uint32_t)T ? 1u : 0u;
                rmark[kTB] = 2 * tid + one;
                rmark[kTB + 1] = one ? ea : ex;
            }
            // thread g starts at window P g: a record of at most 16 windows holds at most 16 / P such windows (one for
            // k-mer sets, two in graph mode where a thread takes 8 windows) and tells every one of those threads
            {
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)(16 / P); ++j)
                {
                    const uint32_t ga = (ex + P - 1) / P + j, gb = (ea + P - 1) / P + j;
                    if (ga * P < ea && ga < (uint32_t)kTB) rmark[ga] = ca;
                    if (gb * P < eb && gb < (uint32_t)kTB) rmark[gb] = cb;
                }
            }
            __syncthreads();
            const uint32_t nfit = rmark[kTB];
            wt = rmark[kTB + 1] == 0xFFFFFFFFu ? (tot2 & 0xFFFFu) : rmark[kTB + 1];
            rc_next += nfit;
            more = rc_next < rc_end;
            if (more) fetch_recs(rc_next);
        }
        else
        {
            if (tid < NV0) { pk[tid] = c0; iv[tid] = b0; }
            if (tid < 4) { pk[NV0 + tid] = c1; iv[NV0 + tid] = b1; }
            __syncthreads();
            more = st + gridDim.x < nsuper;
            if (more)
            {
                const uint64_t tb = (st + gridDim.x) * (uint64_t)T;
                if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
                if (tid < 4) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
            }
        }

        // ---- phase B: windows out of registers, keys, rank inside their digit --------------------
        // Written without branches around the LDS operations: a window that is not valid still gets a
        // (meaningless) key and ranks itself in a spare counter, so that the sixteen returning atomics
        // of a thread are issued back to back and waited for once, not one round trip after the other.
#if !defined(GOSS_E1_SYNC_BLOCKS)
        if (!has_resv && ((B - ((uint32_t)wpos & (B - 1))) & (B - 1)) < kAhead)
        {
            resv = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)B);
            has_resv = true;
        }
#endif
        GOSS_STAMP(0);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(0);
#endif
        Key1 kreg[NK];
        uint32_t rk[NK];
        uint32_t bin4[NK];                        // byte offset of the key's counter in dh (and of its entry in tab)
        uint32_t vm;
        {
            uint32_t m;
            uint64_t blo = 0, bhi = 0;
            // (REC) the staged record the thread is at: its two images, twice the window inside it, twice its windows
            [[maybe_unused]] uint32_t r_at = 0, r_s2 = 0, r_nw2 = 0, rc0 = 0, rc1 = 0, rc2 = 0, rv0 = 0, rv1 = 0, rv2 = 0;
            [[maybe_unused]] uint32_t r_before = 0;
            // (kTurned: the forms that cut their windows out of both images.  The others -- even lengths, the canonical form per
            // window, graphs -- are within a few registers of what three workgroups per CU allow and spill 12 to 44 bytes
            // with the six words of the images live in the loop: they take the bases back out of the complemented words)
            constexpr bool kTurned = REC && MODE == 0 && REPK == 1;
            auto load_rec = [&]() {
                const uint4* const q = reinterpret_cast<const uint4*>(rbuf + 8u * r_at);
                const uint4 a = q[0], b = q[1];
                r_before = b.z; r_nw2 = b.w;
                if constexpr (kTurned) { rc0 = a.x; rc1 = a.y; rc2 = a.z; rv0 = a.w; rv1 = b.x; rv2 = b.y; }
                else { blo = ~((uint64_t)a.x | ((uint64_t)a.y << 32)); bhi = ~a.z & 0x0FFFFFFFu; }
            };
            // (a look-ahead -- the record behind the current one kept in registers, its LDS reads issued a record early --
            // changed nothing: 85.4 against 84.6 ms)
            auto next_rec = [&]() { ++r_at; r_s2 = 0; load_rec(); };
            if constexpr (REC)
            {
                // windows P tid .. P tid + P - 1 of the tile's sequence, starting in record rmark[tid]
                const uint32_t j0 = tid * P;
                const uint32_t left = wt > j0 ? wt - j0 : 0;
                m = left >= (uint32_t)P ? (uint32_t)((1ULL << P) - 1ULL) : ((1u << left) - 1u);
                r_at = left ? rmark[tid] : 0u;                     // (marked by the record itself in phase A)
                load_rec();
                r_s2 = 2u * (j0 - r_before);
            }
            else
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
            const uint64_t lo = w0 | (w1 << 32), hi = w2 | (w3 << 32);
            const uint32_t s2 = 2 * sh;
            blo = s2 ? ((lo >> s2) | (hi << (64 - s2))) : lo;
            bhi = hi >> s2;
            }
            vm = m;
            nvalid += __popc(m);
            // the key(s) of window i from its forward form f and its reverse complement r
            auto emit = [&](int i, uint64_t f, uint64_t r) {
                const Key1 fk{f}, rck{r};
                if (MODE == 0)
                    // odd length: the central base decides (its low bit differs between the strands)
                    kreg[i] = REPK == 2 ? canonical(fk, rck) : REPK == 1 ? (((f >> (len - 1)) & 1ULL) ? rck : fk) : strand_rep(fk, rck, len, lmask);
                else { kreg[i * 2] = fk; kreg[i * 2 + 1] = rck; }
            };
            // the counter of key j: its digit, or a spare one (window not valid); ranked at once
            const uint32_t badm = ~m;
            const uint32_t spare4 = (256u + (tid & 31u)) << 2;
            auto rank_key = [&](int j) {
                const uint32_t nok = (uint32_t)__builtin_amdgcn_sbfe((int32_t)badm, j / S, 1);
                const uint32_t d4 = FAST ? ((uint32_t)(kreg[j].lo >> 32) >> (shift - 34)) & 0x3FCu
                                         : ((uint32_t)(kreg[j].lo >> shift) & 0xFFu) << 2;
                bin4[j] = bit_select(nok, spare4, d4);
                rk[j] = atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(dh) + bin4[j]), 1u);
            };
            // The bases lie in (bhi:blo) lowest first, two bits each: the reverse complement of window i is the field
            // [2 i, 2 i + bits) of the complemented buffer, its forward form the base-4 reverse of the same field.
            if constexpr (REC)
            {
                // (the record's images come from LDS, turned when the record was staged.  Turning it here, when it becomes the
                // current one, needs six more live registers in this loop: 33 spilled to scratch, 137.8 against 85 ms)
                const uint32_t kmlo = (uint32_t)kmask, kmhi = (uint32_t)(kmask >> 32);
                if constexpr (!kTurned)
                {
#pragma unroll
                    for (int i = 0; i < P; ++i)
                    {
                        // the window at bit r_s2 of record r_at, cut out of the record's bases and reversed
                        const uint64_t x = (r_s2 ? ((blo >> r_s2) | (bhi << (64 - r_s2))) : blo) & kmask;
                        emit(i, rev64(x) >> (64 - bits), (~x) & kmask);
                        if (i + 1 < P)
                        {
                            r_s2 += 2u;
                            if (r_s2 >= r_nw2 && r_at < 511) next_rec();
                        }
                    }
                }
                else
#pragma unroll
                for (int i = 0; i < P; ++i)
                {
                    // the window at bit r_s2 of record r_at: two funnel shifts of each image (shifts of 0 .. 30 bits); then
                    // on to the next window, which may be the first of the next record
                    const uint32_t t2 = 30u - r_s2;
                    const uint32_t rlo = __builtin_amdgcn_alignbit(rc1, rc0, r_s2) & kmlo, rhi = __builtin_amdgcn_alignbit(rc2, rc1, r_s2) & kmhi;
                    const uint32_t flo = __builtin_amdgcn_alignbit(rv1, rv0, t2) & kmlo, fhi = __builtin_amdgcn_alignbit(rv2, rv1, t2) & kmhi;
                    if (MODE == 0 && REPK == 1)
                    {
                        // (odd length: the low bit of the central base, bit len - 1 <= 30 of the forward form, picks the strand)
                        const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int32_t)flo, len - 1u, 1u);        // all ones: the reverse complement
                        kreg[i].lo = ((uint64_t)bit_select(sel, rhi, fhi) << 32) | bit_select(sel, rlo, flo);
                    }
                    else emit(i, ((uint64_t)fhi << 32) | flo, ((uint64_t)rhi << 32) | rlo);
                    if (i + 1 < P)
                    {
                        r_s2 += 2u;
                        if (r_s2 >= r_nw2 && r_at < 511) next_rec();
                    }
                }
            }
            else if constexpr (FAST)
            {
                // 32-bit halves, constant shifts: the forward form rolled (the bases it takes in, pre-shifted once:
                // base i + len - 1 at bits 2 i of nx), the reverse complement by funnel shifts of the complemented words
                const uint32_t kmhi = (uint32_t)(kmask >> 32);
                const uint32_t cw0 = ~(uint32_t)blo, cw1 = ~(uint32_t)(blo >> 32), cw2 = ~(uint32_t)bhi;
                const uint32_t pn = bits - 2;                                  // 30 .. 60
                const uint32_t nx = (uint32_t)((blo >> pn) | (bhi << (64 - pn)));
                const uint32_t mid = (uint32_t)(blo >> (len - 1));              // bit 2 i: the low bit of window i's central base (odd len)
                const uint64_t f0 = rev64(blo & kmask) >> (64 - bits);
                uint32_t flo = (uint32_t)f0, fhi = (uint32_t)(f0 >> 32);
#pragma unroll
                for (int i = 0; i < P; ++i)
                {
                    if (i)
                    {
                        const uint32_t nb = (nx >> (2 * i)) & 3u;
                        fhi = __builtin_amdgcn_alignbit(fhi, flo, 30) & kmhi;
                        flo = (flo << 2) | nb;
                    }
                    const uint32_t rlo = i ? __builtin_amdgcn_alignbit(cw1, cw0, 2 * i) : cw0;
                    const uint32_t rhi = (i ? __builtin_amdgcn_alignbit(cw2, cw1, 2 * i) : cw1) & kmhi;
                    if (MODE == 0 && REPK == 1)
                    {
                        const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int32_t)mid, 2 * i, 1);        // all ones: the reverse complement
                        kreg[i].lo = ((uint64_t)bit_select(sel, rhi, fhi) << 32) | bit_select(sel, rlo, flo);
                    }
                    else emit(i, ((uint64_t)fhi << 32) | flo, ((uint64_t)rhi << 32) | rlo);
                    // (ranked at once: the LDS atomics of the first windows run beside the arithmetic of the later ones)
#pragma unroll
                    for (int u = 0; u < S; ++u) rank_key(i * S + u);
                }
            }
            else
            {
                // forward key f and reverse complement r of window 0, then one base rolled in per window
                uint64_t f = rev64(blo & kmask) >> (64 - bits);
                uint64_t r = (~blo) & kmask;
                const uint32_t top = bits - 2;
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
                    emit(i, f, r);
                }
            }
            if constexpr (REC || !FAST)
            {
#pragma unroll
                for (int j = 0; j < NK; ++j) rank_key(j);
            }
        }
        __syncthreads();
        GOSS_STAMP(1);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_C);
#endif

        // ---- phase C: bookkeeping of bucket tid: its new keys' place in LDS, where its granules go, what is carried ----
        // A bucket's new keys lie in one piece that starts on a granule, by rank.  Order inside a bucket does not
        // matter, so the carried keys do not go in front: the bucket's last, partial granule (b new keys) is TOPPED UP
        // with carried keys when there are enough of them (b + carried >= 8: the rest stay in the registers where
        // they are); otherwise it is not stored and its b keys join the carried ones after the scatter.
        uint32_t total_slots = 0, part_at = 0, absorb = 0;
        if constexpr (NARROW)
        {
            // carried + new keys of the bucket: whole granules leave, the rest is the granule carried on
            const uint32_t cnt = dh[tid];
            uint32_t tot = ccnt + cnt;
            uint32_t fg = (tot * 0xAAABu) >> 19;                       // tot / 12 (tot < 2^13)
            uint32_t rest = tot - 12u * fg;
            uint32_t g_at = block_excl_scan_open_u32(fg + (rest ? 1u : 0u), sh_scan, &total_slots);      // (granules; the barrier behind phase C closes it)
            GOSS_STAMP(6);
            dh[tid] = 0;                               // ready for the next tile (its ranking starts behind two barriers)
            if (tid < 32) dh[256 + tid] = 0;
            // room for `fl` 8-byte slots of bucket tid: the rest of the open block (thr of them, from granule tbx on), then
            // new block(s) from granule tby on
            uint32_t thr, tbx, tby;
            auto take_room = [&](uint32_t fl) {
                const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
                thr = fl;
                tbx = (uint32_t)(wpos >> 3);
                tby = 0;
                if (fl > room)
                {
                    thr = room;
                    const uint32_t need = fl - room;
                    unsigned long long at, want;
                    if (has_resv && need <= B) { at = resv; want = B; has_resv = false; }          // (reserved in phase A)
                    else
                    {
                        want = ((uint64_t)(need + B - 1) >> blk_log2) << blk_log2;
                        at = atomicAdd(&pc->cursors[tid * kCursorStride], want);
                    }
                    // a region that is too small: nothing of this tile is stored, the host redoes the chunk
                    if (at + want > my_cap) { atomicOr(&pc->overflow, 1ULL); sh_ovf = 1; }
                    tby = (uint32_t)((my_start + at) >> 3);
                    wpos = my_start + at + need;
                }
                else wpos += fl;
            };
            if (total_slots > nr_capg)
            {
                // (more granules than the layout holds -- never with keys that spread: see kSlots.  Every carried granule
                // goes to its bucket short, its chunks saying how many keys they hold, and the tile is laid out without)
                if (ccnt)
                {
                    take_room(8u);
                    if (sh_ovf == 0) store_short_granule(out + ((uint64_t)(thr ? tbx : tby) << 3));
                    ccnt = 0;
                }
                __syncthreads();                       // (everybody has read the first scan's sums)
                tot = cnt;
                fg = (tot * 0xAAABu) >> 19;
                rest = tot - 12u * fg;
                g_at = block_excl_scan_open_u32(fg + (rest ? 1u : 0u), sh_scan, &total_slots);
            }
            const uint32_t fl = fg << 3;               // 8-byte slots stored now
            part_at = 12u * (g_at + fg);               // the granule carried on, if any
            take_room(fl);
            tab[tid] = (12u * g_at + ccnt) << 2;       // byte offset of the first NEW key's remainder
            {
                const uint32_t t8 = thr >> 3;
                const uint32_t yb = tby - t8;
#pragma unroll
                for (uint32_t g = 0; g < 3; ++g)
                    if (g < fg) gaddr[g_at + g] = (g < t8 ? tbx : yb) + g;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (uint32_t g = 3; g < fg; ++g) gaddr[g_at + g] = (g < t8 ? tbx : yb) + g;
                if (rest) gaddr[g_at + fg] = kSkip;
            }
            // the carried granule to the head of the piece (phase A's arrays in `sorted` are dead: every thread is past
            // phase B); whole, the new keys follow behind the `ccnt` that count.  (Not without carried keys: a bucket that
            // holds nothing has no piece, and the granule at g_at is the next bucket's.)
            if (ccnt)
            {
                unsigned char* const lp = lds_all + 48u * g_at;
                unsigned char* const dp = lds_all + kDigBase + 24u * g_at;
                reinterpret_cast<uint4*>(lp)[0] = clo0; reinterpret_cast<uint4*>(lp)[1] = clo1; reinterpret_cast<uint4*>(lp)[2] = clo2;
                reinterpret_cast<uint64_t*>(dp)[0] = cdig0; reinterpret_cast<uint64_t*>(dp)[1] = cdig1; reinterpret_cast<uint64_t*>(dp)[2] = cdig2;
            }
            ccnt = rest;
        }
        else
        {
            const uint32_t cnt = dh[tid];
            const uint32_t a8 = cnt & ~7u, b = cnt & 7u;
            const uint32_t s_at = block_excl_scan_open_u32((cnt + 7u) & ~7u, sh_scan, &total_slots);               // (the barrier behind phase C closes it)
            GOSS_STAMP(6);
            dh[tid] = 0;                               // ready for the next tile (its ranking starts behind two barriers)
            if (tid < 32) dh[256 + tid] = 0;
            const bool top = b + ccnt >= 8u;
            const uint32_t take = top ? 8u - b : 0u;
            const uint32_t fl = a8 + (top ? 8u : 0u);  // keys stored now
            absorb = top ? 0u : b;
            part_at = s_at + a8;
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            uint32_t thr = fl;
            const uint32_t tbx = (uint32_t)(wpos >> 3);
            uint32_t tby = 0;
            if (fl > room)
            {
                thr = room;
                const uint32_t need = fl - room;
                unsigned long long at, want;
                if (has_resv && need <= B) { at = resv; want = B; has_resv = false; }          // (reserved in phase A)
                else
                {
                    want = ((uint64_t)(need + B - 1) >> blk_log2) << blk_log2;
                    at = atomicAdd(&pc->cursors[tid * kCursorStride], want);
                }
                // a region that is too small: nothing of this tile is stored, the host redoes the chunk
                if (at + want > my_cap) { atomicOr(&pc->overflow, 1ULL); sh_ovf = 1; }
                tby = (uint32_t)((my_start + at) >> 3);
                wpos = my_start + at + need;
            }
            else wpos += fl;
            tab[tid] = s_at << 3;
            // where the bucket's granules go: the rest of the open block, then the new block(s); a partial one stays
            {
                const uint32_t g0 = s_at >> 3, ng = fl >> 3, t8 = thr >> 3;
                const uint32_t yb = tby - t8;
                // (four whole granules or more are rare: 16 new keys per bucket and tile on average)
#pragma unroll
                for (uint32_t g = 0; g < 4; ++g)
                    if (g < ng) gaddr[g0 + g] = (g < t8 ? tbx : yb) + g;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (uint32_t g = 4; g < ng; ++g) gaddr[g0 + g] = (g < t8 ? tbx : yb) + g;
                if (absorb) gaddr[g0 + ng] = kSkip;
            }
            // topping up: the last `take` carried keys behind the b new ones (phase A's arrays in `sorted` are dead:
            // every thread is past phase B); written without a branch -- no bucket tops up: take = 0, all seven
            // keys go to slots nobody reads (0.7 % faster than the writes under `if (top)`)
            {
                const uint32_t from = ccnt - take;
#pragma unroll
                for (int j = 0; j < kCarry; ++j)
                    sorted[((uint32_t)j >= from && (uint32_t)j < ccnt ? part_at + b - from : kGarb + (tid & 63u)) + j] = kc[j];
                ccnt = from;
            }
        }
        GOSS_STAMP(7);
        __syncthreads();
        GOSS_STAMP(2);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_S);
#endif
        // new keys to their place: the stream's first new slot + rank (the table reads of
        // all sixteen keys first, then the writes: no round trip per key)
        {
            uint32_t tb[NK];
#pragma unroll
            for (int i = 0; i < NK; ++i) tb[i] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tab) + bin4[i]);
            if constexpr (NARROW)
            {
                // remainder and digit of every key to their slot: the low nr_rbits bits with the always-clear bit nr_sqbit
                // taken out (if any), the nr_dmask bits above them.  Four forms of the loop, picked by the (uniform)
                // parameters: the 64-bit shifts and selects of one form for all cost 9 instructions per key, these 4-6
                auto put = [&](auto sq_tag, auto hi_tag) {
                    constexpr bool kSq = decltype(sq_tag)::value, kHi = decltype(hi_tag)::value;
                    const uint32_t lomask = kSq ? (1u << nr_sqbit) - 1u : (nr_rbits >= 32 ? 0xFFFFFFFFu : (1u << nr_rbits) - 1u);
                    const uint32_t dbits = (uint32_t)__popc(nr_dmask);
#pragma unroll
                    for (int i = 0; i < NK; ++i)
                    {
                        const uint32_t klo = (uint32_t)kreg[i].lo, khi = (uint32_t)(kreg[i].lo >> 32);
                        uint32_t rem = klo & lomask;
                        // (the bits above the squeezed one, moved down by one; what leaves the word at the top is the digit's)
                        if (kSq) rem |= __builtin_amdgcn_alignbit(khi, klo, nr_sqbit + 1u) << nr_sqbit;
                        const uint32_t dg = kHi ? __builtin_amdgcn_ubfe(khi, nr_rbits - 32u, dbits)
                                                : (__builtin_amdgcn_alignbit(khi, klo, nr_rbits) & nr_dmask);
                        const uint32_t la = tb[i] + (rk[i] << 2);
                        *reinterpret_cast<uint32_t*>(lds_all + la) = rem;
                        *reinterpret_cast<uint16_t*>(lds_all + kDigBase + (la >> 1)) = (uint16_t)dg;
                    }
                };
                if (nr_sqbit) { if (nr_rbits >= 32) put(std::true_type{}, std::true_type{}); else put(std::true_type{}, std::false_type{}); }
                else { if (nr_rbits >= 32) put(std::false_type{}, std::true_type{}); else put(std::false_type{}, std::false_type{}); }
            }
            else
#pragma unroll
            for (int i = 0; i < NK; ++i)
            {
                *reinterpret_cast<Key1*>(reinterpret_cast<char*>(sorted) + (tb[i] + (rk[i] << 3))) = kreg[i];
                if (NH > 0)
                {
                    const bool ok = (vm >> (i / S)) & 1u;
                    const Key1 k = kreg[i];
                    atomicAdd(&lh[(uint32_t)(k.lo >> (shift + 8)) & 0xFFu], ok ? 1u : 0u);
                    if (NH > 1) atomicAdd(&lh[256u + ((uint32_t)(key_shr64(k, shift + 16)) & 0xFFu)], ok ? 1u : 0u);
                }
            }
        }
        GOSS_STAMP(10);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_E);
#endif
        if (more && !REC)
        {
            encode(q0, c0, b0);
            if (tid < 4) encode(q1, c1, b1);
        }
        GOSS_STAMP(11);
        // (REC: the next records must have arrived before this tile's stores are issued -- loads and stores share one
        // in-order counter, and a wait at their first use in the next tile would wait for those stores as well.  They are
        // LOOKED AT here, so that the wait is the compiler's own: behind a wait it does not see -- inline assembly -- it
        // still put `s_waitcnt vmcnt(0)` in front of the next tile's staging, i.e. waited for the stores all the same:
        // phase A of the record form took 7 700 of a tile's 22 400 cycles)
        if constexpr (REC) asm volatile("" ::"v"(ra0), "v"(ra1), "v"(ra2), "v"(rb0), "v"(rb1), "v"(rb2));
        __syncthreads();
        GOSS_STAMP(3);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_D);
#endif

        // ---- phase D: whole granules to the bucket blocks; every 4 aligned lanes store one (two keys each) -------
        // (two keys per lane and store: the store path takes 16 bytes per lane as quickly as 8 --
        // experiments/storegran: 4.5 against 4.1 TB/s for this pattern alone -- and the loop has half the instructions)
        // (the reads of a partial granule that joins the carried keys go ahead of the stores)
        if constexpr (NARROW)
        {
            // the granule that is carried on (whatever lies there when there is none): read ahead of the stores
            {
                const unsigned char* const lp = lds_all + 4u * part_at;
                const unsigned char* const dp = lds_all + kDigBase + 2u * part_at;
                clo0 = reinterpret_cast<const uint4*>(lp)[0]; clo1 = reinterpret_cast<const uint4*>(lp)[1]; clo2 = reinterpret_cast<const uint4*>(lp)[2];
                cdig0 = reinterpret_cast<const uint64_t*>(dp)[0]; cdig1 = reinterpret_cast<const uint64_t*>(dp)[1]; cdig2 = reinterpret_cast<const uint64_t*>(dp)[2];
            }
            if (sh_ovf == 0)
            {
                // a lane stores chunk c = tid & 3 of a granule: slots c, c + 4, c + 8 -- three remainders and their digits.
                // Chunk i = tid + 256 u of a round lies 3 072 u bytes (remainders) / 1 536 u bytes (digits) / 256 u bytes
                // (granule table) behind chunk tid: every LDS read of a round is one address register + a constant
                constexpr int kCh = GOSS_E1_NCH;                             // chunks per lane and round
                static_assert(kSlots / 12u * 4u <= 2624u + 64u && kCh == 4, "the reads past the layout stay inside the allocation (kNG)");
                const uint32_t nchunk = total_slots << 2;                    // (total_slots: granules of the layout)
                const uint32_t c = tid & 3u;
                Key1* const lane_out = out + 2u * c;
                const unsigned char* lp = lds_all + (12u * tid - 8u * c);                  // remainder of slot c of granule tid / 4
                const unsigned char* dp = lds_all + kDigBase + (6u * tid - 4u * c);
                const unsigned char* gp = reinterpret_cast<const unsigned char*>(gaddr) + (tid & ~3u);
                for (uint32_t base = 0; base < nchunk; base += kCh * kTB, lp += 12u * kCh * kTB, dp += 6u * kCh * kTB, gp += kCh * kTB)
                {
                    uint32_t r0[kCh], r1[kCh], r2[kCh], d0[kCh], d1[kCh], d2[kCh], ga[kCh];
#pragma unroll
                    for (int u = 0; u < kCh; ++u)
                    {
                        r0[u] = *reinterpret_cast<const uint32_t*>(lp + 3072 * u);
                        r1[u] = *reinterpret_cast<const uint32_t*>(lp + 3072 * u + 16);
                        r2[u] = *reinterpret_cast<const uint32_t*>(lp + 3072 * u + 32);
                        d0[u] = *reinterpret_cast<const uint16_t*>(dp + 1536 * u);
                        d1[u] = *reinterpret_cast<const uint16_t*>(dp + 1536 * u + 8);
                        d2[u] = *reinterpret_cast<const uint16_t*>(dp + 1536 * u + 16);
                        ga[u] = *reinterpret_cast<const uint32_t*>(gp + 256 * u);
                    }
#pragma unroll
                    for (int u = 0; u < kCh; ++u) asm volatile("" : "+v"(r0[u]), "+v"(r1[u]), "+v"(r2[u]), "+v"(d0[u]), "+v"(d1[u]), "+v"(d2[u]), "+v"(ga[u]));
                    const uint32_t left = nchunk - base;                     // chunks of this round and behind it
#pragma unroll
                    for (int u = 0; u < kCh; ++u)
                    {
                        const uint4 ch = make_uint4(r0[u], r1[u], r2[u], d0[u] | (d1[u] << 10) | (d2[u] << 20) | (3u << 30));
                        if (ga[u] != kSkip && tid + u * kTB < left) *reinterpret_cast<uint4*>(lane_out + ((uint64_t)ga[u] << 3)) = ch;
                    }
                }
            }
            GOSS_STAMP(8);
            GOSS_STAMP(9);
        }
        else
        {
        Key1 ab[kCarry];
#pragma unroll
        for (int j = 0; j < kCarry; ++j)
        {
            const bool in = (uint32_t)j >= ccnt && (uint32_t)j < ccnt + absorb;
            ab[j] = sorted[in ? part_at + j - ccnt : kGarb];
        }
        if (sh_ovf == 0)
        {
            Key1* const lane_out = out + ((2 * tid) & 7u);
            for (uint32_t base = 0; base < total_slots; base += kStep)
            {
                // five pairs at a time: their LDS reads, then their stores
                uint4 kk[kPairs];
                uint32_t ga[kPairs];
#pragma unroll
                for (int u = 0; u < kPairs; ++u)
                {
                    const uint32_t i = min(base + 2 * (tid + u * kTB), kSlots);        // (beyond the layout: the entry that is always kSkip)
                    kk[u] = *reinterpret_cast<const uint4*>(&sorted[i]);
                    ga[u] = gaddr[i >> 3];
                }
                // (all ten reads are issued here, not each under the test of its store)
#pragma unroll
                for (int u = 0; u < kPairs; ++u) asm volatile("" : "+v"(kk[u].x), "+v"(kk[u].y), "+v"(kk[u].z), "+v"(kk[u].w), "+v"(ga[u]));
#pragma unroll
                for (int u = 0; u < kPairs; ++u)
                {
#if defined(GOSS_E1_EXP) && GOSS_E1_EXP == 1
                    // (timing experiment: everything but the stores themselves)
                    asm volatile("" ::"v"(kk[u].x), "v"(kk[u].y), "v"(kk[u].z), "v"(kk[u].w), "v"(ga[u]));
#elif defined(GOSS_E1_EXP) && GOSS_E1_EXP == 2
                    // (timing experiment: two granules of three stored -- what a tile would write with 5.33-byte keys; results wrong)
                    if (ga[u] != kSkip && (ga[u] % 3u) != 0u && base + 2 * (tid + u * kTB) < total_slots) *reinterpret_cast<uint4*>(lane_out + ((uint64_t)ga[u] << 3)) = kk[u];
#else
                    // (entries between this tile's layout and kSlots / 8 are stale)
                    if (ga[u] != kSkip && base + 2 * (tid + u * kTB) < total_slots) *reinterpret_cast<uint4*>(lane_out + ((uint64_t)ga[u] << 3)) = kk[u];
#endif
                }
            }
        }
        GOSS_STAMP(8);
        // a partial granule that was not topped up: its keys join the carried ones
        {
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
            {
                const bool in = (uint32_t)j >= ccnt && (uint32_t)j < ccnt + absorb;
                kc[j] = in ? ab[j] : kc[j];
            }
            ccnt += absorb;
        }
        GOSS_STAMP(9);
        }
        __syncthreads();
        GOSS_STAMP(4);
    }
#if defined(GOSS_STAMPS)
    if (tid == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&pc->hist[500 + i], st_acc[i]);
#endif
#undef GOSS_STAMP

    // ---- the end: carried keys, the unused tail of every open block, the block reserved and not opened ----------
    if (sh_ovf == 0)
    {
        bool spare_blk = has_resv;
        if (ccnt)
        {
            // one more granule: the carried keys, padding behind them
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            if (room == 0)
            {
                unsigned long long at;
                if (has_resv) { at = resv; spare_blk = false; }
                else at = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)B);
                if (at + B > my_cap) { atomicOr(&pc->overflow, 1ULL); ccnt = 0; wpos = 0; }
                else wpos = my_start + at;
            }
            if constexpr (NARROW)
            {
                if (ccnt)          // (0: the region overflowed)
                {
                    store_short_granule(out + wpos);
                    wpos += 8;
                }
            }
            else
            {
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
                if ((uint32_t)j < ccnt) out[wpos + j] = kc[j];
            wpos += ccnt;
            }
        }
        const uint32_t tail = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
        // a reserved block nobody opened counts as handed out: pads, or (beyond the region) the chunk is redone
        uint32_t blk_n = 0;
        if (spare_blk)
        {
            if (resv + B > my_cap) atomicOr(&pc->overflow, 1ULL);
            else blk_n = B;
        }
        // pad [wpos, end of block) and the spare block of every bucket: all threads share the work through LDS
        __syncthreads();
        tab[tid] = tail;
        dh[tid] = blk_n;
        reinterpret_cast<uint64_t*>(sorted)[tid] = wpos;
        reinterpret_cast<uint64_t*>(sorted)[256 + tid] = my_start + resv;
        __syncthreads();
        for (uint32_t d = 0; d < 256; ++d)
        {
            const uint32_t n = tab[d], n2 = dh[d];
            const uint64_t from = reinterpret_cast<const uint64_t*>(sorted)[d], from2 = reinterpret_cast<const uint64_t*>(sorted)[256 + d];
            // (NARROW: a chunk of zeros holds no key)
            for (uint32_t j = tid; j < n; j += kTB) out[from + j] = Key1{NARROW ? 0ULL : kPadKey};
            for (uint32_t j = tid; j < n2; j += kTB) out[from2 + j] = Key1{NARROW ? 0ULL : kPadKey};
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

// Raw keys handed over by a caller that kmerises itself (goss_gpu_push_keys_*: the templated
// GossCmdBuildKmerSet::operator()(cxt, KmerSrc&), GossCmdBuildKmerSet.tcc:246-249): mode 0 normalises every key
// (position_type::normalize, RankSelect.hh:126-140), mode 1 takes the keys as they are (the graph builder inserts
// what its adapter yields).  A key with bits at or above 2*len is not a len-mer: flagged, the push is refused.
template <class K, int MODE>
__global__ __launch_bounds__(kTB) void normalize_keys_kernel(const K* __restrict__ in, K* __restrict__ out, uint64_t n, uint32_t len,
                                                              uint32_t* __restrict__ flags)
{
    const uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    const K x = in[i];
    bool wide;
    if constexpr (sizeof(K) == 8) wide = (x.lo >> (2 * len)) != 0;
    else wide = 2 * len < 128 && (x.hi >> (2 * len - 64)) != 0;
    if (wide) atomicOr(flags + 2, 1u);
    if constexpr (MODE == 0) out[i] = canonical(x, revcomp(x, len));
    else out[i] = x;
}

// The same fusion for two-word keys (32 <= len <= 63): windows out of a 192-bit register buffer
// (extract2_kernel), NKEYS keys per thread, a tile of 256*NKEYS keys partitioned on the digit at `shift`,
// in the form of extract1_part_kernel: private blocks of B slots per workgroup and bucket (a cursor is
// touched once per block), whole 64-byte granules (4 keys) stored from 4 aligned lanes, the remainder of a
// bucket (<= 3 keys) carried in the registers of the thread that owns it, kPadKey pairs behind the last
// keys of every last block, the next tile's bytes fetched one tile ahead.  MODE 0 still computes gossamer's
// canonical form in the kernel (two FNV chains over 16 bytes).
// REC: the input is not bases but two-word super-k-mer records (kernels_route.hpp: SkRec2, 20 bytes = 1..16 windows of
// 32..63 bases); bases_aligned = the records, navail = their number -- the record form of extract1_part_kernel: a
// workgroup walks its own share of the records, stages up to 512 per tile in LDS, takes as many whole records as hold at
// most T windows, and thread t extracts windows P t .. P t + P - 1 of the tile's window sequence wherever the record
// boundaries fall (MODE 0 only: k-mer sets, and graphs counted as strand pairs).
template <int MODE, int NH, int NKEYS, int NBH = 8, bool REC = false, bool PACKED = false>
__global__ __launch_bounds__(kTB, 2) void extract2_part_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                               uint64_t nstarts, uint64_t navail, uint32_t len,
                                                               Key2* __restrict__ out, PartCounters* __restrict__ pc,
                                                               const GapTable* __restrict__ gt, uint32_t shift, uint64_t nsuper,
                                                               uint32_t blk_log2, const uint16_t* __restrict__ pbad = nullptr)
{
    static_assert(!(PACKED && REC), "records are not bases");          // (PACKED: as in extract1_part_kernel)
    constexpr int S = MODE == 1 ? 2 : 1;
    constexpr int P = NKEYS / S;                 // windows per thread; NKEYS keys per thread
    constexpr int T = kTB * P;
    constexpr int NVEC = T / 16 + 6;
    constexpr int NK = P * S;
    static_assert(!REC || MODE == 0, "the record form extracts one key per window");
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
    // a bucket's next block is reserved ahead of the tile that opens it (extract1_part_kernel)
    constexpr uint32_t kAhead = 40;
    unsigned long long resv = 0;
    bool has_resv = false;
    const uint64_t mask_hi = bits == 128 ? ~0ULL : ((1ULL << (bits - 64)) - 1);
    const uint64_t lmask = (1ULL << len) - 1;
    unsigned long long nvalid = 0;
    uint64_t wpos = 0;
    uint32_t ccnt = 0;
    Key2 kc[kCarry];
#pragma unroll
    for (int j = 0; j < kCarry; ++j) kc[j] = Key2{0, 0};

    auto fetch = [&](uint64_t byte0, uint4& q) {
        if constexpr (PACKED)
        {
            q = make_uint4(0u, 0xFFFFu, 0u, 0u);
            if (byte0 < navail + mis)
            {
                const uint64_t g = byte0 >> 4;
                q.x = reinterpret_cast<const uint32_t*>(bases_aligned)[g];
                q.y = pbad[g];
                const uint64_t left = navail + mis - byte0;
                q.z = left < 16 ? (0xFFFFu << (uint32_t)left) & 0xFFFFu : 0u;
            }
            return;
        }
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
        if constexpr (PACKED) { codes = q.x; bads = q.y | q.z; return; }
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        codes = 0; bads = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            uint32_t bad;
            const uint32_t x = base_codes(w[i], bad);
            codes |= pack_codes(x) << (8 * i);
            bads |= pack_flags(bad) << (4 * i);
        }
    };
    constexpr uint32_t NV0 = T / 16;             // thread tid < NV0 encodes vector tid, threads 0..5 also vector NV0 + tid
    static_assert(NVEC == NV0 + 6 && NV0 <= kTB, "one vector per thread and six more");
    uint32_t c0 = 0, b0 = 0, c1 = 0, b1 = 0;
    // REC: this workgroup's records [rc_next, rc_end), two per thread and tile staged (five words each)
    const uint32_t* recw = reinterpret_cast<const uint32_t*>(bases_aligned);
    const uint64_t nrec = navail;
    uint64_t rc_next = 0, rc_end = 0;
    uint32_t ra[5] = {0, 0, 0, 0, kSkPadWord2}, rb[5] = {0, 0, 0, 0, kSkPadWord2};      // (a pad: no window)
    // Staged records, 48 bytes each (phase A's share of `sorted`), turned ONCE by the thread that stages them (the one-word
    // kernel's form): words 0..4 the record's bases complemented (the reverse complement of window w is the field at bit
    // 2 w), words 5..9 the bases in reverse order, base j at bit 2 (len + 14 - j) (the forward form of window w is the
    // field at bit 30 - 2 w), word 10 the windows staged in front of the record, word 11 twice its windows.
    uint32_t* rbuf = pk;                                     // [512][12]
    uint32_t* rmark = pk + 512 * 12;                         // [256] the staged record that holds thread g's first window; [256] = records taken, [257] = their windows
    auto stage_rec = [&](uint32_t at, const uint32_t (&x)[5], uint32_t before) {
        const uint32_t x4m = x[4] & 0x0FFFFFFFu, nw = (x[4] >> 28) + 1u;
        auto rev4 = [](uint32_t w) { const uint32_t r = __brev(w); return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1); };
        // the 160 bits in reverse base order: base j at bit 2 (79 - j); moved down by 2 (65 - len) bits (4 .. 66)
        uint32_t y[6] = {rev4(x4m), rev4(x[3]), rev4(x[2]), rev4(x[1]), rev4(x[0]), 0u};
        uint32_t sh = 2u * (65u - len);
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd)
            if (sh >= 32u) { y[0] = y[1]; y[1] = y[2]; y[2] = y[3]; y[3] = y[4]; y[4] = 0u; sh -= 32u; }          // (uniform)
        uint32_t v[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) v[j] = __builtin_amdgcn_alignbit(y[j + 1], y[j], sh);
        uint4* const q = reinterpret_cast<uint4*>(rbuf + 12u * at);
        q[0] = make_uint4(~x[0], ~x[1], ~x[2], ~x[3]);
        q[1] = make_uint4(~x4m, v[0], v[1], v[2]);
        q[2] = make_uint4(v[3], v[4], before, 2u * nw);
    };
    auto fetch_recs = [&](uint64_t base) {
        const uint64_t ia = base + 2 * (uint64_t)tid, ib = ia + 1;
        // (loads only: the window counts are taken where the records are staged -- extract1_part_kernel)
#pragma unroll
        for (int j = 0; j < 5; ++j) { ra[j] = 0; rb[j] = 0; }
        ra[4] = rb[4] = kSkPadWord2;
        if (ia < rc_end) {
#pragma unroll
            for (int j = 0; j < 5; ++j) ra[j] = recw[5 * ia + j];
        }
        if (ib < rc_end) {
#pragma unroll
            for (int j = 0; j < 5; ++j) rb[j] = recw[5 * ib + j];
        }
    };
    if constexpr (REC)
    {
        const uint64_t per = ((nrec + gridDim.x - 1) / gridDim.x + 1) & ~1ULL;
        rc_next = (uint64_t)blockIdx.x * per;
        rc_end = rc_next + per < nrec ? rc_next + per : nrec;
        if (rc_next < rc_end) fetch_recs(rc_next);
    }
    else if (blockIdx.x < nsuper)
    {
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        const uint64_t tb = (uint64_t)blockIdx.x * T;
        if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
        if (tid < 6) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        encode(q0, c0, b0);
        if (tid < 6) encode(q1, c1, b1);
    }

#if defined(GOSS_STAMPS)
    // (timing build: cycles of wave 0 per phase, summed over its tiles, into the unused histogram words)
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define GOSS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define GOSS_STAMP(i)
#endif
    for (uint64_t st = blockIdx.x;; st += gridDim.x)
    {
        if constexpr (REC) { if (rc_next >= rc_end) break; }
        else { if (st >= nsuper) break; }
        const uint64_t tile_base = st * (uint64_t)T;
#if defined(GOSS_STAMPS)
        st_acc[5] += 1;
#endif
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        bool more;
        uint32_t wt = 0;                          // (REC) windows of this tile
        if constexpr (REC)
        {
            // the records that hold windows staged back to back (pads and what lies beyond the share hold none), with
            // the running sum of their windows: one scan carries both sums
            if (tid == 0) { rmark[kTB] = 512; rmark[kTB + 1] = 0xFFFFFFFFu; }
            const uint32_t rna = rec_windows(ra[4]), rnb = rec_windows(rb[4]);
            const uint32_t pa = rna ? 1u : 0u, pb = rnb ? 1u : 0u;
            uint32_t tot2;
            const uint32_t sc = block_excl_scan_u32((rna + rnb) | ((pa + pb) << 16), sh_scan, &tot2);
            const uint32_t ex = sc & 0xFFFFu, ca = sc >> 16, cb = ca + pa;
            const uint32_t ea = ex + rna, eb = ea + rnb;
            if (pa) stage_rec(ca, ra, ex);
            if (pb) stage_rec(cb, rb, ea);
            if (ex <= (uint32_t)T && eb > (uint32_t)T)
            {
                const uint32_t one = ea <= (uint32_t)T ? 1u : 0u;
                rmark[kTB] = 2 * tid + one;
                rmark[kTB + 1] = one ? ea : ex;
            }
            // thread g starts at window P g: a record of at most 16 windows holds at most two such windows
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)((16 + P - 1) / P); ++j)
            {
                const uint32_t ga = (ex + P - 1) / P + j, gb = (ea + P - 1) / P + j;
                if (ga * P < ea && ga < (uint32_t)kTB) rmark[ga] = ca;
                if (gb * P < eb && gb < (uint32_t)kTB) rmark[gb] = cb;
            }
            __syncthreads();
            const uint32_t nfit = rmark[kTB];
            wt = rmark[kTB + 1] == 0xFFFFFFFFu ? (tot2 & 0xFFFFu) : rmark[kTB + 1];
            rc_next += nfit;
            more = rc_next < rc_end;
            if (more) fetch_recs(rc_next);
        }
        else
        {
        if (tid < NV0) { pk[tid] = c0; iv[tid] = b0; }
        if (tid < 6) { pk[NV0 + tid] = c1; iv[NV0 + tid] = b1; }
        __syncthreads();
        more = st + gridDim.x < nsuper;
        if (more)
        {
            const uint64_t tb = (st + gridDim.x) * (uint64_t)T;
            if (tid < NV0) fetch(tb + (uint64_t)tid * 16, q0);
            if (tid < 6) fetch(tb + (uint64_t)(NV0 + tid) * 16, q1);
        }
        }

        // ---- windows out of registers, keys, rank inside their digit (no branch around the LDS atomics) ----
#if !defined(GOSS_E1_SYNC_BLOCKS)
        if (!has_resv && ((B - ((uint32_t)wpos & (B - 1))) & (B - 1)) < kAhead)
        {
            resv = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)B);
            has_resv = true;
        }
#endif
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(0);              // (priorities by phase as in extract1_part_kernel)
#endif
        GOSS_STAMP(0);
        Key2 kreg[NK];
        uint32_t rk[NK];
        uint32_t vm = 0;
        if constexpr (REC)
        {
            // windows P tid .. P tid + P - 1 of the tile's sequence, starting in record rmark[tid]: every window is cut out
            // of its record's bases (field of 2 len bits at twice its offset), its forward key the base-4 reversal of the
            // field, its reverse complement the complemented field
            const uint32_t j0 = tid * P;
            const uint32_t left = wt > j0 ? wt - j0 : 0;
            vm = left >= (uint32_t)P ? (uint32_t)((1ULL << P) - 1ULL) : ((1u << left) - 1u);
            nvalid += __popc(vm);
            uint32_t r_at = left ? rmark[tid] : 0u;
            uint32_t r_s2 = 0, r_nw2 = 0, r_before = 0;
            uint32_t rc[5] = {0, 0, 0, 0, 0}, rv[5] = {0, 0, 0, 0, 0};
            auto load_rec = [&]() {
                const uint4* const q = reinterpret_cast<const uint4*>(rbuf + 12u * r_at);
                const uint4 a = q[0], b = q[1], c = q[2];
                rc[0] = a.x; rc[1] = a.y; rc[2] = a.z; rc[3] = a.w; rc[4] = b.x;
                rv[0] = b.y; rv[1] = b.z; rv[2] = b.w; rv[3] = c.x; rv[4] = c.y;
                r_before = c.z; r_nw2 = c.w;                       // (no pad is staged)
            };
            auto next_rec = [&]() { ++r_at; r_s2 = 0; load_rec(); };
            load_rec();
            r_s2 = 2u * (j0 - r_before);
            const uint32_t spare = 256u + (tid & 31u);
            uint32_t bin[NK];
            const uint32_t mh2 = (uint32_t)mask_hi, mh3 = (uint32_t)(mask_hi >> 32);
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                // the window at bit r_s2 of record r_at: four funnel shifts of each image (shifts of 0 .. 30 bits)
                const uint32_t t2 = 30u - r_s2;
                const uint32_t r0 = __builtin_amdgcn_alignbit(rc[1], rc[0], r_s2), r1 = __builtin_amdgcn_alignbit(rc[2], rc[1], r_s2);
                const uint32_t r2 = __builtin_amdgcn_alignbit(rc[3], rc[2], r_s2) & mh2, r3 = __builtin_amdgcn_alignbit(rc[4], rc[3], r_s2) & mh3;
                const uint32_t f0 = __builtin_amdgcn_alignbit(rv[1], rv[0], t2), f1 = __builtin_amdgcn_alignbit(rv[2], rv[1], t2);
                const uint32_t f2 = __builtin_amdgcn_alignbit(rv[3], rv[2], t2) & mh2, f3 = __builtin_amdgcn_alignbit(rv[4], rv[3], t2) & mh3;
                const Key2 f{(uint64_t)f0 | ((uint64_t)f1 << 32), (uint64_t)f2 | ((uint64_t)f3 << 32)};
                const Key2 rck{(uint64_t)r0 | ((uint64_t)r1 << 32), (uint64_t)r2 | ((uint64_t)r3 << 32)};
                if (i + 1 < P)
                {
                    r_s2 += 2u;
                    if (r_s2 >= r_nw2 && r_at < 511) next_rec();
                }
                const bool ok = (vm >> i) & 1u;
                Key2 k;
                if constexpr (NBH == 0) k = strand_rep2(f, rck, len, lmask); else k = canonical_tail<NBH>(f, rck);
                kreg[i] = k;
                bin[i] = ok ? key_digit(k, shift) : spare;
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) rk[i] = atomicAdd(&dh[bin[i]], 1u);
        }
        else
        {
            // The bytes' form, round 4: windows out of the register buffer in 32-bit words by CONSTANT funnel shifts, as in
            // the one-word kernel -- the buffer shifted to the thread's first base once, the reverse complement of window i
            // the field [2 i, 2 i + bits) of the complemented words, the forward strand rolled (the bases it takes in are
            // pre-shifted once: base i + len - 1 at bits 2 i of nx), validity of all P windows at once by run doubling.
            // (Before: every window cut its next base out of three 64-bit words by variable shifts and rolled two 128-bit
            // values, and tested its own len flags: phase B was 27 500 of the 39 600 cycles of a tile.)
            const uint32_t q0i = tid * P + mis;
            const uint32_t v0 = q0i >> 4, sh = q0i & 15u;
            const uint64_t p0 = tile_base + (uint64_t)tid * P;
            // ---- validity: window i is valid iff flags [i, i + len) are clear (96 flags from the thread's first base)
            {
                const uint64_t inv_lo = (uint64_t)iv[v0] | ((uint64_t)iv[v0 + 1] << 16) | ((uint64_t)iv[v0 + 2] << 32) | ((uint64_t)iv[v0 + 3] << 48);
                const uint64_t inv_hi = (uint64_t)iv[v0 + 4] | ((uint64_t)iv[v0 + 5] << 16);
                // (bits above the 96 that were read count as good: they belong to windows beyond P)
                uint64_t rl = ~(sh ? ((inv_lo >> sh) | (inv_hi << (64 - sh))) : inv_lo), rh = ~(inv_hi >> sh);
                uint64_t al = ~0ULL;
                uint32_t covered = 0;
                auto shr_lo = [](uint64_t lo, uint64_t hi, uint32_t s) -> uint64_t {          // low word of (hi:lo) >> s, s < 128 (uniform)
                    return s == 0 ? lo : s < 64 ? ((lo >> s) | (hi << (64 - s))) : (hi >> (s - 64));
                };
#pragma unroll
                for (int j = 0; j < 6; ++j)               // len <= 63
                {
                    if ((len >> j) & 1u) { al &= shr_lo(rl, rh, covered); covered += 1u << j; }
                    const uint32_t s = 1u << j;           // run &= run >> s (constant s < 64)
                    const uint64_t nl = (rl >> s) | (rh << (64 - s)), nh = rh >> s;
                    // (the bits shifted in at the top are zero = "not good": only windows near bit 128 - len see them, none of ours)
                    rl &= nl; rh &= nh;
                }
                const uint64_t left = nstarts > p0 ? nstarts - p0 : 0;
                const uint32_t lim = left >= (uint64_t)P ? (uint32_t)((1ULL << P) - 1ULL) : ((1u << (uint32_t)left) - 1u);
                vm = (uint32_t)al & lim;
            }
            nvalid += __popc(vm);
            const uint32_t spare = 256u + (tid & 31u);
            uint32_t bin[NK];
            // ---- the buffer from the thread's first base on: five words (P + len - 1 <= 76 bases)
            uint32_t x[5], cx[5];
            {
                const uint32_t s2 = 2 * sh;
                uint32_t w[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) w[j] = pk[v0 + j];
#pragma unroll
                for (int j = 0; j < 5; ++j) { x[j] = __builtin_amdgcn_alignbit(w[j + 1], w[j], s2); cx[j] = ~x[j]; }
            }
            // the bases rolled in: base len - 1 + i at bits 2 i (2 (len - 1) = 62 .. 124: words 1 .. 3 of x)
            uint32_t nx;
            {
                const uint32_t pn = 2 * (len - 1), wi = pn >> 5, bo = pn & 31u;
                const uint32_t lo = wi == 1 ? x[1] : wi == 2 ? x[2] : x[3];
                const uint32_t hi = wi == 1 ? x[2] : wi == 2 ? x[3] : x[4];
                nx = __builtin_amdgcn_alignbit(hi, lo, bo);
            }
            const uint32_t mh2 = (uint32_t)mask_hi, mh3 = (uint32_t)(mask_hi >> 32);
            // forward key of window 0 (base-4 reversal of its field)
            uint32_t f0, f1, f2, f3;
            {
                const uint64_t elo = (uint64_t)x[0] | ((uint64_t)x[1] << 32);
                const uint64_t ehi = ((uint64_t)x[2] | ((uint64_t)x[3] << 32)) & mask_hi;
                const uint64_t rlo = rev64(ehi), rhi = rev64(elo);
                const uint32_t sft = 128 - bits;
                uint64_t flo, fhi;
                if (sft == 64) { flo = rhi; fhi = 0; }
                else { flo = (rlo >> sft) | (rhi << (64 - sft)); fhi = rhi >> sft; }
                f0 = (uint32_t)flo; f1 = (uint32_t)(flo >> 32); f2 = (uint32_t)fhi; f3 = (uint32_t)(fhi >> 32);
            }
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                if (i)
                {
                    const uint32_t nb = (nx >> (2 * i)) & 3u;
                    f3 = __builtin_amdgcn_alignbit(f3, f2, 30) & mh3;
                    f2 = __builtin_amdgcn_alignbit(f2, f1, 30) & mh2;
                    f1 = __builtin_amdgcn_alignbit(f1, f0, 30);
                    f0 = (f0 << 2) | nb;
                }
                const uint32_t r0 = i ? __builtin_amdgcn_alignbit(cx[1], cx[0], 2 * i) : cx[0];
                const uint32_t r1 = i ? __builtin_amdgcn_alignbit(cx[2], cx[1], 2 * i) : cx[1];
                const uint32_t r2 = (i ? __builtin_amdgcn_alignbit(cx[3], cx[2], 2 * i) : cx[2]) & mh2;
                const uint32_t r3 = (i ? __builtin_amdgcn_alignbit(cx[4], cx[3], 2 * i) : cx[3]) & mh3;
                const Key2 f{(uint64_t)f0 | ((uint64_t)f1 << 32), (uint64_t)f2 | ((uint64_t)f3 << 32)};
                const Key2 rck{(uint64_t)r0 | ((uint64_t)r1 << 32), (uint64_t)r2 | ((uint64_t)r3 << 32)};
                const bool ok = (vm >> i) & 1u;
                if (MODE == 0)
                {
                    Key2 k;
                    if constexpr (NBH == 0) k = strand_rep2(f, rck, len, lmask); else k = canonical_tail<NBH>(f, rck);          // (NBH 0: the strand representative)
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
        GOSS_STAMP(1);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_C);
#endif

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
                unsigned long long at, want;
                if (has_resv && need <= B) { at = resv; want = B; has_resv = false; }          // (reserved before the ranking)
                else
                {
                    want = ((uint64_t)(need + B - 1) >> blk_log2) << blk_log2;
                    at = atomicAdd(&pc->cursors[tid * kCursorStride], want);
                }
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
        GOSS_STAMP(2);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_S);
#endif
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
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_E);
#endif
        if (more && !REC)
        {
            encode(q0, c0, b0);
            if (tid < 6) encode(q1, c1, b1);
        }
        // (REC: the next records must have arrived before this tile's stores are issued -- loads and stores share one in-order
        // counter; they are looked at here so that the wait is the compiler's own: extract1_part_kernel)
        if constexpr (REC)
            asm volatile("" ::"v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(ra[4]), "v"(rb[0]), "v"(rb[1]), "v"(rb[2]), "v"(rb[3]), "v"(rb[4]));
        __syncthreads();
        GOSS_STAMP(3);
#if !defined(GOSS_E1_NOPRIO)
        __builtin_amdgcn_s_setprio(GOSS_E1_PRIO_D);
#endif

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
        GOSS_STAMP(4);
    }
#if defined(GOSS_STAMPS)
    if (tid == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&pc->hist[500 + i], st_acc[i]);
#endif
#undef GOSS_STAMP

    // ---- the end: carried keys and the unused tail of every open block ----
    if (sh_ovf == 0)
    {
        bool spare_blk = has_resv;
        if (ccnt)
        {
            const uint32_t room = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
            if (room == 0)
            {
                unsigned long long at;
                if (has_resv) { at = resv; spare_blk = false; }
                else at = atomicAdd(&pc->cursors[tid * kCursorStride], (unsigned long long)B);
                if (at + B > my_cap) { atomicOr(&pc->overflow, 1ULL); ccnt = 0; wpos = 0; }
                else wpos = my_start + at;
            }
#pragma unroll
            for (int j = 0; j < kCarry; ++j)
                if ((uint32_t)j < ccnt) out[wpos + j] = kc[j];
            wpos += ccnt;
        }
        const uint32_t tail = (B - ((uint32_t)wpos & (B - 1))) & (B - 1);
        // a reserved block nobody opened counts as handed out: pads, or (beyond the region) the chunk is redone
        uint32_t blk_n = 0;
        if (spare_blk)
        {
            if (resv + B > my_cap) atomicOr(&pc->overflow, 1ULL);
            else blk_n = B;
        }
        __syncthreads();
        t_base[tid] = make_uint2(tail, blk_n);
        reinterpret_cast<uint64_t*>(sorted)[tid] = wpos;
        reinterpret_cast<uint64_t*>(sorted)[256 + tid] = my_start + resv;
        __syncthreads();
        for (uint32_t d = 0; d < 256; ++d)
        {
            const uint32_t n = t_base[d].x, n2 = t_base[d].y;
            const uint64_t from = reinterpret_cast<const uint64_t*>(sorted)[d], from2 = reinterpret_cast<const uint64_t*>(sorted)[256 + d];
            for (uint32_t j = tid; j < n; j += kTB) out[from + j] = Key2{~0ULL, ~0ULL};
            for (uint32_t j = tid; j < n2; j += kTB) out[from2 + j] = Key2{~0ULL, ~0ULL};
        }
    }
    if (NH > 0) { if (lh[tid]) atomicAdd(&pc->hist[tid], (unsigned long long)lh[tid]); }
    if (NH > 1) { if (lh[tid + 256]) atomicAdd(&pc->hist[tid + 256], (unsigned long long)lh[tid + 256]); }
    for (int o = 32; o > 0; o >>= 1) nvalid += __shfl_down(nvalid, o, 64);
    if (lane_id() == 0 && nvalid) { atomicAdd(&pc->keys_out, nvalid * S); atomicAdd(&pc->windows, nvalid); }
}

}  // namespace goss
