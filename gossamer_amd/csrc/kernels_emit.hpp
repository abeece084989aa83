// kernels_emit.hpp -- on-disk arrays: Elias-Fano split, DenseSelect, VariableByteArray, SparseArray decode, synthetic reads.
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_common.hpp"

namespace goss {

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

// The same bitmap built from the KEYS' side: a workgroup takes 2 048 consecutive keys, sets their bits h_i - h_first in a
// window of words in LDS and writes the window out -- whole words with plain stores, its first and last word (shared
// with the neighbouring workgroups' windows) with an atomic OR.  Keys are read once and coalesced where the kernel above
// does a binary search over all keys per word (27 dependent loads: 1.3 ms of C2's 2 ms emit).  The bitmap must be
// zeroed before.  A window is 2.4 bits per key on average (SparseArray's D: SparseArray.cc:47-103); keys that lie
// further apart than the LDS window holds set their bits in memory directly.
constexpr int kEfChunk = 2048, kEfWords = 1024;          // keys per workgroup; words of its LDS window (64 K bit positions)
template <class K>
__global__ __launch_bounds__(kTB) void ef_high_bits_keys_kernel(const K* __restrict__ keys, uint64_t m, uint32_t D, uint64_t first_index,
                                                                 uint64_t nwords, unsigned long long* __restrict__ words_at, uint64_t wbase)
{
    // (a range's span of the bitmap: words_at[0] is word `wbase` of the whole; nwords bounds the whole's word numbers)
    unsigned long long* const words = words_at - wbase;
    __shared__ unsigned long long win[kEfWords];
    const uint64_t i0 = (uint64_t)blockIdx.x * kEfChunk;
    if (i0 >= m) return;
    const uint64_t i1 = i0 + kEfChunk < m ? i0 + kEfChunk : m;
    const uint64_t w0 = (ef_hi(keys, i0, D) + first_index + i0) >> 6;          // (uniform)
    for (uint32_t j = threadIdx.x; j < (uint32_t)kEfWords; j += kTB) win[j] = 0;
    __syncthreads();
    uint64_t wmax = w0;
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += kTB)
    {
        const uint64_t h = ef_hi(keys, i, D) + first_index + i;
        const uint64_t w = h >> 6;
        if (w - w0 < (uint64_t)kEfWords) { atomicOr(&win[w - w0], 1ULL << (h & 63)); wmax = w > wmax ? w : wmax; }
        else if (w < nwords) atomicOr(&words[w], 1ULL << (h & 63));
    }
    // the last word of the window that holds a bit (a maximum over the workgroup)
    __shared__ unsigned long long sh_max;
    if (threadIdx.x == 0) sh_max = 0;
    __syncthreads();
    atomicMax(&sh_max, (unsigned long long)(wmax - w0));
    __syncthreads();
    const uint32_t last = (uint32_t)sh_max;
    for (uint32_t j = threadIdx.x; j <= last; j += kTB)
    {
        const unsigned long long v = win[j];
        if (w0 + j >= nwords) continue;
        if (j == 0 || j == last) { if (v) atomicOr(&words[w0 + j], v); }
        else words[w0 + j] = v;
    }
}

// Distributed emission: a range's span of the bitmap ORed into the whole (spans of neighbouring ranges share at most
// their boundary words; the launches follow one another on one stream).
__global__ __launch_bounds__(kTB) void ef_or_span_kernel(unsigned long long* __restrict__ words, const unsigned long long* __restrict__ span, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) { const unsigned long long v = span[i]; if (v) words[i] |= v; }
}
// ... ones per word of the assembled bitmap (scanned by the host's exclusive scan), and the high parts read back from
// it: one i sits at position p, so key_i >> D = p - i -- what the DenseSelect builders take as their keys (D = 0).
__global__ __launch_bounds__(kTB) void ef_word_ones_kernel(const unsigned long long* __restrict__ words, uint64_t nwords, uint64_t* __restrict__ ones)
{
    const uint64_t w = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (w < nwords) ones[w] = (uint64_t)__popcll(words[w]);
}
__global__ __launch_bounds__(kTB) void ef_high_from_bits_kernel(const unsigned long long* __restrict__ words, uint64_t nwords,
                                                                 const uint64_t* __restrict__ before, Key1* __restrict__ hk, uint64_t total)
{
    const uint64_t w = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (w >= nwords) return;
    unsigned long long v = words[w];
    uint64_t i = before[w];
    while (v && i < total)
    {
        const uint32_t b = (uint32_t)__builtin_ctzll(v);
        hk[i].lo = w * 64 + b - i;
        v &= v - 1;
        ++i;
    }
}

// --------------------------------------------------------------------------------------
// K8: DenseSelect image (DenseSelect::Builder, DenseArray.cc:446-694)
// --------------------------------------------------------------------------------------
//
// The indexed positions are never materialised: for sense 1 (ones) position i is h_i; for
// sense 0 (zeros) the j-th zero sits at j + #{i : (key_i >> D) <= j}.

// Where the positions come from (round 5: a range's owner builds the whole blocks inside its share of the ones / zeros,
// DenseArray.cc:446-647 per block; the assembler the blocks that straddle two ranges, from the bitmap):
//   keys, m, D       the keys of THIS range (all keys: first = 0), their high parts key >> D;
//   first            ones of the whole array in the ranges below this one: one i of the whole is key (i - first);
//                    the j-th zero of the whole sits at j + first + #{local keys with high part <= j} -- as long as j is
//                    at or above the last high part of the ranges below and below the first of the ranges above (the
//                    zeros a range owns);
//   direct           (straddling blocks) the positions themselves, direct[idx - direct_first].
template <class K>
struct DsSrc {
    const K* keys; uint64_t m; uint32_t D; int invert; uint64_t first;
    const uint64_t* direct; uint64_t direct_first;
};
template <class K>
__device__ __forceinline__ uint64_t ds_pos(const DsSrc<K>& s, uint64_t idx)
{
    if (s.direct) return s.direct[idx - s.direct_first];
    if (!s.invert) return ef_hi(s.keys, idx - s.first, s.D) + idx;
    uint64_t a = 0, b = s.m;                // upper_bound of idx among the high parts
    while (a < b)
    {
        uint64_t mid = a + ((b - a) >> 1);
        if (ef_hi(s.keys, mid, s.D) <= idx) a = mid + 1; else b = mid;
    }
    return idx + s.first + a;
}

enum : uint32_t { kDsSmall = 0, kDsSpill64 = 1, kDsSpill32 = 2, kDsSpill16 = 3, kDsSpill8 = 4, kDsIntermediate = 5 };

// Pass 1: one thread per block of 8192 indexed positions: block type and byte size (already
// padded to 8).  count = number of indexed positions.
// (blocks b0 .. b0 + nblocks - 1 of the whole array; the arrays are indexed from b0)
template <class K>
__global__ void ds_classify_kernel(const DsSrc<K> src, uint64_t count, uint64_t b0, uint64_t nblocks,
                                   uint32_t* __restrict__ btype, uint64_t* __restrict__ bbytes,
                                   uint64_t* __restrict__ brank)
{
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint64_t first = (b0 + b) << 13;
    uint64_t cnt = count - first < 8192 ? count - first : 8192;
    uint64_t pp = ds_pos(src, first);
    uint64_t p = ds_pos(src, first + cnt - 1);
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
            uint64_t r = ds_pos(src, first + s * 64 + 63) - ds_pos(src, first + s * 64);
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
// (index: the master index entries of these blocks, or null when the assembler writes them)
template <class K>
__global__ __launch_bounds__(128) void ds_fill_kernel(const DsSrc<K> src, uint64_t count, uint64_t b0, const uint32_t* __restrict__ btype,
                                                      const uint64_t* __restrict__ boff, const uint64_t* __restrict__ brank,
                                                      uint8_t* __restrict__ image, uint64_t* __restrict__ index)
{
    __shared__ uint32_t sh_sub[128];
    const uint64_t b = blockIdx.x;
    const uint32_t s = threadIdx.x;
    const uint64_t first = (b0 + b) << 13;
    const uint64_t cnt = count - first < 8192 ? count - first : 8192;
    const uint32_t t = btype[b];
    const uint64_t off = boff[b];
    const uint64_t pp = brank[b];
    uint8_t* blk = image + off;
    if (s == 0 && index) index[b] = off | t;
    if (t == kDsSmall)
    {
        uint16_t v = (uint16_t)(ds_pos(src, first + (uint64_t)s * 64) - pp);
        reinterpret_cast<uint16_t*>(blk)[s] = v;
    }
    else if (t == kDsSpill32)
    {
        for (uint64_t i = s; i < cnt; i += 128)
            reinterpret_cast<uint32_t*>(blk)[i] = (uint32_t)(ds_pos(src, first + i) - pp);
    }
    else if (t == kDsSpill64)
    {
        for (uint64_t i = s; i < cnt; i += 128)
            reinterpret_cast<uint64_t*>(blk)[i] = ds_pos(src, first + i);
    }
    else
    {
        // intermediate: 128 x u32 sample offsets, 128 x u16 internal pointers, sub-blocks
        uint64_t p0 = ds_pos(src, first + (uint64_t)s * 64);
        uint64_t p1 = ds_pos(src, first + (uint64_t)s * 64 + 63);
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
                uint64_t d = ds_pos(src, first + (uint64_t)s * 64 + j) - p0;
                if (ty == kDsSpill8) blk[base + j] = (uint8_t)d;
                else if (ty == kDsSpill16) reinterpret_cast<uint16_t*>(blk + base)[j] = (uint16_t)d;
                else reinterpret_cast<uint32_t*>(blk + base)[j] = (uint32_t)d;
            }
        }
    }
}

// Positions of elements idx0 .. idx0 + n - 1 (ones, or -- invert -- zeros) of an assembled bitmap: `before` = ones in the
// words below each word (exclusive scan of ef_word_ones_kernel's output).  For the few blocks that straddle two ranges.
__global__ __launch_bounds__(kTB) void ef_select_range_kernel(const unsigned long long* __restrict__ words, uint64_t nwords,
                                                              const uint64_t* __restrict__ before, uint64_t idx0, uint64_t n, int invert,
                                                              uint64_t* __restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    const uint64_t idx = idx0 + j;
    // the last word with at most idx such elements below it
    uint64_t a = 0, b = nwords;
    while (b - a > 1)
    {
        const uint64_t mid = a + ((b - a) >> 1);
        const uint64_t below = invert ? mid * 64 - before[mid] : before[mid];
        if (below <= idx) a = mid; else b = mid;
    }
    unsigned long long v = invert ? ~words[a] : words[a];
    uint64_t r = idx - (invert ? a * 64 - before[a] : before[a]);          // rank inside the word
    while (r--) v &= v - 1;
    out[j] = a * 64 + (uint64_t)__builtin_ctzll(v);
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
