// goss_key.hpp -- k-mer key arithmetic shared by the HIP kernels and the host code.
//
// Semantics (what must be bit-identical to the reference):
//   * base codes A=0 C=1 G=2 T=3, first base in the most significant used bits
//     (GossReadBaseString.hh:133-188);
//   * reverse complement of a len-mer held in 128 bits (BigInteger.hh:204-217 with
//     Gossamer::rev, Utils.hh:377-396);
//   * canonical form = whichever of {x, rc(x)} has the smaller FNV-1a-64 hash of its 16
//     little-endian bytes, ties -> smaller value (RankSelect.hh:126-140,
//     BigInteger.hh:528-536,572-582).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define GK_HD __host__ __device__ __forceinline__
#else
#define GK_HD inline
#endif

namespace goss {

// One-word key: len <= 31 (2*len <= 62 bits; the all-ones word is free as a sentinel).
struct Key1 {
    uint64_t lo;
    static constexpr int kWords = 1;
};
// Two-word key: len <= 64.
struct Key2 {
    uint64_t lo, hi;
    static constexpr int kWords = 2;
};

GK_HD bool operator==(const Key1& a, const Key1& b) { return a.lo == b.lo; }
GK_HD bool operator!=(const Key1& a, const Key1& b) { return a.lo != b.lo; }
GK_HD bool operator<(const Key1& a, const Key1& b) { return a.lo < b.lo; }
GK_HD bool operator==(const Key2& a, const Key2& b) { return a.lo == b.lo && a.hi == b.hi; }
GK_HD bool operator!=(const Key2& a, const Key2& b) { return a.lo != b.lo || a.hi != b.hi; }
GK_HD bool operator<(const Key2& a, const Key2& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// key >> s, returning the low 64 bits (s < 128).
GK_HD uint64_t key_shr64(const Key1& k, uint32_t s) { return s >= 64 ? 0 : (k.lo >> s); }
GK_HD uint64_t key_shr64(const Key2& k, uint32_t s)
{
    if (s == 0) return k.lo;
    if (s < 64) return (k.lo >> s) | (k.hi << (64 - s));
    return s >= 128 ? 0 : (k.hi >> (s - 64));
}
// 8-bit digit of a key starting at bit `shift` (any alignment).
GK_HD uint32_t key_digit(const Key1& k, uint32_t shift) { return (uint32_t)key_shr64(k, shift) & 0xFFu; }
GK_HD uint32_t key_digit(const Key2& k, uint32_t shift) { return (uint32_t)key_shr64(k, shift) & 0xFFu; }

// true if (key >> s) does not fit 64 bits
GK_HD bool key_shr_overflows(const Key1&, uint32_t) { return false; }
GK_HD bool key_shr_overflows(const Key2& k, uint32_t s) { return s < 64 && (k.hi >> s) != 0; }

// bits [s, s+64) of key & mask(D) where D >= s : used by the low-bits column split.
GK_HD uint64_t key_lo_word(const Key1& k) { return k.lo; }
GK_HD uint64_t key_lo_word(const Key2& k) { return k.lo; }
GK_HD uint64_t key_hi_word(const Key1&) { return 0; }
GK_HD uint64_t key_hi_word(const Key2& k) { return k.hi; }

// Base-4 reverse of a 64-bit word.
GK_HD uint64_t rev64(uint64_t x)
{
    x = ((x & 0x3333333333333333ULL) << 2) | ((x >> 2) & 0x3333333333333333ULL);
    x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL);
#if defined(__HIP_DEVICE_COMPILE__)
    // byte reversal of the two halves, swapped: v_perm_b32 does each half in one op
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __builtin_bswap32(lo);
    hi = __builtin_bswap32(hi);
    return ((uint64_t)lo << 32) | hi;
#else
    return __builtin_bswap64(x);
#endif
}

GK_HD Key1 revcomp(const Key1& k, uint32_t len)
{
    Key1 r;
    r.lo = rev64(~k.lo) >> (64 - 2 * len);
    return r;
}
GK_HD Key2 revcomp(const Key2& k, uint32_t len)
{
    // 128-bit: swap words, rev(~w) each, shift right by 128 - 2*len.
    uint64_t nlo = rev64(~k.hi), nhi = rev64(~k.lo);
    uint32_t s = 128 - 2 * len;
    Key2 r;
    if (s == 0) { r.lo = nlo; r.hi = nhi; }
    else if (s < 64) { r.lo = (nlo >> s) | (nhi << (64 - s)); r.hi = nhi >> s; }
    else { r.lo = nhi >> (s - 64); r.hi = 0; }
    return r;
}

constexpr uint64_t kFnvSeed = 14695981039346656037ULL;
constexpr uint64_t kFnvPrime = 1099511628211ULL;
// kFnvPrime^8 mod 2^64: folding in the eight zero bytes of an all-zero high word.
constexpr uint64_t fnv_pow(int n) { uint64_t r = 1; for (int i = 0; i < n; ++i) r *= kFnvPrime; return r; }
constexpr uint64_t kFnvPrime8 = fnv_pow(8);

// h * kFnvPrime (= 2^40 + 0x1B3) mod 2^64.
GK_HD uint64_t fnv_mul(uint64_t h)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(GOSS_FNV_U24)
    // Variant with 16-bit limbs on the 24-bit multiplier (v_mul_u32_u24) instead of
    // v_mad_u64_u32 + v_mul_lo_u32.  Measured slower on MI355X (extract 98.6 vs 92.4 ms on C2),
    // kept for reference only.
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    const uint32_t pa = __umul24(lo & 0xFFFFu, 0x1B3u), pb = __umul24(lo >> 16, 0x1B3u);
    const uint32_t pc = __umul24(hi & 0xFFFFu, 0x1B3u), pd = __umul24(hi >> 16, 0x1B3u);
    const uint64_t p = (uint64_t)pa + ((uint64_t)pb << 16);            // lo * 0x1B3, 41 bits
    const uint32_t nhi = pc + (pd << 16) + (uint32_t)(p >> 32) + (lo << 8);   // + (h << 40)
    return ((uint64_t)nhi << 32) | (uint32_t)p;
#else
    return h * kFnvPrime;
#endif
}

GK_HD uint64_t fnv_word(uint64_t w, uint64_t h)
{
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        h ^= (w >> (8 * i)) & 0xFFULL;
        h = fnv_mul(h);
    }
    return h;
}

GK_HD uint64_t key_hash(const Key1& k) { return fnv_word(k.lo, kFnvSeed) * kFnvPrime8; }
GK_HD uint64_t key_hash(const Key2& k) { return fnv_word(k.hi, fnv_word(k.lo, kFnvSeed)); }

// One-word keys with nbytes significant bytes (2*len <= 8*nbytes): the remaining 16 - nbytes
// bytes of the 16-byte image are zero, and a zero byte only multiplies by the prime, so they fold
// into one multiplication by prime^(16 - nbytes) (tail).
template <int NB>
GK_HD uint64_t key_hash_short(uint64_t lo)
{
    uint64_t h = kFnvSeed;
#pragma unroll
    for (int i = 0; i < NB; ++i)
    {
        h ^= (lo >> (8 * i)) & 0xFFULL;
        h = fnv_mul(h);
    }
    return h * fnv_pow(16 - NB);
}
template <int NB>
GK_HD Key1 canonical_short(const Key1& x, const Key1& rc)
{
    uint64_t h0 = key_hash_short<NB>(x.lo), h1 = key_hash_short<NB>(rc.lo);
    if (h0 > h1) return rc;
    if (h0 == h1 && rc < x) return rc;
    return x;
}

// Two-word keys whose high word has at most NBH significant bytes (NBH = 2, 4, 6 or 8): the
// zero bytes above them fold into one multiplication by prime^(8 - NBH).
template <int NBH>
GK_HD uint64_t key_hash_tail(const Key2& k)
{
    uint64_t h = fnv_word(k.lo, kFnvSeed);
#pragma unroll
    for (int i = 0; i < NBH; ++i)
    {
        h ^= (k.hi >> (8 * i)) & 0xFFULL;
        h = fnv_mul(h);
    }
    return NBH < 8 ? h * fnv_pow(8 - NBH) : h;
}
template <int NBH>
GK_HD Key2 canonical_tail(const Key2& x, const Key2& rc)
{
    uint64_t h0 = key_hash_tail<NBH>(x), h1 = key_hash_tail<NBH>(rc);
    if (h0 > h1) return rc;
    if (h0 == h1 && rc < x) return rc;
    return x;
}

// canonical(x) given rc = revcomp(x)
template <class K>
GK_HD K canonical(const K& x, const K& rc)
{
    uint64_t h0 = key_hash(x), h1 = key_hash(rc);
    if (h0 > h1) return rc;
    if (h0 == h1 && rc < x) return rc;
    return x;
}

// ASCII -> 2-bit code; returns 4 for a non-base.
GK_HD uint32_t base_code(uint8_t c)
{
    uint32_t l = c | 0x20u;
    uint32_t x = (l >> 1) & 3u;          // a->0 c->1 t->2 g->3
    x ^= x >> 1;                         // a->0 c->1 g->2 t->3
    bool ok = (l == 'a') | (l == 'c') | (l == 'g') | (l == 't');
    return ok ? x : 4u;
}

// counter-based RNG used by the synthetic read generator
GK_HD uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// Synthetic genome base i (i.i.d. uniform ACGT): 32 bases per hash.
GK_HD uint32_t synth_genome_base(uint64_t seed, uint64_t i)
{
    uint64_t h = splitmix64(seed ^ (0x5851F42D4C957F2DULL * ((i >> 5) + 1)));
    return (uint32_t)(h >> (2 * (i & 31))) & 3u;
}

// Synthetic read r, position j in [0, read_len]: byte value (read_len -> '\n').
GK_HD char synth_read_byte(uint64_t seed, uint64_t genome_len, uint32_t read_len, uint64_t r, uint32_t j)
{
    if (j >= read_len) return '\n';
    uint64_t h = splitmix64(seed * 0x2545F4914F6CDD1DULL + 0xD1B54A32D192ED03ULL * (r + 1));
    uint64_t pos = (h >> 1) % (genome_len - read_len + 1);
    bool flip = h & 1;
    if (r % 97 == 96)
    {
        uint64_t h2 = splitmix64(h);
        if ((uint32_t)(h2 % read_len) == j) return 'N';
    }
    uint32_t b;
    if (!flip) b = synth_genome_base(seed, pos + j);
    else       b = 3u - synth_genome_base(seed, pos + (read_len - 1 - j));
    return "ACGT"[b];
}

}  // namespace goss
