// goss_reader.hpp -- the READ side of the on-disk arrays, on the device: select / rank / access of
// a SparseArray through its DenseSelect indexes, exactly as the reference evaluates them
// (WordyBitVector.tcc:17-54, DenseArray.cc:134-258, SparseArray.hh:246-364).  Used by lint-graph's
// second pass to check the index structures of an existing object against its decoded edge list;
// it is also what a device-resident graph traversal would call.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "goss_key.hpp"

namespace goss {

struct RdBits { const uint64_t* w; uint64_t nwords; };

// DenseSelect::Header (DenseArray.hh:98-136) + the file image it describes
struct RdDenseSelect {
    const uint8_t* data;
    uint64_t size;
    uint64_t flags, indexArrayOffset, rankArrayOffset;
    uint64_t logBlockSize, blockSize, logSampleRate, sampleRate, numBlocks;
};

struct RdSparse {
    uint64_t D, count;
    uint64_t size_lo, size_hi;             // N
    RdBits hi;
    RdDenseSelect d0, d1;                  // zeros (inverted sense) and ones
    uint32_t ncols;
    const uint8_t* col[4];
    uint32_t col_bytes[4], col_shift[4];
};

enum : uint64_t { kRdSmall = 0, kRdFull64 = 1, kRdFull32 = 2, kRdFull16 = 3, kRdFull8 = 4, kRdIntermediate = 5, kRdTypeMask = 7 };

// position of the (rank)-th set bit of x (Utils.hh select1)
__device__ inline uint32_t rd_select1(uint64_t x, uint32_t rank)
{
    for (uint32_t i = 0; i < rank; ++i) x &= x - 1;
    return (uint32_t)__ffsll((long long)x) - 1u;
}

// WordyBitVector::select<Sense>(from, count): the count-th bit of the wanted sense at or after
// position `from` (WordyBitVector.tcc:17-54).  false = ran off the end.
__device__ inline bool rd_bits_select(const RdBits& v, bool invert, uint64_t from, uint64_t count, uint64_t* out)
{
    uint64_t w = from >> 6, b = from & 63;
    if (w >= v.nwords) return false;
    uint64_t x = (invert ? ~v.w[w] : v.w[w]) >> b;
    uint64_t p = (uint64_t)__popcll(x);
    while (count >= p)
    {
        count -= p; ++w; b = 0;
        if (w >= v.nwords) return false;
        x = invert ? ~v.w[w] : v.w[w];
        p = (uint64_t)__popcll(x);
    }
    *out = w * 64 + b + rd_select1(x, (uint32_t)count);
    return true;
}

__device__ inline uint64_t rd_u(const uint8_t* p, uint32_t bytes, uint64_t i)
{
    switch (bytes)
    {
        case 1: return p[i];
        case 2: return reinterpret_cast<const uint16_t*>(p)[i];
        case 4: return reinterpret_cast<const uint32_t*>(p)[i];
        default: return reinterpret_cast<const uint64_t*>(p)[i];
    }
}

// DenseSelect::select(i): the format (DenseArray.hh:98-169; written here by ds_classify_kernel /
// ds_fill_kernel) is a two-level directory.  Per block of 8192 indexed positions: a start position
// (rank array) and a 64-bit reference (index array); a reference, at either level, is a byte offset
// with the kind of what it points at in its low three bits.  Three shapes exist:
//   * "scan"      -- only sample positions are stored (one per 64 indexed positions); the answer is
//                    found by walking the bit vector from the sample;
//   * "explicit"  -- every position is stored, as 8/16/32-bit offsets from a base or as absolute u64;
//   * "two-level" -- per sample a 32-bit offset and a 16-bit reference to a scan or explicit table of
//                    its own.
// The reference walks this in DenseSelect::select + lookupSubBlock (DenseArray.cc:134-258).
struct RdRef { const uint8_t* at; uint32_t kind; };
__device__ inline RdRef rd_ref(const uint8_t* origin, uint64_t word)
{
    return RdRef{origin + (word & ~(uint64_t)kRdTypeMask), (uint32_t)(word & kRdTypeMask)};
}
// entry j of an explicit table, relative to `base` (absolute for 64-bit entries)
__device__ inline bool rd_explicit(const RdRef& t, uint64_t base, uint64_t j, uint64_t* out)
{
    switch (t.kind)
    {
        case kRdFull64: *out = reinterpret_cast<const uint64_t*>(t.at)[j]; return true;
        case kRdFull32: *out = base + reinterpret_cast<const uint32_t*>(t.at)[j]; return true;
        case kRdFull16: *out = base + reinterpret_cast<const uint16_t*>(t.at)[j]; return true;
        case kRdFull8:  *out = base + t.at[j]; return true;
        default: return false;
    }
}
__device__ inline bool rd_dense_select(const RdDenseSelect& d, const RdBits& bits, uint64_t i, uint64_t* out)
{
    const uint64_t blk = i >> d.logBlockSize;
    if (blk >= d.numBlocks) return false;
    const uint64_t in_blk = i & (d.blockSize - 1);                 // which of the block's positions
    const uint64_t sample = in_blk >> d.logSampleRate;             // which sample interval
    const uint64_t behind = in_blk & (d.sampleRate - 1);           // how far behind the sample
    const bool zeros = d.flags & 1;                                // the index is over the zero bits
    const uint64_t base = reinterpret_cast<const uint64_t*>(d.data + d.rankArrayOffset)[blk];
    const RdRef top = rd_ref(d.data, reinterpret_cast<const uint64_t*>(d.data + d.indexArrayOffset)[blk]);
    if (top.kind == kRdSmall)                                      // scan from a 16-bit sample offset
        return rd_bits_select(bits, zeros, base + reinterpret_cast<const uint16_t*>(top.at)[sample], behind, out);
    if (top.kind != kRdIntermediate) return rd_explicit(top, base, in_blk, out);
    // two-level: 32-bit sample offsets, then 16-bit references relative to the block
    const uint64_t nsamples = 1ULL << (d.logBlockSize - d.logSampleRate);
    const uint64_t from = base + reinterpret_cast<const uint32_t*>(top.at)[sample];
    const uint16_t word = reinterpret_cast<const uint16_t*>(top.at + 4 * nsamples)[sample];
    if (word == 0) return rd_bits_select(bits, zeros, from, behind, out);
    const RdRef sub = rd_ref(top.at, word);
    return sub.kind == kRdFull64 ? false : rd_explicit(sub, from, behind, out);
}

// low D bits of element i, from the IntegerArray column files (IntegerArray.cc:259-357)
template <class K> __device__ inline K rd_low(const RdSparse& s, uint64_t i);
template <> __device__ inline Key1 rd_low<Key1>(const RdSparse& s, uint64_t i)
{
    uint64_t v = 0;
    for (uint32_t c = 0; c < s.ncols; ++c)
        if (s.col_shift[c] < 64) v |= rd_u(s.col[c], s.col_bytes[c], i) << s.col_shift[c];
    return Key1{v};
}
template <> __device__ inline Key2 rd_low<Key2>(const RdSparse& s, uint64_t i)
{
    Key2 v{0, 0};
    for (uint32_t c = 0; c < s.ncols; ++c)
    {
        const uint64_t x = rd_u(s.col[c], s.col_bytes[c], i);
        const uint32_t sh = s.col_shift[c];
        if (sh < 64) { v.lo |= x << sh; if (sh) v.hi |= x >> (64 - sh); }
        else v.hi |= x << (sh - 64);
    }
    return v;
}

template <class K> __device__ inline K rd_make(uint64_t high, uint32_t D, const K& low);
template <> __device__ inline Key1 rd_make<Key1>(uint64_t high, uint32_t D, const Key1& low)
{
    return Key1{(D < 64 ? high << D : 0) | low.lo};
}
template <> __device__ inline Key2 rd_make<Key2>(uint64_t high, uint32_t D, const Key2& low)
{
    Key2 k = low;
    if (D < 64) { k.lo |= high << D; if (D) k.hi |= high >> (64 - D); }
    else if (D < 128) k.hi |= high << (D - 64);
    return k;
}

// SparseArray::select(rank) (SparseArray.hh:311-325)
template <class K>
__device__ inline bool rd_sparse_select(const RdSparse& s, uint64_t rnk, K* out)
{
    uint64_t p = 0;
    if (s.D < 128)
    {
        if (!rd_dense_select(s.d1, s.hi, rnk, &p)) return false;
        p -= rnk;
    }
    *out = rd_make<K>(p, (uint32_t)s.D, rd_low<K>(s, rnk));
    return true;
}

template <class K> __device__ inline uint64_t rd_high(const K& k, uint32_t D) { return D >= 128 ? 0 : key_shr64(k, D); }
template <class K> __device__ inline K rd_mask(const K& k, uint32_t D);
template <> __device__ inline Key1 rd_mask<Key1>(const Key1& k, uint32_t D) { return Key1{D >= 64 ? k.lo : (k.lo & ((1ULL << D) - 1))}; }
template <> __device__ inline Key2 rd_mask<Key2>(const Key2& k, uint32_t D)
{
    if (D >= 128) return k;
    if (D >= 64) return Key2{k.lo, D == 64 ? 0 : (k.hi & ((1ULL << (D - 64)) - 1))};
    return Key2{k.lo & ((1ULL << D) - 1), 0};
}

// SparseArray::findLowOrderGroup (SparseArray.hh:345-364): the index range of the elements whose
// high part is posD
template <class K>
__device__ inline bool rd_group(const RdSparse& s, uint64_t posD, uint64_t* b, uint64_t* e)
{
    if (s.D >= 128) { *b = 0; *e = s.count; return true; }
    uint64_t r2;
    if (!posD)
    {
        if (!rd_dense_select(s.d0, s.hi, 0, &r2)) return false;
        *b = 0; *e = r2;
        return true;
    }
    uint64_t r1;
    if (!rd_dense_select(s.d0, s.hi, posD - 1, &r1) || !rd_dense_select(s.d0, s.hi, posD, &r2)) return false;
    r1 += 1;
    *b = r1 >= posD ? r1 - posD : 0;
    *e = r2 >= posD ? r2 - posD : 0;
    return true;
}

// SparseArray::rank(pos) (SparseArray.hh:296-309) and access(pos) (SparseArray.hh:246-260)
template <class K>
__device__ inline bool rd_sparse_rank(const RdSparse& s, const K& pos, uint64_t* rank, bool* present)
{
    uint64_t b, e;
    if (!rd_group<K>(s, rd_high(pos, (uint32_t)s.D), &b, &e)) return false;
    if (e > s.count) e = s.count;
    const K want = rd_mask(pos, (uint32_t)s.D);
    const uint64_t end = e;
    while (b < e)
    {
        const uint64_t m = b + ((e - b) >> 1);
        if (rd_low<K>(s, m) < want) b = m + 1; else e = m;
    }
    *rank = b;
    *present = b < end && rd_low<K>(s, b) == want;
    return true;
}

// lint-graph pass 2 (GossCmdLintGraph.cc:201-243): the i-th element by select must be the i-th
// decoded element, and rank of that element must be i (and the element must be present).
struct IndexReport {
    unsigned long long select_mismatch, rank_mismatch, access_miss, failures;
    uint32_t nexamples, pad;
    unsigned long long ex_index[16];
    uint32_t ex_kind[16];
};

template <class K>
__global__ void check_index_kernel(RdSparse s, const K* __restrict__ keys, uint64_t m, IndexReport* __restrict__ rep)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t kind = 0;
    K got;
    if (!rd_sparse_select<K>(s, i, &got)) { atomicAdd(&rep->failures, 1ULL); kind = 4; }
    else if (got != keys[i]) { atomicAdd(&rep->select_mismatch, 1ULL); kind = 1; }
    uint64_t r; bool present;
    if (!rd_sparse_rank<K>(s, keys[i], &r, &present)) { atomicAdd(&rep->failures, 1ULL); kind = kind ? kind : 4; }
    else
    {
        if (r != i) { atomicAdd(&rep->rank_mismatch, 1ULL); kind = kind ? kind : 2; }
        if (!present) { atomicAdd(&rep->access_miss, 1ULL); kind = kind ? kind : 3; }
    }
    if (kind)
    {
        const uint32_t slot = atomicAdd(&rep->nexamples, 1u);
        if (slot < 16u) { rep->ex_index[slot] = i; rep->ex_kind[slot] = kind; }
    }
}

}  // namespace goss
