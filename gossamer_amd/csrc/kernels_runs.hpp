// kernels_runs.hpp -- run compaction, distinct-key estimate, bitmap helpers.
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_partition.hpp"

namespace goss {

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
        key[j] = K{};                  // (every element set on every path: a conditionally filled array of two-word keys lives in scratch memory)
        val[j] = 0;
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

}  // namespace goss
