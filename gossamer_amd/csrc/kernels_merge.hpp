// kernels_merge.hpp -- merge of sorted (key,count) runs by segments.
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_count.hpp"

namespace goss {

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

// Graph mode counts ONE representative per strand pair (the windows' strand_rep: half the keys through the partition
// and the tables) although ReverseComplementAdapter (ReverseComplementAdapter.hh:34-55) yields both strands of every
// window.  Both strands of a window always come together, so the edge e and its reverse complement have the same
// multiplicity: count(e) = count(rc e) = windows(e) + windows(rc e) = the representative's count -- except a
// palindromic edge (e == rc e), which the adapter yields TWICE per window: twice the count.  This kernel makes the
// other half of the run: out[i] = (rc keys[i], counts[i]); a palindrome gets a pad there (all ones: sorts behind every
// key) and its own count doubled in place.  A doubled count that no longer fits 32 bits is kept exactly in `big`
// ({key lo, key hi, count} triples behind a counter) and the marker 0xFFFFFFFF stored; counts that ARE the marker
// already (exact value in the run's map) are left to the host, which mirrors / doubles the map's entries.
template <class K>
__global__ __launch_bounds__(kTB) void graph_expand_kernel(const K* __restrict__ keys, uint32_t* __restrict__ counts, uint64_t m, uint32_t len,
                                                          K* __restrict__ out_keys, uint32_t* __restrict__ out_counts,
                                                          unsigned long long* __restrict__ big, uint32_t big_cap, unsigned long long* __restrict__ npal)
{
    const uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= m) return;
    const K k = keys[i];
    const uint32_t c = counts[i];
    const K r = revcomp(k, len);
    if (r == k)
    {
        K pad;
        if constexpr (sizeof(K) == 8) pad = K{~0ULL}; else pad = K{~0ULL, ~0ULL};
        out_keys[i] = pad;
        out_counts[i] = 0;
        atomicAdd(npal, 1ULL);
        if (c != 0xFFFFFFFFu)
        {
            const unsigned long long d = 2ULL * c;
            if (d >= 0xFFFFFFFFULL)
            {
                const unsigned long long at = atomicAdd(&big[0], 1ULL);
                if (at < big_cap) { big[1 + 3 * at] = key_lo_word(k); big[2 + 3 * at] = key_hi_word(k); big[3 + 3 * at] = d; }
                counts[i] = 0xFFFFFFFFu;
            }
            else counts[i] = (uint32_t)d;
        }
    }
    else { out_keys[i] = r; out_counts[i] = c; }
}

}  // namespace goss
