// kernels_text.hpp -- text form of an object (dump-*) and the graph self-check (lint-graph).
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_common.hpp"

namespace goss {

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

}  // namespace goss
