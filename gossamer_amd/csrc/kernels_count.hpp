// kernels_count.hpp -- per-segment counting in LDS hash tables, ordering of (key,count) groups.
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_partition.hpp"

namespace goss {

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
#if defined(GOSS_STAMPS)
    unsigned long long stamps[8];  // (timing build) wave 0's cycles per phase of the counting kernel, summed over the segments; [7] = segments
#endif
};

// seg_cnt of a unit whose table overflowed (round 6): the unit is then counted by itself, by sort (count_overflowed_units in
// goss_gpu.hip) -- a few giant segments are what homopolymer stretches make of a real read set (every window that begins
// with nine T's lies in ONE 17-bit segment), and redoing the whole chunk with more bits does not split them.
constexpr unsigned long long kSegOverflowed = ~0ULL;

// the keys of an overflowed unit as full keys: 32-bit remainders (subpart32_kernel's output) / 12-byte records
template <bool SQ>
__global__ __launch_bounds__(kTB) void expand_rem32_kernel(const uint32_t* __restrict__ rems, uint64_t n, uint64_t prefix, uint32_t sqbit,
                                                           Key1* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) out[i].lo = prefix | rem32_unpack<SQ>(rems[i], sqbit);
}
__global__ __launch_bounds__(kTB) void expand_rem96_kernel(const Rem96* __restrict__ recs, uint64_t n, uint64_t seg, uint32_t rem_bits,
                                                           Key2* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    const Rem96 r = recs[i];
    const unsigned __int128 full = ((unsigned __int128)seg << rem_bits) | ((unsigned __int128)r.r2 << 64) | ((uint64_t)r.r0 | ((uint64_t)r.r1 << 32));
    out[i] = Key2{(uint64_t)full, (uint64_t)(full >> 64)};
}

// Where the overflowed units' keys lie in the sorted distinct keys of all of them: unit i = keys in [low_i, high_i)
// (last[i]: no upper bound) -> its place in the staging area and its number of distinct keys
template <class K>
__global__ __launch_bounds__(kTB) void unit_bounds_kernel(const K* __restrict__ keys, uint64_t m, const K* __restrict__ lows, const K* __restrict__ highs,
                                                          const uint8_t* __restrict__ last, const uint32_t* __restrict__ units, uint32_t nunits,
                                                          unsigned long long cursor, uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt)
{
    const uint32_t i = blockIdx.x * kTB + threadIdx.x;
    if (i >= nunits) return;
    auto lower = [&](const K& x) {
        uint64_t lo = 0, hi = m;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (keys[mid] < x) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const uint64_t p0 = lower(lows[i]);
    const uint64_t p1 = last[i] ? m : lower(highs[i]);
    seg_pos[units[i]] = cursor + p0;
    seg_cnt[units[i]] = p1 - p0;
}

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
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = kSegOverflowed; }
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

// --------------------------------------------------------------------------------------
// Counting of 32-bit remainders (subpart32_kernel's output): 131 072 segments, one workgroup each
// --------------------------------------------------------------------------------------
//
// A slot is two 32-bit words, remainder and count; count 0 = empty.  A bucket is two slots = 16 bytes, read with one
// LDS load.  A key has a home bucket and an independent second one (both from one 32-bit mix; second = home ^ an odd
// field of the mix, so never the home); behind the second the buckets that follow it.  A key is inserted with its first
// occurrence counted (64-bit CAS of {marker, 0} -> {rem, 1}); a hit adds 1 to the count word (32-bit LDS atomic).
//
// Every 32-bit pattern is a remainder, so "empty" cannot be a key value -- but it can be a value that never MATCHES:
// the empty slots of bucket b hold the key E_b whose home is b ^ 1 and whose second bucket is b ^ 2.  The fast path
// compares a key only with the slots of its own two buckets, so it can never take an empty slot for its key, and
// needs no look at the counts: four compares pick the address of the count word to bump (or a word of the lane's own
// behind the table, for a miss), one unconditional LDS add does the rest.  ~28 vector instructions per key where the
// 8-byte form takes 56 -- these kernels are bound by what they issue, not by what they read (profiles/r04).
// Misses wait in the lane's bit mask for the slow path, which looks at counts and starts at the home bucket.
// The table is a quarter (SLOTS = 2048: 16 KB) or half (4096) of the 8-byte form's, the keys half the bytes.
// Remainders are loaded four per lane (16 bytes); a sub-region starts on a 16-byte boundary and its capacity is a
// multiple of four, so the last vector may be read whole.
constexpr uint32_t kR32Mul = 0x9E3779B1u, kR32MulInv = 0x0E8B2F51u;          // kR32Mul * kR32MulInv = 1 mod 2^32
__host__ __device__ __forceinline__ uint32_t r32_mix(uint32_t k) { return (k ^ (k >> 15)) * kR32Mul; }
__host__ __device__ __forceinline__ uint32_t r32_unmix(uint32_t f) { const uint32_t y = f * kR32MulInv; return y ^ (y >> 15) ^ (y >> 30); }

#ifndef GOSS_R32_OCC
#define GOSS_R32_OCC 5
#endif
#ifndef GOSS_R32_G
#define GOSS_R32_G 4
#endif
template <int SLOTS, bool SQ>
__global__ __launch_bounds__(kTB, SLOTS == 2048 ? GOSS_R32_OCC : 4) void seg_hash_reduce32_kernel(const uint32_t* __restrict__ rems, const uint64_t* __restrict__ seg_off,
                                                                const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                Key1* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                uint32_t rbits, uint32_t sqbit, uint32_t split_bits)
{
    // split_bits: the segments are the sub-segments of subsplit32_kernel -- 2^split_bits per second-level segment, told
    // apart by the top split_bits bits of the remainder (which they keep) and starting anywhere, not on a 16-byte boundary
    constexpr int NT = kTB;
    constexpr int kLimit = SLOTS / 4 * 3;
    constexpr int BB = SLOTS == 4096 ? 11 : SLOTS == 2048 ? 10 : -1;       // log2(buckets)
    constexpr uint32_t NB = SLOTS / 2;
    static_assert(BB > 0, "table size");
    // the table, then a bucket per lane (64 x 16 bytes) for the accesses that must not land in the table
    __shared__ __attribute__((aligned(16))) unsigned long long tab[SLOTS + 128];
    __shared__ uint32_t ndist, ovf;
    __shared__ unsigned long long sh_base;
    const uint32_t s = unit_block(), tid = threadIdx.x;
    const uint64_t b = seg_off[s], e = seg_end[s];
    if (b == e)
    {
        if (tid == 0) { seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    if (e - b > 0xFFFFFFF0ULL)
    {
        if (tid == 0) { atomicOr(&so->overflow, 2u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) uint8_t* lds_bytes;
    typedef const __attribute__((address_space(3))) u32x4* lds_bucket_plain;
    typedef const volatile __attribute__((address_space(3))) u32x4* lds_bucket_ptr;
    typedef __attribute__((address_space(3))) uint32_t* lds_word;
    const lds_bytes tb = (lds_bytes)tab;
    // (f of the marker: home b ^ 1, odd field 3 -> second bucket b ^ 2)
    auto marker = [](uint32_t bkt) -> uint32_t { return r32_unmix(((bkt ^ 1u) << (32 - BB)) | (3u << (32 - 2 * BB))); };
    for (uint32_t i = tid; i < NB; i += NT)
    {
        const uint32_t m = marker(i);
        *(__attribute__((address_space(3))) u32x4*)(tb + 16 * i) = u32x4{m, 0u, m, 0u};
    }
    if (tid < 128) tab[SLOTS + tid] = 0;
    if (tid == 0) { ndist = 0; ovf = 0; }
    __syncthreads();

    lds_vu32 vovf = (lds_vu32)&ovf;
    // behind the table: a bucket (16 bytes) per lane that nobody else touches -- its second word takes the lane's misses
    const uint32_t own16 = 8u * (uint32_t)SLOTS + 16u * (tid & 63u);
    const uint32_t dummy = own16 + 4u;
    auto home_of = [](uint32_t f) -> uint32_t { return f >> (32 - BB); };
    auto second_of = [](uint32_t f, uint32_t h) -> uint32_t { return h ^ (((f >> (32 - 2 * BB)) & (NB - 1u)) | 1u); };

    constexpr int kVec = 4;                                  // 16-byte loads in flight per lane
    constexpr int kG = GOSS_R32_G;                           // keys whose buckets are read together
    // (vectors from the 16-byte boundary at or below the segment's start: the first `head` elements are not its own)
    const uint32_t head = (uint32_t)(b & 3ULL);
    const u32x4* const v4 = reinterpret_cast<const u32x4*>(rems + (b - head));
    const uint32_t n = (uint32_t)(e - b) + head;
    const uint32_t nvec = (n + 3u) >> 2, nfull = n >> 2;      // vectors, and vectors of four live remainders
    u32x4 nxt[kVec];
#pragma unroll
    for (int u = 0; u < kVec; ++u)
    {
        const uint32_t i = (uint32_t)u * NT + tid;
        nxt[u] = __builtin_nontemporal_load(&v4[i < nvec ? i : nvec - 1]);
    }
    for (uint32_t i0 = 0; i0 < nvec; i0 += (uint32_t)NT * kVec)
    {
        u32x4 cur[kVec];
#pragma unroll
        for (int u = 0; u < kVec; ++u) cur[u] = nxt[u];
        // software pipeline: the next batch's loads are in flight while this one is inserted
#pragma unroll
        for (int u = 0; u < kVec; ++u)
        {
            const uint32_t i = i0 + (uint32_t)(kVec + u) * NT + tid;
            nxt[u] = __builtin_nontemporal_load(&v4[i < nvec ? i : nvec - 1]);
        }
#if defined(GOSS_R32_EXP) && GOSS_R32_EXP == 1
        // (timing experiment: the loads alone)
#pragma unroll
        for (int u = 0; u < kVec; ++u) asm volatile("" ::"v"(cur[u].x), "v"(cur[u].y), "v"(cur[u].z), "v"(cur[u].w));
        continue;
#endif
        // every vector of the batch whole?  (all but a segment's last batch: no validity arithmetic in the fast path)
        const bool whole = i0 + (uint32_t)NT * kVec <= nfull && (i0 != 0 || head == 0);
        uint32_t pend = 0;                                   // bit 4 u + j: remainder j of vector u missed
#pragma unroll
        for (int u = 0; u < kVec; ++u)
        {
            const uint32_t kk[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
            uint32_t live = 0xFu;
            if (!whole)
            {
                const uint32_t i = i0 + (uint32_t)u * NT + tid;
                const uint32_t have = i < nvec ? (n - 4u * i >= 4u ? 4u : n - 4u * i) : 0u;
                live = (1u << have) - 1u;
                if (i == 0) live &= ~((1u << head) - 1u);
            }
            // kG keys at a time: their home buckets read and looked at; the second bucket only by the lanes whose key
            // was not at home (6 % at a load of 0.37): a 16-byte LDS read costs what its busiest bank takes, and with a
            // few lanes active that is one cycle per lane group instead of three -- the LDS pipe is this kernel's bound
#pragma unroll
            for (int g0 = 0; g0 < 4; g0 += kG)
            {
                uint32_t a1[kG], a2[kG], at[kG];
                u32x4 q[kG];
#pragma unroll
                for (int j = 0; j < kG; ++j)
                {
                    const uint32_t f = r32_mix(kk[g0 + j]);
                    const uint32_t h = home_of(f);
                    a1[j] = h << 4;
                    a2[j] = second_of(f, h) << 4;
                    q[j] = *(lds_bucket_plain)(tb + a1[j]);
                }
#pragma unroll
                for (int j = 0; j < kG; ++j)
                {
                    // the count word of the slot that holds the key, else the lane's own word
                    const uint32_t k1 = kk[g0 + j];
                    uint32_t t = dummy;
                    t = q[j].z == k1 ? a1[j] + 12u : t;
                    t = q[j].x == k1 ? a1[j] + 4u : t;
                    at[j] = t;
                }
                {
                    // (a lane whose key was at home reads a bucket of its own behind the table instead: 64 such reads
                    // are one conflict-free sweep, so the instruction costs what the few lanes that missed make it cost)
                    u32x4 q2[kG];
#pragma unroll
                    for (int j = 0; j < kG; ++j)
                    {
#if !defined(GOSS_R32_BOTH)
                        a2[j] = at[j] == dummy ? a2[j] : own16;
#endif
                        q2[j] = *(lds_bucket_plain)(tb + a2[j]);
                    }
#pragma unroll
                    for (int j = 0; j < kG; ++j)
                    {
                        const uint32_t k1 = kk[g0 + j];
                        uint32_t t = dummy;
                        t = q2[j].z == k1 ? a2[j] + 12u : t;
                        t = q2[j].x == k1 ? a2[j] + 4u : t;
                        at[j] = at[j] == dummy ? t : at[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < kG; ++j)
                {
                    uint32_t t = at[j];
                    uint32_t miss = t == dummy ? 1u : 0u;
                    if (!whole) { const uint32_t lv = (live >> (g0 + j)) & 1u; t = lv ? t : dummy; miss &= lv; }
                    __hip_atomic_fetch_add((lds_word)(tb + t), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pend |= miss << (4 * u + g0 + j);
                }
            }
        }
        // slow path: every lane walks its own queue of leftover keys, one probe per wave iteration
        if (__ballot(pend != 0))
        {
            uint32_t key = 0, bk = 0, st = 0;
            bool busy = false;
            for (;;)
            {
                if (!busy && pend)
                {
                    const uint32_t u = __ffs(pend) - 1;
                    pend &= pend - 1;
#pragma unroll
                    for (int uu = 0; uu < kVec; ++uu)
                        if ((u >> 2) == (uint32_t)uu)
                        {
                            const uint32_t j = u & 3u;
                            key = j == 0 ? cur[uu].x : j == 1 ? cur[uu].y : j == 2 ? cur[uu].z : cur[uu].w;
                        }
                    bk = home_of(r32_mix(key));
                    st = 0;
                    busy = true;
                }
                if (!__ballot(busy)) break;
                if (busy)
                {
                    const u32x4 q01 = *(lds_bucket_ptr)(tb + 16u * bk);
                    uint32_t hit = ~0u;                       // slot that holds the key
                    if (q01.y != 0u && q01.x == key) hit = 2 * bk;
                    else if (q01.w != 0u && q01.z == key) hit = 2 * bk + 1;
                    else if (q01.y == 0u || q01.w == 0u)
                    {
                        const uint32_t slot = 2 * bk + (q01.y == 0u ? 0u : 1u);
                        const unsigned long long old = atomicCAS(&tab[slot], (unsigned long long)marker(bk), (unsigned long long)key | (1ULL << 32));
                        if (old == (unsigned long long)marker(bk))
                        {
                            const uint32_t nd = atomicAdd(&ndist, 1u);
                            if (nd + 1 > (uint32_t)kLimit) *vovf = 1;
                            busy = false;                     // (counted by the insertion itself)
                        }
                        else if ((uint32_t)old == key && (old >> 32) != 0) hit = slot;
                        // else: somebody else took the slot; look at this bucket again
                    }
                    else if (st == 0) { bk = second_of(r32_mix(key), bk); st = 1; }
                    else bk = (bk + 1) & (NB - 1u);
                    if (hit != ~0u) { atomicAdd(reinterpret_cast<uint32_t*>(tab) + 2 * hit + 1, 1u); busy = false; }
                }
                if (*vovf) break;
            }
        }
        if (*vovf) break;
    }
    __syncthreads();
    if (ovf)
    {
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = kSegOverflowed; }
        return;
    }

    // Order the occupied slots on the remainder: every thread takes its slots into registers, a bucket sort on the top
    // bits of the remainder (rank inside the bin by an LDS atomic, scan of the bin sizes, scatter as remainder << 32 |
    // count so that a 64-bit compare orders by remainder), an insertion sort of every bin (1.5 entries on average at
    // half the limit); a bin of more than 24 entries (skewed low bits) -> bitonic sort of the compacted entries.
    constexpr int kPer = SLOTS / NT;
    constexpr int kBins = SLOTS / 4, kBinsPer = kBins / NT, kBinBits = BB - 1;
    __shared__ uint32_t bins[kBins];
    __shared__ uint32_t sh_scan2[NT / 64 + 1];
    __shared__ uint32_t big;
    unsigned long long ck[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) ck[j] = tab[tid * kPer + j];
    for (uint32_t i = tid; i < kBins; i += NT) bins[i] = 0;
    if (tid == 0) big = 0;
    __syncthreads();
    const uint32_t rem_bits = rbits - (SQ ? 1u : 0u);
    // (the top split_bits bits are the same for the whole sub-segment: the bins are cut below them)
    const uint32_t bsh = rem_bits > (uint32_t)kBinBits + split_bits ? rem_bits - split_bits - kBinBits : 0;
    uint32_t rnk[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if ((ck[j] >> 32) != 0) rnk[j] = atomicAdd(&bins[((uint32_t)ck[j] >> bsh) & (kBins - 1)], 1u);
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
        if ((ck[j] >> 32) != 0)
            tab[bins[((uint32_t)ck[j] >> bsh) & (kBins - 1)] + rnk[j]] = (ck[j] << 32) | (ck[j] >> 32);
    __syncthreads();
    if (!big)
    {
#pragma unroll
        for (int q = 0; q < kBinsPer; ++q)
            for (uint32_t i = 1; i < bn[q]; ++i)
            {
                const unsigned long long kk = tab[bs[q] + i];
                uint32_t j = i;
                while (j > 0 && tab[bs[q] + j - 1] > kk) { tab[bs[q] + j] = tab[bs[q] + j - 1]; --j; }
                tab[bs[q] + j] = kk;
            }
        __syncthreads();
    }
    else
    {
        uint32_t nsort = 512;
        while (nsort < tot_occ) nsort <<= 1;
        for (uint32_t i = tot_occ + tid; i < nsort; i += NT) tab[i] = ~0ULL;
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= nsort; k2 <<= 1)
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
            {
                for (uint32_t t = tid; t < nsort / 2; t += NT)
                {
                    const uint32_t i = 2 * t - (t & (j - 1));
                    const uint32_t p = i + j;
                    const bool up = (i & k2) == 0;
                    const unsigned long long a = tab[i], c2 = tab[p];
                    if ((a > c2) == up) { tab[i] = c2; tab[p] = a; }
                }
                __syncthreads();
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
    const uint64_t prefix = (uint64_t)(s >> split_bits) << rbits;
    for (uint32_t i = tid; i < d; i += NT)
    {
        const unsigned long long v = tab[i];
        stage_keys[ob + i].lo = prefix | rem32_unpack<SQ>((uint32_t)(v >> 32), sqbit);
        stage_counts[ob + i] = (uint32_t)v;
    }
}

// Which of a bucket's four slots holds the key: the byte offset of the slot (0, 4, 8, 12), or 16 for none.  Four compares
// into four scalar masks, then four selects: written out because the compiler runs every compare and its select through
// vcc back to back, with an `s_nop 1` between them for the two wait states a mask needs before a vector instruction may
// read it -- 16 issue slots where these 8 instructions hide each other's.
__device__ __forceinline__ uint32_t r32b_slot_offset(const uint32_t __attribute__((ext_vector_type(4)))& q, uint32_t k)
{
    uint32_t off;
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_eq_u32_e64 %1, %5, %9\n\t"
        "v_cmp_eq_u32_e64 %2, %6, %9\n\t"
        "v_cmp_eq_u32_e64 %3, %7, %9\n\t"
        "v_cmp_eq_u32_e64 %4, %8, %9\n\t"
        "v_cndmask_b32_e64 %0, 16, 12, %1\n\t"
        "v_cndmask_b32_e64 %0, %0, 8, %2\n\t"
        "v_cndmask_b32_e64 %0, %0, 4, %3\n\t"
        "v_cndmask_b32_e64 %0, %0, 0, %4"
        : "=&v"(off), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(q.w), "v"(q.z), "v"(q.y), "v"(q.x), "v"(k));
    return off;
}

// t[j] = off[j] >= 16 ? dummy : t[j] for four keys: the compares first, then the selects (see above)
__device__ __forceinline__ void r32b_pick4(uint32_t (&t)[4], const uint32_t (&off)[4], uint32_t dummy)
{
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_lt_u32_e64 %4, 15, %8\n\t"
        "v_cmp_lt_u32_e64 %5, 15, %9\n\t"
        "v_cmp_lt_u32_e64 %6, 15, %10\n\t"
        "v_cmp_lt_u32_e64 %7, 15, %11\n\t"
        "v_cndmask_b32_e64 %0, %0, %12, %4\n\t"
        "v_cndmask_b32_e64 %1, %1, %12, %5\n\t"
        "v_cndmask_b32_e64 %2, %2, %12, %6\n\t"
        "v_cndmask_b32_e64 %3, %3, %12, %7"
        : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(dummy));
}

// The same counting with what round 5's look at the instruction stream took out (profiles/r05/SUMMARY.md): the kernel
// above issues 51 vector instructions and three LDS operations per key -- two 16-byte bucket reads and the add -- for a
// table whose LDS pipe is 70-85 % busy.  Here a bucket is FOUR remainders = 16 bytes of a key array, with the four
// counts at the same offset of a count array: one 16-byte read shows a key four candidates where the pair layout
// shows two, so at the same table size far fewer keys live outside their home bucket (load 0.19: 0.13 % of them
// against 1.8 %) and the fast path looks at the home bucket ONLY: mix, one read, four compare-selects for the offset of
// the count word, one add.  A key that is not at home (that per-mille, and every first occurrence) looks at its second
// bucket under a wave-level branch and otherwise waits for the slow path, which walks home, second, second + 1, ...
// Empty slots of bucket b hold a key that never looks at b in either of those two places (home b ^ 1, second b ^ 2), so
// neither needs a look at the counts; the slow path's chain can reach b with that very key, and skips the bucket.
// A slot is claimed by a 32-bit CAS on its key word (marker -> key); counts are only ever added to.
#ifndef GOSS_R32B_VEC
#define GOSS_R32B_VEC 4
#endif
// Tables of 8 192 and 16 384 slots (reads with errors: segments of up to 4 600 / 9 200 distinct keys counted without a
// third partition level, which costs a pass over all keys) run with 512 / 1 024 threads, two / one workgroups per CU:
// the same 16 waves per CU and the same 16 slots per thread in the ordering as the 4 096-slot table's.
template <int SLOTS> struct R32bCfg { static constexpr int kThreads = SLOTS <= 4096 ? kTB : SLOTS == 8192 ? 2 * kTB : 4 * kTB;
                                      static constexpr int kOcc = SLOTS == 2048 ? 5 : SLOTS == 4096 ? 4 : SLOTS == 8192 ? 2 : 1; };
template <int SLOTS, bool SQ>
__global__ __launch_bounds__(R32bCfg<SLOTS>::kThreads, R32bCfg<SLOTS>::kOcc) void seg_hash_reduce32b_kernel(const uint32_t* __restrict__ rems, const uint64_t* __restrict__ seg_off,
                                                                const uint64_t* __restrict__ seg_end, SegOut* __restrict__ so,
                                                                uint64_t* __restrict__ seg_pos, uint64_t* __restrict__ seg_cnt,
                                                                Key1* __restrict__ stage_keys, uint32_t* __restrict__ stage_counts,
                                                                uint32_t rbits, uint32_t sqbit, uint32_t split_bits)
{
    constexpr int NT = R32bCfg<SLOTS>::kThreads;
    constexpr int kLimit = SLOTS / 4 * 3;
    constexpr int BB = SLOTS == 16384 ? 12 : SLOTS == 8192 ? 11 : SLOTS == 4096 ? 10 : SLOTS == 2048 ? 9 : -1;       // log2(buckets of four slots)
    constexpr uint32_t NB = SLOTS / 4;
    constexpr uint32_t kCnt = 4u * SLOTS;                                 // byte offset of the count array
    static_assert(BB > 0, "table size");
    // keys, counts, then a word per lane (64) for the adds of keys that missed; the ordering below reuses all of it as pairs
    __shared__ __attribute__((aligned(16))) unsigned long long tab[SLOTS + 32];
    __shared__ uint32_t ndist, ovf;
    __shared__ unsigned long long sh_base;
    const uint32_t s = unit_block(), tid = threadIdx.x;
#if defined(GOSS_STAMPS)
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define GOSS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define GOSS_STAMP(i)
#endif
    const uint64_t b = seg_off[s], e = seg_end[s];
    if (b == e)
    {
        if (tid == 0) { seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    if (e - b > 0xFFFFFFF0ULL)
    {
        if (tid == 0) { atomicOr(&so->overflow, 2u); seg_pos[s] = 0; seg_cnt[s] = 0; }
        return;
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) uint8_t* lds_bytes;
    typedef const __attribute__((address_space(3))) u32x4* lds_bucket_plain;
    typedef const volatile __attribute__((address_space(3))) u32x4* lds_bucket_ptr;
    typedef __attribute__((address_space(3))) uint32_t* lds_word;
    const lds_bytes tb = (lds_bytes)tab;
    uint32_t* const tkey = reinterpret_cast<uint32_t*>(tab);
    uint32_t* const tcnt = tkey + SLOTS;
    // (f of the marker: home b ^ 1, odd field 3 -> second bucket b ^ 2)
    auto marker = [](uint32_t bkt) -> uint32_t { return r32_unmix(((bkt ^ 1u) << (32 - BB)) | (3u << (32 - 2 * BB))); };
    for (uint32_t i = tid; i < NB; i += NT)
    {
        const uint32_t m = marker(i);
        *(__attribute__((address_space(3))) u32x4*)(tb + 16 * i) = u32x4{m, m, m, m};
        *(__attribute__((address_space(3))) u32x4*)(tb + kCnt + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    }
    if (tid < 32) tab[SLOTS + tid] = 0;
    if (tid == 0) { ndist = 0; ovf = 0; }
    __syncthreads();
    GOSS_STAMP(0);

    lds_vu32 vovf = (lds_vu32)&ovf;
    const uint32_t dummy = 8u * (uint32_t)SLOTS + 4u * (tid & 63u);          // the lane's own word behind the table
    auto home_of = [](uint32_t f) -> uint32_t { return f >> (32 - BB); };
    auto second_of = [](uint32_t f, uint32_t h) -> uint32_t { return h ^ (((f >> (32 - 2 * BB)) & (NB - 1u)) | 1u); };

    constexpr int kVec = GOSS_R32B_VEC;                      // 16-byte loads in flight per lane
    const uint32_t head = (uint32_t)(b & 3ULL);
    const u32x4* const v4 = reinterpret_cast<const u32x4*>(rems + (b - head));
    const uint32_t n = (uint32_t)(e - b) + head;
    const uint32_t nvec = (n + 3u) >> 2, nfull = n >> 2;      // vectors, and vectors of four live remainders
    u32x4 nxt[kVec];
#pragma unroll
    for (int u = 0; u < kVec; ++u)
    {
        const uint32_t i = (uint32_t)u * NT + tid;
        nxt[u] = __builtin_nontemporal_load(&v4[i < nvec ? i : nvec - 1]);
    }
    for (uint32_t i0 = 0; i0 < nvec; i0 += (uint32_t)NT * kVec)
    {
        u32x4 cur[kVec];
#pragma unroll
        for (int u = 0; u < kVec; ++u) cur[u] = nxt[u];
        // software pipeline: the next batch's loads are in flight while this one is inserted
#pragma unroll
        for (int u = 0; u < kVec; ++u)
        {
            const uint32_t i = i0 + (uint32_t)(kVec + u) * NT + tid;
            nxt[u] = __builtin_nontemporal_load(&v4[i < nvec ? i : nvec - 1]);
        }
#if defined(GOSS_R32_EXP) && GOSS_R32_EXP == 1
        // (timing experiment: the loads alone)
#pragma unroll
        for (int u = 0; u < kVec; ++u) asm volatile("" ::"v"(cur[u].x), "v"(cur[u].y), "v"(cur[u].z), "v"(cur[u].w));
        continue;
#endif
        GOSS_STAMP(1);
        // four keys at a time: their home buckets read together and looked at; a key that is not at home looks at its
        // second bucket under a wave-level branch of its own (a per-mille of the keys once the table is filled: the branch
        // is scalar work unless some lane needs it); not there either -> a bit of the result.  `live`: which of the four are
        // the segment's
        auto four_keys = [&](const u32x4& v, uint32_t live, auto whole_tag) -> uint32_t {
            constexpr bool kWhole = decltype(whole_tag)::value;
            const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
            uint32_t f[4], a1[4], off[4];
            u32x4 q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                f[j] = r32_mix(kk[j]);
                a1[j] = home_of(f[j]) << 4;
                q[j] = *(lds_bucket_plain)(tb + a1[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) off[j] = r32b_slot_offset(q[j], kk[j]);
            uint32_t t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                t[j] = kCnt + a1[j] + off[j];
                bool miss = off[j] == 16u;
                if (!kWhole) miss = miss && ((live >> j) & 1u);
                if (__builtin_expect(__ballot(miss) != 0, 0))
                {
                    if (miss)
                    {
                        const uint32_t a2 = second_of(f[j], a1[j] >> 4) << 4;
                        const u32x4 q2 = *(lds_bucket_plain)(tb + a2);
                        off[j] = r32b_slot_offset(q2, kk[j]);
                        t[j] = kCnt + a2 + off[j];
                    }
                }
            }
            // (off = 16: not found -- the add goes to the lane's own word and the key is left to the slow path)
            if (!kWhole)
            {
#pragma unroll
                for (int j = 0; j < 4; ++j) off[j] = ((live >> j) & 1u) ? off[j] : 32u;          // (not the segment's: no add, no wait either)
            }
            r32b_pick4(t, off, dummy);
            uint32_t left = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                __hip_atomic_fetch_add((lds_word)(tb + t[j]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                left |= (off[j] & 16u) << j;          // (bits 4 .. 7)
            }
            return left >> 4;
        };
        // Slow path, per vector (what it inserts is there for the next vector's fast path: of a key's copies in a batch only
        // those of one vector walk through here): every lane takes its leftover keys in turn, one probe per wave iteration
        // -- home, second, then the buckets behind the second.  A bucket is looked at with the same two compare chains as
        // in the fast path (the key, and the bucket's marker = an empty slot); a slot is claimed by a CAS on its key word.
        // Nothing in an iteration waits for more than that read and that CAS: the number of distinct keys is only added
        // to (and looked at once per batch), and a walk that has passed every bucket ends the segment (table full).
        auto slow_keys = [&](const u32x4& v, uint32_t pd) {
            uint32_t key = 0, bk = 0, f = 0, st = 0, moves = 0;
            bool busy = false;
            for (;;)
            {
                if (!busy && pd)
                {
                    const uint32_t j = __ffs(pd) - 1;
                    pd &= pd - 1;
                    key = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
                    f = r32_mix(key);
                    bk = home_of(f);
                    st = 0;
                    busy = true;
                }
                if (!__ballot(busy)) break;
                if (busy)
                {
                    const u32x4 q = *(lds_bucket_ptr)(tb + 16u * bk);
                    const uint32_t mk = marker(bk);
                    const u32x4 qq = u32x4{q.x, q.y, q.z, q.w};
                    const uint32_t ho = r32b_slot_offset(qq, key);           // where the key is
                    const uint32_t eo = r32b_slot_offset(qq, mk);            // the first empty slot
                    uint32_t hit = ~0u;                                      // slot that holds the key
                    // (key == mk: this bucket's empty slots look like the key -- it lives elsewhere)
                    if (key != mk && ho < 16u) hit = 4 * bk + (ho >> 2);
                    else if (key != mk && eo < 16u)
                    {
                        const uint32_t slot = 4 * bk + (eo >> 2);
                        const uint32_t old = atomicCAS(&tkey[slot], mk, key);
                        if (old == mk) { atomicAdd(&ndist, 1u); hit = slot; }
                        else if (old == key) hit = slot;
                        // else: somebody else took the slot; look at this bucket again
                    }
                    else
                    {
                        if (st == 0) { bk = second_of(f, bk); st = 1; }
                        else bk = (bk + 1) & (NB - 1u);
                        if (++moves > NB) { *vovf = 1; busy = false; pd = 0; }
                    }
                    if (hit != ~0u) { atomicAdd(&tcnt[hit], 1u); busy = false; }
                }
            }
        };
        // every vector of the batch whole?  (all but a segment's last batch: no validity arithmetic in that form)
        const bool whole = i0 + (uint32_t)NT * kVec <= nfull && (i0 != 0 || head == 0);
        if (whole)
        {
#pragma unroll
            for (int u = 0; u < kVec; ++u)
            {
                const uint32_t left = four_keys(cur[u], 0xFu, std::true_type{});
                if (__builtin_expect(__ballot(left != 0) != 0, 0)) slow_keys(cur[u], left);
            }
        }
        else
        {
#pragma unroll
            for (int u = 0; u < kVec; ++u)
            {
                const uint32_t i = i0 + (uint32_t)u * NT + tid;
                const uint32_t have = i < nvec ? (n - 4u * i >= 4u ? 4u : n - 4u * i) : 0u;
                uint32_t live = (1u << have) - 1u;
                if (i == 0) live &= ~((1u << head) - 1u);
                const uint32_t left = four_keys(cur[u], live, std::false_type{});
                if (__ballot(left != 0)) slow_keys(cur[u], left);
            }
        }
        GOSS_STAMP(2);
        // (once per batch: more distinct keys than the table is meant to hold, or a walk that found it full)
        if (*(lds_vu32)&ndist > (uint32_t)kLimit) *vovf = 1;
        if (*vovf) break;
        GOSS_STAMP(3);
    }
    __syncthreads();
    GOSS_STAMP(4);
    if (ovf)
    {
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = kSegOverflowed; }
        return;
    }

    // Order the occupied slots on the remainder (as above): bucket sort on the top bits, insertion sort of every bin
    constexpr int kPer = SLOTS / NT;
    constexpr int kBins = SLOTS / 4, kBinsPer = kBins / NT, kBinBits = BB;
    __shared__ uint32_t bins[kBins];
    __shared__ uint32_t sh_scan2[NT / 64 + 1];
    __shared__ uint32_t big;
    unsigned long long ck[kPer];              // count << 32 | remainder
#pragma unroll
    for (int j = 0; j < kPer; ++j) ck[j] = ((unsigned long long)tcnt[tid * kPer + j] << 32) | tkey[tid * kPer + j];
    for (uint32_t i = tid; i < kBins; i += NT) bins[i] = 0;
    if (tid == 0) big = 0;
    __syncthreads();
    const uint32_t rem_bits = rbits - (SQ ? 1u : 0u);
    // (the top split_bits bits are the same for the whole sub-segment: the bins are cut below them)
    const uint32_t bsh = rem_bits > (uint32_t)kBinBits + split_bits ? rem_bits - split_bits - kBinBits : 0;
    uint32_t rnk[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if ((ck[j] >> 32) != 0) rnk[j] = atomicAdd(&bins[((uint32_t)ck[j] >> bsh) & (kBins - 1)], 1u);
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
        if ((ck[j] >> 32) != 0)
            tab[bins[((uint32_t)ck[j] >> bsh) & (kBins - 1)] + rnk[j]] = (ck[j] << 32) | (ck[j] >> 32);
    __syncthreads();
    if (!big)
    {
#pragma unroll
        for (int q = 0; q < kBinsPer; ++q)
            for (uint32_t i = 1; i < bn[q]; ++i)
            {
                const unsigned long long kk = tab[bs[q] + i];
                uint32_t j = i;
                while (j > 0 && tab[bs[q] + j - 1] > kk) { tab[bs[q] + j] = tab[bs[q] + j - 1]; --j; }
                tab[bs[q] + j] = kk;
            }
        __syncthreads();
    }
    else
    {
        uint32_t nsort = 512;
        while (nsort < tot_occ) nsort <<= 1;
        for (uint32_t i = tot_occ + tid; i < nsort; i += NT) tab[i] = ~0ULL;
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= nsort; k2 <<= 1)
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
            {
                for (uint32_t t = tid; t < nsort / 2; t += NT)
                {
                    const uint32_t i = 2 * t - (t & (j - 1));
                    const uint32_t p = i + j;
                    const bool up = (i & k2) == 0;
                    const unsigned long long a = tab[i], c2 = tab[p];
                    if ((a > c2) == up) { tab[i] = c2; tab[p] = a; }
                }
                __syncthreads();
            }
    }
    GOSS_STAMP(5);
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
    const uint64_t prefix = (uint64_t)(s >> split_bits) << rbits;
    for (uint32_t i = tid; i < d; i += NT)
    {
        const unsigned long long v = tab[i];
        stage_keys[ob + i].lo = prefix | rem32_unpack<SQ>((uint32_t)(v >> 32), sqbit);
        stage_counts[ob + i] = (uint32_t)v;
    }
    GOSS_STAMP(6);
#if defined(GOSS_STAMPS)
    if (tid == 0)
    {
        for (int i = 0; i < 7; ++i) atomicAdd(&so->stamps[i], st_acc[i]);
        atomicAdd(&so->stamps[7], 1ULL);
    }
#endif
#undef GOSS_STAMP
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
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = kSegOverflowed; }
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
            if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = kSegOverflowed; }
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
#ifndef GOSS_SEG96_PROBE
#define GOSS_SEG96_PROBE 3          // (C4: 204 ms with 3, 209 with 2, 211 with 4)
#endif
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
    // the wave's queue of keys still to be inserted (see below), and the insertion of what it holds
    constexpr uint32_t kQueue = 64;
    __shared__ uint32_t wqueue[NT / 64][kQueue * 4];
    lds_vu32 wq = (lds_vu32)wqueue[tid >> 6];
    uint32_t qn = 0;                          // entries waiting (the same value in every lane of the wave)
    auto drain = [&]() {
        for (uint32_t base = 0; base < qn; base += 64)
        {
            const uint32_t idx = base + (tid & 63u);
            bool have = idx < qn;
            uint32_t k0 = 0, k1 = 0, k2 = 0, kw = 1;
            if (have) { k0 = wq[4 * idx]; k1 = wq[4 * idx + 1]; k2 = wq[4 * idx + 2]; kw = wq[4 * idx + 3]; }
            const uint32_t fh = k0 ^ __builtin_rotateleft32(k1, 15) ^ __builtin_rotateleft32(k2, 7);
            uint32_t slot = (fh * 0x9E3779B1u) >> (32 - kSlotBits);
            for (;;)
            {
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
        qn = 0;
    };
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
            constexpr int kProbe = GOSS_SEG96_PROBE;          // slots read behind the home slot (the rest waits in the wave's queue)
            static_assert(kU % kQ == 0, "pairs");
#pragma unroll
            for (int h = 0; h < kU / kQ; ++h)
            {
                uint4 f[kQ][kProbe];
#pragma unroll
                for (int j = 0; j < kQ; ++j)
#pragma unroll
                    for (int q = 0; q < kProbe; ++q) f[j][q] = tbl4(tbl, (slots[h * kQ + j] + q) & (SLOTS - 1));
#pragma unroll
                for (int j = 0; j < kQ; ++j)
                {
                    const int u = h * kQ + j;
                    // (integer arithmetic, no && chains: the compiler turns those into a branch per term, and
                    // the scalar instructions of exec-mask bookkeeping then outnumber the vector ones)
                    uint32_t off = 4;
#pragma unroll
                    for (int q = kProbe - 1; q >= 0; --q)
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
        // What the fast path did not find waits in a queue of its WAVE in LDS (key remainder + weight) and is inserted
        // when more than a wave's worth has gathered, one entry per lane: in steady state ~1 % of the keys come here,
        // and a state-machine loop run by every wave for its few lanes cost more scalar than vector instructions
        // (73.5e9 against 50.4e9 on C4, profiles/r02); gathered, the loop runs with all lanes busy.  Nothing is lost
        // by waiting: a key that is not yet in the table merely sends its later copies here as well.
        if (__ballot(pend != 0))
        {
#pragma unroll
            for (int u = 0; u < kU; ++u)
            {
                const bool mine = (pend >> u) & 1u;
                const unsigned long long mm = __ballot(mine);
                if (mm)
                {
                    const uint32_t nm = (uint32_t)__popcll(mm);
                    if (qn + nm > kQueue) drain();                  // (room for a wave's worth)
                    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                    if (mine) { const uint32_t at = 4 * (qn + before); wq[at] = r0[u]; wq[at + 1] = r1[u]; wq[at + 2] = r2[u]; wq[at + 3] = wt[u]; }
                    qn += nm;
                }
            }
        }
        if (*vovf) break;
    }
    drain();
    if (*vovf) break;
    }   // runs
    __syncthreads();
    if (ovf)
    {
        if (tid == 0) { atomicOr(&so->overflow, 1u); seg_pos[s] = 0; seg_cnt[s] = kSegOverflowed; }
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
    // (ties: only between a slot whose 96 remainder bits are all ones -- a key that ends in 48 T's -- and the pads of the
    // bitonic branch below, which carry the same words and the count 0: the key goes first.  Without that the network
    // left them in any order and the cut at `d` entries could hand out a pad -- the key with the count 0 -- in the key's
    // place: found by tests/fuzz_parity.py in round 6 on reads with poly-A stretches)
    auto rem_less = [](const uint4& a, const uint4& b2) {
        return a.z < b2.z || (a.z == b2.z && (a.y < b2.y || (a.y == b2.y && (a.x < b2.x || (a.x == b2.x && a.w != 0u && b2.w == 0u)))));
    };
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

}  // namespace goss
