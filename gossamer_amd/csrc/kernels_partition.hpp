// kernels_partition.hpp -- radix partition passes (histogram-table form, look-back form, sub-region second level).
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp, in this order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"
#include "kernels_common.hpp"

namespace goss {

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

// Histogram of the 16 (nparts = 2) or 17 (nparts = 4) bits at `shift` of every key (65 536 / 131 072 bins): the joint
// histogram of the two partition digits the fused path sizes its sub-regions from (exact when the sample is the
// chunk).  A workgroup holds 32 768 of the bins in LDS (128 KB, one workgroup per CU), so the bins are cut into nparts
// parts and the grid (a multiple of 8 * nparts workgroups) into groups of nparts workgroups that read the SAME keys,
// each counting its own part of the bins.  The members of a group are 8 apart in blockIdx -- one XCD -- and do the
// same amount of work, so they run side by side and the keys come from HBM once and from that XCD's L2 nparts - 1
// times (as nparts sweeps of every workgroup over all keys the four parts took 2.1 ms on C2's 196 M sample keys: 6.3 GB).
constexpr int kJointThreads = 1024;
template <class K>
__global__ __launch_bounds__(kJointThreads) void joint_hist_kernel(const K* __restrict__ keys, uint64_t n, uint32_t shift,
                                                                   unsigned long long* __restrict__ hist, uint32_t nparts)
{
    __shared__ uint32_t lh[32768];
    __shared__ uint32_t spare[64];               // one word per lane for the keys of other parts (ONE word for all of them is a 64-way conflict)
    const uint32_t mask = 32768u * nparts - 1u;
    const uint32_t part = (blockIdx.x >> 3) % nparts;
    const uint32_t stream = (blockIdx.x & 7u) + 8u * (blockIdx.x / (8u * nparts)), nstreams = gridDim.x / nparts;
    const uint64_t stride = (uint64_t)nstreams * kJointThreads;
    uint32_t* const mine = &spare[threadIdx.x & 63u];
    for (uint32_t i = threadIdx.x; i < 32768; i += kJointThreads) lh[i] = 0;
    __syncthreads();
    constexpr int kU = 8;                        // independent loads per thread in flight
    for (uint64_t i0 = (uint64_t)stream * kJointThreads + threadIdx.x; i0 < n; i0 += stride * kU)
    {
        uint32_t bb[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
            const uint64_t i = i0 + (uint64_t)u * stride;
            bb[u] = i < n ? ((uint32_t)key_shr64(keys[i], shift) & mask) : 0xFFFFFFFFu;          // (beyond the end: no bin)
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            // (a bin of another part: a word of the lane's own that nobody reads, instead of a branch around the atomic)
            atomicAdd((bb[u] >> 15) == part ? &lh[bb[u] & 32767u] : mine, 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < 32768; j += kJointThreads)
        if (lh[j]) atomicAdd(&hist[part * 32768u + j], (unsigned long long)lh[j]);
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

    // (the loads of a workgroup that has just started go ahead of the others' ranking and staging: subpart32_kernel)
    __builtin_amdgcn_s_setprio(3);
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
    __builtin_amdgcn_s_setprio(0);
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

// --------------------------------------------------------------------------------------
// Second level, 32-bit-remainder form (one-word keys whose bits below a 17-bit prefix fit 32)
// --------------------------------------------------------------------------------------
//
// What the second level writes and the counting kernel reads need not be the key: inside sub-region (b, d) the top
// bits are implied.  With NINE bits at the second level -- 256 x 512 = 131 072 sub-regions -- a k-mer of k = 25 has
// 33 bits left, and the strand representative of an odd-length k-mer has one bit that is always clear (bit len - 1,
// the low bit of its middle base: extract1_part_kernel picks the strand by it), which is squeezed out: 32 bits.
// This kernel reads 8-byte keys and writes 4-byte remainders; seg_hash_reduce32_kernel counts those: 12 + 4 bytes per
// key behind the first level instead of 16 + 8.  Applies while 2 len - 17 - (odd k-mer set ? 1 : 0) <= 32.
// Ten bits at the second level (2^18 sub-regions) where nine leave 33 bits and there is no bit to squeeze out (graphs of
// k = 24, k-mer sets of k = 26 would need twelve: not served).  More distinct keys than the tables of 2^17 / 2^18 segments
// hold (reads with sequencing errors: every error makes up to k new k-mers) do NOT get more second-level bits -- runs of
// a few bytes per sub-region made that pass 1.5 x (10 bits), 3 x (11) and 7 x (12) slower -- but a THIRD level inside
// every segment (subsplit32_kernel below), which is exact and cheap.
constexpr int kSub32BitsMin = 9, kSub32BitsMax = 10;
constexpr int kSub32SplitMax = 4;                                    // third-level bits at most: 16 sub-segments per segment
constexpr uint32_t kSub32RegionsMax = 256u << kSub32BitsMax;         // 2^18 sub-regions = second-level segments at most
struct SubTable32 {
    unsigned long long start[kSub32RegionsMax];  // first u32 slot of sub-region (b, d), index (b << bits) + d; a multiple of 4
    uint32_t cap[kSub32RegionsMax];
};

// low `rbits` bits of a key, bit `sqbit` (always clear) taken out when SQ
template <bool SQ>
__host__ __device__ __forceinline__ uint32_t rem32_pack(uint64_t key, uint32_t rbits, uint32_t sqbit)
{
    const uint64_t x = key & ((1ULL << rbits) - 1ULL);
    if (!SQ) return (uint32_t)x;
    return (uint32_t)(((x >> (sqbit + 1)) << sqbit) | (x & ((1ULL << sqbit) - 1ULL)));
}
template <bool SQ>
__host__ __device__ __forceinline__ uint64_t rem32_unpack(uint32_t r, uint32_t sqbit)
{
    if (!SQ) return r;
    return (((uint64_t)r >> sqbit) << (sqbit + 1)) | ((uint64_t)r & ((1ULL << sqbit) - 1ULL));
}

// Where the second level's tiles lie: tile t is the (t - tile_first[b])-th run of kTile slots of region b.  Worked out once
// per tile by this kernel (a binary search) so that a tile's workgroup learns its place with ONE scalar load
// instead of a search over the regions, a barrier and a second round of loads before its first key load is issued.
struct Tile32 { unsigned long long base; uint32_t n, bucket; };
#ifndef GOSS_S32_ITEMS
#define GOSS_S32_ITEMS 22
#endif
constexpr int kSub32Items = GOSS_S32_ITEMS;          // keys per thread of subpart32_kernel
constexpr int kSub32Tile = kTB * kSub32Items;
template <int TILE>
__global__ void tiles32_kernel(const GapTable* __restrict__ gt, Tile32* __restrict__ desc, uint32_t total)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint32_t lo = 0, hi = 255;                   // the last region whose first tile is <= t
    while (lo < hi)
    {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (gt->tile_first[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const uint64_t j = (uint64_t)t - gt->tile_first[lo];
    const uint64_t left = gt->cnt[lo] - j * TILE;
    desc[t] = Tile32{gt->reg_start[lo] + j * TILE, (uint32_t)(left < (uint64_t)TILE ? left : (uint64_t)TILE), lo};
}

// Tile = 5 632 keys of ONE first-level region; tiles dealt out by XCD (the ~44-byte runs a tile appends to a sub-region
// meet their neighbours in one L2).  Rank by one LDS atomic per key on 512 digit counters; a thread owns two
// neighbouring digits and takes the tile's room in both sub-regions with ONE 64-bit atomic on the pair of 32-bit
// cursors (a wave's 64 atomics cover 512 contiguous bytes); the REMAINDERS (4 bytes) and their digits (2 bytes) go to
// their digit-sorted place in LDS -- 40 KB per workgroup, four per CU where the 8-byte staging of the other form
// allows three; remainders out in coalesced runs.
#ifndef GOSS_S32_OCC
#define GOSS_S32_OCC 4
#endif
// NARROW (round 5): the regions hold what extract1_part_kernel<.., NARROW> wrote -- 16-byte chunks {remainder, remainder,
// remainder, D} with the three second-level digits at bits 0, 10, 20 of D and the number of keys the chunk holds at
// bits 30-31: a lane takes a chunk with one 16-byte load, three keys, and has nothing to pack (5.33 bytes read per
// key instead of 8).
// Chunks per thread of the narrow form: 10 at nine digit bits (30 keys a thread, 7 680 a tile, 50 KB of LDS: three
// workgroups per CU), 9 at ten bits (whose 1 024 counters and bases take 4 KB more).  Measured on C2 (round 5, tiles of
// 5 / 6 / 7 / 8 / 9 / 10 chunks on 5 / 5 / 4 / 3 / 3 / 3 workgroups per CU): 30.9 / 28.8 / 25.8 / 25.1 / 24.1 / 23.6 ms --
// longer runs per sub-region and fewer cursor atomics per key win over workgroups in flight.
#ifndef GOSS_S32_CH
#define GOSS_S32_CH 10
#define GOSS_S32_OCCN 3
#endif
template <int B2> struct Sub32N { static constexpr int kChunks = B2 == 9 ? GOSS_S32_CH : GOSS_S32_CH - 1; static constexpr int kTileSlots = kTB * kChunks * 2; };
template <bool SQ, int B2, bool NARROW = false>
__global__ __launch_bounds__(kTB, NARROW ? GOSS_S32_OCCN : GOSS_S32_OCC) void subpart32_kernel(const Key1* __restrict__ keys_in, uint32_t* __restrict__ out,
                                                           uint32_t rbits, uint32_t sqbit, unsigned long long* __restrict__ cursors,
                                                           const Tile32* __restrict__ desc, uint32_t total_tiles,
                                                           const SubTable32* __restrict__ sub, LookbackCtl* __restrict__ ctl)
{
    constexpr int kSub32ChunksN = Sub32N<B2>::kChunks;
    constexpr int kItems = NARROW ? 3 * kSub32ChunksN : kSub32Items;
    static_assert(NARROW || kItems <= 32, "one bit of `have` per key");
    constexpr int kTile = kTB * kItems;
    constexpr uint32_t ND = 1u << B2;                        // second-level digits
    constexpr int DPT = ND / kTB;                            // digits a thread owns: 2, 4, 8 or 16 neighbours
    static_assert(DPT >= 2 && DPT % 2 == 0, "a thread owns pairs of digits");
    __shared__ uint32_t stage[kTile];
    __shared__ uint16_t sdig[kTile];
    __shared__ uint32_t hist[ND];                            // keys per digit, then the digit's first slot in `stage`
    __shared__ uint32_t gbase[ND];                           // the digit's first slot in `out` minus its first slot in `stage`, relative to the region's first sub-region
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ uint32_t sh_skip, sh_total;

    const uint32_t tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t per = (total_tiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (tile >= total_tiles) return;
    const Tile32 td = desc[tile];                            // (uniform: one scalar load)
    const uint32_t b = td.bucket, tile_n = td.n;
    const uint64_t tile_base = td.base;
    if (tid == 0) sh_skip = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) hist[tid + j * kTB] = 0;

    // (a workgroup that has just started issues its key loads ahead of the others' ranking and staging: 30.85 -> 30.05 ms)
    __builtin_amdgcn_s_setprio(3);
    Key1 key[NARROW ? 1 : kItems];
    [[maybe_unused]] uint32_t rem[NARROW ? kItems : 1], dw[NARROW ? kSub32ChunksN : 1];
    uint16_t rank[kItems];
    uint32_t have = 0;
    const uint32_t wbase = w * 64 * kItems;
    if constexpr (NARROW)
    {
        const uint32_t nchunk = tile_n >> 1;                     // (the tile's 8-byte slots, two to a chunk)
        const uint4* const cin = reinterpret_cast<const uint4*>(keys_in + tile_base);
        const uint32_t cbase = w * 64 * kSub32ChunksN;
#pragma unroll
        for (int r = 0; r < kSub32ChunksN; ++r)
        {
            const uint32_t ci = cbase + r * 64 + lane;
            dw[r] = 0;
            rem[3 * r] = rem[3 * r + 1] = rem[3 * r + 2] = 0;
            if (ci < nchunk)
            {
                const uint4 v = cin[ci];
                rem[3 * r] = v.x; rem[3 * r + 1] = v.y; rem[3 * r + 2] = v.z; dw[r] = v.w;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        // (a second loop: looking at a chunk inside the load loop would wait for every load in turn)
        // (key r of the thread = field r % 3 of its chunk r / 3: a key iff the chunk holds more than r % 3 of them)
#pragma unroll
        for (int r = 0; r < kItems; ++r)
            if ((dw[r / 3] >> 30) > (uint32_t)(r % 3)) rank[r] = (uint16_t)atomicAdd(&hist[(dw[r / 3] >> (10 * (r % 3))) & (ND - 1u)], 1u);
    }
    else
    {
#if !defined(GOSS_S32_LOAD8)
    // (16 bytes per lane and load: two neighbouring keys -- a tile starts on a granule of the first level and holds an
    // even number of slots; which keys a thread takes does not matter, they rank themselves by atomics)
    static_assert(kItems % 2 == 0, "pairs of keys per load");
#pragma unroll
    for (int r = 0; r < kItems / 2; ++r)
    {
        const uint32_t li = wbase + r * 128 + 2 * lane;
        if (li < tile_n)
        {
#if defined(GOSS_S32_NT)
            typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ vv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(keys_in + tile_base + li));
            const uint4 v = make_uint4(vv.x, vv.y, vv.z, vv.w);
#else
            const uint4 v = *reinterpret_cast<const uint4*>(keys_in + tile_base + li);
#endif
            key[2 * r].lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
            key[2 * r + 1].lo = (uint64_t)v.z | ((uint64_t)v.w << 32);
            have |= 3u << (2 * r);
        }
    }
#else
#pragma unroll
    for (int r = 0; r < kItems; ++r)
    {
        const uint32_t li = wbase + r * 64 + lane;
        if (li < tile_n) { key[r] = keys_in[tile_base + li]; have |= 1u << r; }
    }
#endif
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    // (a second loop: looking at a key inside the load loop would wait for every load in turn)
#pragma unroll
    for (int r = 0; r < kItems; ++r)
        if (((have >> r) & 1u) && is_pad_key(key[r])) have &= ~(1u << r);
#pragma unroll
    for (int r = 0; r < kItems; ++r)
        if ((have >> r) & 1u) rank[r] = (uint16_t)atomicAdd(&hist[(uint32_t)(key[r].lo >> rbits) & (ND - 1u)], 1u);
    }
    __syncthreads();
    const uint64_t region_first = sub->start[b * ND];       // (uniform)
    {
        // thread tid owns digits DPT tid .. DPT tid + DPT - 1; a pair of neighbouring 32-bit cursors is one 64-bit atomic
        uint32_t cn[DPT], ex[DPT], mine = 0;
        const uint32_t sidx = b * ND + DPT * tid;
#pragma unroll
        for (int j = 0; j < DPT; ++j) { cn[j] = hist[DPT * tid + j]; mine += cn[j]; }
        bool over = false;
#pragma unroll
        for (int j = 0; j < DPT; j += 2)
        {
            unsigned long long old = 0;
            if (cn[j] | cn[j + 1]) old = atomicAdd(&cursors[(sidx + j) >> 1], (unsigned long long)cn[j] | ((unsigned long long)cn[j + 1] << 32));
            ex[j] = (uint32_t)old; ex[j + 1] = (uint32_t)(old >> 32);
            // too small a sub-region: nothing of this tile is stored, the host redoes the chunk
            over |= (cn[j] && ex[j] + cn[j] > sub->cap[sidx + j]) || (cn[j + 1] && ex[j + 1] + cn[j + 1] > sub->cap[sidx + j + 1]);
        }
        if (over) { atomicOr(&ctl->error, 2u); sh_skip = 1; }
        uint32_t tile_total;
        uint32_t start = block_excl_scan<uint32_t>(mine, sh_scan, &tile_total);
        // (a region's sub-regions span less than 2^32 slots: the host checks it)
#pragma unroll
        for (int j = 0; j < DPT; ++j)
        {
            hist[DPT * tid + j] = start;
            gbase[DPT * tid + j] = (mine ? (uint32_t)(sub->start[sidx + j] - region_first) : 0u) + ex[j] - start;
            start += cn[j];
        }
        if (tid == 0) sh_total = tile_total;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kItems; ++r)
        if (NARROW ? (dw[NARROW ? r / 3 : 0] >> 30) > (uint32_t)(r % 3) : ((have >> (r & 31)) & 1u) != 0)
        {
            if constexpr (NARROW)
            {
                const uint32_t d = (dw[r / 3] >> (10 * (r % 3))) & (ND - 1u);
                const uint32_t at = hist[d] + rank[r];
                stage[at] = rem[r];
                sdig[at] = (uint16_t)d;
            }
            else
            {
            const uint32_t d = (uint32_t)(key[r].lo >> rbits) & (ND - 1u);
            const uint32_t at = hist[d] + rank[r];
            stage[at] = rem32_pack<SQ>(key[r].lo, rbits, sqbit);
            sdig[at] = (uint16_t)d;
            }
        }
    __syncthreads();
    if (sh_skip) return;
    const uint32_t tile_keys = sh_total;
    uint32_t* const out_r = out + region_first;
    for (uint32_t i = tid; i < tile_keys; i += kTB)
    {
#if defined(GOSS_S32_EXP) && GOSS_S32_EXP == 1
        // (timing experiment: everything but the stores)
        asm volatile("" ::"v"(gbase[sdig[i]]), "v"(stage[i]));
#else
        out_r[(uint32_t)(gbase[sdig[i]] + i)] = stage[i];
#endif
    }
}

// Segment bounds of the 32-bit-remainder layout: segment s holds (the 32-bit half s of the cursor words) remainders from start[s].
__global__ void sub_bounds32_kernel(const SubTable32* __restrict__ sub, const uint32_t* __restrict__ cursors, uint32_t nseg,
                                    uint64_t* __restrict__ seg_beg, uint64_t* __restrict__ seg_end)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    seg_beg[s] = sub->start[s];
    seg_end[s] = sub->start[s] + cursors[s];
}

// Third level of the 32-bit-remainder form: every second-level segment (a contiguous run of remainders, ~96 K of them on
// C2) is split on the top b3 <= 4 bits of its remainders into 2^b3 sub-segments -- the segments the counting kernel
// then takes, when a second-level segment holds more distinct keys than an LDS table (reads with sequencing errors).
// One workgroup per segment, two passes over it: counts per wave and digit (every lane keeps its own counters in LDS:
// no atomics), then every wave appends its keys of digit d to ITS OWN share of sub-segment d (rank inside the wave by
// ballots over the digit's bits).  The output is a permutation of the segment at the same offsets of another buffer:
// exact sizes, no slack, no overflow.  A wave's 2^b3 append streams advance by a few keys per instruction and meet
// again in the L2 at once; 4 + 4 bytes read (the second time mostly from the Infinity Cache) and 4 written per key.
__global__ __launch_bounds__(kTB) void subsplit32_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         const uint64_t* __restrict__ seg_beg, const uint64_t* __restrict__ seg_end,
                                                         uint32_t rem_bits, uint32_t b3, uint64_t* __restrict__ sub_beg,
                                                         uint64_t* __restrict__ sub_end)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t cnt[kWaves][64][17];             // per lane: its count of every digit (17: rows on different banks)
    __shared__ uint32_t wcnt[kWaves][16];                // per wave and digit: count, then the wave's cursor inside the segment
    const uint32_t s = unit_block(), tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t nd = 1u << b3, dsh = rem_bits - b3;
    const uint64_t b = seg_beg[s], e = seg_end[s];
    if (b == e)
    {
        if (tid < nd) { sub_beg[(uint64_t)s * nd + tid] = b; sub_end[(uint64_t)s * nd + tid] = b; }
        return;
    }
    for (uint32_t j = 0; j < 17; ++j) cnt[w][lane][j] = 0;
    const uint64_t n = e - b;
    const uint64_t nvec = (n + 3) >> 2;
    const u32x4* const in4 = reinterpret_cast<const u32x4*>(in + b);          // (a second-level sub-region starts on a 16-byte boundary)
    // ---- pass 1: how many keys of every digit each wave will see ----
    for (uint64_t i = tid; i < nvec; i += kTB)
    {
        const u32x4 v = in4[i];
        const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * i + j < n) cnt[w][lane][(kk[j] >> dsh) & (nd - 1u)] += 1u;          // (the lane's own row: no atomic)
    }
    __syncthreads();
    if (tid < (uint32_t)kWaves * 16u)
    {
        const uint32_t ww = tid >> 4, d = tid & 15u;
        uint32_t sum = 0;
        for (uint32_t l = 0; l < 64; ++l) sum += cnt[ww][l][d];
        wcnt[ww][d] = d < nd ? sum : 0u;
    }
    __syncthreads();
    if (tid == 0)
    {
        // sub-segment d = the waves' shares one after the other; the sub-segments in digit order
        uint64_t at = 0;
        for (uint32_t d = 0; d < nd; ++d)
        {
            sub_beg[(uint64_t)s * nd + d] = b + at;
            for (uint32_t ww = 0; ww < (uint32_t)kWaves; ++ww) { const uint32_t c = wcnt[ww][d]; wcnt[ww][d] = (uint32_t)at; at += c; }
            sub_end[(uint64_t)s * nd + d] = b + at;
        }
    }
    __syncthreads();
    // ---- pass 2: the same keys in the same order, every wave appending to its shares ----
    uint32_t* const o = out + b;
    lds_vu32 cur = (lds_vu32)wcnt[w];
    const uint64_t lt_mask = (1ULL << lane) - 1ULL;
    for (uint64_t base = 0; base < nvec; base += kTB)
    {
        const uint64_t i = base + tid;
        const bool have = i < nvec;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (have) v = __builtin_nontemporal_load(&in4[i]);
        const uint32_t kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const bool live = have && 4 * i + j < n;
            const uint32_t d = (kk[j] >> dsh) & (nd - 1u);
            uint64_t peers = __ballot(live);
            for (uint32_t bit = 0; bit < b3; ++bit)
            {
                const bool one = (d >> bit) & 1u;
                const uint64_t m = __ballot(one);
                peers &= one ? m : ~m;
            }
            const uint32_t rank = __popcll(peers & lt_mask);
            uint32_t at = 0;
            if (live) at = cur[d];
            // (all reads of this round happen before the leader's update: one wave, in order)
            __builtin_amdgcn_wave_barrier();
            if (live)
            {
                o[at + rank] = kk[j];
                if (rank == 0) cur[d] = at + (uint32_t)__popcll(peers);
            }
            __builtin_amdgcn_wave_barrier();
        }
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

}  // namespace goss
