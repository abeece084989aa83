// kernels_route.hpp -- reads -> super-k-mer records, routed by minimizer (the exchange BEFORE counting of a
// multi-GPU build), and records -> keys.
//
// The reference has no distributed path; what these kernels must preserve is the key
// stream of its adapters: every valid window of every read exactly once (KmerizingAdapter.hh:20-86; for graphs
// every (k+1)-mer, which the counting side turns into both strands, ReverseComplementAdapter.hh:20-93), and
// that all copies of a key -- whatever strand they were read from -- are counted by ONE rank, so that the
// per-rank counts are final (position_type::normalize, RankSelect.hh:126-140, picks one strand per k-mer).
//
// Routing function: the MINIMIZER of a window = the smallest (under a multiplicative hash) canonical m-mer
// among its W = len - m + 1 m-mers, canonical = min(m-mer, reverse complement).  The set of canonical m-mers of
// a window and of its reverse complement are the same set, so both strands of a k-mer have the same minimizer:
// destination = the low 16 bits of the minimizer's hash, scaled to [0, nparts).  Consecutive windows of a read mostly share
// their minimizer (~(W + 1) / 2 windows in a row), so a run of windows with one destination travels as ONE
// record holding its bases once: 12 bytes for up to 16 windows instead of 8 bytes per window.
//
// Record (SkRec, 12 bytes): bits 0..91 = the run's (nwin + len - 1 <= 46) bases as 2-bit codes, base j of the
// run at bits [2j, 2j + 2) (the bit order of the extraction kernels' window registers); bits 92..95 = nwin - 1.
// A record never spans a non-base, a read boundary, or a routing tile (4 096 window starts).
// Part of the kernel set of libgossgpu.so (gfx950); included through goss_kernels.hpp.
#pragma once

#include "kernels_common.hpp"
#include "kernels_extract.hpp"

namespace goss {

#ifndef GOSS_ROUTE_OCC
#define GOSS_ROUTE_OCC 5          // workgroups per CU the routing kernel is compiled for (96 VGPRs, 25 KB of LDS): 19.2 ms against 20.3 with 4
#endif
#ifndef GOSS_ROUTE_BLOCK
#define GOSS_ROUTE_BLOCK 512      // record slots a routing workgroup takes ahead per part at 8 parts (fewer per part with more parts)
#endif
struct SkRec { uint32_t w0, w1, w2; };
static_assert(sizeof(SkRec) == 12, "records are 12 bytes");
// A PAD holds no window: words 0, 0, 1 << 27 -- one window (bits 92..95 zero) whose bases would end below bit 64, and a
// bit above them, which no record of windows has.  The routing kernel fills the unused ends of its blocks with them.
// (kSkPadWord2, rec_windows: kernels_extract.hpp)

constexpr int kRouteMaxParts = 256;
struct RouteCounters {
    unsigned long long records[kRouteMaxParts];   // records asked for by part (also past the capacity: the caller learns the need)
    unsigned long long windows[kRouteMaxParts];   // windows routed to part p
    unsigned long long overflow;                  // a part's buffer was too small: nothing of that tile was stored for it
};

// minimizer length m and positions per window W = len - m + 1 for a window of `len` bases: W from a small set
// (the kernel is instantiated per W), m at least 7 where the window allows it and at most 15 (30 bits)
__host__ __device__ inline uint32_t route_positions(uint32_t len)
{
    const uint32_t ws[5] = {17, 13, 9, 5, 1};
    for (int i = 0; i < 5; ++i)
        if (len >= ws[i] && len - ws[i] + 1 >= (len < 7 ? len : 7) && len - ws[i] + 1 <= 15) return ws[i];
    return 1;      // (len <= 15: the window is its own minimizer)
}

#ifndef GOSS_ROUTE_MUL24
#define GOSS_ROUTE_MUL24 1
#endif
// order of the canonical m-mers / scaling of a 16-bit hash to the parts
__device__ __forceinline__ uint32_t route_hash(uint32_t c, uint32_t m)
{
#if GOSS_ROUTE_MUL24
    // (the low 12 bases of a longer m-mer order it: m-mers that agree there tie, and ties go to the same part)
    (void)m;
    return __umul24(c, 0x3779B1u);
#else
    (void)m;
    return c * 0x9E3779B1u;
#endif
}
__device__ __forceinline__ uint32_t route_scale(uint32_t h16, uint32_t nparts)
{
#if GOSS_ROUTE_MUL24
    return __umul24(h16, nparts);
#else
    return h16 * nparts;
#endif
}

// reads (ASCII, any non-ACGT byte ends a run of windows) -> records appended to `nparts` buffers.
// out + part_first[p] = first record slot of part p, part_cap[p] its capacity.  Tile = 4096 window starts,
// 16 per thread; phase A (bytes -> 2-bit codes + non-base flags in LDS) is the extraction kernels'.
template <int W, bool PACKED = false>
__global__ __launch_bounds__(kTB, GOSS_ROUTE_OCC) void route_records_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                               uint64_t nstarts, uint64_t navail, uint32_t len, uint32_t maxwin,
                                                               uint32_t nparts, uint32_t block, SkRec* __restrict__ out,
                                                               const unsigned long long* __restrict__ part_first,
                                                               const unsigned long long* __restrict__ part_cap,
                                                               RouteCounters* __restrict__ rc, uint64_t ntiles,
                                                               const uint16_t* __restrict__ pbad = nullptr)
{
    constexpr int P = 16;
    constexpr int T = kTB * P;
    constexpr int NVEC = T / 16 + 4;
    constexpr int NPOS = P + W - 1;                 // m-mer positions a thread looks at
    __shared__ uint32_t pk[NVEC], iv[NVEC];
    __shared__ uint32_t rkbuf[T];                   // part << 16 | rank of every record inside its part's share of the tile (a record per window at worst)
    __shared__ uint32_t cnt[kRouteMaxParts], win[kRouteMaxParts];
    __shared__ unsigned long long gbase[kRouteMaxParts], gbase2[kRouteMaxParts];   // first slot (minus rank) of a tile's records: in the block's rest / in the new room
    __shared__ uint32_t grem[kRouteMaxParts];                                       // records of the tile that fit the block's rest
    __shared__ unsigned long long bpos[kRouteMaxParts];                             // the workgroup's block: next free slot of the part
    __shared__ uint32_t bleft[kRouteMaxParts], wintot[kRouteMaxParts];              // slots left in it | bit 31: it lies inside the part's buffer; the workgroup's windows so far
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ int sh_max[kWaves];
    __shared__ uint32_t lastd[kTB];                 // destination of the thread's last window, or ~0 when it is not valid
    __shared__ uint16_t vmask[kTB], bmask[kTB];     // valid windows / windows that start a run, per thread
    const uint32_t tid = threadIdx.x;
    if (tid < nparts) { bpos[tid] = 0; wintot[tid] = 0; bleft[tid] = 0; }
    const uint32_t m = len - W + 1;
    const uint32_t mmask = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    (void)maxwin;                                   // (records hold up to 16 windows in both modes)

#if defined(GOSS_STAMPS)
    // (timing build: cycles of wave 0 per phase, summed over its tiles, into the unused counters of parts 128..; with
    // five workgroups per CU a phase's cycles say what all twenty waves of the CU issued meanwhile, not what it waited for)
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define GOSS_RSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define GOSS_RSTAMP(i)
#endif
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
        const uint64_t tb = tile * (uint64_t)T;
#if defined(GOSS_STAMPS)
        st_acc[7] += 1;
#endif
        __builtin_amdgcn_s_setprio(3);              // (priorities by phase, as in extract1_part_kernel: loads first, stores next, the minimizers last)
        // ---- phase A: bytes of the tile -> codes + non-base flags ------------------------------------------
        for (uint32_t v = tid; v < (uint32_t)NVEC; v += kTB)
        {
            const uint64_t byte0 = tb + (uint64_t)v * 16;
            uint32_t codes, bads;
            load_group16<PACKED>(bases_aligned, pbad, byte0, navail + mis, codes, bads);
            if (byte0 < mis) bads |= (1u << (uint32_t)(mis - byte0 > 16 ? 16 : mis - byte0)) - 1u;
            pk[v] = codes; iv[v] = bads;
        }
        if (tid < nparts) { cnt[tid] = 0; win[tid] = 0; }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        GOSS_RSTAMP(0);

        // ---- phase B: this thread's 16 windows: validity, minimizers, destinations, runs --------------------
        const uint32_t q0 = tid * P + mis;
        const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
        const uint64_t p0 = tb + (uint64_t)tid * P;
        const uint64_t i0 = iv[v0], i1 = iv[v0 + 1], i2 = iv[v0 + 2], i3 = iv[v0 + 3];
        const uint64_t w0 = pk[v0], w1 = pk[v0 + 1], w2 = pk[v0 + 2], w3 = pk[v0 + 3];
        const uint64_t inv = (i0 | (i1 << 16) | (i2 << 32) | (i3 << 48)) >> sh;
        uint32_t valid;
        {
            uint64_t run = ~inv, acc = ~0ULL;
            uint32_t covered = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j)               // len <= 31
            {
                if ((len >> j) & 1u) { acc &= run >> covered; covered += 1u << j; }
                run &= run >> (1u << j);
            }
            const uint64_t left = nstarts > p0 ? nstarts - p0 : 0;
            valid = (uint32_t)acc & (left >= (uint64_t)P ? 0xFFFFu : ((1u << (uint32_t)left) - 1u));
        }
        const uint64_t lo = w0 | (w1 << 32), hi = w2 | (w3 << 32);
        const uint32_t s2 = 2 * sh;
        // base j of this thread at bits [2j, 2j + 2) of (blo, bhi): all 64 of them -- a record that starts at the
        // thread's last window and runs 16 windows into the next thread's needs up to 15 + 16 + len - 1 <= 61 bases,
        // so where the thread does not start on a vector boundary the fifth vector fills the top
        const uint64_t w4 = pk[v0 + 4];
        const uint64_t blo = s2 ? ((lo >> s2) | (hi << (64 - s2))) : lo;
        const uint64_t bhi = s2 ? ((hi >> s2) | (w4 << (64 - s2))) : hi;

        uint32_t nrec = 0, starts = 0, bnd = 0, last = 0xFFFFFFFFu;
        uint32_t dest[P];
#pragma unroll
        for (int i = 0; i < P; ++i) dest[i] = 0;
        if (valid)
        {
            // hashed canonical m-mers at positions 0 .. NPOS-1
            uint32_t val[NPOS];
            // forward m-mer: rolled (first base most significant); primed with the first m - 1 bases = their 2-bit
            // groups in reverse order.  Reverse complement at position pos = the complemented bases pos .. pos + m - 1
            // as they lie in the registers (first base least significant): no roll
            uint32_t fm;
            {
                const uint32_t x = (uint32_t)blo & (mmask >> 2);                     // (m - 1 <= 14 bases: all in blo)
                const uint32_t r = m > 1 ? __brev(x) >> (32 - 2 * (m - 1)) : 0u;
                fm = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
            }
            const uint32_t ms = 2 * (m - 1);
            const uint64_t clo = ms ? ((blo >> ms) | (bhi << (64 - ms))) : blo;
            const uint64_t chi = bhi >> ms;
            const uint32_t nw[4] = {~(uint32_t)blo, ~(uint32_t)(blo >> 32), ~(uint32_t)bhi, ~(uint32_t)(bhi >> 32)};
#pragma unroll
            for (int pos = 0; pos < NPOS; ++pos)
            {
                const uint32_t nb = (uint32_t)(pos < 32 ? (clo >> (2 * pos)) : (chi >> (2 * (pos - 32)))) & 3u;
                fm = ((fm << 2) | nb) & mmask;
                const int wi = (2 * pos) >> 5, bs = (2 * pos) & 31;
                const uint32_t rm = (bs ? __builtin_amdgcn_alignbit(nw[wi + 1 < 4 ? wi + 1 : 3], nw[wi], bs) : nw[wi]) & mmask;
                val[pos] = route_hash(fm < rm ? fm : rm, m);
            }
            // minimum over W positions for every window
            uint32_t mn[P];
            if constexpr (W >= P)
            {
                uint32_t suf[W];
                suf[W - 1] = val[W - 1];
#pragma unroll
                for (int j = W - 2; j >= 0; --j) suf[j] = min(val[j], suf[j + 1]);
                mn[0] = suf[0];
                uint32_t pre = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 1; i < P; ++i) { pre = min(pre, val[W - 1 + i]); mn[i] = min(suf[i], pre); }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < P; ++i)
                {
                    uint32_t x = val[i];
#pragma unroll
                    for (int j = 1; j < W; ++j) x = min(x, val[i + j]);
                    mn[i] = x;
                }
            }
            // destination: the low 16 bits of the minimizer's hash (the product's low half: a bijection of the m-mer's
            // last 8 bases, independent of the high bits that made it the minimum), scaled to the parts
#pragma unroll
            for (int i = 0; i < P; ++i) dest[i] = route_scale(mn[i] & 0xFFFFu, nparts) >> 16;
            // windows 1 .. 15 that start a run: valid, and the window before is not or goes elsewhere
            uint32_t neq = 0;
#pragma unroll
            for (int i = 1; i < P; ++i) neq |= (dest[i] != dest[i - 1] ? 1u : 0u) << i;
            bnd = valid & (~(valid << 1) | neq) & 0xFFFEu;
            if ((valid >> (P - 1)) & 1u) last = dest[P - 1];
        }
        // Runs are followed ACROSS threads (a record holds up to 16 windows of a run wherever its first lies): window 0
        // starts a run unless the thread before ends in a valid window with the same destination; a record starts at a
        // run's windows 0, 16, 32, ..; its length is bounded by the next run start or invalid window, which may lie
        // among the next thread's windows (a thread has the bases of 64 positions: enough for 16 windows from any of
        // its own).  Records never span tiles.
        lastd[tid] = last;
        vmask[tid] = (uint16_t)valid;
        GOSS_RSTAMP(1);
        __syncthreads();
        GOSS_RSTAMP(2);
        if (valid & 1u)
        {
            const uint32_t prev = tid ? lastd[tid - 1] : 0xFFFFFFFFu;
            if (prev != dest[0]) bnd |= 1u;
        }
        bmask[tid] = (uint16_t)bnd;
        // position (in the tile) of the last run start before this thread's windows, or -1
        const int mine = bnd ? (int)(tid * P + (31 - __clz(bnd))) : -1;
        int cur;
        {
            int inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1)
            {
                const int o = __shfl_up(inc, d, 64);
                if ((int)lane_id() >= d) inc = max(inc, o);
            }
            if (lane_id() == 63) sh_max[wave_id()] = inc;
            __syncthreads();
            int before = -1;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) if ((int)wave_id() > w) before = max(before, sh_max[w]);
            const int up = __shfl_up(inc, 1, 64);
            cur = max(before, lane_id() ? up : -1);
        }
        const uint32_t nb = tid + 1 < (uint32_t)kTB ? bmask[tid + 1] : 0xFFFFu;
        const uint32_t nv = tid + 1 < (uint32_t)kTB ? vmask[tid + 1] : 0u;
        // bit j set: window j (of this thread's 16 and the next thread's 16) starts a run or is not valid
        const uint32_t stop32 = (bnd | (~valid & 0xFFFFu)) | ((nb | (~nv & 0xFFFFu)) << 16);
        // records start at a run's windows 0, 16, 32, ..: the run starts among this thread's windows, and the one
        // window at that distance from the run that comes in from the threads before (cur), unless a run of this
        // thread starts at or before it
        {
            const uint32_t i0 = (uint32_t)cur & 15u;
            const uint32_t inc = (cur >= 0 && (bnd & ((2u << i0) - 1u)) == 0u) ? 1u << i0 : 0u;
            starts = valid & (bnd | inc);
        }
        nrec = __popc(starts);
        // the windows' destinations as bytes of two words: picked by shifting in the loops over the records
        uint64_t dp0 = 0, dp1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { dp0 |= (uint64_t)dest[i] << (8 * i); dp1 |= (uint64_t)dest[i + 8] << (8 * i); }
        uint32_t tot;
        const uint32_t at = block_excl_scan<uint32_t>(nrec, sh_scan, &tot);
        (void)tot;
        GOSS_RSTAMP(3);

        // ---- phase C: every record of this thread takes a rank inside its part's share of the tile ----------------
        {
            uint32_t todo = starts, r = 0;
            while (todo)
            {
                const uint32_t s = __ffs(todo) - 1;
                todo &= todo - 1;
                const uint32_t stop = (stop32 >> s) & ~1u;                         // bit j set: window s + j ends the run
                const uint32_t n = stop ? min(16u, (uint32_t)__ffs(stop) - 1u) : 16u;
                const uint32_t d = (uint32_t)((s < 8 ? dp0 >> (8 * s) : dp1 >> (8 * (s - 8))) & 0xFFu);
                // rank in the part's share of the tile (< 4 096) | windows - 1 << 12 | part << 16 | first window << 24
                rkbuf[at + r] = atomicAdd(&cnt[d], 1u) | ((n - 1) << 12) | (d << 16) | (s << 24);
                atomicAdd(&win[d], n);
                ++r;
            }
        }
        GOSS_RSTAMP(4);
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);
        // ---- phase D: room in every part's buffer, then the records leave ------------------------------------------
        // A workgroup takes room in blocks: what the tile needs beyond the rest of its block, and `block` slots for the
        // tiles to come (none after its last tile), with ONE atomic -- one returning atomic per part and tile on 2 nparts
        // addresses was a third of the kernel's time (13.5 ms against 19 per 40 M reads without them).  A tile's records
        // of a part are split between the old block's rest (ranks < rem) and the new room.  A block that does not fit the
        // part's buffer is walked through all the same (nothing stored), so that records[] ends as what a big enough
        // buffer would have taken, pads included: the caller learns the need.
        if (tid < nparts)
        {
            const uint32_t c = cnt[tid];
            unsigned long long g0 = ~0ULL, g1 = ~0ULL;
            uint32_t rem = 0;
            if (c)
            {
                unsigned long long pos = bpos[tid];
                uint32_t left = bleft[tid] & 0x7FFFFFFFu, ok = bleft[tid] >> 31;
                rem = c < left ? c : left;
                if (rem) { g0 = ok ? part_first[tid] + pos : ~0ULL; pos += rem; left -= rem; }
                if (c > rem)
                {
                    const uint32_t more = c - rem;
                    const uint32_t take = more + (tile + gridDim.x < ntiles ? block : 0u);
                    const unsigned long long g = atomicAdd(&rc->records[tid], (unsigned long long)take);
                    ok = g + take <= part_cap[tid] ? 1u : 0u;
                    if (!ok) atomicOr(&rc->overflow, 1ULL);
                    g1 = ok ? part_first[tid] + g - rem : ~0ULL;             // (rank r >= rem goes to slot g + r - rem)
                    pos = g + more; left = take - more;
                }
                bpos[tid] = pos; bleft[tid] = left | (ok << 31);
                wintot[tid] += win[tid];
            }
            gbase[tid] = g0; gbase2[tid] = g1; grem[tid] = rem;
        }
        __syncthreads();
        GOSS_RSTAMP(5);
        {
            const uint32_t b0 = (uint32_t)blo, b1 = (uint32_t)(blo >> 32), b2 = (uint32_t)bhi, b3 = (uint32_t)(bhi >> 32);
            for (uint32_t r = 0; r < nrec; ++r)
            {
                const uint32_t dr = rkbuf[at + r];
                const uint32_t rank = dr & 0xFFFu, n1 = (dr >> 12) & 15u, d = (dr >> 16) & 0xFFu, ss = (dr >> 24) * 2;
                // bases s .. s + n + len - 2 of this thread (2 s <= 30: three funnel shifts)
                const uint32_t f0 = __builtin_amdgcn_alignbit(b1, b0, ss), f1 = __builtin_amdgcn_alignbit(b2, b1, ss);
                const uint32_t f2 = __builtin_amdgcn_alignbit(b3, b2, ss);
                const uint32_t nb2 = 2 * (n1 + len);                               // <= 92
                const uint64_t flo = (uint64_t)f0 | ((uint64_t)f1 << 32);
                const uint64_t klo = nb2 >= 64 ? flo : (flo & ((1ULL << nb2) - 1ULL));
                const uint32_t khi = nb2 > 64 ? (f2 & ((1u << (nb2 - 64)) - 1u)) : 0u;
                const unsigned long long g = rank < grem[d] ? gbase[d] : gbase2[d];
                if (g != ~0ULL)
                {
                    SkRec rec{(uint32_t)klo, (uint32_t)(klo >> 32), khi | (n1 << 28)};
                    out[g + rank] = rec;
                }
            }
        }
        __syncthreads();
        GOSS_RSTAMP(6);
    }
#if defined(GOSS_STAMPS)
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&rc->records[128 + i], st_acc[i]);
#endif
#undef GOSS_RSTAMP
    // what is left of the workgroup's blocks is filled with pads (records of no window), and the windows are told
    for (uint32_t p = 0; p < nparts; ++p)
    {
        const uint32_t left = bleft[p] & 0x7FFFFFFFu;
        if (left && (bleft[p] >> 31))
        {
            SkRec* o = out + part_first[p] + bpos[p];
            for (uint32_t i = tid; i < left; i += kTB) o[i] = SkRec{0u, 0u, kSkPadWord2};
        }
    }
    if (tid < nparts && wintot[tid]) atomicAdd(&rc->windows[tid], (unsigned long long)wintot[tid]);
}

constexpr int kRecGroup = kTB;          // records per group of the plain record kernels

// --------------------------------------------------------------------------------------
// Two-word keys (32 <= len <= 63): 20-byte records
// --------------------------------------------------------------------------------------
//
// The same routing for windows of 32 .. 63 bases (k-mer sets of k >= 32, graphs of k >= 31: BASELINE's C4, every
// realistic build-graph).  The minimizer is taken over the CENTRAL c bases of a window, c = 31 for odd len and 30 for
// even: the central part of a window's reverse complement is the reverse complement of its central part (len - c is
// even), so both strands still see the same set of canonical m-mers and all copies of a key reach one part; and the
// central parts of consecutive windows are consecutive c-mers, so the run structure and the arithmetic (W = 17
// positions of m = c - 16 bases) are the one-word kernel's, on the bases shifted by o = (len - c) / 2.
//
// Record (SkRec2, 20 bytes): bits 0..155 = the run's (nwin + len - 1 <= 78) bases as 2-bit codes, base j at bits
// [2j, 2j + 2); bits 156..159 = nwin - 1.  A PAD is {0, 0, 0, 0, 1 << 27}: one window whose bases would end below bit
// 128 and a bit above them.
struct SkRec2 { uint32_t w[5]; };
static_assert(sizeof(SkRec2) == 20, "two-word records are 20 bytes");
constexpr uint32_t kSkPad2Word4 = 1u << 27;
__host__ __device__ inline uint32_t rec2_windows(uint32_t w4) { return (w4 >> 27) == 1u ? 0u : (w4 >> 28) + 1u; }
// central bases the minimizer of a long window is taken from, and where they start
__host__ __device__ inline uint32_t route_central(uint32_t len) { return len <= 31 ? len : ((len & 1u) ? 31u : 30u); }

template <int W, bool PACKED = false>
__global__ __launch_bounds__(kTB, 4) void route_records2_kernel(const uint8_t* __restrict__ bases_aligned, uint32_t mis,
                                                                uint64_t nstarts, uint64_t navail, uint32_t len,
                                                                uint32_t nparts, uint32_t block, SkRec2* __restrict__ out,
                                                                const unsigned long long* __restrict__ part_first,
                                                                const unsigned long long* __restrict__ part_cap,
                                                                RouteCounters* __restrict__ rc, uint64_t ntiles,
                                                                const uint16_t* __restrict__ pbad = nullptr)
{
    constexpr int P = 16;
    constexpr int T = kTB * P;
    constexpr int NVEC = T / 16 + 7;                // a thread reads the codes of 7 vectors: 96 positions behind any start
    constexpr int NPOS = P + W - 1;
    __shared__ uint32_t pk[NVEC], iv[NVEC];
    __shared__ uint32_t rkbuf[T];
    __shared__ uint32_t cnt[kRouteMaxParts], win[kRouteMaxParts];
    __shared__ unsigned long long gbase[kRouteMaxParts], gbase2[kRouteMaxParts];
    __shared__ uint32_t grem[kRouteMaxParts];
    __shared__ unsigned long long bpos[kRouteMaxParts];
    __shared__ uint32_t bleft[kRouteMaxParts], wintot[kRouteMaxParts];
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ int sh_max[kWaves];
    __shared__ uint32_t lastd[kTB];
    __shared__ uint16_t vmask[kTB], bmask[kTB];
    const uint32_t tid = threadIdx.x;
    if (tid < nparts) { bpos[tid] = 0; wintot[tid] = 0; bleft[tid] = 0; }
    const uint32_t c = route_central(len), o = (len - c) >> 1;
    const uint32_t m = c - W + 1;
    const uint32_t mmask = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);

    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
        const uint64_t tb = tile * (uint64_t)T;
        // ---- phase A: bytes of the tile -> codes + non-base flags ------------------------------------------
        for (uint32_t v = tid; v < (uint32_t)NVEC; v += kTB)
        {
            const uint64_t byte0 = tb + (uint64_t)v * 16;
            uint32_t codes, bads;
            load_group16<PACKED>(bases_aligned, pbad, byte0, navail + mis, codes, bads);
            if (byte0 < mis) bads |= (1u << (uint32_t)(mis - byte0 > 16 ? 16 : mis - byte0)) - 1u;
            pk[v] = codes; iv[v] = bads;
        }
        if (tid < nparts) { cnt[tid] = 0; win[tid] = 0; }
        __syncthreads();

        // ---- phase B: this thread's 16 windows: validity, minimizers of their central parts, destinations, runs ----
        const uint32_t q0 = tid * P + mis;
        const uint32_t v0 = q0 >> 4, sh = q0 & 15u;
        const uint64_t p0 = tb + (uint64_t)tid * P;
        // non-base flags of 96 positions from the thread's first, as 128 bits
        uint64_t ilo, ihi;
        {
            const uint64_t a = (uint64_t)iv[v0] | ((uint64_t)iv[v0 + 1] << 16) | ((uint64_t)iv[v0 + 2] << 32) | ((uint64_t)iv[v0 + 3] << 48);
            const uint64_t b = (uint64_t)iv[v0 + 4] | ((uint64_t)iv[v0 + 5] << 16) | ((uint64_t)iv[v0 + 6] << 32);
            ilo = sh ? ((a >> sh) | (b << (64 - sh))) : a;
            ihi = b >> sh;
        }
        uint32_t valid;
        {
            // window i is valid iff bits [i, i + len) of the flags are zero: runs of good bases of length 1, 2, 4, .. by
            // doubling, and the AND of the runs that make up len at their offsets (128-bit forms of the one-word kernel's)
            uint64_t rlo = ~ilo, rhi = ~ihi, alo = ~0ULL;
            uint32_t covered = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j)               // len <= 63
            {
                if ((len >> j) & 1u)
                {
                    alo &= covered ? ((rlo >> covered) | (rhi << (64 - covered))) : rlo;          // (covered < 64; only the low 16 bits are used)
                    covered += 1u << j;
                }
                const uint32_t s = 1u << j;
                const uint64_t nlo = (rlo >> s) | (rhi << (64 - s)), nhi = rhi >> s;
                rlo &= nlo; rhi &= nhi;
            }
            const uint64_t left = nstarts > p0 ? nstarts - p0 : 0;
            valid = (uint32_t)alo & (left >= (uint64_t)P ? 0xFFFFu : ((1u << (uint32_t)left) - 1u));
        }
        // base j of this thread at bits [2j, 2j + 2) of (b0, b1, b2): 96 of them -- a record that starts at the thread's
        // last window and holds 16 windows needs 15 + 16 + len - 1 <= 93
        uint64_t b0, b1, b2;
        {
            const uint64_t x0 = (uint64_t)pk[v0] | ((uint64_t)pk[v0 + 1] << 32), x1 = (uint64_t)pk[v0 + 2] | ((uint64_t)pk[v0 + 3] << 32);
            const uint64_t x2 = (uint64_t)pk[v0 + 4] | ((uint64_t)pk[v0 + 5] << 32), x3 = pk[v0 + 6];
            const uint32_t s2 = 2 * sh;
            b0 = s2 ? ((x0 >> s2) | (x1 << (64 - s2))) : x0;
            b1 = s2 ? ((x1 >> s2) | (x2 << (64 - s2))) : x1;
            b2 = s2 ? ((x2 >> s2) | (x3 << (64 - s2))) : x2;
        }
        // the central parts: bases o .. o + 63 of the thread (o <= 16: the minimizers of 16 windows end at base o + 45)
        const uint32_t o2 = 2 * o;
        const uint64_t blo = o2 ? ((b0 >> o2) | (b1 << (64 - o2))) : b0;
        const uint64_t bhi = o2 ? ((b1 >> o2) | (b2 << (64 - o2))) : b1;

        uint32_t nrec = 0, starts = 0, bnd = 0, last = 0xFFFFFFFFu;
        uint32_t dest[P];
#pragma unroll
        for (int i = 0; i < P; ++i) dest[i] = 0;
        if (valid)
        {
            uint32_t val[NPOS];
            uint32_t fm;
            {
                const uint32_t x = (uint32_t)blo & (mmask >> 2);
                const uint32_t r = m > 1 ? __brev(x) >> (32 - 2 * (m - 1)) : 0u;
                fm = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
            }
            const uint32_t ms = 2 * (m - 1);
            const uint64_t clo = ms ? ((blo >> ms) | (bhi << (64 - ms))) : blo;
            const uint64_t chi = bhi >> ms;
            const uint32_t nw[4] = {~(uint32_t)blo, ~(uint32_t)(blo >> 32), ~(uint32_t)bhi, ~(uint32_t)(bhi >> 32)};
#pragma unroll
            for (int pos = 0; pos < NPOS; ++pos)
            {
                const uint32_t nb = (uint32_t)(pos < 32 ? (clo >> (2 * pos)) : (chi >> (2 * (pos - 32)))) & 3u;
                fm = ((fm << 2) | nb) & mmask;
                const int wi = (2 * pos) >> 5, bs = (2 * pos) & 31;
                const uint32_t rm = (bs ? __builtin_amdgcn_alignbit(nw[wi + 1 < 4 ? wi + 1 : 3], nw[wi], bs) : nw[wi]) & mmask;
                val[pos] = route_hash(fm < rm ? fm : rm, m);
            }
            uint32_t mn[P];
            {
                uint32_t suf[W];
                suf[W - 1] = val[W - 1];
#pragma unroll
                for (int j = W - 2; j >= 0; --j) suf[j] = min(val[j], suf[j + 1]);
                mn[0] = suf[0];
                uint32_t pre = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 1; i < P; ++i) { pre = min(pre, val[W - 1 + i]); mn[i] = min(suf[i], pre); }
            }
#pragma unroll
            for (int i = 0; i < P; ++i) dest[i] = route_scale(mn[i] & 0xFFFFu, nparts) >> 16;
            uint32_t neq = 0;
#pragma unroll
            for (int i = 1; i < P; ++i) neq |= (dest[i] != dest[i - 1] ? 1u : 0u) << i;
            bnd = valid & (~(valid << 1) | neq) & 0xFFFEu;
            if ((valid >> (P - 1)) & 1u) last = dest[P - 1];
        }
        lastd[tid] = last;
        vmask[tid] = (uint16_t)valid;
        __syncthreads();
        if (valid & 1u)
        {
            const uint32_t prev = tid ? lastd[tid - 1] : 0xFFFFFFFFu;
            if (prev != dest[0]) bnd |= 1u;
        }
        bmask[tid] = (uint16_t)bnd;
        const int mine = bnd ? (int)(tid * P + (31 - __clz(bnd))) : -1;
        int cur;
        {
            int inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1)
            {
                const int up = __shfl_up(inc, d, 64);
                if ((int)lane_id() >= d) inc = max(inc, up);
            }
            if (lane_id() == 63) sh_max[wave_id()] = inc;
            __syncthreads();
            int before = -1;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) if ((int)wave_id() > w) before = max(before, sh_max[w]);
            const int up = __shfl_up(inc, 1, 64);
            cur = max(before, lane_id() ? up : -1);
        }
        const uint32_t nb = tid + 1 < (uint32_t)kTB ? bmask[tid + 1] : 0xFFFFu;
        const uint32_t nv = tid + 1 < (uint32_t)kTB ? vmask[tid + 1] : 0u;
        const uint32_t stop32 = (bnd | (~valid & 0xFFFFu)) | ((nb | (~nv & 0xFFFFu)) << 16);
        {
            const uint32_t i0 = (uint32_t)cur & 15u;
            const uint32_t inc = (cur >= 0 && (bnd & ((2u << i0) - 1u)) == 0u) ? 1u << i0 : 0u;
            starts = valid & (bnd | inc);
        }
        nrec = __popc(starts);
        uint64_t dp0 = 0, dp1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { dp0 |= (uint64_t)dest[i] << (8 * i); dp1 |= (uint64_t)dest[i + 8] << (8 * i); }
        uint32_t tot;
        const uint32_t at = block_excl_scan<uint32_t>(nrec, sh_scan, &tot);
        (void)tot;

        // ---- phase C: every record of this thread takes a rank inside its part's share of the tile ----------------
        {
            uint32_t todo = starts, r = 0;
            while (todo)
            {
                const uint32_t s = __ffs(todo) - 1;
                todo &= todo - 1;
                const uint32_t stop = (stop32 >> s) & ~1u;
                const uint32_t n = stop ? min(16u, (uint32_t)__ffs(stop) - 1u) : 16u;
                const uint32_t d = (uint32_t)((s < 8 ? dp0 >> (8 * s) : dp1 >> (8 * (s - 8))) & 0xFFu);
                rkbuf[at + r] = atomicAdd(&cnt[d], 1u) | ((n - 1) << 12) | (d << 16) | (s << 24);
                atomicAdd(&win[d], n);
                ++r;
            }
        }
        __syncthreads();
        // ---- phase D: room in every part's buffer (in blocks, as the one-word kernel), then the records leave --------
        if (tid < nparts)
        {
            const uint32_t cn = cnt[tid];
            unsigned long long g0 = ~0ULL, g1 = ~0ULL;
            uint32_t rem = 0;
            if (cn)
            {
                unsigned long long pos = bpos[tid];
                uint32_t left = bleft[tid] & 0x7FFFFFFFu, ok = bleft[tid] >> 31;
                rem = cn < left ? cn : left;
                if (rem) { g0 = ok ? part_first[tid] + pos : ~0ULL; pos += rem; left -= rem; }
                if (cn > rem)
                {
                    const uint32_t more = cn - rem;
                    const uint32_t take = more + (tile + gridDim.x < ntiles ? block : 0u);
                    const unsigned long long g = atomicAdd(&rc->records[tid], (unsigned long long)take);
                    ok = g + take <= part_cap[tid] ? 1u : 0u;
                    if (!ok) atomicOr(&rc->overflow, 1ULL);
                    g1 = ok ? part_first[tid] + g - rem : ~0ULL;
                    pos = g + more; left = take - more;
                }
                bpos[tid] = pos; bleft[tid] = left | (ok << 31);
                wintot[tid] += win[tid];
            }
            gbase[tid] = g0; gbase2[tid] = g1; grem[tid] = rem;
        }
        __syncthreads();
        {
            const uint32_t B[7] = {(uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32), (uint32_t)b2, (uint32_t)(b2 >> 32), 0u};
            for (uint32_t r = 0; r < nrec; ++r)
            {
                const uint32_t dr = rkbuf[at + r];
                const uint32_t rank = dr & 0xFFFu, n1 = (dr >> 12) & 15u, d = (dr >> 16) & 0xFFu, ss = (dr >> 24) * 2;
                // bases s .. s + n + len - 2 of this thread (2 s <= 30: five funnel shifts), cut at 2 (n + len - 1) bits
                const uint32_t nb2 = 2 * (n1 + len);                               // <= 156
                uint32_t f[5];
#pragma unroll
                for (int j = 0; j < 5; ++j)
                {
                    const uint32_t x = __builtin_amdgcn_alignbit(B[j + 1], B[j], ss);
                    const uint32_t lo_bit = 32u * j;
                    f[j] = nb2 >= lo_bit + 32u ? x : (nb2 > lo_bit ? (x & ((1u << (nb2 - lo_bit)) - 1u)) : 0u);
                }
                const unsigned long long g = rank < grem[d] ? gbase[d] : gbase2[d];
                if (g != ~0ULL)
                {
                    SkRec2 rec;
                    rec.w[0] = f[0]; rec.w[1] = f[1]; rec.w[2] = f[2]; rec.w[3] = f[3]; rec.w[4] = f[4] | (n1 << 28);
                    out[g + rank] = rec;
                }
            }
        }
        __syncthreads();
    }
    for (uint32_t p = 0; p < nparts; ++p)
    {
        const uint32_t left = bleft[p] & 0x7FFFFFFFu;
        if (left && (bleft[p] >> 31))
        {
            SkRec2* op = out + part_first[p] + bpos[p];
            SkRec2 pad;
            pad.w[0] = pad.w[1] = pad.w[2] = pad.w[3] = 0u; pad.w[4] = kSkPad2Word4;
            for (uint32_t i = tid; i < left; i += kTB) op[i] = pad;
        }
    }
    if (tid < nparts && wintot[tid]) atomicAdd(&rc->windows[tid], (unsigned long long)wintot[tid]);
}

// two-word records -> keys, densely: one thread per record (the plain kernel of a two-word record source).  MODE 0:
// gossamer's canonical form of every window (position_type::normalize, RankSelect.hh:126-140); MODE 1: every window's
// key and its reverse complement (ReverseComplementAdapter.hh:34-55).
template <int MODE, bool REP = false>
__global__ __launch_bounds__(kTB) void extract_records2_kernel(const SkRec2* __restrict__ recs, uint64_t nrecs, uint32_t len,
                                                               Key2* __restrict__ out, ExtractCounters* __restrict__ ctr, uint64_t ngroups,
                                                               uint64_t slice_groups = 0, uint64_t slice_stride = 0)
{
    // (REP, MODE 0: the strand representative of the fused pipeline's key space instead of the canonical form; sampling
    // mode as extract_records_kernel's: group g of slice s = g / slice_groups starts at record s * slice_stride + ...)
    constexpr int S = MODE == 1 ? 2 : 1;
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ unsigned long long sh_base;
    const uint32_t tid = threadIdx.x;
    const uint32_t bits = 2 * len;                                         // 64 .. 126
    const uint64_t mask_hi = bits == 128 ? ~0ULL : ((1ULL << (bits - 64)) - 1);
    const uint32_t top = bits - 2;
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        uint64_t r0 = g * (uint64_t)kRecGroup;
        if (slice_groups) r0 = (g / slice_groups) * slice_stride + (g % slice_groups) * (uint64_t)kRecGroup;
        const uint64_t ri = r0 + tid;
        uint32_t nw = 0;
        SkRec2 rec;
        rec.w[0] = rec.w[1] = rec.w[2] = rec.w[3] = rec.w[4] = 0;
        if (ri < nrecs) { rec = recs[ri]; nw = rec2_windows(rec.w[4]); }
        uint32_t tot;
        const uint32_t at = block_excl_scan<uint32_t>(nw * S, sh_scan, &tot);
        if (tid == 0) sh_base = tot ? atomicAdd(&ctr->keys_out, (unsigned long long)tot) : 0ULL;
        __syncthreads();
        const uint64_t ob = sh_base + at;
        const uint64_t w0 = (uint64_t)rec.w[0] | ((uint64_t)rec.w[1] << 32), w1 = (uint64_t)rec.w[2] | ((uint64_t)rec.w[3] << 32);
        const uint64_t w2 = rec.w[4] & 0x0FFFFFFFu;
        // forward key of window 0: the base-4 reversal of its field; reverse complement: the complemented field
        Key2 f, r;
        {
            const uint64_t elo = w0, ehi = w1 & mask_hi;
            const uint64_t rlo = rev64(ehi), rhi = rev64(elo);
            const uint32_t sft = 128 - bits;
            if (sft == 64) { f.lo = rhi; f.hi = 0; }
            else { f.lo = (rlo >> sft) | (rhi << (64 - sft)); f.hi = rhi >> sft; }
            r.lo = ~elo; r.hi = (~ehi) & mask_hi;
        }
        for (uint32_t i = 0; i < nw; ++i)
        {
            if (i)
            {
                const uint32_t pos = 2 * (i + len - 1);
                const uint64_t nb = (pos < 64 ? (w0 >> pos) : pos < 128 ? (w1 >> (pos - 64)) : (w2 >> (pos - 128))) & 3u;
                f.hi = ((f.hi << 2) | (f.lo >> 62)) & mask_hi;
                f.lo = (f.lo << 2) | nb;
                const uint64_t cb = nb ^ 3u;
                r.lo = (r.lo >> 2) | (r.hi << 62);
                r.hi = (r.hi >> 2) | (top >= 64 ? cb << (top - 64) : 0ULL);
                if (top < 64) r.lo |= cb << top;
            }
            if (MODE == 0) out[ob + i] = REP ? strand_rep2(f, r, len, (1ULL << len) - 1) : canonical(f, r);
            else { out[ob + 2 * i] = f; out[ob + 2 * i + 1] = r; }
        }
        if (tid == 0 && tot) atomicAdd(&ctr->windows, (unsigned long long)(tot / S));
        __syncthreads();
    }
}

// records -> keys, densely (the role of extract1_kernel for a record source): one thread per record, its
// keys written at the cursor position of its workgroup.  MODE 0: one key per window -- gossamer's canonical form
// (position_type::normalize), or the strand representative of the fused pipeline's key space when REP; MODE 1:
// every window's key and its reverse complement.  Sampling mode as extract1_kernel's: `nsuper` groups of
// kRecGroup records, group g of slice s = g / slice_groups starts at record s * slice_stride + (g % slice_groups) *
// kRecGroup.
template <int MODE, bool REP>
__global__ __launch_bounds__(kTB) void extract_records_kernel(const SkRec* __restrict__ recs, uint64_t nrecs, uint32_t len,
                                                              Key1* __restrict__ out, ExtractCounters* __restrict__ ctr,
                                                              uint64_t ngroups, uint64_t slice_groups, uint64_t slice_stride)
{
    constexpr int S = MODE == 1 ? 2 : 1;
    __shared__ uint32_t sh_scan[kWaves + 1];
    __shared__ unsigned long long sh_base;
    const uint32_t tid = threadIdx.x;
    const uint32_t bits = 2 * len;
    const uint64_t kmask = (1ULL << bits) - 1;
    const uint64_t lmask = (1ULL << len) - 1;
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        uint64_t r0 = g * (uint64_t)kRecGroup;
        if (slice_groups) r0 = (g / slice_groups) * slice_stride + (g % slice_groups) * (uint64_t)kRecGroup;
        const uint64_t ri = r0 + tid;
        uint32_t nw = 0;
        SkRec rec{0, 0, 0};
        if (ri < nrecs) { rec = recs[ri]; nw = rec_windows(rec.w2); }
        uint32_t tot;
        const uint32_t at = block_excl_scan<uint32_t>(nw * S, sh_scan, &tot);
        if (tid == 0) sh_base = tot ? atomicAdd(&ctr->keys_out, (unsigned long long)tot) : 0ULL;
        __syncthreads();
        const uint64_t ob = sh_base + at;
        const uint64_t blo = (uint64_t)rec.w0 | ((uint64_t)rec.w1 << 32), bhi = rec.w2 & 0x0FFFFFFFu;
        uint64_t f = rev64(blo & kmask) >> (64 - bits);
        uint64_t r = (~blo) & kmask;
        const uint32_t top = bits - 2;
        for (uint32_t i = 0; i < nw; ++i)
        {
            if (i)
            {
                const uint32_t pos = 2 * (i + len - 1);
                const uint32_t nb = (uint32_t)(pos < 64 ? (blo >> pos) : (bhi >> (pos - 64))) & 3u;
                f = ((f << 2) | nb) & kmask;
                r = (r >> 2) | ((uint64_t)(nb ^ 3u) << top);
            }
            const Key1 fk{f}, rck{r};
            if (MODE == 0)
            {
                if (REP) out[ob + i] = (len & 1u) ? (((f >> (len - 1)) & 1ULL) ? rck : fk) : strand_rep(fk, rck, len, lmask);
                else out[ob + i] = canonical(fk, rck);
            }
            else { out[ob + 2 * i] = fk; out[ob + 2 * i + 1] = rck; }
        }
        if (tid == 0 && tot) atomicAdd(&ctr->windows, (unsigned long long)(tot / S));
        __syncthreads();
    }
}

}  // namespace goss
