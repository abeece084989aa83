// goss_gpu.hip -- implementation of the C ABI in include/goss_gpu.h on top of the HIP
// kernels in goss_kernels.hpp.  gfx950 only; no CPU fallback: every entry point that needs
// the device fails with GOSS_ERR_NO_DEVICE / GOSS_ERR_HIP when it is not usable.
#include "../../include/goss_gpu.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "goss_kernels.hpp"
#include "goss_reader.hpp"

using namespace goss;

namespace {

constexpr uint32_t kAbiVersion = 1;

struct HipError { hipError_t e; const char* what; };

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) throw HipError{_e, #expr};                      \
    } while (0)

struct StatusError { int status; std::string msg; };

// Two-ended bump arena over one hipMalloc: permanent blocks grow from the bottom, temporary
// blocks from the top (released by mark).
struct Arena {
    uint8_t* base = nullptr;
    uint64_t size = 0;
    uint64_t lo = 0;        // permanent top
    uint64_t hi = 0;        // temporary bottom (offset from base)

    void* perm(uint64_t bytes)
    {
        uint64_t a = (lo + 255) & ~255ULL;
        if (a + bytes > hi) throw StatusError{GOSS_ERR_OOM, "HBM budget exceeded (permanent)"};
        lo = a + bytes;
        return base + a;
    }
    void* temp(uint64_t bytes)
    {
        if (bytes + 256 > hi) throw StatusError{GOSS_ERR_OOM, "HBM budget exceeded (temporary)"};
        uint64_t a = (hi - bytes) & ~255ULL;
        if (a < lo) throw StatusError{GOSS_ERR_OOM, "HBM budget exceeded (temporary)"};
        hi = a;
        return base + a;
    }
    uint64_t mark() const { return hi; }
    void release(uint64_t m) { hi = m; }
    uint64_t avail() const { return hi > lo ? hi - lo : 0; }
};

struct OutFile {
    std::string suffix;
    uint64_t size = 0;
    const uint8_t* dev = nullptr;      // device-resident image, or
    std::vector<uint8_t> host;         // host-built bytes
};

typedef std::map<std::pair<uint64_t, uint64_t>, uint64_t> BigMap;     // key (hi, lo) -> count that does not fit 32 bits
struct Run { void* keys; uint32_t* counts; uint64_t m; int big = -1; bool rep = false; };
// big: index into ctx.big_maps, or -1.  rep: the keys are strand representatives (extract1_part_kernel), not yet
// gossamer's canonical forms -- runs of ONE space are merged as they are, and the result is canonicalised once

struct PhaseEvents { hipEvent_t a, b; int phase; uint64_t units; };

}  // namespace

struct goss_gpu_ctx {
    int device = 0;
    uint32_t k = 0, len = 0;
    int mode = 0;
    int words = 1;
    int path = 0;                       // 0 auto, 1 LSD sort only
    bool lookback = true;               // single-pass radix scatter (GOSS_GPU_NO_LOOKBACK=1 disables)
    double est_scale = 1.0;             // GOSS_GPU_EST_SCALE: factor on the distinct-key estimate of the fused path (tests)
    uint32_t order_bits = 0;            // group bits of the canonical re-ordering (GOSS_GPU_ORDER_BITS=16|20; 0 = by size)
    bool ordered_tiles = false;         // take tile numbers from a ticket instead of blockIdx
    uint32_t lookback_failures = 0;
    uint64_t emit_estimate = 0;         // SparseArray estimate M for the next emit (merges)
    bool has_emit_estimate = false;
    struct Pending { hipEvent_t ev; void (*fn)(void*); void* user; };
    std::vector<Pending> pending;       // asynchronous host pushes whose buffers the caller has not got back yet
    uint64_t flush_wait_us = 0, flush_count_us = 0, flushes = 0;   // staging buffer counted: waiting for queued copies / counting (host wall)
    // Host pushes are staged in one of TWO buffers (device memory of their own, beside the arena) by copies on a stream of
    // their own; a full buffer is counted by a thread of the library while the caller goes on filling the other
    // A packed staging buffer (round 5): what goss_gpu_push_packed_host* hands over stays 2-bit codes + flags in HBM --
    // u32 of codes per sixteen positions from the buffer's start, u16 of flags per sixteen positions behind them -- and the
    // kernels that read bases take it as it is (load_group16, kernels_common.hpp).  A buffer holds bytes OR packed
    // groups (stage_pk[i]); `pk` describes the packed string a count is running on: the plumbing between the entry
    // points and the kernels goes on passing byte addresses -- made-up ones (pk.fake + position) that only the launch
    // sites turn into the two real pointers (pk_ptrs).
    struct PackedSrc { bool on = false; const uint32_t* codes = nullptr; const uint16_t* bad = nullptr; } pk;
    bool stage_pk[2] = {false, false};
    uint8_t* stage_buf[2] = {nullptr, nullptr};
    int stage_cur = 0;
    hipStream_t copy_stream = nullptr;
    std::thread bg;                     // counts a full staging buffer (one at a time)
    int bg_status = GOSS_OK;
    std::string bg_error;
    std::vector<hipEvent_t> pend_pool;  // events of `pending` (the counting thread has event_pool to itself)
    uint8_t* stage = nullptr;           // the staging buffer being filled (= stage_buf[stage_cur])
    uint64_t stage_cap = 0, stage_fill = 0;
    bool cursor_pass0 = true;           // GOSS_GPU_NO_CURSOR_PASS0=1: look-back chain in every pass
    bool mute_timing = false;           // set around auxiliary launches (the distinct-count estimate)
    uint32_t segment_retries = 0;       // segment path attempts that overflowed an LDS table
    uint32_t extract_hist_shift = 0xFFFFFFFFu;   // digits histogrammed by the last extraction (or none)
    bool extract_rep = false;           // the next one-word k-mer extraction stores strand representatives (fused path's sample)
    uint32_t rep_chunks = 0;            // chunks counted in strand-representative space and mapped to canonical order afterwards
    bool ef_by_words = false;           // GOSS_GPU_EF_BY_WORDS=1: the high-bits bitmap by a binary search per word (round 1's kernel)
    bool graph_rep = true;              // GOSS_GPU_NO_GRAPH_REP=1: the fused path of a graph build counts both strands of every window (round 3's form)
    int canon_l1 = 1;                   // GOSS_GPU_CANON_L1=0|1|2: the fused first level computes gossamer's canonical form itself never / from canon_l1_at distinct keys per window on / always
    // (0.05: re-ordering M distinct pairs into the canonical order costs 53 ms per 10^9 of them, two FNV hashes per window
    // 34 ms per 12.6 G windows -- even at M = 0.051 x windows, and the counting of canonical keys is the faster above that
    // (C2 with 0.3 % / 0.5 % errors: 160 -> 148 / 216 -> 170 ms); 0.10 until the end of round 5)
    double canon_l1_at = 0.05;          // GOSS_GPU_CANON_L1_AT=<fraction>
    // more of the same input is known to follow the push being counted (a staging buffer that filled up while the caller
    // keeps pushing): its runs will be merged with the runs of the others
    // in representative space and the re-ordering into canonical order paid ONCE, on the merged run -- the chunk's own
    // share of distinct keys then says little about what that costs (C2 from FASTQ: thirteen chunks of 0.8 G windows
    // that each see all 10^8 k-mers of the genome, 12 %, took the canonical form per window -- 67 ms of first level
    // where the representatives take 31, ten second-level bits, and a re-ordering per run)
    bool more_follows = false;
    uint64_t more_starts = 0;           // ... and how many window starts of the push being counted lie behind this chunk (a device push knows)
    uint64_t expect_bases = 0;          // goss_gpu_expect_bases: bases the caller means to push in all (0: not said)
    int space_choice = -1;              // the key space the build's fused chunks count in once one has chosen: 0 representatives, 1 canonical forms (-1: none yet)
    uint32_t canon_chunks = 0;          // chunks counted that way
    bool extract_v1 = false;            // GOSS_GPU_EXTRACT_V1=1: per-base LDS extraction kernel for one-word keys
    bool fused = true;                  // GOSS_GPU_NO_FUSED=1: never fuse the first partition pass into the extraction
    uint64_t fused_min = 32u << 20;     // GOSS_GPU_FUSED_MIN=<window starts>: smallest chunk the fused path takes
    uint32_t fused_overflows = 0;       // fused chunks redone because a bucket region was too small
    uint32_t valid_sized_chunks = 0;    // chunks counted in key buffers sized from the estimated share of valid windows
    uint32_t valid_resizes = 0;         // chunks whose estimate was too low: redone in full-size buffers
    uint64_t dump_lo = 0;               // permanent top before the current dump piece
    bool dump_live = false;
    uint64_t budget_limit = 0;          // the arena may grow up to this many bytes (goss_gpu_set_budget_limit; 0 = fixed)
    uint32_t arena_grows = 0;
    uint64_t arena_ms = 0;              // time hipMalloc took to map the arena
    std::thread arena_thread;           // goss_gpu_prepare: the arena is being mapped in the background
    int arena_status = GOSS_OK;         // ... and how that went
    std::string arena_error;
    uint32_t fused_grid = 0;            // GOSS_GPU_FUSED_GRID: workgroups of the fused extraction kernel (0 = 1024)
    bool seg_merge = true;              // GOSS_GPU_NO_SEG_MERGE=1: merge runs by sorting their concatenation
    uint32_t seg_merges = 0;            // merges done by segments
    uint32_t hash_merges = 0;           // ... of them through the counting table (seg_hash_merge96_kernel)
    uint64_t hash_merge_min = 1u << 20; // GOSS_GPU_HASH_MERGE_MIN=<entries>: smallest merge that goes that way (65 536 workgroups)
    bool fused_msd = true;              // GOSS_GPU_NO_MSD=1: never use the two-level (sub-region) form
    uint32_t fused_msd_chunks = 0;      // chunks counted by the two-level form
    bool rem32 = true;                  // GOSS_GPU_NO_REM32=1: never take the 32-bit-remainder form of the second level and the counting
    int rem32_slots = 0;                // GOSS_GPU_REM32_SLOTS=2048|4096|8192|16384: counting table of that form (0 = by the distinct-key estimate)
    uint32_t narrow_capg = 656;         // GOSS_GPU_NARROW_CAPG (tests): granules a tile of the narrow form may lay out before it sends its carried granules off short
    bool ds_parts = true;               // GOSS_GPU_DS_PARTS=0: a range builds no DenseSelect blocks of its own (round 4: the assembler builds them all)
    bool narrow = true;                 // GOSS_GPU_NARROW=0: 8-byte keys between the two levels of the 32-bit-remainder form (rounds 3-4)
    int r32_form = 1;                   // GOSS_GPU_R32_FORM=0: the pair layout of rounds 3-4 (seg_hash_reduce32_kernel), 1: buckets of four (round 5)
    uint32_t r32_small_max = 0;         // distinct keys per segment up to which the 2048-slot table is taken (GOSS_GPU_R32_SMALL_MAX; 0 = the form's default)
    bool big_r32 = true;                // GOSS_GPU_R32_BIG=0: no 8 192- / 16 384-slot tables of remainders (a third partition level instead, as before)
    bool overflow_by_sort = true;       // GOSS_GPU_OVERFLOW_BY_SORT=0: a table that overflows sends the whole chunk up the ladder of forms (rounds 1-5)
    uint64_t overflow_units = 0;        // segments counted by sort because their table overflowed
    uint32_t rem32_chunks = 0;          // chunks counted in that form
    uint32_t narrow_chunks = 0;         // ... of them with remainder + digit (5.33 bytes a key) between the two levels
    uint64_t assemble_us = 0;           // host clock of the last goss_gpu_emit_assemble
    uint64_t ds_blocks_ranges = 0, ds_blocks_own = 0;      // DenseSelect blocks of the last assembly: taken from the ranges' records / built here from the bitmap
    uint32_t pk_fused_chunks = 0;       // chunks of a packed string the fused kernels read as they were
    uint32_t pk_unpacked_chunks = 0;    // ... and chunks (or samples) unpacked to bytes for the plain kernels
    uint32_t rem32_bits_min = 0;        // GOSS_GPU_REM32_BITS=<9..12>: at least that many second-level bits (tests)
    uint32_t rem32_bits_last = 0;       // second-level bits of the last chunk counted in that form
    uint32_t rem32_split_min = 0;       // GOSS_GPU_REM32_SPLIT=<0..4>: at least that many third-level bits (tests; raised when tables overflow)
    uint32_t rem32_split_last = 0;      // third-level bits of the last chunk counted in that form
    uint32_t fused_chunks = 0;          // chunks counted by the fused path
    bool debug = false;                 // GOSS_GPU_DEBUG=1: say on stderr why a fast path was not taken
    double fused_capscale = 1.0;        // GOSS_GPU_FUSED_CAPSCALE: multiplies the bucket regions (tests force overflows)
    uint32_t blk_log2_max = 0;          // GOSS_GPU_BLK_LOG2=<b>: blocks of the fused extraction of at most 2^b slots (experiments)
    bool no_fast32 = false;             // GOSS_GPU_NO_FAST32=1: the fused extraction of one-word keys in its 64-bit form (tests of both forms)
    bool table96 = true;                // GOSS_GPU_NO_TABLE96=1: never count two-word keys as 96-bit remainders in 16-byte slots
    uint32_t table96_chunks = 0;
    bool wide_table = true;             // GOSS_GPU_NO_WIDE_TABLE=1: never count two-word keys in the 6144-slot table
    bool big_table = true;              // GOSS_GPU_NO_BIG_TABLE=1: never count 16-bit segments in the 8192-slot table
    int big_rounds_max = 2;             // GOSS_GPU_BIG_ROUNDS=<r>: at most 2^r workgroups share a segment of that form
    int big_rounds_min = 0;             // GOSS_GPU_BIG_ROUNDS_MIN=<r>: at least 2^r (tests)
    uint32_t big_table_chunks = 0;      // chunks counted that way
    uint32_t wide_table_chunks = 0;     // ... of them, two-word keys in the 6144-slot table
    bool rec_mode = false;              // the current push is a string of super-k-mer records (goss_gpu_push_records_device): "bases" point at
                                        // SkRec records, a "window start" is one of a record's 16 window slots
    uint32_t rec_chunks = 0;            // chunks counted from records by the fused path
    bool deferred = false;              // goss_gpu_set_deferred: host pushes only stage; a full staging buffer is reported, not counted
                                        // (the caller then runs goss_gpu_group_route_exchange: the exchange before counting)
    uint8_t* grp_send = nullptr;        // goss_gpu_group_route_exchange: this member's routed records (device memory of its own) ...
    uint64_t grp_send_cap = 0;          // ... record slots
    uint8_t* grp_inbox = nullptr;       // ... and what the other members sent it
    uint64_t grp_inbox_cap = 0;
    hipStream_t xstream = nullptr;      // stream of the transfers into this member's inbox
    double valid_frac = 1.0;            // estimated valid windows per window start of the current push (sizes the key buffers)
    bool size_by_valid = true;          // GOSS_GPU_NO_VALID_SIZING=1: key buffers always hold one key per window start
    uint64_t budget = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    Arena arena;
    std::vector<Run> runs;
    std::vector<BigMap> big_maps;       // per run that has them: the counts >= 2^32 - 1 (graph mode; marker 0xFFFFFFFF in the run)
    BigMap res_big;                     // ... of the result (its u32 counts hold the value modulo 2^32, as the reference stores it)
    uint64_t windows = 0, keys_total = 0;
    bool finished = false, emitted = false;
    // a group's route-and-exchange round failed half way: what this context had staged may be in records that were
    // lost with the round -- every later push, exchange and finish is refused (GOSS_ERR_STATE) until goss_gpu_reset
    bool broken = false;
    double grp_count_ms = 0;            // how long the background thread of the last exchange round took to count this member's records
    void* res_keys = nullptr;
    uint32_t* res_counts = nullptr;
    uint64_t M = 0;
    std::vector<OutFile> files;
    ExtractCounters* d_ctr = nullptr;     // device counters
    uint32_t* d_flags = nullptr;          // device error flags [0]=count overflow [1]=ef overflow
    void* d_route = nullptr;              // RouteCounters + the parts' first slots and capacities (goss_gpu_route_records_device)
    void* h_pinned = nullptr;             // pinned scratch (>= 64 bytes)
    std::vector<PhaseEvents> events;
    std::vector<hipEvent_t> event_pool;
    goss_gpu_timing timing{};
};

namespace {

// HIP-event timer around the launches of one kernel class (GOSS_T_*).
struct PhaseTimer {
    goss_gpu_ctx* c;
    PhaseEvents pe;
    PhaseTimer(goss_gpu_ctx* ctx, int phase, uint64_t units = 0) : c(ctx)
    {
        muted = c->mute_timing;
        if (muted) return;
        pe.units = units;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
            else HIP_TRY(hipEventCreate(&e));
            return e;
        };
        pe.a = get(); pe.b = get(); pe.phase = phase;
        HIP_TRY(hipEventRecord(pe.a, c->stream));
    }
    void stop()
    {
        if (muted) return;
        HIP_TRY(hipEventRecord(pe.b, c->stream));
        c->events.push_back(pe);
    }
    bool muted = false;
};

void resolve_timing(goss_gpu_ctx* c)
{
    for (auto& pe : c->events)
    {
        HIP_TRY(hipEventSynchronize(pe.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, pe.a, pe.b));
        c->timing.ms[pe.phase] += ms;
        c->timing.launches[pe.phase] += 1;
        c->timing.units[pe.phase] += pe.units;
        c->event_pool.push_back(pe.a);
        c->event_pool.push_back(pe.b);
    }
    c->events.clear();
}

inline uint32_t grid_for(uint64_t n, uint32_t per_block)
{
    uint64_t g = (n + per_block - 1) / per_block;
    if (g == 0) g = 1;
    if (g > 0x7FFFFFFFULL) throw StatusError{GOSS_ERR_INVALID_ARG, "launch grid too large"};
    return (uint32_t)g;
}

void map_arena(goss_gpu_ctx* c)
{
    uint64_t budget = c->budget;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (budget == 0) budget = (uint64_t)((double)free_b * 0.8);
    if (budget > (uint64_t)((double)free_b * 0.92)) budget = (uint64_t)((double)free_b * 0.92);   // a request, not a demand
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, budget);
    if (e != hipSuccess) throw StatusError{GOSS_ERR_OOM, std::string("hipMalloc(budget) failed: ") + hipGetErrorString(e)};
    c->arena_ms = (uint64_t)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->arena.base = (uint8_t*)p;
    c->arena.size = budget;
    c->arena.lo = 0;
    c->arena.hi = budget;
    c->budget = budget;
}

void ensure_arena(goss_gpu_ctx* c)
{
    if (c->arena_thread.joinable())
    {
        c->arena_thread.join();
        if (c->arena_status != GOSS_OK) throw StatusError{c->arena_status, c->arena_error};
    }
    if (c->arena.base) return;
    map_arena(c);
}

// Grow the arena so that at least `want_avail` bytes are free (or double it when want_avail is 0),
// up to budget_limit (0 = the budget is fixed): a new mapping, the permanent part copied over, every
// pointer into the arena rebased.  Only legal where no temporary is live (start of a chunk, start of a
// merge; the staging buffers of host pushes are memory of their own).  False = not possible.
bool grow_arena(goss_gpu_ctx* c, uint64_t want_avail)
{
    Arena& a = c->arena;
    if (!a.base || c->budget_limit <= a.size) return false;
    const uint64_t top = a.size - a.hi;                    // live temporaries at the top: none where growing is legal
    if (top != 0) return false;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t target = std::max<uint64_t>(a.size * 2, a.lo + top + want_avail + (1ULL << 30));
    target = std::min<uint64_t>(target, c->budget_limit);
    target = std::min<uint64_t>(target, (uint64_t)((double)free_b * 0.95));        // the old mapping is still there
    if (target < a.size + (1ULL << 30) || (want_avail && target < a.lo + top + want_avail)) return false;
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipMalloc(&p, target) != hipSuccess) { (void)hipGetLastError(); return false; }
    c->arena_ms += (uint64_t)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint8_t* nb = (uint8_t*)p;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (a.lo) HIP_TRY(hipMemcpyAsync(nb, a.base, a.lo, hipMemcpyDeviceToDevice, c->stream));
    if (top) HIP_TRY(hipMemcpyAsync(nb + target - top, a.base + a.hi, top, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    auto rebase = [&](auto*& ptr) {
        if (!ptr) return;
        uint8_t* q = (uint8_t*)ptr;
        if (q >= a.base && q < a.base + a.size) ptr = (std::remove_reference_t<decltype(ptr)>)(nb + (q - a.base));
    };
    for (auto& r : c->runs) { rebase(r.keys); rebase(r.counts); }
    rebase(c->res_keys); rebase(c->res_counts);
    for (auto& f : c->files) rebase(f.dev);          // file images already emitted (stand-alone SparseArray builds)
    (void)hipFree(a.base);
    a.base = nb; a.hi = target - top; a.size = target;
    c->budget = target;
    c->arena_grows++;
    if (c->debug) std::fprintf(stderr, "libgossgpu: arena grown to %llu GB\n", (unsigned long long)(target >> 30));
    return true;
}

// ---- device-wide exclusive scan (in place) -------------------------------------------
void exclusive_scan_u64(goss_gpu_ctx* c, uint64_t* a, uint64_t n)
{
    if (n == 0) return;
    uint64_t nchunks = (n + kScanChunk - 1) / kScanChunk;
    if (nchunks == 1)
    {
        hipLaunchKernelGGL(scan_apply_kernel, dim3(1), dim3(kTB), 0, c->stream, a, n, (const uint64_t*)nullptr);
        return;
    }
    uint64_t mark = c->arena.mark();
    uint64_t* partial = (uint64_t*)c->arena.temp(nchunks * 8);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(grid_for(n, kScanChunk)), dim3(kTB), 0, c->stream, (const uint64_t*)a, n, partial);
    exclusive_scan_u64(c, partial, nchunks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(grid_for(n, kScanChunk)), dim3(kTB), 0, c->stream, a, n, (const uint64_t*)partial);
    // the stream orders the kernels; the scratch can be reused by later launches on the
    // same stream once released
    c->arena.release(mark);
}

// ---- LSD radix sort -------------------------------------------------------------------
// Stable passes on the 8-bit digits at bit first_shift, first_shift+8, ... (ndigits of them).
// Returns true if the result is in (kb, vb).
//
// One pass with per-tile histogram table: histogram kernel, device scan, stable scatter.
template <class K, bool HAS_VAL>
void radix_pass_table(goss_gpu_ctx* c, const K* src, const uint32_t* vs, K* dst, uint32_t* vd, uint64_t n, uint32_t d,
                      uint64_t ntiles, uint64_t* table)
{
    constexpr int tile = SortCfg<K, HAS_VAL>::kTile;
    {
        PhaseTimer t(c, GOSS_T_HIST, n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_hist_kernel<K, HAS_VAL>), dim3(grid_for(n, tile)), dim3(kTB), 0, c->stream,
                           src, n, d, ntiles, table);
        t.stop();
    }
    {
        PhaseTimer t(c, GOSS_T_SCAN, 256ULL * ntiles);
        exclusive_scan_u64(c, table, 256ULL * ntiles);
        t.stop();
    }
    {
        PhaseTimer t(c, GOSS_T_SCATTER, n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_scatter_kernel<K, HAS_VAL>), dim3(grid_for(n, tile)), dim3(kTB), 0, c->stream,
                           src, vs, dst, vd, n, d, ntiles, (const uint64_t*)table);
        t.stop();
    }
}

template <class K, bool HAS_VAL>
bool radix_sort(goss_gpu_ctx* c, K* ka, K* kb, uint32_t* va, uint32_t* vb, uint64_t n, uint32_t ndigits,
                uint32_t first_shift = 0, const unsigned long long* prehist = nullptr)
{
    if (n < 2) return false;
    constexpr int tile = SortCfg<K, HAS_VAL>::kTile;
    const uint64_t ntiles = (n + tile - 1) / tile;
    uint64_t mark = c->arena.mark();
    bool in_b = false;
    // the per-tile table (classic passes) and the look-back status words share one buffer
    uint64_t* table = (uint64_t*)c->arena.temp(256ULL * ntiles * 8);
    unsigned long long* status = (unsigned long long*)table;
    unsigned long long* hist = nullptr;
    unsigned long long* cursors = nullptr;
    LookbackCtl* ctl = nullptr;
    LookbackCtl* hctl = (LookbackCtl*)((uint8_t*)c->h_pinned + 128);
    bool lookback = c->lookback && ndigits <= 16;
    if (lookback)
    {
        // single-pass form: all digit histograms in one read of the keys, then one look-back
        // scatter per digit
        hist = (unsigned long long*)c->arena.temp(ndigits * 256 * 8);
        ctl = (LookbackCtl*)c->arena.temp(sizeof(LookbackCtl));
        cursors = (unsigned long long*)c->arena.temp(256 * kCursorStride * 8);
        HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(LookbackCtl), c->stream));
        if (prehist)
            HIP_TRY(hipMemcpyAsync(hist, prehist, ndigits * 256 * 8, hipMemcpyDeviceToDevice, c->stream));
        else
        {
            HIP_TRY(hipMemsetAsync(hist, 0, ndigits * 256 * 8, c->stream));
            PhaseTimer t(c, GOSS_T_HIST, n);
            uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n + kTB - 1) / kTB);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(global_hist_kernel<K>), dim3(grid), dim3(kTB), 0, c->stream,
                               (const K*)ka, n, first_shift, ndigits, hist);
            t.stop();
        }
        {
            PhaseTimer t(c, GOSS_T_SCAN, ndigits * 256);
            hipLaunchKernelGGL(scan_rows256_kernel, dim3(ndigits), dim3(kTB), 0, c->stream, hist);
            t.stop();
        }
    }
    for (uint32_t di = 0; di < ndigits; ++di)
    {
        const uint32_t d = first_shift + 8 * di;       // bit offset of this pass's digit
        K* src = in_b ? kb : ka; K* dst = in_b ? ka : kb;
        uint32_t* vs = in_b ? vb : va; uint32_t* vd = in_b ? va : vb;
        bool done = false;
        if (lookback)
        {
            // pass 0 may place tiles in any order: atomic bucket cursors instead of the chain
            unsigned long long* cur = (di == 0 && c->cursor_pass0) ? cursors : nullptr;
            if (cur) HIP_TRY(hipMemsetAsync(cursors, 0, 256 * kCursorStride * 8, c->stream));
            else HIP_TRY(hipMemsetAsync(status, 0, ntiles * 256 * 8, c->stream));
            {
                PhaseTimer t(c, GOSS_T_SCATTER, n);
                if (c->ordered_tiles)
                {
                    HIP_TRY(hipMemsetAsync(&ctl->ticket, 0, 4, c->stream));
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_onesweep_kernel<K, HAS_VAL, true>), dim3(grid_for(n, tile)), dim3(kTB), 0,
                                       c->stream, (const K*)src, (const uint32_t*)vs, dst, vd, n, d, first_shift,
                                       (const unsigned long long*)(hist + di * 256), status, ctl, cur);
                }
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_onesweep_kernel<K, HAS_VAL, false>), dim3(grid_for(n, tile)), dim3(kTB), 0,
                                       c->stream, (const K*)src, (const uint32_t*)vs, dst, vd, n, d, first_shift,
                                       (const unsigned long long*)(hist + di * 256), status, ctl, cur);
                t.stop();
            }
            // the source buffer is still intact: a chain that gave up is redone below
            HIP_TRY(hipMemcpyAsync(hctl, ctl, sizeof(LookbackCtl), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
#if defined(GOSS_LB_STATS)
            std::fprintf(stderr, "lookback pass d=%u: tiles=%llu walk_steps=%llu (%.2f/tile) spin_polls=%llu (%.2f/tile) max_depth=%llu\n", d,
                         hctl->tiles, hctl->walk_steps, (double)hctl->walk_steps / (double)(hctl->tiles ? hctl->tiles : 1),
                         hctl->spin_polls, (double)hctl->spin_polls / (double)(hctl->tiles ? hctl->tiles : 1), hctl->max_depth);
            HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(LookbackCtl), c->stream));
#endif
            if (hctl->error)
            {
                std::fprintf(stderr, "libgossgpu: radix look-back chain gave up; redoing the pass with histogram tables\n");
                c->lookback_failures++;
                if (!c->ordered_tiles) c->ordered_tiles = true;     // next passes take tickets
                HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(LookbackCtl), c->stream));
            }
            else done = true;
        }
        if (!done) radix_pass_table<K, HAS_VAL>(c, src, vs, dst, vd, n, d, ntiles, table);
        in_b = !in_b;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
    return in_b;
}

// ---- counts beyond 32 bits (graph mode) ----------------------------------------------------
// After a step that summed counts into `out` (run lengths, run sums, segment merge) and raised the
// overflow flag: the entries of `out` that carry the marker get their exact u64 count from the inputs
// the step read (`keys`/`vals` in `nruns` sorted pieces; vals == nullptr: raw keys, each worth 1) plus
// what the input runs' own big maps hold for the key.  A k-mer set stores no counts: nothing to do.
constexpr uint32_t kMaxBig = 256;
template <class K>
void resolve_big_counts(goss_gpu_ctx* c, Run& out, const K* keys, const uint32_t* vals, const std::vector<uint64_t>& run_off,
                        const std::vector<int>& in_bigs)
{
    if (c->mode != GOSS_MODE_GRAPH || out.m == 0) return;
    uint32_t* hf = (uint32_t*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(hf, c->d_flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!hf[0]) return;
    uint64_t mark = c->arena.mark();
    unsigned long long* found = (unsigned long long*)c->arena.temp((kMaxBig + 1) * 8);
    HIP_TRY(hipMemsetAsync(found, 0, 8, c->stream));
    hipLaunchKernelGGL(find_saturated_kernel, dim3(grid_for(out.m, 256)), dim3(256), 0, c->stream, (const uint32_t*)out.counts, out.m, found, kMaxBig);
    std::vector<unsigned long long> hfound(kMaxBig + 1);
    HIP_TRY(hipMemcpyAsync(hfound.data(), found, (kMaxBig + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t nq = (uint32_t)hfound[0];
    if (nq > kMaxBig) throw StatusError{GOSS_ERR_COUNT_OVERFLOW, "more than 256 keys occurred 2^32 times or more"};
    const uint32_t nruns = (uint32_t)run_off.size() - 1;
    if (nruns > 1024) throw StatusError{GOSS_ERR_COUNT_OVERFLOW, "a key occurred 2^32 times or more in a merge of more than 1024 runs"};
    if (nq)
    {
        K* dq = (K*)c->arena.temp(nq * sizeof(K));
        uint64_t* doff = (uint64_t*)c->arena.temp((nruns + 1) * 8);
        unsigned long long* dsum = (unsigned long long*)c->arena.temp(2 * nq * 8);
        for (uint32_t q = 0; q < nq; ++q)
            HIP_TRY(hipMemcpyAsync(dq + q, (const K*)out.keys + hfound[1 + q], sizeof(K), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(doff, run_off.data(), (nruns + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(dsum, 0, 2 * nq * 8, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sum_equal_kernel<K>), dim3(nq), dim3(std::max<uint32_t>(64, (nruns + 63) / 64 * 64)), 0, c->stream,
                           keys, vals, (const uint64_t*)doff, nruns, (const K*)dq, nq, dsum, dsum + nq);
        std::vector<K> hq(nq);
        std::vector<unsigned long long> hs(2 * nq);
        HIP_TRY(hipMemcpyAsync(hq.data(), dq, nq * sizeof(K), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(hs.data(), dsum, 2 * nq * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        BigMap bm;
        for (uint32_t q = 0; q < nq; ++q)
        {
            const std::pair<uint64_t, uint64_t> kk(key_hi_word(hq[q]), key_lo_word(hq[q]));
            uint64_t exact = hs[q], markers = 0;
            for (int b : in_bigs)
                if (b >= 0)
                {
                    auto it = c->big_maps[b].find(kk);
                    if (it != c->big_maps[b].end()) { exact += it->second; ++markers; }
                }
            if (markers != hs[nq + q]) throw StatusError{GOSS_ERR_HIP, "count bookkeeping: a saturated entry without its exact count"};
            bm[kk] = exact;
        }
        c->big_maps.push_back(std::move(bm));
        out.big = (int)c->big_maps.size() - 1;
    }
    HIP_TRY(hipMemsetAsync(c->d_flags, 0, 4, c->stream));
    c->arena.release(mark);
}

// ---- run compaction ---------------------------------------------------------------------
// keys sorted (n).  Produces a permanent Run (distinct keys + u32 counts).  If vals != null
// the counts are sums of vals over each run, else run lengths.  `scratch_keys` is a buffer
// of >= n keys that may be clobbered (the free half of the sort ping-pong).
template <class K>
Run reduce_runs(goss_gpu_ctx* c, const K* keys, const uint32_t* vals, uint64_t n, K* scratch_keys, const std::vector<int>* in_bigs = nullptr)
{
    Run r{nullptr, nullptr, 0};
    if (n == 0) return r;
    uint64_t mark = c->arena.mark();
    const uint64_t ntiles = (n + kRedTile - 1) / kRedTile;
    uint64_t* tile_counts = (uint64_t*)c->arena.temp((ntiles + 1) * 8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_count_kernel<K>), dim3(grid_for(n, kRedTile)), dim3(kTB), 0, c->stream,
                       keys, n, tile_counts);
    // total = last offset + last count: append a zero element and scan ntiles+1 entries
    HIP_TRY(hipMemsetAsync(tile_counts + ntiles, 0, 8, c->stream));
    exclusive_scan_u64(c, tile_counts, ntiles + 1);
    uint64_t* h = (uint64_t*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, tile_counts + ntiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t m = h[0];
    uint64_t* starts = (uint64_t*)c->arena.temp(m * 8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_write_kernel<K>), dim3(grid_for(n, kRedTile)), dim3(kTB), 0, c->stream,
                       keys, n, (const uint64_t*)tile_counts, scratch_keys, starts);
    // permanent storage
    r.m = m;
    r.keys = c->arena.perm(m * sizeof(K));
    r.counts = (uint32_t*)c->arena.perm(m * 4);
    HIP_TRY(hipMemcpyAsync(r.keys, scratch_keys, m * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
    if (vals)
        hipLaunchKernelGGL(run_sums_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream,
                           (const uint64_t*)starts, m, n, vals, r.counts, c->d_flags);
    else
        hipLaunchKernelGGL(run_lengths_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream,
                           (const uint64_t*)starts, m, n, r.counts, c->d_flags);
    HIP_TRY(hipStreamSynchronize(c->stream));   // temporaries are released below
    resolve_big_counts<K>(c, r, keys, vals, std::vector<uint64_t>{0, n}, in_bigs ? *in_bigs : std::vector<int>());
    c->arena.release(mark);
    return r;
}

// ---- packed strings (ctx.pk) ---------------------------------------------------------------------------------------
constexpr uintptr_t kPkFake = (uintptr_t)1 << 44;          // "address" of position 0 of the packed string being counted (a multiple of 16)
// groups of a packed staging buffer of `cap` positions, and its two arrays
static inline uint64_t pk_groups(uint64_t cap) { return cap / 16 + 4; }
static inline uint32_t* pk_codes_of(uint8_t* buf) { return (uint32_t*)buf; }
static inline uint16_t* pk_bad_of(uint8_t* buf, uint64_t cap) { return (uint16_t*)(buf + pk_groups(cap) * 4); }
// a 16-byte aligned made-up address -> the codes and flags of its group
static inline void pk_ptrs(const goss_gpu_ctx* c, const uint8_t* aligned, const uint8_t** src, const uint16_t** bad)
{
    const uint64_t g = ((uintptr_t)aligned - kPkFake) >> 4;
    *src = (const uint8_t*)(c->pk.codes + g);
    *bad = c->pk.bad + g;
}
// Bytes of `n` positions of the packed string from made-up address `p` on, for the kernels that have no packed form (the
// plain extraction of small inputs and of the fallback sequence -- never the fused path's): unpacked into a temporary
// of the arena (the caller releases its mark).  Returns the aligned byte address; *mis as for a byte string.
static const uint8_t* pk_unpack_temp(goss_gpu_ctx* c, const uint8_t* p, uint64_t n, uint32_t* mis)
{
    const uintptr_t addr = (uintptr_t)p;
    *mis = (uint32_t)(addr & 15u);
    const uint8_t* src; const uint16_t* bad;
    pk_ptrs(c, (const uint8_t*)(addr - *mis), &src, &bad);
    const uint64_t groups = (n + *mis + 15) / 16;
    c->pk_unpacked_chunks++;
    uint8_t* out = (uint8_t*)c->arena.temp((groups + 1) * 16 + 64);
    hipLaunchKernelGGL(unpack_bases_kernel, dim3(grid_for(groups + 1, kTB)), dim3(kTB), 0, c->stream, (const uint32_t*)src, bad, groups,
                       n + *mis, out);
    return out;
}

template <class K, int MODE, int P>
void launch_extract(goss_gpu_ctx* c, const uint8_t* aligned, uint32_t mis, uint64_t nstarts, uint64_t navail, K* out)
{
    constexpr int T = kTB * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_kernel<K, MODE, P>), dim3(grid_for(nstarts, T)), dim3(kTB), 0, c->stream,
                       aligned, mis, nstarts, navail, c->len, out, c->d_ctr);
}

template <class K>
void extract_dispatch(goss_gpu_ctx* c, const uint8_t* aligned, uint32_t mis, uint64_t nstarts, uint64_t navail, K* out);
template <class K> bool use_segment_path(const goss_gpu_ctx* c);
template <int MODE, int P, int G>
void launch_extract1(goss_gpu_ctx* c, const uint8_t* aligned, uint32_t mis, uint64_t nstarts, uint64_t navail, Key1* out)
{
    constexpr int T = kTB * P * G;
    const uint64_t nsuper = (nstarts + T - 1) / T;
    // persistent grid: 4 workgroups per CU (LDS-limited), 256 CUs
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nsuper ? nsuper : 1, 1024 * 2);
    // fused digit histograms for the 16-bit partition that follows on the segment path
    c->extract_hist_shift = use_segment_path<Key1>(c) ? 2 * c->len - kSegBits : 0xFFFFFFFFu;
#define GOSS_LAUNCH_E1(NB)                                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(extract1_kernel<MODE, P, G, NB>), dim3(grid), dim3(kTB), 0, c->stream, aligned, mis, \
                       nstarts, navail, c->len, out, c->d_ctr, c->extract_hist_shift, nsuper)
    if (MODE == 1) { GOSS_LAUNCH_E1(8); return; }          // graph mode does not hash
    if (MODE == 0 && c->extract_rep)
    {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract1_kernel<MODE, P, G, 8, true>), dim3(grid), dim3(kTB), 0, c->stream, aligned, mis,
                           nstarts, navail, c->len, out, c->d_ctr, c->extract_hist_shift, nsuper);
        return;
    }
    switch ((2 * c->len + 7) / 8)
    {
        case 1: GOSS_LAUNCH_E1(1); break;
        case 2: GOSS_LAUNCH_E1(2); break;
        case 3: GOSS_LAUNCH_E1(3); break;
        case 4: GOSS_LAUNCH_E1(4); break;
        case 5: GOSS_LAUNCH_E1(5); break;
        case 6: GOSS_LAUNCH_E1(6); break;
        case 7: GOSS_LAUNCH_E1(7); break;
        default: GOSS_LAUNCH_E1(8); break;
    }
#undef GOSS_LAUNCH_E1
}
// window slots of a record (the most windows one holds): a string of n records counts as 16 n window starts
inline uint32_t rec_slots(const goss_gpu_ctx*) { return 16u; }
// bytes of a record: 12 for one-word keys (SkRec), 20 for two-word keys (SkRec2)
inline uint64_t rec_bytes(const goss_gpu_ctx* c) { return c->words == 1 ? sizeof(SkRec) : sizeof(SkRec2); }

// records -> keys with the plain kernel: all records (slice_groups = 0) or slices of them (the fused path's sample)
void launch_extract_records(goss_gpu_ctx* c, const SkRec* recs, uint64_t nrecs, Key1* out, uint64_t ngroups, uint64_t slice_groups,
                            uint64_t slice_stride, bool rep)
{
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ngroups, 1), 256 * 16);
    if (c->mode == GOSS_MODE_GRAPH && !rep)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_records_kernel<1, false>), dim3(grid), dim3(kTB), 0, c->stream, recs, nrecs, c->len, out,
                           c->d_ctr, ngroups, slice_groups, slice_stride);
    else if (rep)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_records_kernel<0, true>), dim3(grid), dim3(kTB), 0, c->stream, recs, nrecs, c->len, out,
                           c->d_ctr, ngroups, slice_groups, slice_stride);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_records_kernel<0, false>), dim3(grid), dim3(kTB), 0, c->stream, recs, nrecs, c->len, out,
                           c->d_ctr, ngroups, slice_groups, slice_stride);
}

// the same for two-word records (rep: graph builds counted as strand pairs)
void launch_extract_records2(goss_gpu_ctx* c, const SkRec2* recs, uint64_t nrecs, Key2* out, uint64_t ngroups, uint64_t slice_groups,
                             uint64_t slice_stride, bool rep)
{
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ngroups, 1), 256 * 16);
    if (c->mode == GOSS_MODE_GRAPH && !rep)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_records2_kernel<1, false>), dim3(grid), dim3(kTB), 0, c->stream, recs, nrecs, c->len, out, c->d_ctr, ngroups,
                           slice_groups, slice_stride);
    else if (rep)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_records2_kernel<0, true>), dim3(grid), dim3(kTB), 0, c->stream, recs, nrecs, c->len, out, c->d_ctr, ngroups,
                           slice_groups, slice_stride);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(extract_records2_kernel<0, false>), dim3(grid), dim3(kTB), 0, c->stream, recs, nrecs, c->len, out, c->d_ctr, ngroups,
                           slice_groups, slice_stride);
}

template <>
void extract_dispatch<Key1>(goss_gpu_ctx* c, const uint8_t* aligned, uint32_t mis, uint64_t nstarts, uint64_t navail, Key1* out)
{
    c->extract_hist_shift = 0xFFFFFFFFu;
    if (c->rec_mode)
    {
        const uint64_t nrecs = nstarts / rec_slots(c);
        launch_extract_records(c, (const SkRec*)aligned, nrecs, out, (nrecs + kRecGroup - 1) / kRecGroup, 0, 0, c->extract_rep);
        return;
    }
    if (c->extract_v1)
    {
        if (c->mode == GOSS_MODE_KMER_SET) launch_extract<Key1, 0, 16>(c, aligned, mis, nstarts, navail, out);
        else launch_extract<Key1, 1, 8>(c, aligned, mis, nstarts, navail, out);
        return;
    }
    // (extract_rep in graph mode: one strand representative per window -- the fused path's key space for graphs)
    if (c->mode == GOSS_MODE_KMER_SET || c->extract_rep) launch_extract1<0, 16, 8>(c, aligned, mis, nstarts, navail, out);
    else launch_extract1<1, 8, 8>(c, aligned, mis, nstarts, navail, out);
}
// significant bytes of a two-word key's high word, rounded up to the instantiated classes 2/4/6/8
inline int key2_nbh(const goss_gpu_ctx* c) { const int nb = (int)(2 * c->len + 7) / 8 - 8; return nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 6 ? 6 : 8; }

template <int MODE, int P, int G>
void launch_extract2(goss_gpu_ctx* c, const uint8_t* aligned, uint32_t mis, uint64_t nstarts, uint64_t navail, Key2* out,
                     uint64_t slice_tiles = 0, uint64_t slice_stride = 0, uint64_t nsuper_override = 0, bool rep = false)
{
    constexpr int T = kTB * P * G;
    const uint64_t nsuper = nsuper_override ? nsuper_override : (nstarts + T - 1) / T;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nsuper ? nsuper : 1, 1024 * 2);
    // (a packed string -- only the fused path's sample comes here with one: the plain extraction of a whole chunk unpacks first)
    const uint8_t* pk_src = nullptr; const uint16_t* pk_bad = nullptr;
    if (c->pk.on) pk_ptrs(c, aligned, &pk_src, &pk_bad);
#define GOSS_LAUNCH_E2(NBH)                                                                                           \
    do {                                                                                                              \
        if (c->pk.on)                                                                                                 \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract2_kernel<MODE, P, G, NBH, true>), dim3(grid), dim3(kTB), 0, c->stream, pk_src, mis, nstarts,  \
                               navail, c->len, out, c->d_ctr, nsuper, slice_tiles, slice_stride, pk_bad);             \
        else                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract2_kernel<MODE, P, G, NBH>), dim3(grid), dim3(kTB), 0, c->stream, aligned, mis, nstarts,  \
                               navail, c->len, out, c->d_ctr, nsuper, slice_tiles, slice_stride, (const uint16_t*)nullptr); \
    } while (0)
    if (MODE == 1) { GOSS_LAUNCH_E2(8); return; }             // graph mode does not hash
    if (rep) { GOSS_LAUNCH_E2(0); return; }                   // strand representatives (NBH 0): no hash either
    switch (key2_nbh(c))
    {
        case 2: GOSS_LAUNCH_E2(2); break;
        case 4: GOSS_LAUNCH_E2(4); break;
        case 6: GOSS_LAUNCH_E2(6); break;
        default: GOSS_LAUNCH_E2(8); break;
    }
#undef GOSS_LAUNCH_E2
}
template <>
void extract_dispatch<Key2>(goss_gpu_ctx* c, const uint8_t* aligned, uint32_t mis, uint64_t nstarts, uint64_t navail, Key2* out)
{
    if (c->rec_mode)
    {
        // two-word records -> keys with the plain kernel (their counting goes through the unfused sequence)
        const uint64_t nrecs = nstarts / rec_slots(c);
        launch_extract_records2(c, (const SkRec2*)aligned, nrecs, out, (nrecs + kRecGroup - 1) / kRecGroup, 0, 0, c->extract_rep);
        return;
    }
    if (c->extract_v1)
    {
        if (c->mode == GOSS_MODE_KMER_SET) launch_extract<Key2, 0, 8>(c, aligned, mis, nstarts, navail, out);
        else launch_extract<Key2, 1, 4>(c, aligned, mis, nstarts, navail, out);
        return;
    }
    if (c->mode == GOSS_MODE_KMER_SET || c->extract_rep) launch_extract2<0, 8, 8>(c, aligned, mis, nstarts, navail, out, 0, 0, 0, c->extract_rep);
    else launch_extract2<1, 4, 8>(c, aligned, mis, nstarts, navail, out);
}

inline uint32_t key_digits(const goss_gpu_ctx* c) { return (2 * c->len + 7) / 8; }

// ---- fast path: radix partition on the top bits + per-segment LDS hash table -------------
template <class K> bool use_segment_path(const goss_gpu_ctx* c)
{
    if (c->path == 1) return false;                   // LSD only
    return 2 * c->len >= 24;                          // enough key bits below the partition bits
}
template bool use_segment_path<Key1>(const goss_gpu_ctx*);
template bool use_segment_path<Key2>(const goss_gpu_ctx*);

template <class K> struct SegCfg;
template <> struct SegCfg<Key1> { static constexpr uint64_t kLimit = kSegLimit; };
template <> struct SegCfg<Key2> { static constexpr uint64_t kLimit = kSegLimit2; };

// Grid of `units` workgroups for the kernels that number their workgroup with unit_block(): x up
// to 2^20, the rest in y (units is a power of two above that).  HIP refuses gridDim.x * blockDim.x
// >= 2^32 -- and a refused launch leaves the output counters at zero, which reads as "no keys".
inline dim3 unit_grid(uint64_t units)
{
    const uint64_t kx = 1u << 20;
    if (units <= kx) return dim3((uint32_t)std::max<uint64_t>(units, 1));
    return dim3((uint32_t)kx, (uint32_t)((units + kx - 1) / kx));
}
inline void check_launch(const char* what)
{
    // the thread's error state was cleared when the entry point was entered (guarded): whatever
    // is there now was raised by this library's own calls -- it never polls, so not even
    // hipErrorNotReady is expected
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw StatusError{GOSS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)};
}

inline void launch_seg_hash(goss_gpu_ctx* c, uint32_t nseg, const Key1* keys, const uint64_t* seg_off, const uint64_t* seg_end,
                            SegOut* so, uint64_t* seg_pos, uint64_t* seg_cnt, Key1* sk, uint32_t* sc, uint32_t rem_bits, int big)
{
    // big: 0 = the 4096-slot table; 1 + r = the 8192-slot table, every segment shared by 2^r workgroups
    if (big > 1)
        hipLaunchKernelGGL(seg_hash_reduce_shared_kernel, unit_grid((uint64_t)nseg << (big - 1)), dim3(kSegBigThreads), 0, c->stream, keys, seg_off, seg_end,
                           so, seg_pos, seg_cnt, sk, sc, rem_bits, (uint32_t)(big - 1));
    else if (big)
        hipLaunchKernelGGL(seg_hash_reduce_big_kernel, unit_grid(nseg), dim3(kSegBigThreads), 0, c->stream, keys, seg_off, seg_end,
                           so, seg_pos, seg_cnt, sk, sc, rem_bits, 0u);
    else
        hipLaunchKernelGGL(seg_hash_reduce_kernel, unit_grid(nseg), dim3(kTB), 0, c->stream, keys, seg_off, seg_end, so, seg_pos, seg_cnt, sk, sc,
                           rem_bits);
}
inline void launch_seg_hash(goss_gpu_ctx* c, uint32_t nseg, const Key2* keys, const uint64_t* seg_off, const uint64_t* seg_end,
                            SegOut* so, uint64_t* seg_pos, uint64_t* seg_cnt, Key2* sk, uint32_t* sc, uint32_t rem_bits, int big)
{
    // big: -2 = the 8192-slot table of 96-bit remainders, -1 = the 6144-slot table, one workgroup per segment each;
    // 1 + r = the 4096-slot table, 2^r workgroups per segment
    if (big == -3)      // the 96-bit-remainder table fed 12-byte records by the second level
        hipLaunchKernelGGL(seg_hash_reduce96p_kernel, unit_grid(nseg), dim3(kSegBigThreads), 0, c->stream, keys, seg_off, seg_end, so,
                           seg_pos, seg_cnt, sk, sc, rem_bits);
    else if (big == -2)
        hipLaunchKernelGGL(seg_hash_reduce96_kernel, unit_grid(nseg), dim3(kSegBigThreads), 0, c->stream, keys, seg_off, seg_end, so,
                           seg_pos, seg_cnt, sk, sc, rem_bits);
    else if (big < 0)
        hipLaunchKernelGGL(seg_hash_reduce2_wide_kernel, unit_grid(nseg), dim3(kSegBigThreads), 0, c->stream, keys, seg_off, seg_end, so,
                           seg_pos, seg_cnt, sk, sc, rem_bits);
    else if (big)
        hipLaunchKernelGGL(seg_hash_reduce2_big_kernel, unit_grid((uint64_t)nseg << (big - 1)), dim3(kSegBigThreads), 0, c->stream, keys, seg_off, seg_end, so,
                           seg_pos, seg_cnt, sk, sc, rem_bits, (uint32_t)(big - 1));
    else
        hipLaunchKernelGGL(seg_hash_reduce2_kernel, unit_grid(nseg), dim3(kTB), 0, c->stream, keys, seg_off, seg_end, so, seg_pos, seg_cnt, sk, sc,
                           rem_bits);
}

// Number of distinct keys in the chunk, estimated from its first keys (reads arrive in no
// particular order, so a prefix is a sample): with s sampled keys of which d are distinct,
// M ~ s^2 / (2 (s - d)) (birthday estimate), never less than d.  Returns 0 when the sample has
// no repeated key at all (M unknown, presumably of the order of n).
// Exact number of distinct keys among the first s keys (sorts a copy of them).
template <class K>
uint64_t count_distinct_sample(goss_gpu_ctx* c, const K* keys, uint64_t s)
{
    if (s < 2) return s;
    uint64_t mark = c->arena.mark();
    K* a = (K*)c->arena.temp(s * sizeof(K));
    K* b = (K*)c->arena.temp(s * sizeof(K));
    HIP_TRY(hipMemcpyAsync(a, keys, s * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
    // auxiliary sort of the sample: histogram-table kernels, kept out of the per-kernel timing
    // so that the partition kernel's launch statistics describe the partition only
    const bool lb = c->lookback;
    c->lookback = false; c->mute_timing = true;
    bool in_b = radix_sort<K, false>(c, a, b, nullptr, nullptr, s, key_digits(c));
    c->lookback = lb; c->mute_timing = false;
    const uint64_t ntiles = (s + kRedTile - 1) / kRedTile;
    uint64_t* tile_counts = (uint64_t*)c->arena.temp((ntiles + 1) * 8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_count_kernel<K>), dim3(grid_for(s, kRedTile)), dim3(kTB), 0, c->stream,
                       (const K*)(in_b ? b : a), s, tile_counts);
    HIP_TRY(hipMemsetAsync(tile_counts + ntiles, 0, 8, c->stream));
    exclusive_scan_u64(c, tile_counts, ntiles + 1);
    uint64_t* h = (uint64_t*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, tile_counts + ntiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t d = h[0];
    c->arena.release(mark);
    return d;
}

// Number of equally likely keys M from which s draws show d distinct ones: the root of
// d = M (1 - exp(-s / M)) (~ s^2 / (2 (s - d)) when almost every draw is new, ~ d when the
// sample saturates the population).  0 = no repeated key at all (unknown, at least of the order of s).
inline uint64_t birthday_estimate(uint64_t s, uint64_t d)
{
    if (d >= s) return 0;
    double lo = (double)d, hi = (double)d;
    auto seen = [&](double m) { return m * -std::expm1(-(double)s / m); };
    while (seen(hi) < (double)d && hi < 1e19) hi *= 2.0;
    for (int it = 0; it < 80; ++it)
    {
        const double mid = 0.5 * (lo + hi);
        if (seen(mid) < (double)d) lo = mid; else hi = mid;
    }
    return std::max<uint64_t>(d, (uint64_t)hi);
}

// Number of distinct keys of a population of `population` keys of which keys[0, avail) are a random part
// (any prefix of a chunk: reads arrive in no particular order).  The birthday estimate above assumes
// that every key is equally frequent; sequencing reads are not like that -- the k-mers of the genome occur
// `coverage` times each, the k-mers that contain a base-calling error once -- and it then reports little more
// than the genome (1e8 for C2 with 0.1 % errors, where 3.3e8 is right): the counting tables overflow and the
// chunk is redone with more partition bits, several times.  So: the multiplicity SPECTRUM of a slice of
// the key space.  The keys whose mixed bits are zero (all copies of a key or none) are sorted and the keys
// seen exactly once / twice / three times counted (f1, f2, f3).  A key that occurs c times in the population
// shows Poisson(c p) copies in the part (p = avail / population): the frequent component is fitted from
// f2 and f3 (lambda = 3 f3 / f2, G = 2 f2 e^lambda / lambda^2), the singletons it does not explain are
// population keys of multiplicity ~1 seen with probability p each.  Never less than the plain estimate.
template <class K>
uint64_t spectrum_estimate(goss_gpu_ctx* c, const K* keys, uint64_t avail, double population, uint64_t* rare_out = nullptr)
{
    // *rare_out: of the estimate, the keys of multiplicity ~1 in the population (more of the same input brings more
    // of THEM; the frequent keys it mostly brings again); the whole estimate where the spectrum could not be fitted
    if (rare_out) *rare_out = avail;
    if (avail < 2) return avail;
    const uint64_t scan = std::min<uint64_t>(avail, 1ULL << 30);
    uint32_t q = 1;
    while ((scan / q) > (3u << 20) && q < (1u << 16)) q <<= 1;          // 1.5 to 3 M keys survive
    const uint64_t cap = scan / q + scan / q / 4 + (1u << 16);
    uint64_t mark = c->arena.mark();
    K* a = (K*)c->arena.temp(cap * sizeof(K));
    K* b = (K*)c->arena.temp(cap * sizeof(K));
    unsigned long long* ctr = (unsigned long long*)c->arena.temp(64);
    HIP_TRY(hipMemsetAsync(ctr, 0, 64, c->stream));
    const bool lb = c->lookback, mute = c->mute_timing;
    c->mute_timing = true;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(slice_filter_kernel<K>), dim3((uint32_t)std::min<uint64_t>(2048, (scan + kTB - 1) / kTB)), dim3(kTB), 0, c->stream,
                       keys, scan, q - 1, a, ctr, cap);
    unsigned long long* h = (unsigned long long*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, ctr, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t sf = std::min<uint64_t>(h[0], cap);
    uint64_t est = 0;
    if (sf >= 2)
    {
        c->lookback = false;
        const bool in_b = radix_sort<K, false>(c, a, b, nullptr, nullptr, sf, key_digits(c));
        c->lookback = lb;
        HIP_TRY(hipMemsetAsync(ctr, 0, 64, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spectrum_kernel<K>), dim3((uint32_t)std::min<uint64_t>(1024, (sf + kTB - 1) / kTB)), dim3(kTB), 0, c->stream,
                           (const K*)(in_b ? b : a), sf, ctr);
        HIP_TRY(hipMemcpyAsync(h, ctr, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const double d = (double)h[0], f1 = (double)h[1], f2 = (double)h[2], f3 = (double)h[3];
        const double p = std::min(1.0, (double)scan / std::max(population, (double)scan));
        const double scale_q = (double)q;
        // plain estimate on the slice (equally frequent keys), scaled to the key space
        const uint64_t plain = scan >= (uint64_t)population ? (uint64_t)(d * scale_q) : (uint64_t)((double)birthday_estimate(sf, (uint64_t)d) * scale_q);
        double model = 0, rare_keys = -1.0;
        if (p >= 1.0) { model = d; rare_keys = f1; }              // the whole population was looked at
        else if (f2 >= 256.0 && f3 >= 64.0)
        {
            const double lambda = 3.0 * f3 / f2;
            const double G = 2.0 * f2 * std::exp(lambda) / (lambda * lambda);
            const double rare = std::max(0.0, f1 - G * lambda * std::exp(-lambda)) / p;
            // (seen but fitted by neither: the keys seen four times and more are part of G already)
            model = G + rare;
            rare_keys = rare;
        }
        est = std::max<uint64_t>(plain, (uint64_t)(model * scale_q));
        est = std::max<uint64_t>(est, (uint64_t)(d * scale_q));
        if (plain == 0 && model == 0) est = 0;                    // no repeated key at all: unknown
        if (rare_out) *rare_out = rare_keys < 0 ? est : std::min<uint64_t>(est, (uint64_t)(rare_keys * scale_q));
        if (c->debug)
            std::fprintf(stderr, "libgossgpu: spectrum of 1/%u of the key space over %llu keys (p = %.4f): d %.0f f1 %.0f f2 %.0f f3 %.0f -> plain %llu, fitted %.0f, estimate %llu\n",
                         q, (unsigned long long)scan, p, d, f1, f2, f3, (unsigned long long)plain, model * scale_q, (unsigned long long)est);
    }
    c->mute_timing = mute;
    c->arena.release(mark);
    return est;
}

template <class K>
uint64_t estimate_distinct(goss_gpu_ctx* c, const K* keys, uint64_t n)
{
    if (n < 2) return n;
    if (n <= (4u << 20)) return count_distinct_sample<K>(c, keys, n);
    // a prefix of the chunk stands for the chunk: reads arrive in no particular order
    return spectrum_estimate<K>(c, keys, std::min<uint64_t>(n, 256u << 20), (double)n);
}

// Partition ka on its top `segbits` bits (result back in ka or kb), count every segment in LDS.
// Returns 0 on success; 1 if some segment holds too many distinct keys (retry with more
// partition bits); 2 if the staging area is too small (use the full sort).  The keys stay,
// permuted, in ka or kb (*in_b_out).
template <class K>
int segment_reduce(goss_gpu_ctx* c, K* part, K* spare, uint64_t n, uint32_t segbits, Run* out, const uint64_t* seg_beg = nullptr,
                   const uint64_t* seg_end = nullptr, int big = 0);

template <class K>
int segment_count(goss_gpu_ctx* c, K* ka, K* kb, uint64_t n, uint32_t segbits, bool* in_b_out, Run* out)
{
    const uint32_t keybits = 2 * c->len;
    const uint32_t shift = keybits - segbits;
    const uint32_t npass = (segbits + 7) / 8;
    uint64_t mark = c->arena.mark();
    const unsigned long long* prehist =
        (segbits == kSegBits && c->extract_hist_shift == shift && c->lookback && !*in_b_out) ? c->d_ctr->hist : nullptr;
    K* src = *in_b_out ? kb : ka;
    K* dst = *in_b_out ? ka : kb;
    bool moved = radix_sort<K, false>(c, src, dst, nullptr, nullptr, n, npass, shift, prehist);
    K* part = moved ? dst : src;          // partitioned keys
    K* spare = moved ? src : dst;         // free half of the ping-pong: staging area
    *in_b_out = (part == kb);
    c->arena.release(mark);
    return segment_reduce<K>(c, part, spare, n, segbits, out);
}

// Units (segments) whose counting table overflowed are counted by themselves -- all of them in one sort -- into the staging
// area the kernel filled for the others (round 6).  A FEW giant segments are what skew looks like -- reads with homopolymer stretches:
// every window that begins with nine T's lies in one 17-bit segment, tens of millions of distinct keys where a table
// holds thousands -- and the ladder of "the whole chunk again with more bits" neither splits them (the bits below the
// prefix are all T as well) nor ends before the full sort of the chunk: 3.8 s for 2.5 G windows that now take 0.1 s.
// `expand(unit, first, n, dst)` queues the unit's n keys as full keys at dst.  Returns 0 (all counted: *h updated, the
// device's seg_pos / seg_cnt patched), or the code the caller hands on: 1 -- too many such units, no room for their
// sort, a count beyond 32 bits: the old ladder takes over -- or 2, the staging area is full.
constexpr size_t kMaxOverflowUnits = 4096;
template <class K, class Expand, class UnitLow>
int count_overflowed_units(goss_gpu_ctx* c, uint32_t nunit, const uint64_t* d_beg, const uint64_t* d_end, uint64_t* d_pos, uint64_t* d_cnt,
                           SegOut* h, K* stage_keys, uint32_t* stage_counts, Expand expand, UnitLow unit_low)
{
    // (`unit_low(u)`: the smallest key a key of unit u can be -- the units are ranges of the key space in unit order)
    std::vector<uint64_t> cnt(nunit), beg(nunit), end(nunit);
    HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nunit * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(beg.data(), d_beg, (size_t)nunit * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(end.data(), d_end, (size_t)nunit * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> units;
    uint64_t total = 0;
    for (uint32_t u = 0; u < nunit; ++u)
        if (cnt[u] == kSegOverflowed) { units.push_back(u); total += end[u] - beg[u]; if (units.size() > kMaxOverflowUnits) return 1; }
    if (units.empty() || total == 0) return 1;          // (the flag without a marked unit: not this function's case)
    // ALL of them in one sort (their keys are full keys: the units stay apart) -- one by one, a few hundred small sorts
    // cost half a millisecond of launches and waits each
    const uint64_t need = total * (3 * sizeof(K) + 16) + units.size() * (2 * sizeof(K) + 16) + (64ULL << 20);
    if (c->arena.avail() < need) return 1;
    const uint64_t mark = c->arena.mark(), lo0 = c->arena.lo;
    const size_t maps0 = c->big_maps.size();
    K* a = (K*)c->arena.temp(total * sizeof(K));
    K* b = (K*)c->arena.temp(total * sizeof(K));
    {
        uint64_t at = 0;
        for (uint32_t u : units) { expand(u, beg[u], end[u] - beg[u], a + at); at += end[u] - beg[u]; }
    }
    Run r{nullptr, nullptr, 0};
    {
        // (the sort's passes belong to the counting phase that is being timed; put back also when the sort gives up)
        struct Mute { goss_gpu_ctx* c; bool was; ~Mute() { c->mute_timing = was; } } muted{c, c->mute_timing};
        c->mute_timing = true;
        const bool moved = radix_sort<K, false>(c, a, b, nullptr, nullptr, total, key_digits(c));
        r = reduce_runs<K>(c, moved ? b : a, nullptr, total, moved ? a : b);
    }
    int rc = 0;
    if (r.big >= 0 || c->big_maps.size() != maps0) { c->big_maps.resize(maps0); rc = 1; }          // (a count of 2^32 - 1 or more: the general way keeps those)
    else if (h->cursor + r.m > h->stage_cap) rc = 2;
    else
    {
        const uint32_t nu = (uint32_t)units.size();
        std::vector<K> lows(nu), highs(nu);
        std::vector<uint8_t> last(nu);
        for (uint32_t i = 0; i < nu; ++i)
        {
            lows[i] = unit_low(units[i]);
            last[i] = units[i] + 1 == nunit ? 1 : 0;
            highs[i] = last[i] ? lows[i] : unit_low(units[i] + 1);
        }
        K* d_lows = (K*)c->arena.temp(nu * sizeof(K));
        K* d_highs = (K*)c->arena.temp(nu * sizeof(K));
        uint32_t* d_units = (uint32_t*)c->arena.temp(nu * 4 + 16);
        uint8_t* d_last = (uint8_t*)c->arena.temp(nu + 16);
        HIP_TRY(hipMemcpyAsync(d_lows, lows.data(), nu * sizeof(K), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_highs, highs.data(), nu * sizeof(K), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_units, units.data(), nu * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_last, last.data(), nu, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(stage_keys + h->cursor, r.keys, r.m * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(stage_counts + h->cursor, r.counts, r.m * 4, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(unit_bounds_kernel<K>), dim3(grid_for(nu, kTB)), dim3(kTB), 0, c->stream, (const K*)r.keys, r.m, (const K*)d_lows,
                           (const K*)d_highs, (const uint8_t*)d_last, (const uint32_t*)d_units, nu, (unsigned long long)h->cursor, d_pos, d_cnt);
        HIP_TRY(hipStreamSynchronize(c->stream));          // (the host vectors go)
        h->cursor += r.m;
        h->overflow = 0;
        c->overflow_units += nu;
        if (c->debug) std::fprintf(stderr, "libgossgpu: %u segment(s) with more distinct keys than a table holds counted by sort (%llu keys, %llu distinct)\n",
                                   nu, (unsigned long long)total, (unsigned long long)r.m);
    }
    c->arena.lo = lo0;               // (the run's storage: it lives on in the staging area)
    c->arena.release(mark);
    return rc;
}

// Count every segment of the partitioned keys `part` in LDS; `spare` (n keys) is the staging area.
// Segment s is part[seg_beg[s], seg_end[s]) when the bounds are given (sub-region layout), else
// the bounds are found by binary search in the dense, partitioned array.
template <class K>
int segment_reduce(goss_gpu_ctx* c, K* part, K* spare, uint64_t n, uint32_t segbits, Run* out, const uint64_t* seg_beg,
                   const uint64_t* seg_end, int big)
{
    const uint32_t keybits = 2 * c->len;
    const uint32_t shift = keybits - segbits;
    const uint32_t nseg = 1u << segbits;
    const uint32_t nunit = big > 1 ? nseg << (big - 1) : nseg;      // (segment, round) units of the staging area
    uint64_t mark = c->arena.mark();
    PhaseTimer t(c, GOSS_T_REDUCE, n);
    uint64_t* seg_off = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
    uint64_t* seg_pos = (uint64_t*)c->arena.temp((uint64_t)nunit * 8);
    uint64_t* seg_cnt = (uint64_t*)c->arena.temp(((uint64_t)nunit + 1) * 8);
    uint64_t* seg_dst = (uint64_t*)c->arena.temp(((uint64_t)nunit + 1) * 8);
    SegOut* so = (SegOut*)c->arena.temp(sizeof(SegOut));
    // staged (key,count) pairs share the spare key buffer: cap entries of keys, then the counts
    const uint64_t cap = n * sizeof(K) / (sizeof(K) + 4);
    K* stage_keys = spare;
    uint32_t* stage_counts = (uint32_t*)(spare + cap);
    SegOut hso{};
    hso.stage_cap = cap;
    HIP_TRY(hipMemcpyAsync(so, &hso, sizeof(SegOut), hipMemcpyHostToDevice, c->stream));
    if (!seg_beg)
    {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_bounds_kernel<K>), dim3(nseg / 256 + 1), dim3(256), 0, c->stream,
                           (const K*)part, n, shift, nseg, seg_off);
        seg_beg = seg_off; seg_end = seg_off + 1;
    }
    launch_seg_hash(c, nseg, (const K*)part, seg_beg, seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, shift, big);
    check_launch("segment counting kernel");
    SegOut* h = (SegOut*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, so, sizeof(SegOut), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h->overflow == 1u && big <= 1 && c->overflow_by_sort)
    {
        // (tables that overflowed, nothing else: those segments by sort -- not the forms in which several workgroups
        // share a segment, whose units are not ranges of `part`)
        SegOut hs = *h;
        const int rc = count_overflowed_units<K>(c, nunit, seg_beg, seg_end, seg_pos, seg_cnt, &hs, stage_keys, stage_counts,
            [&](uint32_t u, uint64_t first, uint64_t cnt_u, K* dst) {
                if constexpr (std::is_same<K, Key2>::value)
                {
                    if (big == -3)          // (12-byte records of the remainders: the second level's packed form)
                    {
                        hipLaunchKernelGGL(expand_rem96_kernel, dim3(grid_for(cnt_u, kTB)), dim3(kTB), 0, c->stream,
                                           reinterpret_cast<const Rem96*>(part) + first, cnt_u, (uint64_t)u, shift, dst);
                        return;
                    }
                }
                HIP_TRY(hipMemcpyAsync(dst, part + first, cnt_u * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
            },
            [&](uint32_t u) -> K {
                if constexpr (std::is_same<K, Key2>::value)
                {
                    const unsigned __int128 v = (unsigned __int128)u << shift;
                    return Key2{(uint64_t)v, (uint64_t)(v >> 64)};
                }
                else return Key1{(uint64_t)u << shift};
            });
        *h = hs;
        if (rc == 2) h->overflow = 2u;
    }
    if (h->overflow)
    {
        const int why = (h->overflow & 1u) ? 1 : 2;
        t.stop();
        c->arena.release(mark);
        return why;
    }
    const uint64_t m = h->cursor;
    HIP_TRY(hipMemcpyAsync(seg_dst, seg_cnt, (uint64_t)nunit * 8, hipMemcpyDeviceToDevice, c->stream));
    exclusive_scan_u64(c, seg_dst, nunit);
    out->m = m;
    out->keys = c->arena.perm(std::max<uint64_t>(m, 1) * sizeof(K));
    out->counts = (uint32_t*)c->arena.perm(std::max<uint64_t>(m, 1) * 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_gather_kernel<K>), unit_grid(nunit), dim3(kTB), 0, c->stream,
                       (const K*)stage_keys, (const uint32_t*)stage_counts, (const uint64_t*)seg_pos,
                       (const uint64_t*)seg_dst, (const uint64_t*)seg_cnt, (K*)out->keys, out->counts);
    t.stop();
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
    return 0;
}

// The same for the 32-bit-remainder form (subpart32_kernel's output): nseg = 2^17 .. 2^20 segments of u32 remainders, counted by
// seg_hash_reduce32_kernel in tables of `slots` slots; `spare` (n one-word keys) is the staging area.
// Returns 0, 1 (a table overflowed) or 2 (staging area too small) like segment_reduce.
int segment_reduce32(goss_gpu_ctx* c, const uint32_t* rems, Key1* spare, uint64_t n, Run* out, const uint64_t* seg_beg,
                     const uint64_t* seg_end, int slots, bool squeeze, uint32_t rbits, uint32_t sqbit, uint32_t nseg, uint32_t split_bits)
{
    uint64_t mark = c->arena.mark();
    PhaseTimer t(c, GOSS_T_REDUCE, n);
    uint64_t* seg_pos = (uint64_t*)c->arena.temp((uint64_t)nseg * 8);
    uint64_t* seg_cnt = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
    uint64_t* seg_dst = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
    SegOut* so = (SegOut*)c->arena.temp(sizeof(SegOut));
    const uint64_t cap = n * sizeof(Key1) / (sizeof(Key1) + 4);
    Key1* stage_keys = spare;
    uint32_t* stage_counts = (uint32_t*)(spare + cap);
    SegOut hso{};
    hso.stage_cap = cap;
    HIP_TRY(hipMemcpyAsync(so, &hso, sizeof(SegOut), hipMemcpyHostToDevice, c->stream));
#define GOSS_LAUNCH_R32(KERNEL, SLOTS, SQ)                                                                               \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<SLOTS, SQ>), unit_grid(nseg), dim3(SLOTS <= 4096 ? kTB : SLOTS / 4096 * kTB), 0, c->stream, rems, seg_beg, \
                       seg_end, so, seg_pos, seg_cnt, stage_keys, stage_counts, rbits, sqbit, split_bits)
    // (round 5: buckets of four remainders, home bucket only in the fast path; GOSS_GPU_R32_FORM=0: the pair layout, whose
    // largest table has 4 096 slots)
    if (c->r32_form)
    {
        if (slots == 2048) { if (squeeze) GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 2048, true); else GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 2048, false); }
        else if (slots == 8192) { if (squeeze) GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 8192, true); else GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 8192, false); }
        else if (slots == 16384) { if (squeeze) GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 16384, true); else GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 16384, false); }
        else { if (squeeze) GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 4096, true); else GOSS_LAUNCH_R32(seg_hash_reduce32b_kernel, 4096, false); }
    }
    else
    {
        if (slots == 2048) { if (squeeze) GOSS_LAUNCH_R32(seg_hash_reduce32_kernel, 2048, true); else GOSS_LAUNCH_R32(seg_hash_reduce32_kernel, 2048, false); }
        else { if (squeeze) GOSS_LAUNCH_R32(seg_hash_reduce32_kernel, 4096, true); else GOSS_LAUNCH_R32(seg_hash_reduce32_kernel, 4096, false); }
    }
#undef GOSS_LAUNCH_R32
    check_launch("32-bit segment counting kernel");
    SegOut* h = (SegOut*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, so, sizeof(SegOut), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
#if defined(GOSS_STAMPS)
    if (h->stamps[7])
        std::fprintf(stderr, "libgossgpu: counting stamps per segment (cycles of wave 0): set-up %.0f  loads+copy %.0f  fast path %.0f  slow path %.0f  closing barrier %.0f  ordering %.0f  write-out %.0f   (%llu segments)\n",
                     (double)h->stamps[0] / h->stamps[7], (double)h->stamps[1] / h->stamps[7], (double)h->stamps[2] / h->stamps[7], (double)h->stamps[3] / h->stamps[7],
                     (double)h->stamps[4] / h->stamps[7], (double)h->stamps[5] / h->stamps[7], (double)h->stamps[6] / h->stamps[7], (unsigned long long)h->stamps[7]);
#endif
    if (h->overflow == 1u && c->overflow_by_sort)
    {
        SegOut hs = *h;
        const int rc = count_overflowed_units<Key1>(c, nseg, seg_beg, seg_end, seg_pos, seg_cnt, &hs, stage_keys, stage_counts,
            [&](uint32_t u, uint64_t first, uint64_t cnt_u, Key1* dst) {
                const uint64_t prefix = (uint64_t)(u >> split_bits) << rbits;          // (as the counting kernel's write-out)
                if (squeeze)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(expand_rem32_kernel<true>), dim3(grid_for(cnt_u, kTB)), dim3(kTB), 0, c->stream, rems + first, cnt_u, prefix, sqbit, dst);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(expand_rem32_kernel<false>), dim3(grid_for(cnt_u, kTB)), dim3(kTB), 0, c->stream, rems + first, cnt_u, prefix, sqbit, dst);
            },
            [&](uint32_t u) -> Key1 {
                // the unit's smallest key: its segment's prefix, and the unit's number in the top split_bits of the remainder
                const uint32_t rem_bits = rbits - (squeeze ? 1u : 0u);
                const uint32_t sub = split_bits ? (u & ((1u << split_bits) - 1u)) << (rem_bits - split_bits) : 0u;
                const uint64_t low = squeeze ? rem32_unpack<true>(sub, sqbit) : rem32_unpack<false>(sub, sqbit);
                return Key1{((uint64_t)(u >> split_bits) << rbits) | low};
            });
        *h = hs;
        if (rc == 2) h->overflow = 2u;
    }
    if (h->overflow)
    {
        const int why = (h->overflow & 1u) ? 1 : 2;
        t.stop();
        c->arena.release(mark);
        return why;
    }
    const uint64_t m = h->cursor;
    HIP_TRY(hipMemcpyAsync(seg_dst, seg_cnt, (uint64_t)nseg * 8, hipMemcpyDeviceToDevice, c->stream));
    exclusive_scan_u64(c, seg_dst, nseg);
    out->m = m;
    out->keys = c->arena.perm(std::max<uint64_t>(m, 1) * sizeof(Key1));
    out->counts = (uint32_t*)c->arena.perm(std::max<uint64_t>(m, 1) * 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_gather_kernel<Key1>), unit_grid(nseg), dim3(kTB), 0, c->stream,
                       (const Key1*)stage_keys, (const uint32_t*)stage_counts, (const uint64_t*)seg_pos,
                       (const uint64_t*)seg_dst, (const uint64_t*)seg_cnt, (Key1*)out->keys, out->counts);
    t.stop();
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
    return 0;
}

// Count a chunk of n extracted keys (in ka): segment path with as many partition bits as the
// estimated number of distinct keys asks for, more bits after an overflow, finally the full
// LSD sort + run compaction.
template <class K>
Run count_keys(goss_gpu_ctx* c, K* ka, K* kb, uint64_t n)
{
    Run r{nullptr, nullptr, 0};
    bool in_b = false;
    if (use_segment_path<K>(c))
    {
        const uint32_t keybits = 2 * c->len;
        const uint64_t limit = SegCfg<K>::kLimit;
        const uint64_t m_est = n <= (uint64_t)(1u << kSegBits) * (limit / 2) ? n : estimate_distinct<K>(c, ka, n);
        // worth it only with real duplication (and the staging area needs M <= ~2n/3)
        if (m_est != 0 && (m_est <= n / 3 || n <= (uint64_t)(1u << kSegBits) * (limit / 2)))
        {
            uint32_t segbits = kSegBits;
            while (segbits < kSegBitsMax && (m_est >> segbits) > limit * 3 / 4) segbits += 4;   // margin for skew
            for (; segbits <= kSegBitsMax && segbits + 8 <= keybits; segbits += 4)
            {
                if ((m_est >> segbits) > limit) continue;
                int rc = segment_count<K>(c, ka, kb, n, segbits, &in_b, &r);
                if (c->debug) std::fprintf(stderr, "libgossgpu: count_keys: %llu keys, %u segment bits -> %d (%llu distinct)\n",
                                           (unsigned long long)n, segbits, rc, (unsigned long long)(rc == 0 ? r.m : 0));
                if (rc == 0) return r;
                if (rc == 2) break;
                c->segment_retries++;
            }
        }
    }
    K* src = in_b ? kb : ka;
    K* dst = in_b ? ka : kb;
    bool moved = radix_sort<K, false>(c, src, dst, nullptr, nullptr, n, key_digits(c));
    PhaseTimer t(c, GOSS_T_REDUCE, n);
    r = reduce_runs<K>(c, moved ? dst : src, nullptr, n, moved ? src : dst);
    t.stop();
    if (c->debug) std::fprintf(stderr, "libgossgpu: count_keys: full sort of %llu keys -> %llu distinct\n", (unsigned long long)n, (unsigned long long)r.m);
    return r;
}

// ---- fused extraction + first partition pass (one-word canonical keys) ----------------------
// Returns true when it counted the chunk (run appended, counters updated); false = not
// applicable or a region overflowed / a later stage asked for a retry: the caller then runs the
// unfused sequence on the same (untouched) input.
//
// Two forms.  LSD (any number of partition digits): the fused kernel partitions on the lowest
// partition digit into 256 bucket regions, the remaining digits are look-back passes (the first
// of them reads the regions).  MSD (exactly two digits, the common case): the fused kernel
// partitions on the HIGH digit and the second pass places the keys of region b by their LOW digit
// into 65 536 sub-regions -- the segments of the counting kernel -- with atomic cursors: no
// look-back chain and no digit histograms at all.  Region and sub-region sizes come from a sample
// of the input (the whole chunk when it is small), with 5 / 6 standard deviations of slack.
template <class K>
int segment_reduce(goss_gpu_ctx* c, K* part, K* spare, uint64_t n, uint32_t segbits, Run* out, const uint64_t* seg_beg,
                   const uint64_t* seg_end, int big);

// A run counted in strand-representative space (extract1_part_kernel, MODE 0) -> the run the rest of
// the library expects: every key replaced by gossamer's canonical form (the strand with the smaller
// FNV-1a hash), then (key,count) pairs put back in key order.  The map is a bijection between the
// two choices of representative, so counts carry over and no two entries collide.
template <class K>
void canonicalize_run(goss_gpu_ctx* c, Run& r)
{
    const uint64_t m = r.m;
    r.rep = false;
    if (m == 0) return;
    PhaseTimer t(c, GOSS_T_ORDER, m);
    // the run's own storage is one side of the sort's ping-pong, a temporary copy the other
    const uint64_t need = m * sizeof(K) + m * 4 + 512;
    if (c->arena.avail() < need + (256ULL << 20)) grow_arena(c, need + (256ULL << 20));      // (rebases r: it lives in c->runs)
    uint64_t mark = c->arena.mark();
    K* ka = (K*)r.keys;
    uint32_t* va = r.counts;
    K* kb = (K*)c->arena.temp(m * sizeof(K));
    uint32_t* vb = (uint32_t*)c->arena.temp(m * 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(canonical_map_kernel<K>), dim3(grid_for(m, kTB)), dim3(kTB), 0, c->stream,
                       (const K*)ka, ka, m, c->len);
    const bool mute = c->mute_timing;
    c->mute_timing = true;                  // the sort's passes belong to this phase, not to the partition classes
    bool in_b = false, ordered = false;
    const uint32_t keybits = 2 * c->len;
    if constexpr (std::is_same<K, Key1>::value)
    {
        // canonical forms are uniform on their leading bits: two (three) radix passes group the pairs by their
        // top 16 (17 .. 24) bits, then every group (~m / 2^bits pairs, at most 4 096) is ordered in LDS -- three or
        // four passes over the pairs instead of one per key byte.  More than 16 bits for runs above 157 M keys
        // (reads with errors; the ranks of a multi-GPU build hold up to the whole k-mer set before the exchange)
        uint32_t sb = 16;
        while (sb < 24 && m > (1ULL << sb) * 2400) ++sb;                // the fewest groups of at most ~2 400 pairs
        if (c->order_bits) sb = c->order_bits;
        if (keybits >= sb + 10 && m >= (1u << 20) && m <= (1ULL << sb) * 2400)
        {
            const uint32_t nseg = 1u << sb;
            in_b = radix_sort<K, true>(c, ka, kb, va, vb, m, (sb + 7) / 8, keybits - sb);
            K* sk = in_b ? kb : ka;
            uint32_t* sv = in_b ? vb : va;
            uint64_t m2 = c->arena.mark();
            uint64_t* soff = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
            uint32_t* flag = (uint32_t*)c->arena.temp(16);
            HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_bounds_kernel<K>), dim3(nseg / 256 + 1), dim3(256), 0, c->stream,
                               (const K*)sk, m, keybits - sb, nseg, soff);
            hipLaunchKernelGGL(seg_sort_pairs_kernel, unit_grid(nseg), dim3(kTB), 0, c->stream, sk, sv, (const uint64_t*)soff,
                               keybits - sb, flag);
            uint32_t* hf = (uint32_t*)c->h_pinned;
            HIP_TRY(hipMemcpyAsync(hf, flag, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            ordered = hf[0] == 0;
            c->arena.release(m2);
            if (c->debug) std::fprintf(stderr, "libgossgpu: canonical order of %llu keys by %u-bit groups: %s\n", (unsigned long long)m, sb, ordered ? "done" : "skewed, full sort");
            if (!ordered)
            {
                // skewed bits: order everything by the remaining digits as well (the top bits are in place,
                // a full stable sort from the current buffer is simply the general answer)
                if (in_b) { std::swap(ka, kb); std::swap(va, vb); }
                in_b = false;
            }
        }
    }
    if (!ordered) in_b = radix_sort<K, true>(c, ka, kb, va, vb, m, key_digits(c));
    c->mute_timing = mute;
    if ((in_b ? kb : ka) != (K*)r.keys)
    {
        HIP_TRY(hipMemcpyAsync(r.keys, in_b ? kb : ka, m * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(r.counts, in_b ? vb : va, m * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    t.stop();
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
}

// A run of a graph build counted in strand-pair space (one representative per window, process_chunk_fused) -> both
// strands: run `ri` keeps its keys (palindromic edges get their count doubled: the adapter yields them twice per
// window), and a second run holds the reverse complements with the same counts, sorted; both are in the full key space
// and disjoint, the caller merges them.  Exact counts beyond 32 bits are mirrored in the runs' maps.
template <class K>
void expand_graph_run(goss_gpu_ctx* c, size_t ri)
{
    const uint64_t m = c->runs[ri].m;
    c->runs[ri].rep = false;
    if (m == 0) return;
    PhaseTimer t(c, GOSS_T_ORDER, m);
    {
        const uint64_t need = 3 * m * (sizeof(K) + 4) + (256ULL << 20);
        if (c->arena.avail() < need) grow_arena(c, need);          // (rebases the runs: fetched by index below)
    }
    uint64_t mark = c->arena.mark();
    K* bk = (K*)c->arena.temp(m * sizeof(K));
    uint32_t* bc = (uint32_t*)c->arena.temp(m * 4);
    K* tk = (K*)c->arena.temp(m * sizeof(K));
    uint32_t* tc = (uint32_t*)c->arena.temp(m * 4);
    unsigned long long* dbig = (unsigned long long*)c->arena.temp((2 + 3 * kMaxBig) * 8);
    unsigned long long* npal = dbig + 1 + 3 * kMaxBig;
    HIP_TRY(hipMemsetAsync(dbig, 0, (2 + 3 * kMaxBig) * 8, c->stream));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(graph_expand_kernel<K>), dim3(grid_for(m, kTB)), dim3(kTB), 0, c->stream, (const K*)c->runs[ri].keys,
                       c->runs[ri].counts, m, c->len, bk, bc, dbig, kMaxBig, npal);
    const bool mute = c->mute_timing;
    c->mute_timing = true;                  // the sort's passes belong to this phase
    // (the pads -- all ones, in the slots of the palindromes -- must end up BEHIND every key: where the key's 2 len bits
    // fill whole digits (len a multiple of four: 12, 16, .. 28, 32, ..) the edge T..T has every digit a pad has, a stable
    // sort on those digits alone leaves the two kinds in input order, and the cut below dropped T..T for a pad whenever a
    // palindrome stood in front of it: one digit more -- zeros for a key, ones for a pad -- tells them apart)
    const uint32_t nd = key_digits(c) + ((2 * c->len) % 8 == 0 ? 1u : 0u);
    const bool in_b = radix_sort<K, true>(c, bk, tk, bc, tc, m, nd);
    c->mute_timing = mute;
    std::vector<unsigned long long> hb(2 + 3 * kMaxBig);
    HIP_TRY(hipMemcpyAsync(hb.data(), dbig, hb.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t npads = hb[1 + 3 * kMaxBig];
    if (hb[0] > kMaxBig) throw StatusError{GOSS_ERR_COUNT_OVERFLOW, "more than 256 keys occurred 2^32 times or more"};
    const uint64_t mb = m - npads;
    // exact counts: the representative's entry stays (doubled for a palindrome), its reverse complement gets the same
    BigMap ma, mbm;
    if (c->runs[ri].big >= 0)
        for (const auto& kv : c->big_maps[c->runs[ri].big])
        {
            K k;
            if constexpr (sizeof(K) == 8) k = K{kv.first.second}; else k = K{kv.first.second, kv.first.first};
            const K r = revcomp(k, c->len);
            if (r == k) ma[kv.first] = 2 * kv.second;
            else { ma[kv.first] = kv.second; mbm[std::make_pair((uint64_t)key_hi_word(r), (uint64_t)key_lo_word(r))] = kv.second; }
        }
    for (uint64_t i = 0; i < hb[0]; ++i) ma[std::make_pair((uint64_t)hb[2 + 3 * i], (uint64_t)hb[1 + 3 * i])] = hb[3 + 3 * i];
    c->runs[ri].big = -1;
    if (!ma.empty()) { c->big_maps.push_back(std::move(ma)); c->runs[ri].big = (int)c->big_maps.size() - 1; }
    if (mb)
    {
        Run b{nullptr, nullptr, mb};
        b.keys = c->arena.perm(mb * sizeof(K));
        b.counts = (uint32_t*)c->arena.perm(mb * 4);
        HIP_TRY(hipMemcpyAsync(b.keys, in_b ? tk : bk, mb * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(b.counts, in_b ? tc : bc, mb * 4, hipMemcpyDeviceToDevice, c->stream));
        if (!mbm.empty()) { c->big_maps.push_back(std::move(mbm)); b.big = (int)c->big_maps.size() - 1; }
        c->runs.push_back(b);
    }
    t.stop();
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
    if (c->debug) std::fprintf(stderr, "libgossgpu: %llu strand pairs expanded (%llu palindromes)\n", (unsigned long long)m, (unsigned long long)npads);
}

// Returns kFusedDone, kFusedDeclined (the caller runs the unfused sequence) or kFusedNeedFull (the
// key buffers were sized for fewer valid windows than the sample shows: the caller retries with
// buffers of one key per window start).
enum { kFusedDeclined = 0, kFusedDone = 1, kFusedNeedFull = 2 };
constexpr uint32_t kFusedGrid = 256 * GOSS_E1_OCC;                         // workgroups of extract1_part_kernel: 3 per CU (52 KB of LDS each)
#ifndef GOSS_FUSED_NKEYS2
#define GOSS_FUSED_NKEYS2 14          // keys per thread of extract2_part_kernel (tile of 3584 keys + carry = 70 KB of LDS)
#endif
#ifndef GOSS_FUSED_GRID2
#define GOSS_FUSED_GRID2 512
#endif
constexpr uint32_t kFusedGrid2 = GOSS_FUSED_GRID2;                        // ... of extract2_part_kernel: 2 per CU (75 KB)
constexpr double kValidSlackA = 1.06, kValidSlackB = 1.17;   // key buffer slots per expected key (bucket regions; sub-regions with their six sigma each: 1.149 measured on C4's two-word keys)
constexpr uint64_t kValidSizingMin = 640u << 20;             // window starts: smaller chunks are sampled whole into a full buffer

template <class K>
int process_chunk_fused(goss_gpu_ctx* c, const uint8_t* d_bases, uint64_t nstarts, uint64_t navail, K* ka, uint64_t ka_slots,
                        K* kb, uint64_t kb_slots)
{
    constexpr bool kOne = std::is_same<K, Key1>::value;          // one-word keys
    const uint32_t keybits = 2 * c->len;
    if (c->rec_mode && !kOne && c->mode == GOSS_MODE_GRAPH && !c->graph_rep) return kFusedDeclined;          // (the record form extracts one key per window)
    // Graph mode wants both strands of every window (ReverseComplementAdapter.hh:34-55), and both always come together:
    // the fused path counts ONE strand representative per window -- half the keys through the partition and the tables
    // -- and the run is expanded into both strands after counting (expand_graph_run).  Inside this function such a chunk
    // is a k-mer-set chunk of (k+1)-mers in representative space.
    const bool rep_graph = c->mode == GOSS_MODE_GRAPH && c->graph_rep;
    const bool rep_kmer = c->mode != GOSS_MODE_GRAPH && kOne;    // (one-word k-mer sets: representatives, canonical forms after counting)
    const bool use_rep = rep_graph || rep_kmer;
    if (!c->fused || c->path != 0 || !c->lookback || c->ordered_tiles ||
        c->extract_v1 || nstarts < c->fused_min || keybits < (uint32_t)kSegBits + 8)
        return kFusedDeclined;
    constexpr int kTile = SortCfg<K, false>::kTile;
    uint64_t mark = c->arena.mark();
    struct Release { goss_gpu_ctx* c; uint64_t m; ~Release() { c->arena.release(m); } } release{c, mark};
    auto decline = [&](const char* why) {
        if (c->debug) std::fprintf(stderr, "libgossgpu: fused path declined (%s), %llu window starts\n", why, (unsigned long long)nstarts);
        return (int)kFusedDeclined;
    };

    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!c->debug) return;
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::fprintf(stderr, "libgossgpu: fused path: %-28s at %8.3f ms\n", what,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };

    // 1. a sample of the keys: slices spread evenly over the chunk, extracted with the plain
    //    kernel.  The two-level form needs the joint histogram of two digits (65 536 bins), hence
    //    a larger sample: 1/64 of the chunk but at least 160 M window starts; a chunk of up to 640 M
    //    window starts is sampled whole (exact sizes, +4 % extraction work at most).
    const bool want_msd = c->fused_msd;
    uint64_t sample_starts = nstarts <= (16u << 20) ? nstarts : (4u << 20);
    if (want_msd) sample_starts = nstarts <= (640u << 20) ? nstarts : std::max<uint64_t>(160u << 20, nstarts / 64);
    // keys that the 32-bit-remainder forms take (by their width): half the sample -- their sub-regions are 4-byte slots,
    // which have the room for the wider six-sigma margins of a smaller sample (22 % instead of 15 % on C2), and the
    // sample's extraction, spectrum and joint histogram are 3 ms of a 97 ms step
    {
        bool width_ok = false;
        if (kOne && c->rem32 && keybits >= 8 + 9 + 8)
            for (uint32_t b2 = kSub32BitsMin; b2 <= (uint32_t)kSub32BitsMax; ++b2)
            {
                const uint32_t rb = keybits - 8 - b2;
                const bool sq = (rep_kmer || rep_graph) && (c->len & 1u) && rb == 33;
                width_ok = width_ok || rb - (sq ? 1u : 0u) <= 32;
            }
        if (want_msd && width_ok && nstarts > (640u << 20)) sample_starts = std::max<uint64_t>(80u << 20, nstarts / 128);
    }
    // slices are whole super-tiles of the plain kernel (32 768 window starts) and lie a multiple
    // of 16 bytes apart, so that ONE strided launch extracts them all
    const bool graph_mode = c->mode == GOSS_MODE_GRAPH && !rep_graph;
    // window starts per super-tile of the plain kernels: extract1_kernel<0,16,8> / <1,8,8>, extract2_kernel<0,8,8> / <1,4,8>
    const uint64_t kPlainSuper = 8ULL * kTB * (kOne ? (graph_mode ? 8 : 16) : (graph_mode ? 4 : 8));
    // a slice is ONE super-tile (~217 reads of 150 bp): thousands of slices follow a drifting
    // k-mer distribution (sorted inputs) far better than a few long ones
    const uint64_t nslices = sample_starts >= nstarts ? 1 : std::max<uint64_t>(64, sample_starts / kPlainSuper);
    const uint64_t slice_starts = sample_starts >= nstarts ? nstarts : kPlainSuper;
    if (nslices > 1 && nstarts < 4 * nslices * slice_starts) return decline("chunk smaller than the sample");
    // key buffers sized from the estimated share of valid windows (process_chunk): they must hold
    // the sample whatever it contains
    const uint64_t kps = graph_mode ? 2 : 1;
    const bool reduced = ka_slots < nstarts * kps || kb_slots < nstarts * kps;
    if (reduced && (nslices == 1 || nslices * slice_starts * kps > std::min(ka_slots, kb_slots))) return (int)kFusedNeedFull;
    const uint64_t slice_stride = nslices > 1 ? ((nstarts - slice_starts) / (nslices - 1)) & ~15ULL : 0;
    // rep: one-word k-mer sets in strand-representative space (what the fused kernel counts in, unless canon_l1 below)
    auto extract_sample = [&](bool rep) {
    c->mute_timing = true;
    HIP_TRY(hipMemsetAsync(c->d_ctr, 0, sizeof(ExtractCounters), c->stream));
    {
        const uintptr_t addr0 = (uintptr_t)d_bases;
        const uint32_t mis0 = c->rec_mode ? 0u : (uint32_t)(addr0 & 15u);          // (records are taken where they lie)
        // (a packed string whose sample is the whole chunk: the slice kernels below, told to take every tile in turn --
        // they have a packed form, the plain kernels behind extract_dispatch read bytes)
        const bool whole_pk = nslices == 1 && c->pk.on && !c->rec_mode;
        if (nslices == 1 && !whole_pk)
        {
            c->extract_rep = use_rep && rep;
            extract_dispatch<K>(c, (const uint8_t*)(addr0 - mis0), mis0, nstarts, navail, ka);
            c->extract_rep = false;
        }
        else if (!kOne && c->rec_mode)
        {
            // slices of kPlainSuper window slots of two-word records
            const uint64_t P = rec_slots(c);
            const uint64_t slice_groups = std::max<uint64_t>(1, slice_starts / P / kRecGroup);
            launch_extract_records2(c, (const SkRec2*)d_bases, nstarts / P, (Key2*)ka, slice_groups * nslices, slice_groups, slice_stride / P,
                                    rep_graph && rep);
        }
        else if constexpr (!kOne)
        {
            const uint64_t slice_tiles = whole_pk ? 0 : slice_starts / kPlainSuper, nsuper = whole_pk ? (nstarts + kPlainSuper - 1) / kPlainSuper : slice_tiles * nslices;
            const uint32_t grid = (uint32_t)std::min<uint64_t>(nsuper, 2048);
            (void)grid;
            if (graph_mode) launch_extract2<1, 4, 8>(c, (const uint8_t*)(addr0 - mis0), mis0, nstarts, navail, ka, slice_tiles, slice_stride, nsuper);
            else launch_extract2<0, 8, 8>(c, (const uint8_t*)(addr0 - mis0), mis0, nstarts, navail, ka, slice_tiles, slice_stride, nsuper, rep_graph && rep);
        }
        else if (c->rec_mode)
        {
            // slices of kPlainSuper window slots = 2048 records each, in strand-representative space for k-mer sets
            const uint64_t P = rec_slots(c);
            const uint64_t slice_groups = slice_starts / P / kRecGroup;
            launch_extract_records(c, (const SkRec*)d_bases, nstarts / P, (Key1*)ka, slice_groups * nslices, slice_groups, slice_stride / P,
                                   use_rep && rep);
        }
        else
        {
            const uint64_t slice_tiles = whole_pk ? 0 : slice_starts / kPlainSuper, nsuper = whole_pk ? (nstarts + kPlainSuper - 1) / kPlainSuper : slice_tiles * nslices;
            const uint32_t grid = (uint32_t)std::min<uint64_t>(nsuper, 2048);
            const uint8_t* src = (const uint8_t*)(addr0 - mis0); const uint16_t* pbad = nullptr;
            if (c->pk.on) pk_ptrs(c, src, &src, &pbad);
#define GOSS_LAUNCH_SAMPLE1(MODE, P, REP, PK)                                                                         \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(extract1_kernel<MODE, P, 8, 8, REP, PK>), dim3(grid), dim3(kTB), 0, c->stream, src, mis0, nstarts, navail, \
                       c->len, ka, c->d_ctr, 0xFFFFFFFFu, nsuper, slice_tiles, slice_stride, pbad)
            if (graph_mode) { if (c->pk.on) GOSS_LAUNCH_SAMPLE1(1, 8, false, true); else GOSS_LAUNCH_SAMPLE1(1, 8, false, false); }
            else if (rep)      // strand representatives: the key space the fused kernel counts in
            { if (c->pk.on) GOSS_LAUNCH_SAMPLE1(0, 16, true, true); else GOSS_LAUNCH_SAMPLE1(0, 16, true, false); }
            else { if (c->pk.on) GOSS_LAUNCH_SAMPLE1(0, 16, false, true); else GOSS_LAUNCH_SAMPLE1(0, 16, false, false); }
#undef GOSS_LAUNCH_SAMPLE1
        }
    }
    c->extract_hist_shift = 0xFFFFFFFFu;
    ExtractCounters* hcs = (ExtractCounters*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(hcs, c->d_ctr, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->mute_timing = false;
    return (uint64_t)hcs->keys_out;
    };
    const uint64_t ns = extract_sample(true);
    if (ns < (1u << 20)) return decline("mostly non-bases");
    lap("sample extracted");
    const bool exact = nslices == 1;                                   // the sample is the chunk
    const double scale = (double)nstarts / (double)(nslices * slice_starts);
    const uint64_t n_exp = (uint64_t)((double)ns * scale);          // expected number of keys
    uint64_t m_rare = 0;
    uint64_t m_est = spectrum_estimate<K>(c, ka, ns, (double)n_exp, &m_rare);
    if (c->est_scale != 1.0) m_est = (uint64_t)((double)m_est * c->est_scale);      // tests: a wrong estimate on purpose
    m_rare = std::min(m_rare, m_est);
    lap("distinct keys estimated");
    if (m_est == 0 || m_est > n_exp / 3) return decline("too little duplication for the segment path");
    // One-word k-mer sets are counted as strand representatives and mapped to gossamer's canonical form afterwards -- a
    // re-ordering of the DISTINCT keys (0.06 ms per million), cheap beside two FNV hashes per WINDOW in the first level
    // (+ ~2.4 ms per 10^9 windows) while distinct keys are few.  Reads with many errors turn that round (2e9 distinct
    // 25-mers of 12.6e9 windows: 116 ms of re-ordering against ~30 ms of hashing): from 5 % distinct keys per window on
    // (canon_l1_at) the first level computes the canonical form itself and the run needs no re-ordering.
    // A chunk that is ONE OF SEVERAL shares the re-ordering with the others -- the runs are merged in representative
    // space and re-ordered once -- so what counts is the whole build: its windows W (what the caller said it will push,
    // goss_gpu_expect_bases; else what has been counted plus this chunk, times four when more is known to follow) and
    // its distinct keys D = the chunk's frequent keys (more of the same input mostly brings THEM again) + its keys of
    // multiplicity ~1 scaled to W (every chunk brings its own).  C2 from FASTQ: thirteen chunks of 0.8 G windows each
    // see all 10^8 k-mers of the genome (12 % of their windows) -- per chunk that read "canonical", 67 ms of first
    // level where representatives take 31, ten second-level bits, a re-ordering per run; of the build's 12.6 G windows
    // they are 0.8 %.  Once a chunk has chosen, the chunks that follow count in the same space while its run waits:
    // a run in canonical space among runs of representatives sends every one of those through a re-ordering of its own.
    bool canon_auto = (double)m_est > c->canon_l1_at * (double)n_exp;
    if (rep_kmer && c->canon_l1 == 1)
    {
        if (c->space_choice >= 0 && !c->runs.empty()) canon_auto = c->space_choice == 1;
        else
        {
            const double w_c = (double)n_exp;
            // (what follows: known for the rest of the push being counted; a guess -- three times as much again -- when
            // that push is a staging buffer that filled up under a caller who goes on pushing)
            const double w_push = w_c + (double)c->more_starts * std::min(1.0, c->valid_frac);          // this push, from this chunk on
            double w_all = (double)c->windows + w_push * (c->more_follows ? 4.0 : 1.0);
            if (c->expect_bases) w_all = std::max(w_all, (double)c->expect_bases * std::min(1.0, c->valid_frac));
            const double d_all = (double)(m_est - m_rare) + (double)m_rare * (w_all / w_c);
            canon_auto = d_all > c->canon_l1_at * w_all;
            if (c->debug) std::fprintf(stderr, "libgossgpu: fused path: %.0f distinct keys (%.0f of multiplicity ~1) of %.0f windows here, %.0f of %.0f in all: %s\n",
                                       (double)m_est, (double)m_rare, w_c, d_all, w_all, canon_auto ? "canonical forms in the first level" : "strand representatives");
        }
    }
    const bool canon_l1 = rep_kmer && (c->canon_l1 == 2 || (c->canon_l1 == 1 && canon_auto));
    if (rep_kmer && c->canon_l1 == 1) c->space_choice = canon_l1 ? 1 : 0;
    if (canon_l1)
    {
        // (the regions are sized from the sample: it must be in the key space the first level writes)
        const uint64_t ns2 = extract_sample(false);
        if (ns2 != ns) throw StatusError{GOSS_ERR_HIP, "fused path: the sample changed between two extractions"};
        lap("sample extracted again (canonical forms)");
    }
    // buffers sized from the estimated share of valid windows must hold what the sample promises
    {
        if (reduced && ((double)n_exp * 1.035 + 262144.0 > (double)ka_slots || (double)n_exp * 1.02 + 6.0e6 > (double)kb_slots))
        {
            if (c->debug) std::fprintf(stderr, "libgossgpu: fused path: %llu keys expected, buffers of %llu / %llu slots too small\n",
                                       (unsigned long long)n_exp, (unsigned long long)ka_slots, (unsigned long long)kb_slots);
            return (int)kFusedNeedFull;
        }
    }
    const uint64_t limit = SegCfg<K>::kLimit;
    uint32_t segbits = kSegBits;
    while (segbits < (uint32_t)kSegBitsMax && (m_est >> segbits) > limit * 3 / 4) segbits += 4;
    // one-word keys: between 3/4 of the small table and 3/4 of the big one per 16-bit segment, the
    // two-level form with the big counting table saves the third partition digit
    // (up to 4 workgroups sharing a segment, each counting the keys of one value of the next bits)
    int big_table = 0;
    if (kOne && c->fused_msd && c->big_table && segbits > (uint32_t)kSegBits && keybits >= (uint32_t)kSegBits + 8 + 2)
        for (int r = std::max(0, c->big_rounds_min); r <= c->big_rounds_max; ++r)
            if ((m_est >> (kSegBits + r)) <= (uint64_t)kSegBigLimit * 17 / 20) { segbits = kSegBits; big_table = 1 + r; break; }   // (an overflow costs one more counting pass, no more)
    // two-word keys: the 4096-slot table, up to two workgroups per segment (a pass over 16-byte
    // keys costs more than one over 8-byte keys)
    // two-word keys whose bits below a 16-bit prefix fit 96: 16-byte slots, 8192 of them
    if (!kOne && c->fused_msd && c->big_table && c->table96 && segbits > (uint32_t)kSegBits && keybits - kSegBits <= 96 &&
        c->big_rounds_min == 0 && (m_est >> kSegBits) <= (uint64_t)kSeg96Limit * 3 / 4)
    { segbits = kSegBits; big_table = -2; }
    if (!kOne && !big_table && c->fused_msd && c->big_table && segbits > (uint32_t)kSegBits)
        for (int r = std::max(0, c->big_rounds_min); r <= std::min(1, c->big_rounds_max); ++r)
        {
            if ((m_est >> (kSegBits + r)) <= (uint64_t)kSegBigLimit2 * 3 / 4) { segbits = kSegBits; big_table = 1 + r; break; }
            // between the two: the 6144-slot table, still one workgroup (and one read) per segment
            if (r == 0 && c->wide_table && (m_est >> kSegBits) <= (uint64_t)kSegWideLimit2 * 3 / 4) { segbits = kSegBits; big_table = -1; break; }
        }
    // one-word keys whose bits below a 17- to 20-bit prefix fit 32 (an odd-length k-mer's strand representative has one
    // bit that is always clear): 9 to 12 bits at the second level, which then writes -- and the counting kernel reads --
    // 4-byte remainders instead of 8-byte keys (kernels_partition.hpp: subpart32_kernel).  The fewest bits whose
    // segments hold the estimated distinct keys in an LDS table (2048 slots at 9 bits when they do, else 4096).
    uint32_t r32_bits = 0, rbits32 = 0, r32_split = 0;
    bool squeeze = false;
    const uint32_t sqbit32 = c->len - 1;
    int r32_slots = 0;
    if (kOne && c->rem32 && c->fused_msd && c->big_rounds_min == 0)
    {
        // (second-level bits, third-level bits) in the order of what they cost on C2's 12.6 G keys: the second level 31 ms
        // with 9 bits and 46 with 10 (shorter runs), the third level ~30 ms whatever it splits into -- so ten bits before a
        // third level, and nine bits + a third level before ten + a third level.  The first pair whose remainder fits 32
        // bits and whose segments hold the estimated distinct keys in an LDS table (2048 slots at (9, 0) when they
        // do, else 4096 with a quarter to spare).
        // Reads with errors (round 5): where ten bits and the 4 096-slot table do not do, the tables of 8 192 and 16 384 slots
        // (512 / 1 024 threads) come BEFORE a third level -- that level is a pass over all keys (~30 ms on C2's 12.6 G), the
        // larger table the same counting on fewer, longer segments.  Third element: the largest table of the candidate.
        static const uint32_t order[][3] = {{9, 0, 4096}, {10, 0, 4096}, {10, 0, 8192}, {10, 0, 16384}, {9, 1, 4096}, {9, 2, 4096}, {9, 3, 4096}, {9, 4, 4096},
                                            {10, 1, 4096}, {10, 2, 4096}, {10, 3, 4096}, {10, 4, 4096}};
        for (const auto& cand : order)
        {
            const uint32_t b2 = cand[0], b3 = cand[1], table = cand[2];
            if (table > 4096u && !(c->r32_form && c->big_r32 && !c->rem32_slots)) continue;
            if (b2 < c->rem32_bits_min || b3 < c->rem32_split_min) continue;
            if (keybits < 8 + b2 + 8) continue;
            const uint32_t rb = keybits - 8 - b2;
            const bool sq = !graph_mode && !canon_l1 && (c->len & 1u) && rb == 33;
            if (rb - (sq ? 1u : 0u) > 32 || rb - (sq ? 1u : 0u) < b3 + 8) continue;
            const uint64_t per = m_est >> (8 + b2 + b3);
            int slots = 0;
            if (c->rem32_slots) slots = per <= (uint64_t)(c->rem32_slots / 4 * 3) || (b2 == 10 && b3 == (uint32_t)kSub32SplitMax) ? c->rem32_slots : 0;
            // (buckets of four: what is not at home costs a second look, and at a load of 0.37 that is 1.5 % of the keys, at
            // 0.19 a per-mille -- the small table only where it stays that empty)
            else if (per <= (c->r32_small_max ? c->r32_small_max : c->r32_form ? 400u : 2048u / 4 * 3 * 3 / 4) && b3 == 0 && b2 == (uint32_t)kSub32BitsMin) slots = 2048;
            else if (per <= (uint64_t)table / 4 * 3 * 3 / 4) slots = (int)table;
            if (!slots) continue;
            r32_slots = slots; r32_bits = b2; rbits32 = rb; squeeze = sq; r32_split = b3;
            break;
        }
    }
    if (r32_slots) { segbits = kSegBits; big_table = 0; }
    const uint32_t r32_digits = 1u << r32_bits, r32_regions = 256u << r32_bits;
    if ((!big_table && !r32_slots && (m_est >> segbits) > limit) || segbits + 8 > keybits)
        return decline("too many distinct keys per segment");
    const uint32_t shift = keybits - segbits;
    const uint32_t npass = (segbits + 7) / 8;

    // 2. histograms of the sample: joint over both digits for the two-level form, else of the
    //    first partition digit only
    std::vector<unsigned long long> hh(256, 0);          // the fused kernel's digit
    std::vector<uint64_t> joint;                         // [high*256 + low], two-level form only
    std::vector<uint64_t> joint17;                       // [high*512 + low9], 32-bit-remainder form
    bool msd = want_msd && segbits == 16;
    if (!msd) r32_slots = 0;
    if (msd)
    {
        // the sample's joint histogram of both digits (kept out of the per-kernel timing): one pass over the sample
        // with the bins in LDS (a 16-bit partition of the sample + segment bounds took 2.4 ms on C2, this 0.9);
        // 17 bits (four sweeps) for the 32-bit-remainder form, whose pairs of bins are the 16-bit form's
        const uint32_t jbins = r32_slots ? r32_regions : 65536u;
        c->mute_timing = true;
        unsigned long long* jh = (unsigned long long*)c->arena.temp((uint64_t)jbins * 8);
        HIP_TRY(hipMemsetAsync(jh, 0, (uint64_t)jbins * 8, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(joint_hist_kernel<K>), dim3(256), dim3(kJointThreads), 0, c->stream, (const K*)ka, ns,
                           r32_slots ? rbits32 : shift, jh, jbins / 32768u);
        c->mute_timing = false;
        std::vector<uint64_t> ho(jbins);
        HIP_TRY(hipMemcpyAsync(ho.data(), jh, (uint64_t)jbins * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        joint.resize(65536);
        if (r32_slots)
        {
            joint17.swap(ho);
            const uint32_t fold = r32_regions / 65536u;          // bins of this form per bin of the 16-bit form
            for (uint32_t i = 0; i < 65536; ++i) { uint64_t a = 0; for (uint32_t j = 0; j < fold; ++j) a += joint17[i * fold + j]; joint[i] = a; }
        }
        else joint.swap(ho);
        for (uint32_t i = 0; i < 65536; ++i) hh[i >> 8] += joint[i];
    }
    else
    {
        unsigned long long* shist = (unsigned long long*)c->arena.temp(256 * 8);
        HIP_TRY(hipMemsetAsync(shist, 0, 256 * 8, c->stream));
        c->mute_timing = true;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(global_hist_kernel<K>), dim3(256), dim3(kTB), 0, c->stream, (const K*)ka, ns, shift, 1u, shist);
        c->mute_timing = false;
        HIP_TRY(hipMemcpyAsync(hh.data(), shist, 256 * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }

    // sub-regions of the second buffer (two-level form): expected size + six standard deviations
    // of the sample count; they must fit, else the one-level form is used
    std::vector<SubTable> hsub;
    std::vector<uint64_t> hsub32_start;          // 32-bit-remainder form: first slot and capacity of every sub-region
    std::vector<uint32_t> hsub32_cap;
    if (msd && r32_slots)
    {
        // 2^17 .. 2^20 sub-regions of 4-byte slots in the second key buffer; every start a multiple of four slots (the
        // counting kernel loads 16 bytes per lane)
        hsub32_start.resize(r32_regions);
        hsub32_cap.resize(r32_regions);
        uint64_t at = 0;
        bool fits = true;
        for (uint32_t i = 0; i < r32_regions; ++i)
        {
            const double h = (double)joint17[i];
            const uint64_t cap = exact ? (((uint64_t)(h * c->fused_capscale)) + 3) & ~3ULL
                                       : (((uint64_t)(((h + 6.0 * std::sqrt(h + 1.0) + 4.0) * scale + 64.0) * c->fused_capscale) + 15) & ~15ULL);
            if (cap > 0xFFFF0000ULL) fits = false;
            hsub32_start[i] = at; hsub32_cap[i] = (uint32_t)cap;
            at += cap;
            // (the second level addresses a region's sub-regions with 32-bit offsets from the region's first)
            if ((i & (r32_digits - 1u)) == r32_digits - 1u && at - hsub32_start[i - (r32_digits - 1u)] > 0xFFFF0000ULL) fits = false;
        }
        if (!fits || at > 2 * kb_slots || (r32_split && at > 2 * ka_slots))          // (the third level writes the remainders into the first buffer)
        {
            if (c->debug) std::fprintf(stderr, "libgossgpu: 32-bit sub-regions need %llu slots of %llu: 8-byte form\n",
                                       (unsigned long long)at, (unsigned long long)(2 * kb_slots));
            r32_slots = 0;
            hsub32_start.clear(); hsub32_cap.clear();
            if ((m_est >> segbits) > limit) return decline("too many distinct keys per segment");
        }
    }
    if (msd && !r32_slots)
    {
        hsub.resize(1);
        uint64_t at = 0;
        for (uint32_t i = 0; i < 65536; ++i)
        {
            const double h = (double)joint[i];
            // exact counts need no slack (and a small chunk cannot afford 65 536 paddings)
            const uint64_t cap = exact ? (uint64_t)(h * c->fused_capscale)
                                       : (((uint64_t)(((h + 6.0 * std::sqrt(h + 1.0) + 4.0) * scale + 64.0) * c->fused_capscale) + 15) & ~15ULL);
            hsub[0].start[i] = at; hsub[0].cap[i] = cap;
            at += cap;
        }
        if (at > kb_slots && reduced)
        {
            if (c->debug) std::fprintf(stderr, "libgossgpu: sub-regions need %llu slots of %llu\n", (unsigned long long)at, (unsigned long long)kb_slots);
            return (int)kFusedNeedFull;
        }
        if (at > kb_slots)
        {
            if (c->debug) std::fprintf(stderr, "libgossgpu: sub-regions need %llu slots of %llu: one-level form\n",
                                       (unsigned long long)at, (unsigned long long)kb_slots);
            if (big_table) return decline("sub-regions do not fit and the big table needs them");
            msd = false;
            // the one-level form partitions on the LOW digit: its marginal histogram
            std::fill(hh.begin(), hh.end(), 0ULL);
            for (uint32_t i = 0; i < 65536; ++i) hh[i & 255u] += joint[i];
            hsub.clear();
        }
    }
    const uint32_t part_shift = msd ? keybits - 8 : shift;           // the fused kernel's digit
    const bool narrow = kOne && msd && r32_slots && c->narrow;       // remainder + digit between the two levels, 5.33 bytes a key
    lap("sample histograms");

    // bucket regions of the first buffer: expected size of every bucket plus five standard
    // deviations of the sample count; whatever room the key buffer has beyond that (up to 25 %)
    // is handed out proportionally, so that a mildly non-stationary input still fits
    // Every workgroup of the fused kernel appends to a private block of B slots per bucket and pads the
    // unused tail of its last blocks, so a region also needs one block per workgroup; B is the largest
    // power of two (one 64-byte granule .. 256 slots) that keeps that padding within 3 % of the keys: a small
    // chunk gets fewer workgroups, then smaller blocks.  Workgroups per CU: 3 for one-word keys (52 KB of
    // LDS each), 2 for two-word keys (75 KB).
    const uint32_t kGranule = kOne ? 8 : 4;                       // keys per 64 bytes
    const uint64_t kSuperFused = kOne ? (uint64_t)kTB * (graph_mode ? GOSS_E1_NK / 2 : GOSS_E1_NK)
                                      : (uint64_t)kTB * (graph_mode ? GOSS_FUSED_NKEYS2 / 2 : GOSS_FUSED_NKEYS2);
    const double pad_budget = 0.03 * (double)n_exp;
    uint32_t fgrid = (uint32_t)std::min<uint64_t>((nstarts + kSuperFused - 1) / kSuperFused, (uint64_t)(kOne ? kFusedGrid : kFusedGrid2));
    if (c->fused_grid) fgrid = std::min(fgrid, c->fused_grid);
    fgrid = (uint32_t)std::max(16.0, std::min((double)fgrid, pad_budget / (256.0 * kGranule)));
    uint32_t blk_log2 = kOne ? 3 : 2;
    while (blk_log2 < 8 && (double)fgrid * 256.0 * (double)(2u << blk_log2) <= pad_budget) ++blk_log2;
    if (c->blk_log2_max) blk_log2 = std::min(blk_log2, std::max(c->blk_log2_max, kOne ? 3u : 2u));
    const uint64_t B = 1ULL << blk_log2;
    // (a workgroup also holds a reserved block per bucket that it may never open)
    const double blk_extra = (double)fgrid * (double)B * 2.0;
    GapTable gt{};
    double base[256], base_sum = 0;
    for (int d = 0; d < 256; ++d)
    {
        const double h = (double)hh[d];
        base[d] = exact ? h + 64.0 : (h + 5.0 * std::sqrt(h + 1.0) + 16.0) * scale + 1024.0;    // exact counts need no slack
        base_sum += base[d];
    }
    double slack = std::min(1.25, ((double)ka_slots - 256.0 * ((double)B + blk_extra)) / base_sum);
    if (slack < (exact ? 1.0 : 1.02))
    {
        // buffers sized from the share of valid windows: the caller retries with one slot per window start
        if (reduced) return (int)kFusedNeedFull;
        return decline("bucket regions do not fit the key buffer");
    }
    slack *= c->fused_capscale;
    uint64_t at = 0;
    for (int d = 0; d < 256; ++d)
    {
        uint64_t cap = ((uint64_t)(base[d] * slack + blk_extra * c->fused_capscale) + B - 1) & ~(B - 1);
        gt.reg_start[d] = at; gt.reg_cap[d] = cap;
        at += cap;
    }

    // 3. extraction that partitions
    GapTable* dgt = (GapTable*)c->arena.temp(sizeof(GapTable));
    PartCounters* pc = (PartCounters*)c->arena.temp(sizeof(PartCounters));
    HIP_TRY(hipMemcpyAsync(dgt, &gt, sizeof(GapTable), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(pc, 0, sizeof(PartCounters), c->stream));
    const uintptr_t addr = (uintptr_t)d_bases;
    const uint32_t mis = (uint32_t)(addr & 15u);
    const uint8_t* aligned = (const uint8_t*)(addr - mis);
    const uint8_t* pk_src = nullptr; const uint16_t* pk_bad = nullptr;          // (a packed string: the group `aligned` stands for)
    if (c->pk.on && !c->rec_mode) { pk_ptrs(c, aligned, &pk_src, &pk_bad); c->pk_fused_chunks++; }
    {
#ifndef GOSS_FUSED_G
#define GOSS_FUSED_G 1
#endif
        const bool graph = graph_mode;
        const uint64_t kSuper = kOne ? (uint64_t)kTB * (graph ? GOSS_E1_NK / 2 : GOSS_E1_NK)
                                     : (uint64_t)kTB * (graph ? GOSS_FUSED_NKEYS2 / 2 : GOSS_FUSED_NKEYS2);
        const uint64_t nsuper = (nstarts + kSuper - 1) / kSuper;
        uint32_t grid = (uint32_t)std::min<uint64_t>(nsuper, 1024);
        if (c->fused_grid) grid = std::min(grid, c->fused_grid);      // experiments: leave room for a second context's kernels
        grid = fgrid;                                                  // the regions were sized for this many workgroups
        const int nh = msd ? 0 : (npass > 2 ? 2 : 1);
        PhaseTimer t(c, GOSS_T_EXTRACT, nstarts);
        if constexpr (kOne)
        {
            // the 32-bit forms of the kernel: the partition digit is the key's top eight bits (msd) and lies at bit 34 or above
            const bool fastk = nh == 0 && 2 * c->len >= 32 && part_shift >= 34 && !c->no_fast32;
            // (round 5: ahead of the 32-bit-remainder form of the second level the keys leave as remainder + digit, twelve
            // to a granule -- kernels_extract.hpp, NARROW)
            const uint32_t nr_rbits = rbits32, nr_sqbit = squeeze ? sqbit32 : 0u, nr_dmask = (1u << r32_bits) - 1u;
            const uint32_t nr_capg = std::min(656u, std::max(576u, c->narrow_capg));          // (granules of the kernel's LDS layout: kernels_extract.hpp, kSlots)
#define GOSS_LAUNCH_EP5(MODE, NH, ODD, FAST, NRW)                                                                     \
    do {                                                                                                              \
        if (c->rec_mode)                                                                                              \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract1_part_kernel<MODE, NH, ODD, true, FAST, NRW>), dim3(grid), dim3(kTB), 0, c->stream, \
                               d_bases, 0u, nstarts, nstarts / rec_slots(c), c->len, ka, pc, (const GapTable*)dgt, part_shift, nsuper, blk_log2, \
                               nr_rbits, nr_sqbit, nr_dmask, nr_capg);                                                \
        else if (c->pk.on)                                                                                            \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract1_part_kernel<MODE, NH, ODD, false, FAST, NRW, true>), dim3(grid), dim3(kTB), 0, c->stream, \
                               pk_src, mis, nstarts, navail, c->len, ka, pc, (const GapTable*)dgt, part_shift, nsuper, blk_log2, \
                               nr_rbits, nr_sqbit, nr_dmask, nr_capg, pk_bad);                                        \
        else                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract1_part_kernel<MODE, NH, ODD, false, FAST, NRW>), dim3(grid), dim3(kTB), 0, c->stream, \
                               aligned, mis, nstarts, navail, c->len, ka, pc, (const GapTable*)dgt, part_shift, nsuper, blk_log2, \
                               nr_rbits, nr_sqbit, nr_dmask, nr_capg, (const uint16_t*)nullptr);                      \
    } while (0)
#define GOSS_LAUNCH_EP4(MODE, NH, ODD, FAST)                                                                          \
    do {                                                                                                              \
        if (narrow) GOSS_LAUNCH_EP5(MODE, NH, ODD, FAST, (NH == 0));                                                  \
        else GOSS_LAUNCH_EP5(MODE, NH, ODD, FAST, false);                                                             \
    } while (0)
#define GOSS_LAUNCH_EP3(MODE, NH, ODD)                                                                                \
    do {                                                                                                              \
        if (fastk) GOSS_LAUNCH_EP4(MODE, NH, ODD, (NH == 0));                                                         \
        else GOSS_LAUNCH_EP4(MODE, NH, ODD, false);                                                                   \
    } while (0)
            if (graph)
            {
                if (nh == 0) GOSS_LAUNCH_EP3(1, 0, 0);
                else if (nh == 1) GOSS_LAUNCH_EP3(1, 1, 0);
                else GOSS_LAUNCH_EP3(1, 2, 0);
            }
            else if (canon_l1)
            {
                // gossamer's canonical form computed per window (many distinct keys: above)
                if (nh == 0) GOSS_LAUNCH_EP3(0, 0, 2);
                else if (nh == 1) GOSS_LAUNCH_EP3(0, 1, 2);
                else GOSS_LAUNCH_EP3(0, 2, 2);
            }
            else if (c->len & 1u)
            {
                // k-mer sets are counted as strand representatives and mapped to the canonical form
                // afterwards (canonicalize_run); odd k: the central base picks the strand
                if (nh == 0) GOSS_LAUNCH_EP3(0, 0, 1);
                else if (nh == 1) GOSS_LAUNCH_EP3(0, 1, 1);
                else GOSS_LAUNCH_EP3(0, 2, 1);
            }
            else
            {
                if (nh == 0) GOSS_LAUNCH_EP3(0, 0, 0);
                else if (nh == 1) GOSS_LAUNCH_EP3(0, 1, 0);
                else GOSS_LAUNCH_EP3(0, 2, 0);
            }
#undef GOSS_LAUNCH_EP3
#undef GOSS_LAUNCH_EP4
#undef GOSS_LAUNCH_EP5
        }
        else
        {
#define GOSS_LAUNCH_E2P(MODE, NH, NBH)                                                                                \
    do {                                                                                                              \
        if constexpr (MODE == 0)                                                                                      \
        {                                                                                                             \
            if (c->rec_mode)                                                                                          \
            {                                                                                                         \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(extract2_part_kernel<0, NH, GOSS_FUSED_NKEYS2, NBH, true>), dim3(grid), dim3(kTB), 0, c->stream, \
                                   d_bases, 0u, nstarts, nstarts / rec_slots(c), c->len, ka, pc, (const GapTable*)dgt, part_shift, nsuper, blk_log2); \
                break;                                                                                                \
            }                                                                                                         \
        }                                                                                                             \
        if (c->pk.on)                                                                                                 \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract2_part_kernel<MODE, NH, GOSS_FUSED_NKEYS2, NBH, false, true>), dim3(grid), dim3(kTB), 0, c->stream, \
                               pk_src, mis, nstarts, navail, c->len, ka, pc, (const GapTable*)dgt, part_shift, nsuper, blk_log2, pk_bad); \
        else                                                                                                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(extract2_part_kernel<MODE, NH, GOSS_FUSED_NKEYS2, NBH>), dim3(grid), dim3(kTB), 0, c->stream, \
                               aligned, mis, nstarts, navail, c->len, ka, pc, (const GapTable*)dgt, part_shift, nsuper, blk_log2, (const uint16_t*)nullptr); \
    } while (0)
#define GOSS_LAUNCH_E2N(MODE, NBH)                                                                                    \
    do { if (nh == 0) GOSS_LAUNCH_E2P(MODE, 0, NBH); else if (nh == 1) GOSS_LAUNCH_E2P(MODE, 1, NBH); else GOSS_LAUNCH_E2P(MODE, 2, NBH); } while (0)
            if (graph) GOSS_LAUNCH_E2N(1, 8);
            else if (rep_graph) GOSS_LAUNCH_E2N(0, 0);          // (NBH 0: strand representatives)
            else
                switch (key2_nbh(c))
                {
                    case 2: GOSS_LAUNCH_E2N(0, 2); break;
                    case 4: GOSS_LAUNCH_E2N(0, 4); break;
                    case 6: GOSS_LAUNCH_E2N(0, 6); break;
                    default: GOSS_LAUNCH_E2N(0, 8); break;
                }
#undef GOSS_LAUNCH_E2N
#undef GOSS_LAUNCH_E2P
        }
        t.stop();
    }
    std::vector<unsigned long long> hpc(sizeof(PartCounters) / 8);
    HIP_TRY(hipMemcpyAsync(hpc.data(), pc, sizeof(PartCounters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const PartCounters* hp = (const PartCounters*)hpc.data();
    if (hp->overflow) { c->fused_overflows++; return decline("a bucket region overflowed"); }
    lap("extraction + first level");
#if defined(GOSS_STAMPS)
    // (timing build: wave 0's cycles per phase and tile, averaged over the workgroups)
    if (hp->hist[505])
        std::fprintf(stderr, "libgossgpu: stamps per tile (cycles): A %.0f  B %.0f  C %.0f  scatter %.0f  D %.0f   (%llu tiles)\n",
                     (double)hp->hist[500] / hp->hist[505], (double)hp->hist[501] / hp->hist[505], (double)hp->hist[502] / hp->hist[505],
                     (double)hp->hist[503] / hp->hist[505], (double)hp->hist[504] / hp->hist[505], (unsigned long long)hp->hist[505]);
    if (hp->hist[505] && hp->hist[506])
        std::fprintf(stderr, "libgossgpu: stamps, finer: C = scan %.0f + bookkeeping %.0f + barrier; scatter = LDS %.0f + encoder %.0f + barrier; D = stores %.0f + absorb %.0f + barrier %.0f\n",
                     (double)hp->hist[506] / hp->hist[505], (double)hp->hist[507] / hp->hist[505], (double)hp->hist[510] / hp->hist[505],
                     (double)hp->hist[511] / hp->hist[505], (double)hp->hist[508] / hp->hist[505], (double)hp->hist[509] / hp->hist[505],
                     (double)hp->hist[504] / hp->hist[505]);
#endif
    const uint64_t n = hp->keys_out;
    if (n == 0) return kFusedDeclined;
    if (n > ka_slots || n > kb_slots) return decline("more keys than the buffers hold");
    uint64_t tiles = 0, sum = 0;
    const uint64_t tile_keys = narrow ? (uint64_t)(r32_bits == 9 ? Sub32N<9>::kTileSlots : Sub32N<10>::kTileSlots) : (msd && r32_slots) ? (uint64_t)kSub32Tile : msd ? (uint64_t)SubCfg<K>::kTile : (uint64_t)kTile;      // of the pass that reads the regions
    for (int d = 0; d < 256; ++d)
    {
        // one-word keys: slots handed out in whole blocks, padding included (the next pass skips it)
        gt.cnt[d] = hp->cursors[d * kCursorStride];
        gt.tile_first[d] = tiles;
        tiles += (gt.cnt[d] + tile_keys - 1) / tile_keys;
        sum += gt.cnt[d];
    }
    gt.tile_first[256] = tiles;
    // (the 8-byte slots handed out hold the keys: one each, or -- narrow form -- twelve to a granule of eight)
    const uint64_t n_slots = narrow ? n / 3 * 2 : n;
    if (sum < n_slots || sum > n_slots + 8 + (uint64_t)fgrid * 256 * (2 * B + 8))
        throw StatusError{GOSS_ERR_HIP, "fused extraction: bucket counts do not add up"};
    HIP_TRY(hipMemcpyAsync(dgt, &gt, sizeof(GapTable), hipMemcpyHostToDevice, c->stream));

    LookbackCtl* ctl = (LookbackCtl*)c->arena.temp(sizeof(LookbackCtl));
    LookbackCtl* hctl = (LookbackCtl*)((uint8_t*)c->h_pinned + 128);
    HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(LookbackCtl), c->stream));
    Run r{nullptr, nullptr, 0};
    if (msd && r32_slots)
    {
        // 4a'. second level, 32-bit-remainder form: keys of region b go to sub-region (b, next 9 bits) as u32 remainders
        SubTable32* dsub = (SubTable32*)c->arena.temp(sizeof(SubTable32));
        unsigned long long* cur2 = (unsigned long long*)c->arena.temp((uint64_t)r32_regions * 4);     // pairs of 32-bit cursors
        uint64_t* seg_beg = (uint64_t*)c->arena.temp((uint64_t)r32_regions * 8);
        uint64_t* seg_end = (uint64_t*)c->arena.temp((uint64_t)r32_regions * 8);
        Tile32* tdesc = (Tile32*)c->arena.temp(std::max<uint64_t>(tiles, 1) * sizeof(Tile32));
        // (only the used part of the table travels: the starts, then the capacities)
        HIP_TRY(hipMemcpyAsync(dsub->start, hsub32_start.data(), (uint64_t)r32_regions * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dsub->cap, hsub32_cap.data(), (uint64_t)r32_regions * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(cur2, 0, (uint64_t)r32_regions * 4, c->stream));
        if (narrow && r32_bits == 9)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(tiles32_kernel<Sub32N<9>::kTileSlots>), dim3(grid_for(tiles, 256)), dim3(256), 0, c->stream,
                               (const GapTable*)dgt, tdesc, (uint32_t)tiles);
        else if (narrow)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(tiles32_kernel<Sub32N<10>::kTileSlots>), dim3(grid_for(tiles, 256)), dim3(256), 0, c->stream,
                               (const GapTable*)dgt, tdesc, (uint32_t)tiles);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(tiles32_kernel<kSub32Tile>), dim3(grid_for(tiles, 256)), dim3(256), 0, c->stream,
                               (const GapTable*)dgt, tdesc, (uint32_t)tiles);
        {
            PhaseTimer t(c, GOSS_T_SCATTER, n);
            const dim3 g2((uint32_t)((tiles + 7) / 8 * 8));
#define GOSS_LAUNCH_S32N(SQ, B2, NRW)                                                                                    \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(subpart32_kernel<SQ, B2, NRW>), g2, dim3(kTB), 0, c->stream, (const Key1*)ka, (uint32_t*)kb, rbits32, \
                       sqbit32, cur2, (const Tile32*)tdesc, (uint32_t)tiles, (const SubTable32*)dsub, ctl)
#define GOSS_LAUNCH_S32(SQ, B2) do { if (narrow) GOSS_LAUNCH_S32N(SQ, B2, true); else GOSS_LAUNCH_S32N(SQ, B2, false); } while (0)
            if (squeeze) GOSS_LAUNCH_S32(true, 9);          // (only the 9-bit form of an odd k-mer set needs the squeeze)
            else if (r32_bits == 9) GOSS_LAUNCH_S32(false, 9);
            else GOSS_LAUNCH_S32(false, 10);
#undef GOSS_LAUNCH_S32
#undef GOSS_LAUNCH_S32N
            t.stop();
        }
        hipLaunchKernelGGL(sub_bounds32_kernel, dim3(r32_regions / 256), dim3(256), 0, c->stream, (const SubTable32*)dsub,
                           (const uint32_t*)cur2, r32_regions, seg_beg, seg_end);
        HIP_TRY(hipMemcpyAsync(hctl, ctl, sizeof(LookbackCtl), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (hctl->error) { c->fused_overflows++; return decline("a 32-bit sub-region overflowed"); }
        lap("second level (32-bit remainders)");
        // third level: every segment split into 2^r32_split sub-segments, from kb into ka (same offsets); the counts are
        // then staged in kb
        const uint32_t* rems = (const uint32_t*)kb;
        Key1* spare32 = (Key1*)ka;
        uint32_t nseg32 = r32_regions;
        if (r32_split)
        {
            nseg32 = r32_regions << r32_split;
            uint64_t* sub_beg = (uint64_t*)c->arena.temp((uint64_t)nseg32 * 8);
            uint64_t* sub_end = (uint64_t*)c->arena.temp((uint64_t)nseg32 * 8);
            {
                PhaseTimer t(c, GOSS_T_SCATTER, n);
                hipLaunchKernelGGL(subsplit32_kernel, unit_grid(r32_regions), dim3(kTB), 0, c->stream, (const uint32_t*)kb, (uint32_t*)ka,
                                   (const uint64_t*)seg_beg, (const uint64_t*)seg_end, rbits32 - (squeeze ? 1u : 0u), r32_split, sub_beg, sub_end);
                t.stop();
            }
            check_launch("third-level split kernel");
            seg_beg = sub_beg; seg_end = sub_end;
            rems = (const uint32_t*)ka;
            spare32 = (Key1*)kb;
            lap("third level (sub-segments)");
        }
        int rc;
        for (;;)
        {
            rc = segment_reduce32(c, rems, spare32, n, &r, seg_beg, seg_end, r32_slots, squeeze, rbits32, sqbit32, nseg32, r32_split);
            const int top_slots = c->r32_form && c->big_r32 ? 16384 : 4096;
            if (rc != 1 || r32_slots >= top_slots || c->rem32_slots) break;
            // the remainders are still in their sub-regions: only the counting is redone, in the next larger table
            c->segment_retries++;
            r32_slots = r32_slots < 4096 ? 4096 : 2 * r32_slots;
        }
        if (rc != 0)
        {
            // more distinct keys per segment than this form takes: the chunk again in the 8-byte form and its ladder of tables
            c->segment_retries++;
            // more second-level bits while the remainder allows them (the chunk's first level is redone: the staging of
            // the counts has overwritten its regions), else the 8-byte form and its ladder of tables
            if (rc == 1 && r32_split < (uint32_t)kSub32SplitMax) c->rem32_split_min = r32_split + 1;
            else c->rem32 = false;
            if (c->debug) std::fprintf(stderr, "libgossgpu: fused path: 32-bit form with %u + %u bits overflowed (%d), redoing the chunk %s\n", r32_bits, r32_split, rc,
                                       c->rem32 ? "with more sub-segments" : "in the 8-byte form");
            c->arena.release(mark);
            return process_chunk_fused<K>(c, d_bases, nstarts, navail, ka, ka_slots, kb, kb_slots);
        }
        c->fused_msd_chunks++;
        c->rem32_chunks++;
        if (narrow) c->narrow_chunks++;
        c->rem32_bits_last = r32_bits;
        c->rem32_split_last = r32_split;
    }
    else if (msd)
    {
        // 4a. second level: keys of region b go to sub-region (b, low digit) by atomic cursors
        SubTable* dsub = (SubTable*)c->arena.temp(sizeof(SubTable));
        unsigned long long* cur2 = (unsigned long long*)c->arena.temp(65536ULL * kSubCursorStride * 8);
        uint64_t* seg_beg = (uint64_t*)c->arena.temp(65536 * 8);
        uint64_t* seg_end = (uint64_t*)c->arena.temp(65536 * 8);
        HIP_TRY(hipMemcpyAsync(dsub, hsub.data(), sizeof(SubTable), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(cur2, 0, 65536ULL * kSubCursorStride * 8, c->stream));
        {
            PhaseTimer t(c, GOSS_T_SCATTER, n);
            // (a multiple of 8 workgroups: the kernel deals the tiles out by XCD)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_onesweep_kernel<K, false, false, true, SubCfg<K>::kItems>), dim3((uint32_t)((tiles + 7) / 8 * 8)), dim3(kTB), 0,
                               c->stream, (const K*)ka, (const uint32_t*)nullptr, kb, (uint32_t*)nullptr, n, shift, shift,
                               (const unsigned long long*)nullptr, (unsigned long long*)nullptr, ctl, cur2,
                               (const GapTable*)dgt, (const SubTable*)dsub, big_table == -2 ? 1u : 0u);
            t.stop();
        }
        hipLaunchKernelGGL(sub_bounds_kernel, dim3(256), dim3(256), 0, c->stream, (const SubTable*)dsub,
                           (const unsigned long long*)cur2, seg_beg, seg_end);
        HIP_TRY(hipMemcpyAsync(hctl, ctl, sizeof(LookbackCtl), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (hctl->error) { c->fused_overflows++; return decline("a sub-region overflowed"); }
        lap("second level");
        // (-3: the second level wrote 12-byte remainders for the 96-bit table)
        int rc;
        for (;;)
        {
            rc = segment_reduce<K>(c, kb, ka, n, segbits, &r, seg_beg, seg_end, big_table == -2 ? -3 : big_table);
            // a table that overflowed (one-word keys): the keys are still in their sub-regions, so only the counting
            // is redone, with the next larger form -- 8192 slots, then 2 and 4 workgroups per segment -- instead of
            // the whole chunk with the unfused kernels
            if (rc != 1 || !kOne || !c->big_table || big_table < 0 || big_table >= 1 + c->big_rounds_max) break;
            c->segment_retries++;
            ++big_table;
            if (c->debug) std::fprintf(stderr, "libgossgpu: fused path: a counting table overflowed, next form %d\n", big_table);
        }
        if (rc != 0)
        {
            c->segment_retries++;
            // this input is too skewed for it: the next smaller form from now on
            if (big_table == -2) c->table96 = false;
            else if (big_table < 0) c->wide_table = false;
            else if (big_table) c->big_table = false;
            return decline("a segment table overflowed");
        }
        c->fused_msd_chunks++;
        if (big_table) c->big_table_chunks++;
        if (big_table == -1) c->wide_table_chunks++;
        if (big_table == -2) c->table96_chunks++;
    }
    else
    {
        // 4b. remaining partition passes: the first reads the bucket regions, the others are dense
        const uint64_t ntiles_dense = (n + kTile - 1) / kTile;
        unsigned long long* status = (unsigned long long*)c->arena.temp(256ULL * std::max(tiles, ntiles_dense) * 8);
        {
            PhaseTimer t(c, GOSS_T_SCAN, 512);
            hipLaunchKernelGGL(scan_rows256_kernel, dim3(2), dim3(kTB), 0, c->stream, pc->hist);
            t.stop();
        }
        K* src = ka; K* dst = kb;
        for (uint32_t di = 1; di < npass; ++di)
        {
            const uint32_t d = shift + 8 * di;
            const bool gapped = di == 1;
            const uint64_t nt = gapped ? tiles : ntiles_dense;
            HIP_TRY(hipMemsetAsync(status, 0, nt * 256 * 8, c->stream));
            {
                PhaseTimer t(c, GOSS_T_SCATTER, n);
                if (gapped)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_onesweep_kernel<K, false, false, true>), dim3((uint32_t)nt), dim3(kTB), 0,
                                       c->stream, (const K*)src, (const uint32_t*)nullptr, dst, (uint32_t*)nullptr, n, d, shift,
                                       (const unsigned long long*)(pc->hist + (di - 1) * 256), status, ctl,
                                       (unsigned long long*)nullptr, (const GapTable*)dgt, (const SubTable*)nullptr);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_onesweep_kernel<K, false, false, false>), dim3((uint32_t)nt), dim3(kTB), 0,
                                       c->stream, (const K*)src, (const uint32_t*)nullptr, dst, (uint32_t*)nullptr, n, d, shift,
                                       (const unsigned long long*)(pc->hist + (di - 1) * 256), status, ctl,
                                       (unsigned long long*)nullptr, (const GapTable*)nullptr, (const SubTable*)nullptr);
                t.stop();
            }
            HIP_TRY(hipMemcpyAsync(hctl, ctl, sizeof(LookbackCtl), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (hctl->error)
            {
                std::fprintf(stderr, "libgossgpu: radix look-back chain gave up in the fused path; redoing the chunk unfused\n");
                c->lookback_failures++;
                c->ordered_tiles = true;
                return kFusedDeclined;
            }
            std::swap(src, dst);
        }
        // src = partitioned keys (dense: npass >= 2), dst = spare
        c->arena.release(mark); release.m = c->arena.mark();
        const int rc = segment_reduce<K>(c, src, dst, n, segbits, &r, nullptr, nullptr, false);
        if (rc != 0) { c->segment_retries++; return decline("a segment table overflowed"); }
    }
    lap("segments counted");
    if (canon_l1) c->canon_chunks++;
    if (use_rep && !canon_l1)
    {
        // the run stays in representative space: it is mapped to gossamer's canonical forms when it meets a run
        // that is not, or at finish -- a build of several chunks pays for the re-ordering once, on the merged run
        r.rep = true;
        c->rep_chunks++;
    }
    c->runs.push_back(r);
    c->windows += hp->windows;
    c->keys_total += rep_graph ? 2 * n : n;          // (the adapter's key stream: two keys per window of a graph)
    c->fused_chunks++;
    if (c->rec_mode) c->rec_chunks++;
    return kFusedDone;
}

// Process window starts [0, nstarts) of a device-resident byte string (navail readable bytes,
// navail >= nstarts): extract -> sort -> reduce -> append a run.
template <class K>
void process_chunk(goss_gpu_ctx* c, const uint8_t* d_bases, uint64_t nstarts, uint64_t navail)
{
    const uint32_t S = c->mode == GOSS_MODE_GRAPH ? 2 : 1;
    const uint64_t cap = nstarts * S;              // upper bound on keys
    uint64_t mark = c->arena.mark();
    // the fused path wants room for its bucket regions (expected keys + slack): up to cap/8 more
    // slots in the first buffer when the arena can spare them beyond the two buffers and the
    // partition / segment tables
    uint64_t ka_slots = cap, kb_slots = cap;
    auto full_slots = [&](uint64_t basis) {
        ka_slots = basis; kb_slots = basis;
        const uint64_t extra = basis / 8 + 256 * 16;
        const uint64_t need = (2 * basis + extra) * sizeof(K) + basis / 2 + (uint64_t)(1u << kSegBits) * kSegLimit * 12 + (64u << 20);
        if (c->arena.avail() >= need) ka_slots = basis + extra;
    };
    bool reduced = false;
    if (c->fused && nstarts >= c->fused_min)
    {
        // (strand pairs: ONE key per window of a graph on the fused path -- what chunk_capacity cut the chunk for; the
        // sequence a declined chunk falls back to emits both strands and gets its buffers then, below)
        const uint64_t capf = c->mode == GOSS_MODE_GRAPH && c->graph_rep ? nstarts : cap;
        // a chunk whose sample is a set of slices (not the whole chunk): buffers for the expected
        // number of keys -- bucket regions with 6 % of slack, sub-regions with 13 %
        if (c->valid_frac < 0.97 && nstarts > kValidSizingMin)
        {
            // (+ the blocks the first level's workgroups hold in every region -- 256 regions x 768 workgroups x two blocks
            // of up to 256 slots: an eighth of a chunk of 0.8 G windows, which the slack alone did not cover -- every
            // staged chunk of a FASTQ build was sampled twice, 2 x 2 ms of joint histogram among it)
            ka_slots = std::min<uint64_t>(cap, (uint64_t)((double)capf * c->valid_frac * kValidSlackA) + (1u << 20) + std::min<uint64_t>(capf / 8, 104u << 20));
            kb_slots = std::min<uint64_t>(cap, (uint64_t)((double)capf * c->valid_frac * kValidSlackB) + (8u << 20));
            reduced = ka_slots < cap || kb_slots < cap;
        }
        if (!reduced) { full_slots(capf); reduced = capf < cap; }
    }
    K* ka = (K*)c->arena.temp(ka_slots * sizeof(K));
    K* kb = (K*)c->arena.temp(kb_slots * sizeof(K));
    int frc = process_chunk_fused<K>(c, d_bases, nstarts, navail, ka, ka_slots, kb, kb_slots);
    if (frc == kFusedDone) { if (reduced) c->valid_sized_chunks++; c->arena.release(mark); return; }
    if (reduced)
    {
        c->valid_resizes++;
        // the unfused kernels (and a fused retry) want one slot per window start
        c->arena.release(mark);
        c->valid_frac = 1.0;
        full_slots(cap);
        ka = (K*)c->arena.temp(ka_slots * sizeof(K));
        kb = (K*)c->arena.temp(kb_slots * sizeof(K));
        if (frc == kFusedNeedFull && process_chunk_fused<K>(c, d_bases, nstarts, navail, ka, ka_slots, kb, kb_slots) == kFusedDone)
        {
            c->arena.release(mark);
            return;
        }
    }
    HIP_TRY(hipMemsetAsync(c->d_ctr, 0, sizeof(ExtractCounters), c->stream));
    uintptr_t addr = (uintptr_t)d_bases;
    uint32_t mis = c->rec_mode ? 0u : (uint32_t)(addr & 15u);
    const uint8_t* aligned = (const uint8_t*)(addr - mis);
    {
        PhaseTimer t(c, GOSS_T_EXTRACT, nstarts);
        if (c->pk.on && !c->rec_mode)
        {
            // (a packed string outside the fused path -- a small input, or the sequence a chunk falls back to: the plain
            // kernels read bytes, and this chunk's are made here; released with the chunk's other temporaries)
            aligned = pk_unpack_temp(c, d_bases, navail, &mis);
            const auto keep = c->pk; c->pk.on = false;
            extract_dispatch<K>(c, aligned, mis, nstarts, navail, ka);
            c->pk = keep;
        }
        else extract_dispatch<K>(c, aligned, mis, nstarts, navail, ka);
        t.stop();
    }
    ExtractCounters* h = (ExtractCounters*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, c->d_ctr, 16 /* keys_out, windows */, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t n = h->keys_out;
    const uint64_t nwin = n / S;              // S keys per valid window
    if (n)
    {
        Run r = count_keys<K>(c, ka, kb, n);
        c->runs.push_back(r);
    }
    c->windows += nwin;           // only once the chunk has succeeded (it may be retried)
    c->keys_total += n;
    c->arena.release(mark);
}

// Merge all runs into one (concatenate, sort pairs, sum equal keys).
template <class K>
void merge_runs(goss_gpu_ctx* c)
{
    if (c->runs.size() <= 1) return;
    bool all_rep = true, any_rep = false;
    for (auto& r : c->runs) { all_rep = all_rep && r.rep; any_rep = any_rep || r.rep; }
    if (any_rep && !all_rep)
    {
        const size_t nr = c->runs.size();
        for (size_t i = 0; i < nr; ++i)
            if (c->runs[i].rep)
            {
                if (c->mode == GOSS_MODE_GRAPH) expand_graph_run<K>(c, i);          // (appends the reverse complements as a run of their own)
                else if constexpr (std::is_same<K, Key1>::value) canonicalize_run<K>(c, c->runs[i]);
                else throw StatusError{GOSS_ERR_STATE, "a two-word run in representative space"};
            }
    }
    const bool rep_out = all_rep;
    uint64_t total = 0;
    for (auto& r : c->runs) total += r.m;
    {
        // two copies of all entries + segment tables; low-duplication inputs do not shrink
        const uint64_t need = 2 * total * (sizeof(K) + 4) + (512ULL << 20);
        if (c->arena.avail() < need) grow_arena(c, need);
    }
    uint64_t mark = c->arena.mark();
    K* ka = (K*)c->arena.temp(total * sizeof(K));
    K* kb = (K*)c->arena.temp(total * sizeof(K));
    uint32_t* va = (uint32_t*)c->arena.temp(total * 4);
    uint32_t* vb = (uint32_t*)c->arena.temp(total * 4);
    uint64_t off = 0;
    for (auto& r : c->runs)
    {
        HIP_TRY(hipMemcpyAsync(ka + off, r.keys, r.m * sizeof(K), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(va + off, r.counts, r.m * 4, hipMemcpyDeviceToDevice, c->stream));
        off += r.m;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<uint64_t> run_off;
    {
        uint64_t o = 0;
        for (auto& r : c->runs) { run_off.push_back(o); o += r.m; }
        run_off.push_back(o);
    }
    const uint32_t nruns = (uint32_t)c->runs.size();
    std::vector<int> in_bigs;                       // the inputs' counts beyond 32 bits (graph mode)
    for (auto& r : c->runs) in_bigs.push_back(r.big);
    // the old runs' permanent storage is dead now: rewind the permanent end to the first run
    c->arena.lo = (uint64_t)((uint8_t*)c->runs.front().keys - c->arena.base);
    c->runs.clear();

    // Two-word keys with at most 96 bits below a 16-bit prefix: the runs' entries are counted into the LDS table
    // of the counting kernel, segment by segment (seg_hash_merge96_kernel).  The runs of a high-coverage build all
    // hold the same keys, so 65 536 tables of 8192 slots take them; a segment with more than 6144 distinct keys, or
    // a count that reaches 2^31, sends the merge down the general path below.
    if constexpr (std::is_same<K, Key2>::value)
    {
        const uint32_t keybits = 2 * c->len;
        if (c->seg_merge && c->table96 && keybits >= 16 + 8 && keybits - 16 <= 96 && nruns <= (uint32_t)kMergeRuns && total >= c->hash_merge_min)
        {
            const uint32_t nseg = 65536, shift = keybits - 16;
            uint64_t m2 = c->arena.mark();
            uint64_t* bounds = (uint64_t*)c->arena.temp((uint64_t)nruns * (nseg + 1) * 8);
            uint64_t* d_off = (uint64_t*)c->arena.temp((nruns + 1) * 8);
            uint64_t* seg_pos = (uint64_t*)c->arena.temp((uint64_t)nseg * 8);
            uint64_t* seg_cnt = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
            uint64_t* seg_dst = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
            SegOut* so = (SegOut*)c->arena.temp(sizeof(SegOut));
            SegOut hso{};
            hso.stage_cap = total;
            HIP_TRY(hipMemcpyAsync(so, &hso, sizeof(SegOut), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_off, run_off.data(), (nruns + 1) * 8, hipMemcpyHostToDevice, c->stream));
            PhaseTimer t(c, GOSS_T_REDUCE, total);
            for (uint32_t r = 0; r < nruns; ++r)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_bounds_kernel<K>), dim3(nseg / 256 + 1), dim3(256), 0, c->stream,
                                   (const K*)(ka + run_off[r]), run_off[r + 1] - run_off[r], shift, nseg,
                                   bounds + (uint64_t)r * (nseg + 1));
            hipLaunchKernelGGL(seg_hash_merge96_kernel, unit_grid(nseg), dim3(kSegBigThreads), 0, c->stream, (const Key2*)ka, (const uint32_t*)va,
                               (const uint64_t*)d_off, (const uint64_t*)bounds, nruns, so, seg_pos, seg_cnt, (Key2*)kb, vb, shift);
            SegOut* h = (SegOut*)c->h_pinned;
            HIP_TRY(hipMemcpyAsync(h, so, sizeof(SegOut), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (!h->overflow)
            {
                const uint64_t m = h->cursor;
                HIP_TRY(hipMemcpyAsync(seg_dst, seg_cnt, (uint64_t)nseg * 8, hipMemcpyDeviceToDevice, c->stream));
                exclusive_scan_u64(c, seg_dst, nseg);
                Run r{nullptr, nullptr, m};
                r.keys = c->arena.perm(std::max<uint64_t>(m, 1) * sizeof(K));
                r.counts = (uint32_t*)c->arena.perm(std::max<uint64_t>(m, 1) * 4);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_gather_kernel<K>), unit_grid(nseg), dim3(kTB), 0, c->stream,
                                   (const K*)kb, (const uint32_t*)vb, (const uint64_t*)seg_pos,
                                   (const uint64_t*)seg_dst, (const uint64_t*)seg_cnt, (K*)r.keys, r.counts);
                t.stop();
                HIP_TRY(hipStreamSynchronize(c->stream));
                r.rep = rep_out;
                c->runs.push_back(r);
                c->seg_merges++;
                c->hash_merges++;
                c->arena.release(mark);
                return;
            }
            t.stop();
            c->arena.release(m2);
        }
    }

    // Merge by segments (seg_merge_kernel): every entry read once, written once.  Needs every
    // segment's entries of all runs to fit kMergeCap; else the concatenation is sorted again.
    if (c->seg_merge && nruns <= (uint32_t)kMergeRuns && total >= 1024)
    {
        const uint32_t keybits = 2 * c->len;
        uint32_t segbits = 8;
        while (segbits < 26 && segbits + 1 <= keybits && (total >> segbits) > (uint64_t)kMergeCap / 2) ++segbits;
        for (int attempt = 0; attempt < 3 && segbits <= 26 && segbits <= keybits; ++attempt, segbits += 2)
        {
            const uint32_t nseg = 1u << segbits, shift = keybits - segbits;
            uint64_t m2 = c->arena.mark();
            const uint64_t need = (uint64_t)nruns * (nseg + 1) * 8 + (uint64_t)nseg * 32 + (1u << 20);
            if (c->arena.avail() < need) break;
            uint64_t* bounds = (uint64_t*)c->arena.temp((uint64_t)nruns * (nseg + 1) * 8);
            uint64_t* d_off = (uint64_t*)c->arena.temp((nruns + 1) * 8);
            unsigned long long* d_max = (unsigned long long*)c->arena.temp(16);
            HIP_TRY(hipMemcpyAsync(d_off, run_off.data(), (nruns + 1) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemsetAsync(d_max, 0, 16, c->stream));
            PhaseTimer t(c, GOSS_T_REDUCE, total);
            for (uint32_t r = 0; r < nruns; ++r)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_bounds_kernel<K>), dim3(nseg / 256 + 1), dim3(256), 0, c->stream,
                                   (const K*)(ka + run_off[r]), run_off[r + 1] - run_off[r], shift, nseg,
                                   bounds + (uint64_t)r * (nseg + 1));
            hipLaunchKernelGGL(seg_totals_kernel, dim3(nseg / 256 + 1), dim3(256), 0, c->stream, (const uint64_t*)bounds, nruns, nseg, d_max);
            unsigned long long* hmax = (unsigned long long*)c->h_pinned;
            HIP_TRY(hipMemcpyAsync(hmax, d_max, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (hmax[0] > (unsigned long long)kMergeCap) { t.stop(); c->arena.release(m2); continue; }
            uint64_t* seg_pos = (uint64_t*)c->arena.temp((uint64_t)nseg * 8);
            uint64_t* seg_cnt = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
            uint64_t* seg_dst = (uint64_t*)c->arena.temp(((uint64_t)nseg + 1) * 8);
            SegOut* so = (SegOut*)c->arena.temp(sizeof(SegOut));
            SegOut hso{};
            hso.stage_cap = total;
            HIP_TRY(hipMemcpyAsync(so, &hso, sizeof(SegOut), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_merge_kernel<K>), unit_grid(nseg), dim3(kTB), 0, c->stream, (const K*)ka, (const uint32_t*)va,
                               (const uint64_t*)d_off, (const uint64_t*)bounds, nruns, nseg, so, seg_pos, seg_cnt, kb, vb, c->d_flags);
            SegOut* h = (SegOut*)c->h_pinned;
            HIP_TRY(hipMemcpyAsync(h, so, sizeof(SegOut), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (h->overflow) { t.stop(); c->arena.release(m2); break; }
            const uint64_t m = h->cursor;
            HIP_TRY(hipMemcpyAsync(seg_dst, seg_cnt, (uint64_t)nseg * 8, hipMemcpyDeviceToDevice, c->stream));
            exclusive_scan_u64(c, seg_dst, nseg);
            Run r{nullptr, nullptr, m};
            r.keys = c->arena.perm(std::max<uint64_t>(m, 1) * sizeof(K));
            r.counts = (uint32_t*)c->arena.perm(std::max<uint64_t>(m, 1) * 4);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(seg_gather_kernel<K>), unit_grid(nseg), dim3(kTB), 0, c->stream,
                               (const K*)kb, (const uint32_t*)vb, (const uint64_t*)seg_pos,
                               (const uint64_t*)seg_dst, (const uint64_t*)seg_cnt, (K*)r.keys, r.counts);
            t.stop();
            HIP_TRY(hipStreamSynchronize(c->stream));
            resolve_big_counts<K>(c, r, ka, va, run_off, in_bigs);
            r.rep = rep_out;
            c->runs.push_back(r);
            c->seg_merges++;
            c->arena.release(mark);
            return;
        }
    }
    bool in_b = radix_sort<K, true>(c, ka, kb, va, vb, total, key_digits(c));
    PhaseTimer t(c, GOSS_T_REDUCE, total);
    Run r = reduce_runs<K>(c, in_b ? kb : ka, in_b ? vb : va, total, in_b ? ka : kb, &in_bigs);
    t.stop();
    r.rep = rep_out;
    c->runs.push_back(r);
    c->arena.release(mark);
}

// Largest number of window starts one chunk may cover with the memory currently free.
// optimistic: size the chunk for the segment-hash path (two key buffers + look-back status;
// its output is bounded by kSegCount * kSegLimit entries).  If a chunk then needs the full
// sort with a large output and runs out of memory, push_device halves it and retries.
uint64_t chunk_capacity(goss_gpu_ctx* c, bool optimistic)
{
    const uint64_t ksz = c->words * 8;
    // (keys per window: two for a graph -- but ONE where the fused path counts strand pairs (round 4) and makes the other
    // strand from the counted pairs: sized for two, C4's chunks were half of what the arena holds, seven where four do;
    // a chunk the fused path declines after all asks for its full buffers and is cut in halves if they are not there)
    const uint32_t S = c->mode == GOSS_MODE_GRAPH && !(optimistic && c->graph_rep && c->fused) ? 2 : 1;
    uint64_t avail = c->arena.avail();
    double per_key;
    if (optimistic)
    {
        const uint64_t fixed = (uint64_t)(1u << kSegBits) * kSegLimit * (ksz + 4) + (256u << 20);
        if (avail <= fixed) return 0;
        avail -= fixed;
        per_key = 2.0 * ksz + 0.6;
        if (c->valid_frac < 0.97) per_key = (kValidSlackA + kValidSlackB) * c->valid_frac * ksz + 0.6;
    }
    else
    {
        // per key: two key buffers + histogram table + worst-case output (keys, counts, starts)
        per_key = 2.0 * ksz + 1.2 + (ksz + 12.0);
    }
    uint64_t keys = (uint64_t)((double)avail * 0.95 / per_key);
    uint64_t starts = keys / S;
    if (optimistic && c->valid_frac < 0.97 && starts <= kValidSizingMin + 4096)
        starts = (uint64_t)((double)avail * 0.95 / (2.0 * ksz + 0.6)) / S;     // such chunks get full buffers
    starts &= ~4095ULL;
    return starts;
}

// Share of the window starts of a pushed string that begin a valid window, estimated from 64 MB of
// evenly spread slices: a non-base removes at most `len` windows.  One point of margin; the
// fused path checks the figure against its own sample and asks for full buffers when it was low.
double estimate_valid_fraction(goss_gpu_ctx* c, const uint8_t* d, uint64_t nbytes)
{
    constexpr uint32_t kSlice = 16384;
    constexpr uint64_t kSlices = 4096;
    const uint8_t* al = (const uint8_t*)(((uintptr_t)d + 15) & ~(uintptr_t)15);
    const uint64_t skip = (uint64_t)(al - d);
    if (nbytes < skip + 2 * kSlices * kSlice) return 1.0;
    const uint64_t stride = ((nbytes - skip - kSlice) / (kSlices - 1)) & ~15ULL;
    c->mute_timing = true;
    HIP_TRY(hipMemsetAsync(c->d_ctr, 0, 16, c->stream));
    if (c->pk.on)
    {
        const uint8_t* src; const uint16_t* bad;
        pk_ptrs(c, al, &src, &bad);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nonbase_sample_kernel<true>), dim3(1024), dim3(kTB), 0, c->stream, (const uint8_t*)bad, kSlices, stride,
                           kSlice, (unsigned long long*)c->d_ctr);
    }
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nonbase_sample_kernel<false>), dim3(1024), dim3(kTB), 0, c->stream, al, kSlices, stride, kSlice,
                           (unsigned long long*)c->d_ctr);
    c->mute_timing = false;
    unsigned long long* h = (unsigned long long*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, c->d_ctr, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h[1] == 0) return 1.0;
    const double f = 1.0 - (double)h[0] / (double)h[1] * (double)c->len + 0.01;
    return std::min(1.0, std::max(0.05, f));
}

template <class K>
void push_device(goss_gpu_ctx* c, const uint8_t* d, uint64_t nbytes)
{
    if (nbytes < c->len) return;
    ensure_arena(c);
    const uint64_t nstarts_total = nbytes - c->len + 1;
    c->valid_frac = 1.0;
    if (c->size_by_valid && c->fused && c->path == 0 && nstarts_total > kValidSizingMin)
    {
        c->valid_frac = estimate_valid_fraction(c, d, nbytes);
        if (c->debug) std::fprintf(stderr, "libgossgpu: valid windows per window start, estimated: %.3f\n", c->valid_frac);
    }
    uint64_t done = 0;
    uint64_t limit = 0;                 // chunk size cap after an out-of-memory retry
    // (staging buffers are outside the arena: growing it moves nothing the bases live in)
    while (done < nstarts_total)
    {
        const bool optimistic = use_segment_path<K>(c);
        uint64_t capn = chunk_capacity(c, optimistic);
        if (capn < 4096) capn = chunk_capacity(c, false);
        if (capn < 4096)
        {
            // try to make room by merging what we have
            if (c->runs.size() > 1) { merge_runs<K>(c); capn = chunk_capacity(c, false); }
            if (capn < 4096 && grow_arena(c, 0)) capn = chunk_capacity(c, false);
            if (capn < 4096) throw StatusError{GOSS_ERR_OOM, "HBM budget too small for one chunk"};
        }
        if (limit && capn > limit) capn = limit;
        // equal chunks rather than full ones and a small remainder: a remainder covers the
        // genome too thinly for the segment path and would go through the full sort
        {
            const uint64_t left = nstarts_total - done;
            if (left > capn)
            {
                const uint64_t parts = (left + capn - 1) / capn;
                capn = std::min<uint64_t>(capn, ((left + parts - 1) / parts + 4095) & ~4095ULL);
            }
        }
        uint64_t ns = std::min(capn, nstarts_total - done);
        uint64_t navail = std::min(nbytes - done, ns + c->len - 1);
        const uint64_t lo0 = c->arena.lo, hi0 = c->arena.hi;
        const size_t runs0 = c->runs.size();
        // (behind this chunk: the rest of this push and what the push's own caller has said lies behind IT)
        struct MoreOff { goss_gpu_ctx* c; uint64_t was; ~MoreOff() { c->more_starts = was; } } moreOff{c, c->more_starts};
        c->more_starts = moreOff.was + (nstarts_total - done - ns);
        try
        {
            process_chunk<K>(c, d + done, ns, navail);
        }
        catch (const StatusError& e)
        {
            if (e.status != GOSS_ERR_OOM || ns <= 8192) throw;
            // undo the partial chunk; retry it in a larger arena if the budget may grow, else in halves
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->arena.lo = lo0; c->arena.hi = hi0;
            c->runs.resize(runs0);
            c->mute_timing = false;             // (a phase that gave up part-way may have left it set)
            if (grow_arena(c, 0)) continue;
            limit = (ns / 2) & ~4095ULL;
            continue;
        }
        done += ns;
        // keep the accumulated runs from eating the budget
        uint64_t run_bytes = 0;
        for (auto& r : c->runs) run_bytes += r.m * (c->words * 8 + 4);
        if (c->runs.size() > 1 && run_bytes > c->arena.size / 4)
        {
            // merging needs two more copies of the runs: possible now, or after growing; else the
            // runs wait for finish
            if (c->arena.avail() >= 2 * run_bytes + (512ULL << 20) || grow_arena(c, 2 * run_bytes + (512ULL << 20))) merge_runs<K>(c);
        }
    }
}

// A string of super-k-mer records (kernels_route.hpp) counted like a string of bases: the same chunk loop, a
// "window start" being one of a record's P window slots.  nwindows (0 = unknown) sizes the key buffers.
template <class K>
void push_records(goss_gpu_ctx* c, const uint8_t* d, uint64_t nrecs, uint64_t nwindows)
{
    if (nrecs == 0) return;
    ensure_arena(c);
    const uint64_t P = rec_slots(c);
    struct Mode { goss_gpu_ctx* c; ~Mode() { c->rec_mode = false; c->valid_frac = 1.0; } } mode{c};
    c->rec_mode = true;
    const uint64_t nstarts_total = nrecs * P;
    c->valid_frac = 1.0;
    if (nwindows && c->size_by_valid && c->fused && c->path == 0 && nstarts_total > kValidSizingMin)
        c->valid_frac = std::min(1.0, std::max(0.05, (double)nwindows / (double)nstarts_total + 0.01));
    uint64_t done = 0, limit = 0;
    while (done < nstarts_total)
    {
        const bool optimistic = use_segment_path<K>(c);
        uint64_t capn = chunk_capacity(c, optimistic);
        if (capn < 4096) capn = chunk_capacity(c, false);
        if (capn < 4096)
        {
            if (c->runs.size() > 1) { merge_runs<K>(c); capn = chunk_capacity(c, false); }
            if (capn < 4096 && grow_arena(c, 0)) capn = chunk_capacity(c, false);
            if (capn < 4096) throw StatusError{GOSS_ERR_OOM, "HBM budget too small for one chunk"};
        }
        if (limit && capn > limit) capn = limit;
        {
            const uint64_t left = nstarts_total - done;
            if (left > capn)
            {
                const uint64_t parts = (left + capn - 1) / capn;
                capn = std::min<uint64_t>(capn, ((left + parts - 1) / parts + 4095) & ~4095ULL);
            }
        }
        const uint64_t ns = std::min(capn, nstarts_total - done);          // (a multiple of 4096 slots = whole records, but for the last)
        const uint64_t lo0 = c->arena.lo, hi0 = c->arena.hi;
        const size_t runs0 = c->runs.size();
        try
        {
            process_chunk<K>(c, d + (done / P) * rec_bytes(c), ns, 0);
        }
        catch (const StatusError& e)
        {
            if (e.status != GOSS_ERR_OOM || ns <= 8192) throw;
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->arena.lo = lo0; c->arena.hi = hi0;
            c->runs.resize(runs0);
            c->mute_timing = false;             // (a phase that gave up part-way may have left it set)
            if (grow_arena(c, 0)) continue;
            limit = (ns / 2) & ~4095ULL;
            continue;
        }
        done += ns;
        uint64_t run_bytes = 0;
        for (auto& r : c->runs) run_bytes += r.m * (c->words * 8 + 4);
        if (c->runs.size() > 1 && run_bytes > c->arena.size / 4 &&
            (c->arena.avail() >= 2 * run_bytes + (512ULL << 20) || grow_arena(c, 2 * run_bytes + (512ULL << 20)))) merge_runs<K>(c);
    }
}

// ---- on-disk arrays ---------------------------------------------------------------------

struct IaCol { std::string suffix; uint32_t bytes; uint32_t shift; };

// IntegerArray::builder column layout (IntegerArray.cc:259-357, StackedArray.hh:152-178).
bool ia_layout(uint32_t bits, const std::string& prefix, uint32_t shift, std::vector<IaCol>& out)
{
    uint32_t ub = 0, lb = 0;
    switch (bits)
    {
        case 8: case 16: case 32: case 64: out.push_back({prefix, bits / 8, shift}); return true;
        case 24: ub = 8; lb = 16; break;
        case 40: ub = 8; lb = 32; break;
        case 48: ub = 16; lb = 32; break;
        case 56: ub = 8; lb = 48; break;
        case 72: ub = 8; lb = 64; break;
        case 80: ub = 16; lb = 64; break;
        case 88: ub = 8; lb = 80; break;
        case 96: ub = 32; lb = 64; break;
        case 104: ub = 8; lb = 96; break;
        case 112: ub = 16; lb = 96; break;
        case 120: ub = 24; lb = 96; break;
        case 128: ub = 64; lb = 64; break;
        default: return false;
    }
    return ia_layout(ub, prefix + ".upr", shift + lb, out) && ia_layout(lb, prefix + ".lwr", shift, out);
}

// SparseArray::Builder::d (SparseArray.cc:47-72).
uint64_t sparse_d(uint64_t n_lo, uint64_t n_hi, uint64_t M)
{
    double scale = 18446744073709551616.0;
    double n = (double)n_hi * scale + (double)n_lo;
    double m = (double)M;
    double d0 = std::log2(n / ((1 + m) * 1.4426950408889634));
    uint64_t d = (uint64_t)std::ceil(d0);
    if (d < 8) d = 8; else if (d > 128) d = 128;
    return d;
}

struct DsHeader {
    uint64_t version, flags, indexArrayOffset, rankArrayOffset;
    uint64_t logBlockSize, blockSize, logSampleRate, sampleRate;
    uint64_t numBlocks, indexSize, smallBlocks, smallBlocksSize;
    uint64_t intermediateBlocks, intermediateBlocksSize, largeBlocks, largeBlocksSize;
};
static_assert(sizeof(DsHeader) == 128, "DenseSelect header is 128 bytes");

// DenseSelect file image over the virtual position sequence (DenseArray.cc:446-694).
template <class K>
void emit_dense_select(goss_gpu_ctx* c, const K* keys, uint64_t m, uint32_t D, int invert, uint64_t count,
                       const std::string& suffix)
{
    const uint64_t nblocks = (count + 8191) >> 13;
    DsHeader h{};
    h.version = 2012092701ULL;
    h.flags = invert ? 1 : 0;
    h.logBlockSize = 13; h.blockSize = 8192; h.logSampleRate = 6; h.sampleRate = 64;
    std::vector<uint32_t> btype(nblocks);
    std::vector<uint64_t> bbytes(nblocks), boff(nblocks);
    uint64_t mark = c->arena.mark();
    uint32_t* d_btype = nullptr; uint64_t* d_bbytes = nullptr; uint64_t* d_brank = nullptr; uint64_t* d_boff = nullptr;
    if (nblocks)
    {
        d_btype = (uint32_t*)c->arena.temp(nblocks * 4);
        d_bbytes = (uint64_t*)c->arena.temp(nblocks * 8);
        d_brank = (uint64_t*)c->arena.temp(nblocks * 8);
        d_boff = (uint64_t*)c->arena.temp(nblocks * 8);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ds_classify_kernel<K>), dim3(grid_for(nblocks, 64)), dim3(64), 0, c->stream,
                           DsSrc<K>{keys, m, D, invert, 0, nullptr, 0}, count, (uint64_t)0, nblocks, d_btype, d_bbytes, d_brank);
        HIP_TRY(hipMemcpyAsync(btype.data(), d_btype, nblocks * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(bbytes.data(), d_bbytes, nblocks * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    uint64_t pos = 4096;
    for (uint64_t b = 0; b < nblocks; ++b)
    {
        boff[b] = pos;
        uint64_t cnt = std::min<uint64_t>(8192, count - (b << 13));
        switch (btype[b])
        {
            case kDsSmall: h.smallBlocks++; h.smallBlocksSize += 256; break;
            case kDsIntermediate: h.intermediateBlocks++; h.intermediateBlocksSize += bbytes[b]; break;
            case kDsSpill32: h.largeBlocks++; h.largeBlocksSize += cnt * 4; break;
            default: h.largeBlocks++; h.largeBlocksSize += cnt * 8; break;
        }
        pos += bbytes[b];
    }
    h.numBlocks = nblocks;
    pos = (pos + 15) & ~15ULL;
    h.indexArrayOffset = pos;
    h.rankArrayOffset = pos + nblocks * 8;
    h.indexSize = nblocks * 16;
    const uint64_t size = pos + nblocks * 16;
    uint8_t* image = (uint8_t*)c->arena.perm(size);
    HIP_TRY(hipMemsetAsync(image, 0, size, c->stream));
    HIP_TRY(hipMemcpyAsync(image, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    if (nblocks)
    {
        HIP_TRY(hipMemcpyAsync(d_boff, boff.data(), nblocks * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ds_fill_kernel<K>), dim3((uint32_t)nblocks), dim3(128), 0, c->stream,
                           DsSrc<K>{keys, m, D, invert, 0, nullptr, 0}, count, (uint64_t)0, (const uint32_t*)d_btype, (const uint64_t*)d_boff,
                           (const uint64_t*)d_brank, image, (uint64_t*)(image + h.indexArrayOffset));
        HIP_TRY(hipMemcpyAsync(image + h.rankArrayOffset, d_brank, nblocks * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));    // host vectors / header go out of scope
    c->arena.release(mark);
    OutFile f; f.suffix = suffix; f.size = size; f.dev = image;
    c->files.push_back(std::move(f));
}

// ---- DenseSelect blocks per range (round 5) ---------------------------------------------------------------------------
// A block of 8192 indexed positions is a function of those positions alone (DenseArray.cc:446-647): type, bytes and body.
// A range's owner builds the blocks that lie wholly inside its share of the ones (or zeros); the assembler builds the few
// that straddle two ranges from the bitmap, adds up the sizes, copies the bodies to their offsets and writes the master
// index and the rank array.  A range's blocks travel as ONE record: {nb, payload bytes, 0, 0}, bytes[nb], rank[nb],
// type[nb] (u32, padded to 8), the bodies back to back.
constexpr uint64_t kPartSpan = 1, kPartDs = 2;          // kinds of the records of a ".part.span" file: {kind, a, b, body bytes}, the body
struct DsPlan { uint64_t b0 = 0, nb = 0, payload = 0; std::vector<uint32_t> type; std::vector<uint64_t> bytes, rank; uint32_t* d_type = nullptr; uint64_t* d_rank = nullptr; };
static inline uint64_t ds_record_bytes(const DsPlan& p) { return 32 + p.nb * 16 + ((p.nb * 4 + 7) & ~7ULL) + p.payload; }

// types, sizes and first positions of blocks b0 .. b0 + nb - 1 (temporaries of the arena stay for ds_fill_record)
template <class K>
DsPlan ds_plan(goss_gpu_ctx* c, const DsSrc<K>& src, uint64_t count, uint64_t b0, uint64_t nb)
{
    DsPlan p; p.b0 = b0; p.nb = nb;
    if (!nb) return p;
    p.type.resize(nb); p.bytes.resize(nb); p.rank.resize(nb);
    p.d_type = (uint32_t*)c->arena.temp(nb * 4);
    uint64_t* d_bytes = (uint64_t*)c->arena.temp(nb * 8);
    p.d_rank = (uint64_t*)c->arena.temp(nb * 8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ds_classify_kernel<K>), dim3(grid_for(nb, 64)), dim3(64), 0, c->stream, src, count, b0, nb, p.d_type, d_bytes, p.d_rank);
    HIP_TRY(hipMemcpyAsync(p.type.data(), p.d_type, nb * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(p.bytes.data(), d_bytes, nb * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(p.rank.data(), p.d_rank, nb * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint64_t b = 0; b < nb; ++b) p.payload += p.bytes[b];
    return p;
}
// the record of a plan at `at` (ds_record_bytes(p) bytes, zeroed by the caller)
template <class K>
void ds_fill_record(goss_gpu_ctx* c, const DsSrc<K>& src, uint64_t count, const DsPlan& p, uint8_t* at)
{
    const uint64_t head[4] = {p.nb, p.payload, 0, 0};
    HIP_TRY(hipMemcpyAsync(at, head, 32, hipMemcpyHostToDevice, c->stream));
    if (p.nb)
    {
        uint8_t* a_bytes = at + 32; uint8_t* a_rank = a_bytes + p.nb * 8; uint8_t* a_type = a_rank + p.nb * 8;
        uint8_t* body = a_type + ((p.nb * 4 + 7) & ~7ULL);
        std::vector<uint64_t> boff(p.nb);
        uint64_t pos = 0;
        for (uint64_t b = 0; b < p.nb; ++b) { boff[b] = pos; pos += p.bytes[b]; }
        uint64_t* d_boff = (uint64_t*)c->arena.temp(p.nb * 8);
        HIP_TRY(hipMemcpyAsync(d_boff, boff.data(), p.nb * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(a_bytes, p.bytes.data(), p.nb * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(a_rank, p.d_rank, p.nb * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(a_type, p.d_type, p.nb * 4, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ds_fill_kernel<K>), dim3((uint32_t)p.nb), dim3(128), 0, c->stream, src, count, p.b0, (const uint32_t*)p.d_type,
                           (const uint64_t*)d_boff, (const uint64_t*)p.d_rank, body, (uint64_t*)nullptr);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));          // (host vectors go out of scope)
}

// A DenseSelect file from the records of the ranges (device addresses on this context's device: `recs` = {first block, the
// record}) and, for every block none of them holds, from the assembled bitmap (words, nwords, ones before every word).
void ds_compose(goss_gpu_ctx* c, int invert, uint64_t count, const std::vector<std::pair<uint64_t, const uint8_t*>>& recs,
                const unsigned long long* words, uint64_t nwords, const uint64_t* before, const std::string& suffix)
{
    const uint64_t NB = (count + 8191) >> 13;
    std::vector<uint32_t> type(NB);
    std::vector<uint64_t> bytes(NB), rank(NB), boff(NB);
    std::vector<uint8_t> have(NB, 0);
    struct Seg { uint64_t b0, nb; const uint8_t* body; uint64_t payload; };
    std::vector<Seg> segs;
    uint64_t mark = c->arena.mark();
    for (auto& r : recs)
    {
        uint64_t head[4];
        HIP_TRY(hipMemcpyAsync(head, r.second, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint64_t b0 = r.first, nb = head[0];
        if (!nb) continue;
        if (b0 + nb > NB) throw StatusError{GOSS_ERR_INVALID_ARG, "emit_assemble: a range's blocks lie beyond the array"};
        const uint8_t* a_bytes = r.second + 32; const uint8_t* a_rank = a_bytes + nb * 8; const uint8_t* a_type = a_rank + nb * 8;
        HIP_TRY(hipMemcpyAsync(bytes.data() + b0, a_bytes, nb * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(rank.data() + b0, a_rank, nb * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(type.data() + b0, a_type, nb * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (uint64_t b = b0; b < b0 + nb; ++b) { if (have[b]) throw StatusError{GOSS_ERR_INVALID_ARG, "emit_assemble: a block built twice"}; have[b] = 1; }
        c->ds_blocks_ranges += nb;
        segs.push_back({b0, nb, a_type + ((nb * 4 + 7) & ~7ULL), head[1]});
    }
    // the blocks nobody built (they straddle two ranges, or their range gave no record): runs of them, from the bitmap
    struct Own { DsPlan plan; uint64_t* pos; uint64_t first; uint8_t* rec; };
    std::vector<Own> own;
    for (uint64_t b = 0; b < NB;)
    {
        if (have[b]) { ++b; continue; }
        uint64_t e = b;
        while (e < NB && !have[e] && e - b < 4096) ++e;          // (at most 32 M positions at a time)
        const uint64_t first = b << 13, n = std::min<uint64_t>(count, e << 13) - first;
        uint64_t* pos = (uint64_t*)c->arena.temp(n * 8);
        hipLaunchKernelGGL(ef_select_range_kernel, dim3(grid_for(n, kTB)), dim3(kTB), 0, c->stream, words, nwords, before, first, n, invert, pos);
        const DsSrc<Key1> src{nullptr, 0, 0, invert, 0, pos, first};
        Own o; o.pos = pos; o.first = first;
        o.plan = ds_plan<Key1>(c, src, count, b, e - b);
        const uint64_t rb = ds_record_bytes(o.plan);
        o.rec = (uint8_t*)c->arena.temp(rb);
        HIP_TRY(hipMemsetAsync(o.rec, 0, rb, c->stream));
        ds_fill_record<Key1>(c, src, count, o.plan, o.rec);
        for (uint64_t x = b; x < e; ++x) { type[x] = o.plan.type[x - b]; bytes[x] = o.plan.bytes[x - b]; rank[x] = o.plan.rank[x - b]; have[x] = 1; }
        segs.push_back({b, e - b, o.rec + 32 + (e - b) * 16 + (((e - b) * 4 + 7) & ~7ULL), o.plan.payload});
        own.push_back(std::move(o));
        c->ds_blocks_own += e - b;
        b = e;
    }
    DsHeader h{};
    h.version = 2012092701ULL;
    h.flags = invert ? 1 : 0;
    h.logBlockSize = 13; h.blockSize = 8192; h.logSampleRate = 6; h.sampleRate = 64;
    uint64_t pos = 4096;
    for (uint64_t b = 0; b < NB; ++b)
    {
        boff[b] = pos;
        const uint64_t cnt = std::min<uint64_t>(8192, count - (b << 13));
        switch (type[b])
        {
            case kDsSmall: h.smallBlocks++; h.smallBlocksSize += 256; break;
            case kDsIntermediate: h.intermediateBlocks++; h.intermediateBlocksSize += bytes[b]; break;
            case kDsSpill32: h.largeBlocks++; h.largeBlocksSize += cnt * 4; break;
            default: h.largeBlocks++; h.largeBlocksSize += cnt * 8; break;
        }
        pos += bytes[b];
    }
    h.numBlocks = NB;
    pos = (pos + 15) & ~15ULL;
    h.indexArrayOffset = pos;
    h.rankArrayOffset = pos + NB * 8;
    h.indexSize = NB * 16;
    const uint64_t size = pos + NB * 16;
    // (the image is permanent, the temporaries above it are released below: it goes first in the arena's order -- taken
    // from the permanent end, which grows from the other side)
    uint8_t* image = (uint8_t*)c->arena.perm(size);
    HIP_TRY(hipMemsetAsync(image, 0, size, c->stream));
    HIP_TRY(hipMemcpyAsync(image, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    for (auto& sg : segs)
        if (sg.payload) HIP_TRY(hipMemcpyAsync(image + boff[sg.b0], sg.body, sg.payload, hipMemcpyDeviceToDevice, c->stream));
    for (uint64_t b = 0; b < NB; ++b) boff[b] |= type[b];          // the master index entries
    if (NB)
    {
        HIP_TRY(hipMemcpyAsync(image + h.indexArrayOffset, boff.data(), NB * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(image + h.rankArrayOffset, rank.data(), NB * 8, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
    OutFile f; f.suffix = suffix; f.size = size; f.dev = image;
    c->files.push_back(std::move(f));
}

struct SaHeader { uint64_t version, D, quantizedD, mask_lo, mask_hi, size_lo, size_hi, count; };
static_assert(sizeof(SaHeader) == 64, "SparseArray header is 64 bytes");

// nd = Nend >> D, which must fit 64 bits (SparseArray.cc:79-86)
uint64_t sparse_nd(uint32_t D, uint64_t Nend_lo, uint64_t Nend_hi)
{
    if (D >= 128) return 0;
    if (D >= 64) return Nend_hi >> (D - 64);
    if (D > 0 && (Nend_hi >> D) != 0) throw StatusError{GOSS_ERR_TOO_LARGE, "Internal error in SparseArray; nd does not fit 64 bits"};
    if (D == 0 && Nend_hi) throw StatusError{GOSS_ERR_TOO_LARGE, "Internal error in SparseArray; nd does not fit 64 bits"};
    return D == 0 ? Nend_lo : ((Nend_lo >> D) | (Nend_hi << (64 - D)));
}

// SparseArray::Header (SparseArray.hh:60-72) as the ".header" file
void emit_sparse_header(goss_gpu_ctx* c, uint32_t D, uint64_t Nend_lo, uint64_t Nend_hi, uint64_t count, const std::string& base)
{
    SaHeader h{};
    h.version = 2012030501ULL; h.D = D; h.quantizedD = 8 * ((D + 7) / 8);
    if (D >= 128) { h.mask_lo = ~0ULL; h.mask_hi = ~0ULL; }
    else if (D >= 64) { h.mask_lo = ~0ULL; h.mask_hi = D == 64 ? 0 : ((1ULL << (D - 64)) - 1); }
    else { h.mask_lo = (1ULL << D) - 1; h.mask_hi = 0; }
    h.size_lo = Nend_lo; h.size_hi = Nend_hi; h.count = count;
    OutFile f; f.suffix = base + ".header"; f.size = sizeof h;
    f.host.assign((uint8_t*)&h, (uint8_t*)&h + sizeof h);
    c->files.push_back(std::move(f));
}

// The index part of a SparseArray: high-bits bitmap, -d0 (zeros), -d1 (ones), from keys whose high part
// is key >> D -- or from the high parts themselves with D = 0 (distributed emission).
template <class K>
void emit_sparse_index(goss_gpu_ctx* c, const K* keys, uint64_t m, uint32_t D, uint64_t nd, const std::string& base, uint64_t* built = nullptr)
{
    const uint64_t nwords = (nd + m + 3) / 64 + 1;
    uint64_t* words = built ? built : (uint64_t*)c->arena.perm(nwords * 8);
    if (built) {}                                // (assembled from the ranges' spans: emit_assemble)
    else if (m >= (1u << 16) && !c->ef_by_words)
    {
        // from the keys' side: one coalesced pass over the keys (GOSS_GPU_EF_BY_WORDS=1: the per-word binary search)
        HIP_TRY(hipMemsetAsync(words, 0, nwords * 8, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_high_bits_keys_kernel<K>), dim3(grid_for(m, kEfChunk)), dim3(kTB), 0, c->stream,
                           keys, m, D, (uint64_t)0, nwords, (unsigned long long*)words, (uint64_t)0);
    }
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_high_bits_kernel<K>), dim3(grid_for(nwords, 256)), dim3(256), 0, c->stream,
                           keys, m, D, nwords, words);
    OutFile f; f.suffix = base + ".high-bits"; f.size = nwords * 8; f.dev = (const uint8_t*)words;
    c->files.push_back(std::move(f));
    emit_dense_select<K>(c, keys, m, D, 1, nd + 2, base + "-d0");
    emit_dense_select<K>(c, keys, m, D, 0, m, base + "-d1");
}

// The low-bits column files of m keys (IntegerArray / StackedArray columns of quantizedD bits)
template <class K>
void emit_sparse_low_bits(goss_gpu_ctx* c, const K* keys, uint64_t m, uint32_t D, const std::string& base)
{
    const uint32_t qD = 8 * ((D + 7) / 8);
    std::vector<IaCol> cols;
    if (!ia_layout(qD, "", 0, cols)) throw StatusError{GOSS_ERR_INVALID_ARG, "IntegerArray::builder: unsupported integer width"};
    EfColumns ec{};
    ec.n = (uint32_t)cols.size();
    for (size_t i = 0; i < cols.size(); ++i)
    {
        uint8_t* dst = (uint8_t*)c->arena.perm(std::max<uint64_t>(m * cols[i].bytes, 8));
        ec.c[i] = EfColumn{dst, cols[i].bytes, cols[i].shift};
        OutFile f; f.suffix = base + ".low-bits" + cols[i].suffix; f.size = m * cols[i].bytes; f.dev = dst;
        c->files.push_back(std::move(f));
    }
    if (m)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_low_bits_kernel<K>), dim3(grid_for(m, 256)), dim3(256), 0, c->stream,
                           keys, m, D, ec);
}

// SparseArray at suffix `base` (SparseArray::Builder ctor + push_back* + end).
template <class K>
void emit_sparse_array(goss_gpu_ctx* c, const K* keys, uint64_t m, uint64_t N_lo, uint64_t N_hi, uint64_t Mest,
                       uint64_t Nend_lo, uint64_t Nend_hi, const std::string& base)
{
    const uint32_t D = (uint32_t)sparse_d(N_lo, N_hi, Mest);
    const uint64_t nd = sparse_nd(D, Nend_lo, Nend_hi);
    // every key's high part must fit too (SparseArray.hh:91-95)
    HIP_TRY(hipMemsetAsync(c->d_flags + 1, 0, 4, c->stream));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_check_kernel<K>), dim3(1), dim3(64), 0, c->stream, keys, m, D, c->d_flags + 1);
    emit_sparse_header(c, D, Nend_lo, Nend_hi, m, base);
    emit_sparse_index<K>(c, keys, m, D, nd, base);
    emit_sparse_low_bits<K>(c, keys, m, D, base);
    uint32_t* hf = (uint32_t*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(hf, c->d_flags + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hf[0]) throw StatusError{GOSS_ERR_TOO_LARGE, "SparseArray::push_back: high bits do not fit 64 bits"};
}

void add_host_file(goss_gpu_ctx* c, const std::string& suffix, const void* p, size_t n)
{
    OutFile f; f.suffix = suffix; f.size = n;
    f.host.assign((const uint8_t*)p, (const uint8_t*)p + n);
    c->files.push_back(std::move(f));
}

// VariableByteArray + counts histogram (VariableByteArray.hh:76-118, Graph.cc:115-134).
void emit_counts(goss_gpu_ctx* c, const uint32_t* counts, uint64_t m, uint64_t num_items, const std::string& out_counts, const std::string& out_hist)
{
    uint64_t mark = c->arena.mark();
    uint8_t* ord0 = (uint8_t*)c->arena.perm(std::max<uint64_t>(m, 8));
    uint64_t n1 = 0, n2 = 0;
    Key1* pos1 = nullptr; uint8_t* ord1 = nullptr; Key1* pos2 = nullptr; uint16_t* ord2 = nullptr;
    uint64_t* h = (uint64_t*)c->h_pinned;
    if (m)
    {
        uint64_t* slot = (uint64_t*)c->arena.temp((m + 1) * 8);
        hipLaunchKernelGGL(vba_ord0_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m, ord0, slot);
        HIP_TRY(hipMemsetAsync(slot + m, 0, 8, c->stream));
        exclusive_scan_u64(c, slot, m + 1);
        HIP_TRY(hipMemcpyAsync(h, slot + m, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        n1 = h[0];
        pos1 = (Key1*)c->arena.perm(std::max<uint64_t>(n1 * 8, 8));
        ord1 = (uint8_t*)c->arena.perm(std::max<uint64_t>(n1, 8));
        if (n1)
        {
            uint32_t* hi16 = (uint32_t*)c->arena.temp(n1 * 4);
            uint64_t* slot2 = (uint64_t*)c->arena.temp((n1 + 1) * 8);
            hipLaunchKernelGGL(vba_ord1_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m,
                               (const uint64_t*)slot, n1, (uint64_t*)pos1, ord1, hi16);
            hipLaunchKernelGGL(vba_flag2_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, c->stream, (const uint32_t*)hi16, n1, slot2);
            HIP_TRY(hipMemsetAsync(slot2 + n1, 0, 8, c->stream));
            exclusive_scan_u64(c, slot2, n1 + 1);
            HIP_TRY(hipMemcpyAsync(h, slot2 + n1, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            n2 = h[0];
            pos2 = (Key1*)c->arena.perm(std::max<uint64_t>(n2 * 8, 8));
            ord2 = (uint16_t*)c->arena.perm(std::max<uint64_t>(n2 * 2, 8));
            if (n2)
                hipLaunchKernelGGL(vba_ord2_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, c->stream, (const uint32_t*)hi16, n1,
                                   (const uint64_t*)slot2, (uint64_t*)pos2, ord2);
        }
    }
    if (!pos1) pos1 = (Key1*)c->arena.perm(8);
    if (!pos2) pos2 = (Key1*)c->arena.perm(8);
    if (!ord1) ord1 = (uint8_t*)c->arena.perm(8);
    if (!ord2) ord2 = (uint16_t*)c->arena.perm(8);
    { OutFile f; f.suffix = out_counts + ".ord0"; f.size = m; f.dev = ord0; c->files.push_back(std::move(f)); }
    const uint64_t mest = (uint64_t)((double)num_items * 0.001);
    emit_sparse_array<Key1>(c, pos1, n1, num_items, 0, mest, m, 0, out_counts + ".ord1p");
    { OutFile f; f.suffix = out_counts + ".ord1"; f.size = n1; f.dev = ord1; c->files.push_back(std::move(f)); }
    emit_sparse_array<Key1>(c, pos2, n2, num_items, 0, mest, n1, 0, out_counts + ".ord2p");
    { OutFile f; f.suffix = out_counts + ".ord2"; f.size = n2 * 2; f.dev = (const uint8_t*)ord2; c->files.push_back(std::move(f)); }

    // histogram of counts: sort the counts as keys, run-length them, format on the host
    std::string text;
    if (m)
    {
        Key1* ka = (Key1*)c->arena.temp(m * 8);
        Key1* kb = (Key1*)c->arena.temp(m * 8);
        hipLaunchKernelGGL(widen_counts_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m, ka);
        bool in_b = radix_sort<Key1, false>(c, ka, kb, nullptr, nullptr, m, 4);
        const Key1* sorted = in_b ? kb : ka;
        Key1* distinct = in_b ? ka : kb;
        const uint64_t ntiles = (m + kRedTile - 1) / kRedTile;
        uint64_t* tile_counts = (uint64_t*)c->arena.temp((ntiles + 1) * 8);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_count_kernel<Key1>), dim3(grid_for(m, kRedTile)), dim3(kTB), 0, c->stream,
                           sorted, m, tile_counts);
        HIP_TRY(hipMemsetAsync(tile_counts + ntiles, 0, 8, c->stream));
        exclusive_scan_u64(c, tile_counts, ntiles + 1);
        HIP_TRY(hipMemcpyAsync(h, tile_counts + ntiles, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint64_t nd = h[0];
        uint64_t* starts = (uint64_t*)c->arena.temp(nd * 8);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_write_kernel<Key1>), dim3(grid_for(m, kRedTile)), dim3(kTB), 0, c->stream,
                           sorted, m, (const uint64_t*)tile_counts, distinct, starts);
        std::vector<uint64_t> hv(nd), hs(nd);
        HIP_TRY(hipMemcpyAsync(hv.data(), distinct, nd * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(hs.data(), starts, nd * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::map<uint64_t, uint64_t> hist;
        for (uint64_t i = 0; i < nd; ++i) hist[hv[i]] = (i + 1 < nd ? hs[i + 1] : m) - hs[i];
        // the result's counts beyond 32 bits sit in the array modulo 2^32; the histogram is keyed by the count
        // itself (Graph.hh:101-106: mHist[count]++ on the u64 before it is narrowed)
        if (counts == c->res_counts)
            for (auto& kv : c->res_big)
            {
                auto it = hist.find(kv.second & 0xFFFFFFFFULL);
                if (it != hist.end() && --it->second == 0) hist.erase(it);
                hist[kv.second] += 1;
            }
        char line[64];
        for (auto& kv : hist)
        {
            int l = snprintf(line, sizeof line, "%llu\t%llu\n", (unsigned long long)kv.first, (unsigned long long)kv.second);
            text.append(line, (size_t)l);
        }
    }
    add_host_file(c, out_hist, text.data(), text.size());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
}

template <class K>
void emit_object(goss_gpu_ctx* c)
{
    const K* keys = (const K*)c->res_keys;
    const uint64_t m = c->M;
    // the SparseArray estimate: the exact distinct count (single-pass build) unless the caller
    // gave one (merges use the sum of the inputs' counts, GossCmdMerge.tcc:256-258,296)
    const uint64_t est = c->has_emit_estimate ? c->emit_estimate : m;
    PhaseTimer t(c, GOSS_T_EMIT, m);
    if (c->mode == GOSS_MODE_KMER_SET)
    {
        // N = 4^K (KmerSet.hh:84,72)
        uint32_t bits = 2 * c->k;
        uint64_t nlo = bits < 64 ? (1ULL << bits) : 0, nhi = bits >= 64 ? (1ULL << (bits - 64)) : 0;
        emit_sparse_array<K>(c, keys, m, nlo, nhi, est, nlo, nhi, ".kmers");
        uint64_t hdr[3] = {2011101701ULL, c->k, m};
        add_host_file(c, ".header", hdr, sizeof hdr);
    }
    else
    {
        uint64_t hdr[3] = {2011101014ULL, c->k, 0};
        add_host_file(c, ".header", hdr, sizeof hdr);
        uint32_t bits = 2 * c->k + 2;
        uint64_t nlo = bits < 64 ? (1ULL << bits) : 0, nhi = bits >= 64 ? (1ULL << (bits - 64)) : 0;
        emit_sparse_array<K>(c, keys, m, nlo, nhi, est, nlo, nhi, "-edges");
        emit_counts(c, c->res_counts, m, est, "-counts", "-counts-hist.txt");
    }
    t.stop();
}

// N = 4^len of the object's SparseArray (KmerSet.hh:84, Graph.cc:120-124) and its file name base
void object_universe(const goss_gpu_ctx* c, uint64_t* nlo, uint64_t* nhi, std::string* base)
{
    const uint32_t bits = c->mode == GOSS_MODE_KMER_SET ? 2 * c->k : 2 * c->k + 2;
    *nlo = bits < 64 ? (1ULL << bits) : 0;
    *nhi = bits >= 64 ? (1ULL << (bits - 64)) : 0;
    *base = c->mode == GOSS_MODE_KMER_SET ? ".kmers" : "-edges";
}

// Distributed emission, the part a range's owner builds from its own keys (goss_gpu_emit_part).
// prev_high: the high part (key >> D) of the last key of the ranges below this one (0 when there is none) -- what tells a
// range which zeros of the bitmap are its own; have_prev: it was given (else the range builds no blocks of "-d0").
template <class K>
void emit_part(goss_gpu_ctx* c, uint64_t first_index, uint64_t total, uint64_t estimate, uint64_t prev_high = 0, bool have_prev = false)
{
    const K* keys = (const K*)c->res_keys;
    const uint64_t m = c->M;
    PhaseTimer t(c, GOSS_T_EMIT, m);
    uint64_t nlo, nhi; std::string base;
    object_universe(c, &nlo, &nhi, &base);
    const uint32_t D = (uint32_t)sparse_d(nlo, nhi, estimate);
    const uint64_t nd = sparse_nd(D, nlo, nhi);
    HIP_TRY(hipMemsetAsync(c->d_flags + 1, 0, 4, c->stream));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_check_kernel<K>), dim3(1), dim3(64), 0, c->stream, keys, m, D, c->d_flags + 1);
    emit_sparse_low_bits<K>(c, keys, m, D, base);
    // This range's SPAN of the high-bits bitmap for the assembling rank: the words its ones fall into (one i of the
    // whole array sits at bit (key_i >> D) + i), built from the range's own keys -- about 2.4 bits per key on the wire
    // where the first form of this path sent key >> D of every key (4 or 8 bytes) -- and, behind it, the DenseSelect blocks
    // that lie wholly inside the range's share of the ones ("-d1") and of the zeros ("-d0": with prev_high).  Records
    // {kind, a, b, body bytes} + body: the span {kPartSpan, first word, words} + the words; blocks {kPartDs, sense (0 ones,
    // 1 zeros), first block} + ds_fill_record's record.
    {
        const uint64_t nwords = (nd + total + 3) / 64 + 1;
        uint64_t w0 = 0, nspan = 0, last_high = prev_high;
        if (m)
        {
            K ends[2];
            HIP_TRY(hipMemcpyAsync(&ends[0], keys, sizeof(K), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(&ends[1], keys + (m - 1), sizeof(K), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const uint64_t h0 = (D >= 128 ? 0 : key_shr64(ends[0], D)) + first_index;
            last_high = D >= 128 ? 0 : key_shr64(ends[1], D);
            const uint64_t h1 = last_high + first_index + (m - 1);
            w0 = h0 >> 6;
            nspan = std::min((h1 >> 6) + 1, nwords) - w0;
        }
        const uint64_t tmark = c->arena.mark();
        // whole blocks of ones [first_index, first_index + m) -- the array's last, partial block is the last range's --
        // and of zeros [prev_high, last_high) (the last range: up to the array's nd + 2 zeros)
        // (the LAST NON-EMPTY range owns the tails: an empty range behind it -- first_index == total, m == 0 -- would satisfy
        // first_index + m == total as well and build the same tail blocks a second time; an empty range builds no block,
        // and when every range is empty the assembler builds them all from the bitmap)
        const bool last_range = m > 0 && first_index + m == total;
        const uint64_t count1 = total, count0 = nd + 2;
        DsPlan p1, p0;
        const DsSrc<K> src1{keys, m, D, 0, first_index, nullptr, 0}, src0{keys, m, D, 1, first_index, nullptr, 0};
        if (m && c->ds_parts)
        {
            const uint64_t b0 = (first_index + 8191) >> 13;
            const uint64_t e = last_range ? (count1 + 8191) >> 13 : (first_index + m) >> 13;
            if (e > b0) p1 = ds_plan<K>(c, src1, count1, b0, e - b0);
        }
        if (have_prev && c->ds_parts && m)
        {
            const uint64_t z0 = prev_high, z1 = last_range ? count0 : last_high;
            const uint64_t b0 = (z0 + 8191) >> 13;
            const uint64_t e = last_range ? (count0 + 8191) >> 13 : z1 >> 13;
            if (e > b0) p0 = ds_plan<K>(c, src0, count0, b0, e - b0);
        }
        const uint64_t r1 = p1.nb ? 32 + ds_record_bytes(p1) : 0, r0 = p0.nb ? 32 + ds_record_bytes(p0) : 0;
        const uint64_t bytes = 32 + nspan * 8 + r1 + r0;
        uint8_t* blob = (uint8_t*)c->arena.perm(bytes);
        HIP_TRY(hipMemsetAsync(blob, 0, bytes, c->stream));
        const uint64_t hdr[4] = {kPartSpan, w0, nspan, nspan * 8};
        HIP_TRY(hipMemcpyAsync(blob, hdr, 32, hipMemcpyHostToDevice, c->stream));
        if (m)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_high_bits_keys_kernel<K>), dim3(grid_for(m, kEfChunk)), dim3(kTB), 0, c->stream,
                               keys, m, D, first_index, nwords, (unsigned long long*)(blob + 32), w0);
        uint8_t* at = blob + 32 + nspan * 8;
        const uint64_t h1[4] = {kPartDs, 0, p1.b0, ds_record_bytes(p1)}, h0[4] = {kPartDs, 1, p0.b0, ds_record_bytes(p0)};
        if (p1.nb)
        {
            HIP_TRY(hipMemcpyAsync(at, h1, 32, hipMemcpyHostToDevice, c->stream));
            ds_fill_record<K>(c, src1, count1, p1, at + 32);
            at += r1;
        }
        if (p0.nb)
        {
            HIP_TRY(hipMemcpyAsync(at, h0, 32, hipMemcpyHostToDevice, c->stream));
            ds_fill_record<K>(c, src0, count0, p0, at + 32);
            at += r0;
        }
        HIP_TRY(hipStreamSynchronize(c->stream));            // (the headers go out of scope)
        c->arena.release(tmark);
        OutFile f; f.suffix = ".part.span"; f.size = bytes; f.dev = (const uint8_t*)blob;
        c->files.push_back(std::move(f));
    }
    if (c->mode == GOSS_MODE_GRAPH)
    {
        const uint32_t* counts = c->res_counts;
        uint64_t mark = c->arena.mark();
        uint8_t* ord0 = (uint8_t*)c->arena.perm(std::max<uint64_t>(m, 8));
        uint64_t nbig = 0;
        BigCount* big = nullptr;
        uint64_t* h = (uint64_t*)c->h_pinned;
        std::vector<uint64_t> hist;                      // (count, frequency) pairs, ascending
        if (m)
        {
            uint64_t* slot = (uint64_t*)c->arena.temp((m + 1) * 8);
            hipLaunchKernelGGL(vba_ord0_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m, ord0, slot);
            HIP_TRY(hipMemsetAsync(slot + m, 0, 8, c->stream));
            exclusive_scan_u64(c, slot, m + 1);
            HIP_TRY(hipMemcpyAsync(h, slot + m, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            nbig = h[0];
            big = (BigCount*)c->arena.perm(std::max<uint64_t>(nbig * sizeof(BigCount), 16));
            if (nbig)
                hipLaunchKernelGGL(vba_big_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m, (const uint64_t*)slot,
                                   first_index, big);
            // histogram of this range's counts: sort them as keys, run-length them
            Key1* ka = (Key1*)c->arena.temp(m * 8);
            Key1* kb = (Key1*)c->arena.temp(m * 8);
            hipLaunchKernelGGL(widen_counts_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m, ka);
            const bool mute = c->mute_timing; c->mute_timing = true;
            const bool in_b = radix_sort<Key1, false>(c, ka, kb, nullptr, nullptr, m, 4);
            c->mute_timing = mute;
            const Key1* sorted = in_b ? kb : ka;
            Key1* distinct = in_b ? ka : kb;
            const uint64_t ntiles = (m + kRedTile - 1) / kRedTile;
            uint64_t* tile_counts = (uint64_t*)c->arena.temp((ntiles + 1) * 8);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_count_kernel<Key1>), dim3(grid_for(m, kRedTile)), dim3(kTB), 0, c->stream,
                               sorted, m, tile_counts);
            HIP_TRY(hipMemsetAsync(tile_counts + ntiles, 0, 8, c->stream));
            exclusive_scan_u64(c, tile_counts, ntiles + 1);
            HIP_TRY(hipMemcpyAsync(h, tile_counts + ntiles, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const uint64_t nv = h[0];
            uint64_t* starts = (uint64_t*)c->arena.temp(std::max<uint64_t>(nv, 1) * 8);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(heads_write_kernel<Key1>), dim3(grid_for(m, kRedTile)), dim3(kTB), 0, c->stream,
                               sorted, m, (const uint64_t*)tile_counts, distinct, starts);
            std::vector<uint64_t> hv(nv), hs(nv);
            HIP_TRY(hipMemcpyAsync(hv.data(), distinct, nv * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(hs.data(), starts, nv * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            std::map<uint64_t, uint64_t> hm;
            for (uint64_t i = 0; i < nv; ++i) hm[hv[i]] = (i + 1 < nv ? hs[i + 1] : m) - hs[i];
            for (auto& kv : c->res_big)              // counts beyond 32 bits: keyed by the count itself (see emit_counts)
            {
                auto it = hm.find(kv.second & 0xFFFFFFFFULL);
                if (it != hm.end() && --it->second == 0) hm.erase(it);
                hm[kv.second] += 1;
            }
            for (auto& kv : hm) { hist.push_back(kv.first); hist.push_back(kv.second); }
        }
        { OutFile f; f.suffix = "-counts.ord0"; f.size = m; f.dev = ord0; c->files.push_back(std::move(f)); }
        { OutFile f; f.suffix = ".part.big"; f.size = nbig * sizeof(BigCount); f.dev = (const uint8_t*)big; if (!big) f.host.clear(); c->files.push_back(std::move(f)); }
        add_host_file(c, ".part.hist", hist.data(), hist.size() * 8);
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
    }
    t.stop();
    uint32_t* hf = (uint32_t*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(hf, c->d_flags + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hf[0]) throw StatusError{GOSS_ERR_TOO_LARGE, "SparseArray::push_back: high bits do not fit 64 bits"};
    (void)total;
}

// Distributed emission, the assembling side (goss_gpu_emit_assemble): everything that needs all ranges.
void emit_assemble(goss_gpu_ctx* c, const void* d_spans, uint64_t span_bytes, uint64_t total, uint64_t estimate,
                   const BigCount* big, uint64_t nbig, const uint64_t* hist, uint64_t nhist)
{
    PhaseTimer t(c, GOSS_T_EMIT, total);
    uint64_t nlo, nhi; std::string base;
    object_universe(c, &nlo, &nhi, &base);
    const uint32_t D = (uint32_t)sparse_d(nlo, nhi, estimate);
    const uint64_t nd = sparse_nd(D, nlo, nhi);
    c->ds_blocks_ranges = c->ds_blocks_own = 0;
    // the bitmap: the ranges' spans ORed together (neighbours share their boundary words); the ranges' DenseSelect blocks
    // noted for the composition below
    const uint64_t nwords = (nd + total + 3) / 64 + 1;
    uint64_t* words = (uint64_t*)c->arena.perm(nwords * 8);
    HIP_TRY(hipMemsetAsync(words, 0, nwords * 8, c->stream));
    std::vector<std::pair<uint64_t, const uint8_t*>> ds_recs[2];          // [sense]: {first block, record}
    for (uint64_t at = 0; at + 32 <= span_bytes;)
    {
        uint64_t hdr[4];
        HIP_TRY(hipMemcpyAsync(hdr, (const uint8_t*)d_spans + at, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint8_t* body = (const uint8_t*)d_spans + at + 32;
        if (at + 32 + hdr[3] > span_bytes) throw StatusError{GOSS_ERR_INVALID_ARG, "emit_assemble: a record runs past the parts"};
        if (hdr[0] == kPartSpan)
        {
            if (hdr[1] + hdr[2] > nwords || hdr[3] != hdr[2] * 8) throw StatusError{GOSS_ERR_INVALID_ARG, "emit_assemble: a span does not fit the bitmap"};
            if (hdr[2])
                hipLaunchKernelGGL(ef_or_span_kernel, dim3(grid_for(hdr[2], kTB)), dim3(kTB), 0, c->stream, (unsigned long long*)words + hdr[1],
                                   (const unsigned long long*)body, hdr[2]);
        }
        else if (hdr[0] == kPartDs && hdr[1] < 2) ds_recs[hdr[1]].push_back({hdr[2], body});
        else throw StatusError{GOSS_ERR_INVALID_ARG, "emit_assemble: not a part's record"};
        at += 32 + hdr[3];
    }
    // ones in the words below every word: what finds the positions of the blocks no range could build alone
    uint64_t mark = c->arena.mark();
    uint64_t* ones = (uint64_t*)c->arena.temp(nwords * 8);
    hipLaunchKernelGGL(ef_word_ones_kernel, dim3(grid_for(nwords, kTB)), dim3(kTB), 0, c->stream, (const unsigned long long*)words, nwords, ones);
    exclusive_scan_u64(c, ones, nwords);
    auto index_files = [&]() {
        OutFile f; f.suffix = base + ".high-bits"; f.size = nwords * 8; f.dev = (const uint8_t*)words;
        c->files.push_back(std::move(f));
        ds_compose(c, 1, nd + 2, ds_recs[1], (const unsigned long long*)words, nwords, ones, base + "-d0");
        ds_compose(c, 0, total, ds_recs[0], (const unsigned long long*)words, nwords, ones, base + "-d1");
    };
    if (c->mode == GOSS_MODE_KMER_SET)
    {
        emit_sparse_header(c, D, nlo, nhi, total, base);
        index_files();
        uint64_t hdr[3] = {2011101701ULL, c->k, total};
        add_host_file(c, ".header", hdr, sizeof hdr);
    }
    else
    {
        uint64_t hdr[3] = {2011101014ULL, c->k, 0};
        add_host_file(c, ".header", hdr, sizeof hdr);
        emit_sparse_header(c, D, nlo, nhi, total, base);
        index_files();
        // VariableByteArray continuation arrays from the entries with count > 255 (VariableByteArray.hh:76-118):
        // ord1p marks their global positions, ord1 holds bits 8..15; ord2p marks, among those, the ones with
        // count > 65535 by their index in ord1, ord2 holds bits 16..31
        std::vector<uint64_t> pos1(nbig), pos2;
        std::vector<uint8_t> ord1(nbig);
        std::vector<uint16_t> ord2;
        for (uint64_t i = 0; i < nbig; ++i)
        {
            pos1[i] = big[i].index;
            ord1[i] = (uint8_t)((big[i].count >> 8) & 0xFF);
            if (big[i].count >> 16) { pos2.push_back(i); ord2.push_back((uint16_t)(big[i].count >> 16)); }
        }
        const uint64_t mest = (uint64_t)((double)estimate * 0.001);
        Key1* d1 = (Key1*)c->arena.temp(std::max<uint64_t>(nbig, 1) * 8);
        Key1* d2 = (Key1*)c->arena.temp(std::max<uint64_t>(pos2.size(), 1) * 8);
        if (nbig) HIP_TRY(hipMemcpyAsync(d1, pos1.data(), nbig * 8, hipMemcpyHostToDevice, c->stream));
        if (!pos2.empty()) HIP_TRY(hipMemcpyAsync(d2, pos2.data(), pos2.size() * 8, hipMemcpyHostToDevice, c->stream));
        emit_sparse_array<Key1>(c, d1, nbig, estimate, 0, mest, total, 0, "-counts.ord1p");
        add_host_file(c, "-counts.ord1", ord1.data(), ord1.size());
        emit_sparse_array<Key1>(c, d2, pos2.size(), estimate, 0, mest, nbig, 0, "-counts.ord2p");
        add_host_file(c, "-counts.ord2", ord2.data(), ord2.size() * 2);
        // histogram: the ranges' (count, frequency) pairs added up (Graph.cc:131-150 writes them ascending)
        std::map<uint64_t, uint64_t> all;
        for (uint64_t i = 0; i < nhist; ++i) all[hist[2 * i]] += hist[2 * i + 1];
        std::string text;
        char line[64];
        for (auto& kv : all)
        {
            int l = snprintf(line, sizeof line, "%llu\t%llu\n", (unsigned long long)kv.first, (unsigned long long)kv.second);
            text.append(line, (size_t)l);
        }
        add_host_file(c, "-counts-hist.txt", text.data(), text.size());
    }
    t.stop();
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
}

void wait_background_thread(goss_gpu_ctx* c);

// wait_bg: the call works on the arena or the runs -- the thread that counts a full staging buffer must have ended
// (its failure becomes this call's).  Only the host pushes, which touch nothing but the other staging buffer, go on
// beside it.
template <class F>
int guarded(goss_gpu_ctx* c, F&& f, bool wait_bg = true)
{
    try
    {
        if (c) HIP_TRY(hipSetDevice(c->device));
        // forget what the calling thread left behind (e.g. hipErrorNotReady from an event or stream
        // poll of the host process): from here on an error is ours
        (void)hipGetLastError();
        if (c && wait_bg) wait_background_thread(c);
        f();
        // a refused kernel launch raises no exception by itself and leaves its outputs untouched:
        // no entry point returns success over one
        check_launch("a kernel launch was refused");
        return GOSS_OK;
    }
    catch (const HipError& e)
    {
        if (c) c->last_error = std::string(e.what) + ": " + hipGetErrorString(e.e);
        return GOSS_ERR_HIP;
    }
    catch (const StatusError& e)
    {
        if (c) c->last_error = e.msg;
        return e.status;
    }
    catch (const std::bad_alloc&)
    {
        if (c) c->last_error = "host allocation failed";
        return GOSS_ERR_OOM;
    }
    catch (const std::system_error& e)
    {
        // (a thread that could not be started: nothing may cross the C boundary)
        if (c) c->last_error = std::string("a host resource is exhausted: ") + e.what();
        return GOSS_ERR_OOM;
    }
    catch (const std::exception& e)
    {
        if (c) c->last_error = std::string("unexpected failure: ") + e.what();
        return GOSS_ERR_STATE;
    }
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
static const char* const kBrokenMsg = "an exchange round of this context's group failed half way: reads staged before it may be lost (goss_gpu_reset starts over)";

extern "C" {

const char* goss_gpu_strerror(int status)
{
    switch (status)
    {
        case GOSS_OK: return "ok";
        case GOSS_ERR_INVALID_ARG: return "invalid argument";
        case GOSS_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
        case GOSS_ERR_OOM: return "out of memory (HBM budget)";
        case GOSS_ERR_HIP: return "HIP runtime error";
        case GOSS_ERR_STATE: return "call out of order";
        case GOSS_ERR_K_RANGE: return "unable to build a graph with that k";
        case GOSS_ERR_COUNT_OVERFLOW: return "a key count does not fit 32 bits";
        case GOSS_ERR_TOO_LARGE: return "SparseArray high bits do not fit 64 bits";
        case GOSS_ERR_BUFFER: return "a buffer of the caller is too small";
        default: return "unknown status";
    }
}

const char* goss_gpu_last_error(const goss_gpu_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

uint32_t goss_gpu_abi_version(void) { return kAbiVersion; }

int goss_gpu_create(goss_gpu_ctx** out, int device, uint32_t k, int mode, uint64_t hbm_budget, void* stream)
{
    if (!out) return GOSS_ERR_INVALID_ARG;
    *out = nullptr;
    if (mode != GOSS_MODE_KMER_SET && mode != GOSS_MODE_GRAPH) return GOSS_ERR_INVALID_ARG;
    if (k == 0) return GOSS_ERR_K_RANGE;
    if (mode == GOSS_MODE_KMER_SET && k > 63) return GOSS_ERR_K_RANGE;   // KmerSet::MaxK
    if (mode == GOSS_MODE_GRAPH && k > 62) return GOSS_ERR_K_RANGE;      // Graph::MaxK
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GOSS_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GOSS_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return GOSS_ERR_NO_DEVICE;
    goss_gpu_ctx* c = new (std::nothrow) goss_gpu_ctx();
    if (!c) return GOSS_ERR_OOM;
    c->device = device; c->k = k; c->mode = mode;
    c->len = mode == GOSS_MODE_GRAPH ? k + 1 : k;
    c->words = (2 * c->len <= 62) ? 1 : 2;
    c->budget = hbm_budget;
    { const char* e = std::getenv("GOSS_GPU_NO_LOOKBACK"); if (e && *e == '1') c->lookback = false; }
    { const char* e = std::getenv("GOSS_GPU_ORDERED_TILES"); if (e && *e == '1') c->ordered_tiles = true; }
    { const char* e = std::getenv("GOSS_GPU_EST_SCALE"); if (e && std::atof(e) > 0) c->est_scale = std::atof(e); }
    { const char* e = std::getenv("GOSS_GPU_ORDER_BITS"); if (e && std::atoi(e) >= 16 && std::atoi(e) <= 24) c->order_bits = (uint32_t)std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_EXTRACT_V1"); if (e && *e == '1') c->extract_v1 = true; }
    { const char* e = std::getenv("GOSS_GPU_NO_CURSOR_PASS0"); if (e && *e == '1') c->cursor_pass0 = false; }
    { const char* e = std::getenv("GOSS_GPU_NO_FUSED"); if (e && *e == '1') c->fused = false; }
    { const char* e = std::getenv("GOSS_GPU_FUSED_GRID"); if (e && *e) c->fused_grid = (uint32_t)std::strtoul(e, nullptr, 10); }
    { const char* e = std::getenv("GOSS_GPU_NO_SEG_MERGE"); if (e && *e == '1') c->seg_merge = false; }
    { const char* e = std::getenv("GOSS_GPU_NO_MSD"); if (e && *e == '1') c->fused_msd = false; }
    { const char* e = std::getenv("GOSS_GPU_EF_BY_WORDS"); if (e && *e && *e != '0') c->ef_by_words = true; }
    { const char* e = std::getenv("GOSS_GPU_NO_GRAPH_REP"); if (e && *e && *e != '0') c->graph_rep = false; }
    { const char* e = std::getenv("GOSS_GPU_CANON_L1"); if (e && *e >= '0' && *e <= '2') c->canon_l1 = *e - '0'; }
    { const char* e = std::getenv("GOSS_GPU_CANON_L1_AT"); if (e && std::atof(e) > 0) c->canon_l1_at = std::atof(e); }
    { const char* e = std::getenv("GOSS_GPU_NO_REM32"); if (e && *e && *e != '0') c->rem32 = false; }
    { const char* e = std::getenv("GOSS_GPU_REM32_BITS"); if (e && std::atoi(e) >= 9 && std::atoi(e) <= 10) c->rem32_bits_min = (uint32_t)std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_REM32_SPLIT"); if (e && std::atoi(e) >= 0 && std::atoi(e) <= 4) c->rem32_split_min = (uint32_t)std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_REM32_SLOTS"); if (e && (std::atoi(e) == 2048 || std::atoi(e) == 4096 || std::atoi(e) == 8192 || std::atoi(e) == 16384)) c->rem32_slots = std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_R32_FORM"); if (e) c->r32_form = std::atoi(e) ? 1 : 0; }
    { const char* e = std::getenv("GOSS_GPU_NARROW"); if (e) c->narrow = std::atoi(e) != 0; }
    { const char* e = std::getenv("GOSS_GPU_OVERFLOW_BY_SORT"); if (e) c->overflow_by_sort = std::atoi(e) != 0; }
    { const char* e = std::getenv("GOSS_GPU_DS_PARTS"); if (e) c->ds_parts = std::atoi(e) != 0; }
    { const char* e = std::getenv("GOSS_GPU_NARROW_CAPG"); if (e && std::atoi(e) > 0) c->narrow_capg = (uint32_t)std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_R32_SMALL_MAX"); if (e) c->r32_small_max = (uint32_t)std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_R32_BIG"); if (e && *e == '0') c->big_r32 = false; }
    { const char* e = std::getenv("GOSS_GPU_NO_BIG_TABLE"); if (e && *e && *e != '0') c->big_table = false; }
    { const char* e = std::getenv("GOSS_GPU_HASH_MERGE_MIN"); if (e && *e) c->hash_merge_min = std::strtoull(e, nullptr, 10); }
    { const char* e = std::getenv("GOSS_GPU_BLK_LOG2"); if (e && *e) c->blk_log2_max = (uint32_t)std::atoi(e); }
    { const char* e = std::getenv("GOSS_GPU_NO_FAST32"); if (e && *e == '1') c->no_fast32 = true; }
    { const char* e = std::getenv("GOSS_GPU_NO_TABLE96"); if (e && *e && *e != '0') c->table96 = false; }
    { const char* e = std::getenv("GOSS_GPU_NO_WIDE_TABLE"); if (e && *e && *e != '0') c->wide_table = false; }
    { const char* e = std::getenv("GOSS_GPU_BIG_ROUNDS"); if (e && *e) c->big_rounds_max = std::min(3, std::max(0, std::atoi(e))); }
    { const char* e = std::getenv("GOSS_GPU_BIG_ROUNDS_MIN"); if (e && *e) c->big_rounds_min = std::min(3, std::max(0, std::atoi(e))); }
    { const char* e = std::getenv("GOSS_GPU_NO_VALID_SIZING"); if (e && *e && *e != '0') c->size_by_valid = false; }
    { const char* e = std::getenv("GOSS_GPU_FUSED_MIN"); if (e && *e) c->fused_min = std::strtoull(e, nullptr, 10); }
    { const char* e = std::getenv("GOSS_GPU_FUSED_CAPSCALE"); if (e && *e) c->fused_capscale = std::atof(e); }
    { const char* e = std::getenv("GOSS_GPU_DEBUG"); if (e && *e == '1') c->debug = true; }
    int rc = guarded(c, [&]() {
        if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
        else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
        HIP_TRY(hipMalloc((void**)&c->d_ctr, sizeof(ExtractCounters)));
        HIP_TRY(hipMalloc((void**)&c->d_flags, 16));
        HIP_TRY(hipMemsetAsync(c->d_flags, 0, 16, c->stream));
        HIP_TRY(hipHostMalloc(&c->h_pinned, 256, hipHostMallocDefault));
    });
    if (rc != GOSS_OK) { goss_gpu_destroy(c); return rc == GOSS_ERR_HIP ? GOSS_ERR_NO_DEVICE : rc; }
    *out = c;
    return GOSS_OK;
}

int goss_gpu_prepare(goss_gpu_ctx* c)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (c->arena.base || c->arena_thread.joinable()) return GOSS_OK;
    c->arena_status = GOSS_OK;
    c->arena_thread = std::thread([c]() {
        try
        {
            HIP_TRY(hipSetDevice(c->device));
            map_arena(c);
        }
        catch (const HipError& e) { c->arena_status = GOSS_ERR_HIP; c->arena_error = std::string(e.what) + ": " + hipGetErrorString(e.e); }
        catch (const StatusError& e) { c->arena_status = e.status; c->arena_error = e.msg; }
    });
    return GOSS_OK;
}

void goss_gpu_destroy(goss_gpu_ctx* c)
{
    if (!c) return;
    if (c->arena_thread.joinable()) c->arena_thread.join();
    (void)hipSetDevice(c->device);
    if (c->bg.joinable()) c->bg.join();
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& pd : c->pending) { if (pd.fn) pd.fn(pd.user); (void)hipEventDestroy(pd.ev); }
    for (auto e : c->pend_pool) (void)hipEventDestroy(e);
    for (auto*& b : c->stage_buf) if (b) (void)hipFree(b);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& pe : c->events) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->arena.base) (void)hipFree(c->arena.base);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->d_route) (void)hipFree(c->d_route);
    if (c->grp_send) (void)hipFree(c->grp_send);
    if (c->grp_inbox) (void)hipFree(c->grp_inbox);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Host pushes are gathered in a staging buffer in HBM and counted together: every counted chunk
// becomes a sorted run that has to be merged later, so many small chunks are far more expensive
// than one large one.  Pushes are separated by one non-base byte (windows never span pushes).
// The counting thread's outcome, taken over by the caller's thread (every entry point but the host pushes does this
// first: one thread at a time works on the context's arena and runs).
static void wait_background(goss_gpu_ctx* c);
namespace { void wait_background_thread(goss_gpu_ctx* c) { wait_background(c); } }
static void wait_background(goss_gpu_ctx* c)
{
    if (!c->bg.joinable()) return;
    c->bg.join();
    if (c->bg_status != GOSS_OK)
    {
        const int st = c->bg_status;
        c->bg_status = GOSS_OK;
        throw StatusError{st, c->bg_error};
    }
}

static void count_staged(goss_gpu_ctx* c, const uint8_t* buf, uint64_t n, bool packed = false)
{
    const auto t1 = std::chrono::steady_clock::now();
    struct PkOff { goss_gpu_ctx* c; ~PkOff() { c->pk.on = false; } } pkOff{c};
    if (packed)
    {
        // (a packed buffer: counted as the string at the made-up address kPkFake, which the launch sites resolve)
        c->pk.on = true; c->pk.codes = pk_codes_of((uint8_t*)buf); c->pk.bad = pk_bad_of((uint8_t*)buf, c->stage_cap);
        buf = (const uint8_t*)kPkFake;
    }
    if (c->words == 1) push_device<Key1>(c, buf, n); else push_device<Key2>(c, buf, n);
    c->flush_count_us += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    c->flushes++;
}

// What is staged is counted now, on the caller's thread (finish, a device push, anything that needs the runs).
static void flush_staging(goss_gpu_ctx* c)
{
    wait_background(c);
    if (!c->stage || c->stage_fill == 0) return;
    const uint64_t n = c->stage_fill;
    c->stage_fill = 0;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipStreamSynchronize(c->copy_stream));      // (the copies queued so far: counted apart from the chunk's own time)
    c->flush_wait_us += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    count_staged(c, c->stage, n, c->stage_pk[c->stage_cur]);
}

// A staging buffer is full while the caller keeps pushing: a thread of the library counts it -- its kernels wait on
// the device for the copies that filled it -- and the caller goes on with the other buffer.
static void flush_staging_background(goss_gpu_ctx* c)
{
    if (c->deferred && c->stage && c->stage_fill)
        throw StatusError{GOSS_ERR_BUFFER, "the staging buffer is full and counting is deferred: goss_gpu_group_route_exchange first (goss_gpu_stage_room says how much fits)"};
    wait_background(c);                                  // (one buffer is counted at a time: this is where a producer that outruns the device waits)
    if (!c->stage || c->stage_fill == 0) return;
    const uint64_t n = c->stage_fill;
    uint8_t* buf = c->stage;
    hipEvent_t e;
    if (!c->pend_pool.empty()) { e = c->pend_pool.back(); c->pend_pool.pop_back(); }
    else HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(e, c->copy_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
    c->pend_pool.push_back(e);                           // (the wait has been queued: the event may be recorded again)
    // (the caller's next copies go to the other buffer: the background run below reads this one.  The buffer being
    // switched to is free: wait_background above returned only after the run that read it had synchronised its stream)
    const int cur0 = c->stage_cur;
    const bool packed = c->stage_pk[cur0];
    c->stage_cur ^= 1;
    c->stage = c->stage_buf[c->stage_cur];
    c->stage_fill = 0;
    c->bg_status = GOSS_OK;
    try {
    c->bg = std::thread([c, buf, n, packed]() {
        try
        {
            HIP_TRY(hipSetDevice(c->device));
            (void)hipGetLastError();
            // (the buffer filled up under a caller that goes on pushing: this chunk is one of several)
            struct MoreOff { goss_gpu_ctx* c; ~MoreOff() { c->more_follows = false; } } moreOff{c};
            c->more_follows = true;
            count_staged(c, buf, n, packed);
            check_launch("a kernel launch was refused");
        }
        catch (const HipError& e) { c->bg_status = GOSS_ERR_HIP; c->bg_error = std::string(e.what) + ": " + hipGetErrorString(e.e); }
        catch (const StatusError& e) { c->bg_status = e.status; c->bg_error = e.msg; }
        catch (const std::bad_alloc&) { c->bg_status = GOSS_ERR_OOM; c->bg_error = "host allocation failed"; }
    });
    }
    catch (const std::system_error& e)
    {
        // no thread to be had: nothing was counted and nothing may be lost -- the staged bases are the current buffer's again
        c->stage_cur = cur0;
        c->stage = c->stage_buf[cur0];
        c->stage_fill = n;
        throw StatusError{GOSS_ERR_OOM, std::string("no thread for the counting of a staging buffer: ") + e.what()};
    }
}

int goss_gpu_push_bases_device(goss_gpu_ctx* c, const void* d_bases, uint64_t nbytes)
{
    if (!c || (!d_bases && nbytes)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        flush_staging(c);
        if (c->words == 1) push_device<Key1>(c, (const uint8_t*)d_bases, nbytes);
        else push_device<Key2>(c, (const uint8_t*)d_bases, nbytes);
    });
}

// What the caller means to push in all (the `goss` commands know their files' sizes): lets the first chunks of a build
// of many choose the key space they count in for the whole build rather than for themselves.
int goss_gpu_expect_bases(goss_gpu_ctx* c, uint64_t total_bases)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    c->expect_bases = total_bases;
    return GOSS_OK;
}

// Packed bases that are already resident in HBM: counted where they lie, like goss_gpu_push_bases_device's bytes -- the
// context's packed string (ctx.pk) is the caller's two arrays for the length of the call.
int goss_gpu_push_packed_device(goss_gpu_ctx* c, const uint32_t* d_codes, const uint16_t* d_nonbase, uint64_t nbases)
{
    if (!c || (nbases && (!d_codes || !d_nonbase))) return GOSS_ERR_INVALID_ARG;
    if (((uintptr_t)d_codes & 3u) || ((uintptr_t)d_nonbase & 1u)) { c->last_error = "packed arrays must be aligned to their elements"; return GOSS_ERR_INVALID_ARG; }
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    if (c->deferred) { c->last_error = "a deferred context takes host pushes only (its staging buffer is what the group routes)"; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        flush_staging(c);
        struct PkOff { goss_gpu_ctx* c; ~PkOff() { c->pk.on = false; } } pkOff{c};
        c->pk.on = true; c->pk.codes = d_codes; c->pk.bad = d_nonbase;
        if (c->words == 1) push_device<Key1>(c, (const uint8_t*)kPkFake, nbases);
        else push_device<Key2>(c, (const uint8_t*)kPkFake, nbases);
    });
}

// The byte form of a base string in HBM -> the packed form in the caller's two arrays (ceil(nbytes / 16) elements
// each), on the context's stream; returns when the arrays are written.
int goss_gpu_pack_bases_device(goss_gpu_ctx* c, const void* d_bases, uint64_t nbytes, uint32_t* d_codes, uint16_t* d_nonbase)
{
    if (!c || (nbytes && (!d_bases || !d_codes || !d_nonbase))) return GOSS_ERR_INVALID_ARG;
    if (((uintptr_t)d_codes & 3u) || ((uintptr_t)d_nonbase & 1u)) { c->last_error = "packed arrays must be aligned to their elements"; return GOSS_ERR_INVALID_ARG; }
    return guarded(c, [&]() {
        if (!nbytes) return;
        const uintptr_t addr = (uintptr_t)d_bases;
        const uint32_t mis = (uint32_t)(addr & 15u);
        const uint64_t groups = (nbytes + 15) / 16;
        hipLaunchKernelGGL(pack_bases_kernel, dim3(grid_for(groups, kTB)), dim3(kTB), 0, c->stream, (const uint8_t*)(addr - mis), mis, nbytes,
                           d_codes, d_nonbase, groups);
        check_launch("a kernel launch was refused");
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
}

// Buffers of asynchronous pushes whose copies have completed go back to the caller (from the caller's own
// thread, inside this call); wait = all of them.
static void release_pending(goss_gpu_ctx* c, bool wait)
{
    size_t done = 0;
    for (; done < c->pending.size(); ++done)
    {
        auto& pd = c->pending[done];
        if (wait) HIP_TRY(hipEventSynchronize(pd.ev));
        else
        {
            const hipError_t e = hipEventQuery(pd.ev);
            if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }          // (the stream is in order: the rest is not ready either)
            if (e != hipSuccess) throw HipError{e, "hipEventQuery(pending push)"};
        }
        if (pd.fn) pd.fn(pd.user);
        c->pend_pool.push_back(pd.ev);
    }
    c->pending.erase(c->pending.begin(), c->pending.begin() + done);
}

static void note_pending(goss_gpu_ctx* c, void (*fn)(void*), void* user)
{
    hipEvent_t e;
    if (!c->pend_pool.empty()) { e = c->pend_pool.back(); c->pend_pool.pop_back(); }
    else HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(e, c->copy_stream));
    c->pending.push_back({e, fn, user});
}

// An asynchronous push registers its buffer and hands back the buffers of earlier pushes whose copies are done.  The
// contract of a failed push (include/goss_gpu.h): the library does NOT call release for it -- the buffer is the
// caller's again on return -- so when handing back fails, the entry just registered is taken out again, after the
// copies queued from the buffer have run (the caller may free it at once).
static void note_pending_checked(goss_gpu_ctx* c, void (*fn)(void*), void* user)
{
    note_pending(c, fn, user);
    try { release_pending(c, false); }
    catch (...)
    {
        (void)hipStreamSynchronize(c->copy_stream);
        if (!c->pending.empty() && c->pending.back().fn == fn && c->pending.back().user == user)
        {
            c->pend_pool.push_back(c->pending.back().ev);
            c->pending.pop_back();
        }
        throw;
    }
}

static void ensure_stage(goss_gpu_ctx* c)
{
    if (c->stage) return;
    wait_background(c);
    ensure_arena(c);
    // 1/24 of the arena each: with ~17 bytes of key workspace per base the staged bases then
    // fill about three quarters of the arena when they are counted
    c->stage_cap = std::max<uint64_t>(c->arena.avail() / 24, 1u << 20) & ~4095ULL;
    {
        // (an arena that was given nearly all of the device leaves less than that beside it: smaller buffers then)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // (a deferred context -- a member of a group that exchanges records -- also keeps its routed records and its
        // inbox beside the arena: 0.27 records per staged byte to send and a ninth more to receive, 12 or 20 bytes each --
        // 3.2 + 3.6 staging buffers' worth for one-word keys, 5.3 + 6 for two-word keys; they are allocated at the first
        // exchange and must still find room then)
        const uint64_t units = !c->deferred ? 2u : c->words == 1 ? 9u : 14u;
        const uint64_t room = free_b > (768ULL << 20) ? (free_b - (512ULL << 20)) / units : (128ULL << 20) / units;
        const uint64_t fits = room > (48ULL << 20) ? (room - (32ULL << 20)) : (1ULL << 20);
        if (c->stage_cap > fits) c->stage_cap = std::max<uint64_t>(fits, 1u << 20) & ~4095ULL;
    }
    // (tests: a small staging buffer makes a small input take several flushes / exchange rounds)
    if (const char* e = std::getenv("GOSS_GPU_STAGE_CAP"))
    {
        const uint64_t v = std::strtoull(e, nullptr, 10);
        if (v) c->stage_cap = std::min<uint64_t>(c->stage_cap, std::max<uint64_t>(v, 1u << 16) & ~4095ULL);
    }
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    // (one block each: stage_cap bytes, or -- packed -- 6 bytes per sixteen positions of them: codes, then flags)
    const uint64_t bytes = c->stage_cap + 16 + 256 + 64;
    for (auto*& b : c->stage_buf)
        if (!b && hipMalloc((void**)&b, bytes) != hipSuccess) { (void)hipGetLastError(); b = nullptr; throw StatusError{GOSS_ERR_OOM, "no device memory for the staging buffers"}; }
    c->stage_cur = 0;
    c->stage = c->stage_buf[0];
    c->stage_fill = 0;
}

static inline bool is_base_byte(char ch) { const char l = (char)(ch | 0x20); return l == 'a' || l == 'c' || l == 'g' || l == 't'; }

// bases -> staging buffer.  release == nullptr && !async: the copy has completed on return (the caller may
// reuse its buffer); else the buffer comes back through release(user) once its copy has.
static void push_bases_host(goss_gpu_ctx* c, const char* bases, uint64_t nbytes, bool async, void (*release)(void*), void* user)
{
    if (nbytes < c->len) { if (release) release(user); return; }
    ensure_stage(c);
    if (nbytes + 1 > c->stage_cap)
    {
        if (c->deferred) throw StatusError{GOSS_ERR_BUFFER, "a push larger than the staging buffer while counting is deferred"};
        // larger than the staging buffer: count it on its own, in pieces that overlap by
        // len-1 bytes so that no window is lost at a cut
        flush_staging(c);
        uint64_t done = 0;
        const uint64_t nstarts_total = nbytes - c->len + 1;
        while (done < nstarts_total)
        {
            uint64_t ns = std::min<uint64_t>(c->stage_cap - c->len, nstarts_total - done);
            uint64_t nb = ns + c->len - 1;
            c->stage_pk[c->stage_cur] = false;
            HIP_TRY(hipMemcpyAsync(c->stage, bases + done, nb, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipStreamSynchronize(c->copy_stream));
            struct MoreOff { goss_gpu_ctx* c; ~MoreOff() { c->more_starts = 0; } } moreOff{c};
            c->more_starts = nstarts_total - done - ns;          // (the pieces behind this one: the chunks choose their key space for all of them)
            count_staged(c, c->stage, nb);
            HIP_TRY(hipStreamSynchronize(c->stream));
            done += ns;
        }
        if (release) release(user);
        return;
    }
    // (a buffer holds bytes or packed groups: what is staged in the other form is counted first -- or, when counting is
    // deferred, is the exchange round's to take)
    if (c->stage_fill && c->stage_pk[c->stage_cur])
    {
        if (c->deferred) throw StatusError{GOSS_ERR_BUFFER, "the staging buffer holds packed bases and counting is deferred: goss_gpu_group_route_exchange first"};
        flush_staging_background(c);
    }
    if (c->stage_fill + nbytes + 1 > c->stage_cap) flush_staging_background(c);
    c->stage_pk[c->stage_cur] = false;
    HIP_TRY(hipMemcpyAsync(c->stage + c->stage_fill, bases, nbytes, hipMemcpyHostToDevice, c->copy_stream));
    // reads of two pushes must not join: a separator unless the caller's bytes end with one already
    const bool sep = is_base_byte(bases[nbytes - 1]);
    if (sep) HIP_TRY(hipMemsetAsync(c->stage + c->stage_fill + nbytes, '\n', 1, c->copy_stream));
    c->stage_fill += nbytes + (sep ? 1 : 0);
    if (async) { note_pending_checked(c, release, user); }
    else { HIP_TRY(hipStreamSynchronize(c->copy_stream)); release_pending(c, false); if (release) release(user); }
}

// packed bases -> the staging buffer, as they are: codes to the buffer's code array, flags to its flag array, sixteen
// positions per group (round 5: no landing area, no unpacking kernel -- the kernels that read bases take the groups)
static void push_packed_host(goss_gpu_ctx* c, const uint32_t* codes, const uint16_t* nonbase, uint64_t nbases, bool async,
                             void (*release)(void*), void* user)
{
    if (nbases < c->len) { if (release) release(user); return; }
    ensure_stage(c);
    // The pieces of one push are staged back to back without separators (all but the last are whole groups), so no
    // window is lost where one piece ends.  When the staging buffer must be counted in between, the last len - 1
    // positions are staged again in front of the rest -- their groups from the start, the positions whose windows
    // have been counted flagged as no bases.
    // (the staged string always ends on a group boundary behind a group of separators: the flag array of a buffer is
    // all ones when it starts to fill, and a push sets the flags behind its own last group back to that)
    uint64_t pos = 0, kill = 0;
    bool cont = false;
    if (c->deferred)
    {
        // (goss_gpu.h: a deferred push that does not fit is refused with NOTHING of it taken -- the caller runs the
        // group's exchange round and pushes the same batch again, so a piece staged before the refusal would be
        // counted twice.  The whole push must fit behind what is staged, or the state is not touched.)
        const uint64_t at0 = (c->stage_fill + 15) & ~15ULL;
        const uint64_t need = ((nbases + 15) / 16 + 1) * 16;
        if (at0 + need + 32 > c->stage_cap)
            throw StatusError{GOSS_ERR_BUFFER, c->stage_fill ? "the staging buffer is full and counting is deferred: goss_gpu_group_route_exchange first (goss_gpu_stage_room says how much fits)"
                                                             : "a push larger than the staging buffer while counting is deferred"};
        if (c->stage_fill && !c->stage_pk[c->stage_cur])
            throw StatusError{GOSS_ERR_BUFFER, "the staging buffer holds bytes and counting is deferred: goss_gpu_group_route_exchange first"};
    }
    // (bytes staged by the byte form: counted first -- a buffer holds one form)
    if (c->stage_fill && !c->stage_pk[c->stage_cur]) flush_staging_background(c);
    const uint16_t ones = 0xFFFFu;
    while (pos < nbases)
    {
        if (c->stage_fill == 0)
        {
            // a buffer that starts to fill: packed from here on, every flag set
            c->stage_pk[c->stage_cur] = true;
            HIP_TRY(hipMemsetAsync(pk_bad_of(c->stage, c->stage_cap), 0xFF, pk_groups(c->stage_cap) * 2, c->copy_stream));
        }
        // continuing a push: over the separator group of the piece before (whole groups but for the last piece)
        uint64_t at = cont ? c->stage_fill - 16 : c->stage_fill;
        uint64_t room = at + 64 < c->stage_cap ? (c->stage_cap - 32 - at) & ~15ULL : 0;
        if (room < std::min<uint64_t>((nbases - pos + 15) & ~15ULL, 65536))
        {
            c->stage_fill = at;
            flush_staging_background(c);
            if (cont)
            {
                const uint64_t keep = pos >= c->len - 1 ? pos - (c->len - 1) : 0;      // first position whose window is still to be counted
                const uint64_t from = keep & ~15ULL;
                kill = keep - from;
                pos = from;
            }
            cont = false;
            continue;          // (the other buffer, from its start)
        }
        const uint64_t n = std::min<uint64_t>(nbases - pos, room);
        const uint64_t groups = (n + 15) / 16;
        uint32_t* dcodes = pk_codes_of(c->stage) + at / 16;
        uint16_t* dbad = pk_bad_of(c->stage, c->stage_cap) + at / 16;
        const uint32_t* hc = codes + pos / 16;
        const uint16_t* hb = nonbase + pos / 16;
        HIP_TRY(hipMemcpyAsync(dcodes, hc, groups * 4, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipMemcpyAsync(dbad, hb, groups * 2, hipMemcpyHostToDevice, c->copy_stream));
        // positions behind the piece's last in its last group are no bases (the caller's flags need not say so), and the
        // positions of a continued push whose windows were counted with the buffer before are none any more: the two
        // flag words concerned are set once more, corrected -- by a 16-bit memset behind the copies on their stream: the
        // value travels in the command, no host memory has to outlive the call
        if ((n & 15ULL) && (uint16_t)(hb[groups - 1] | (ones << (n & 15ULL))) != hb[groups - 1])
        {
            const uint16_t w = (uint16_t)(hb[groups - 1] | (ones << (n & 15ULL)));
            HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)(dbad + (groups - 1)), w, 1, c->copy_stream));
        }
        if (kill)
        {
            const uint16_t w = (uint16_t)(hb[0] | ((1u << kill) - 1u) | ((groups == 1 && (n & 15ULL)) ? (ones << (n & 15ULL)) : 0u));
            HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)dbad, w, 1, c->copy_stream));
            kill = 0;
        }
        // (the group behind the piece is a group of separators as it stands: a buffer fills from its start, nothing has
        // been written at or behind that group since the flags were set to ones)
        c->stage_fill = at + (groups + 1) * 16;
        pos += n;
        cont = true;
    }
    if (async) { note_pending_checked(c, release, user); }
    else { HIP_TRY(hipStreamSynchronize(c->copy_stream)); release_pending(c, false); if (release) release(user); }
}

int goss_gpu_push_bases_host(goss_gpu_ctx* c, const char* bases, uint64_t nbytes)
{
    if (!c || (!bases && nbytes)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    return guarded(c, [&]() { push_bases_host(c, bases, nbytes, false, nullptr, nullptr); }, false);
}

int goss_gpu_push_bases_host_async(goss_gpu_ctx* c, const char* bases, uint64_t nbytes, goss_gpu_release_fn release, void* user)
{
    if (!c || (!bases && nbytes)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        // (a failed push hands the buffer back to the caller on return: what was queued from it must have run)
        try { push_bases_host(c, bases, nbytes, true, release, user); }
        catch (...) { if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); throw; }
    }, false);
}

int goss_gpu_push_packed_host(goss_gpu_ctx* c, const uint32_t* codes, const uint16_t* nonbase, uint64_t nbases)
{
    if (!c || (nbases && (!codes || !nonbase))) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    return guarded(c, [&]() { push_packed_host(c, codes, nonbase, nbases, false, nullptr, nullptr); }, false);
}

int goss_gpu_push_packed_host_async(goss_gpu_ctx* c, const uint32_t* codes, const uint16_t* nonbase, uint64_t nbases,
                                    goss_gpu_release_fn release, void* user)
{
    if (!c || (nbases && (!codes || !nonbase))) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        try { push_packed_host(c, codes, nonbase, nbases, true, release, user); }
        catch (...) { if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); throw; }
    }, false);
}

int goss_gpu_flush(goss_gpu_ctx* c)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    return guarded(c, [&]() { if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream)); release_pending(c, true); }, false);
}

int goss_gpu_finish(goss_gpu_ctx* c, goss_gpu_counts* out)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "finish called twice"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    int rc = guarded(c, [&]() {
        ensure_arena(c);
        flush_staging(c);
        release_pending(c, true);          // (every asynchronous push has been copied by now: the buffers go back)
        if (c->words == 1) merge_runs<Key1>(c); else merge_runs<Key2>(c);
        if (!c->runs.empty() && c->runs[0].rep)
        {
            if (c->mode == GOSS_MODE_GRAPH)
            {
                // strand pairs -> both strands: the reverse complements as a second run, merged with the first
                if (c->words == 1) { expand_graph_run<Key1>(c, 0); merge_runs<Key1>(c); }
                else { expand_graph_run<Key2>(c, 0); merge_runs<Key2>(c); }
            }
            else
            {
                if (c->words != 1) throw StatusError{GOSS_ERR_STATE, "a two-word run in representative space"};
                canonicalize_run<Key1>(c, c->runs[0]);
            }
        }
        if (!c->runs.empty()) { c->res_keys = c->runs[0].keys; c->res_counts = c->runs[0].counts; c->M = c->runs[0].m; }
        else { c->res_keys = c->arena.perm(16); c->res_counts = (uint32_t*)c->arena.perm(16); c->M = 0; }
        uint32_t* hf = (uint32_t*)c->h_pinned;
        HIP_TRY(hipMemcpyAsync(hf, c->d_flags, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        // u32 counts saturate at 2^32-1.  A k-mer set does not store counts.  A graph keeps the exact u64 counts
        // of such keys beside the run (resolve_big_counts); its u32 count becomes the value modulo 2^32, which is
        // what the reference stores (Graph::Builder::push_back hands a u64 to VariableByteArray's u32 value_type,
        // Graph.hh:101-106, VariableByteArray.hh:72,81), and its histogram is keyed by the exact count.
        c->res_big.clear();
        if (c->mode == GOSS_MODE_GRAPH)
        {
            if (hf[0]) throw StatusError{GOSS_ERR_COUNT_OVERFLOW, "a key occurred 2^32 times or more (unresolved)"};
            if (!c->runs.empty() && c->runs[0].big >= 0 && !c->big_maps[c->runs[0].big].empty())
            {
                c->res_big = c->big_maps[c->runs[0].big];
                uint64_t mark = c->arena.mark();
                unsigned long long* found = (unsigned long long*)c->arena.temp((kMaxBig + 1) * 8);
                HIP_TRY(hipMemsetAsync(found, 0, 8, c->stream));
                hipLaunchKernelGGL(find_saturated_kernel, dim3(grid_for(c->M, 256)), dim3(256), 0, c->stream, (const uint32_t*)c->res_counts, c->M, found, kMaxBig);
                std::vector<unsigned long long> hfound(kMaxBig + 1);
                HIP_TRY(hipMemcpyAsync(hfound.data(), found, (kMaxBig + 1) * 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                const uint32_t nq = (uint32_t)std::min<unsigned long long>(hfound[0], kMaxBig);
                std::vector<uint32_t> vals(nq);
                for (uint32_t q = 0; q < nq; ++q)
                {
                    uint64_t kw[2] = {0, 0};
                    HIP_TRY(hipMemcpy(kw, (const uint8_t*)c->res_keys + hfound[1 + q] * c->words * 8, c->words * 8, hipMemcpyDeviceToHost));
                    auto it = c->res_big.find(std::make_pair(c->words == 2 ? kw[1] : 0ULL, kw[0]));
                    if (it == c->res_big.end()) throw StatusError{GOSS_ERR_HIP, "count bookkeeping: a saturated result entry without its exact count"};
                    vals[q] = (uint32_t)(it->second & 0xFFFFFFFFULL);
                }
                if (nq)
                {
                    uint32_t* dv = (uint32_t*)c->arena.temp(nq * 4);
                    HIP_TRY(hipMemcpyAsync(dv, vals.data(), nq * 4, hipMemcpyHostToDevice, c->stream));
                    hipLaunchKernelGGL(patch_counts_kernel, dim3(grid_for(nq, 64)), dim3(64), 0, c->stream, c->res_counts,
                                       (const unsigned long long*)(found + 1), (const uint32_t*)dv, nq);
                    HIP_TRY(hipStreamSynchronize(c->stream));
                }
                c->arena.release(mark);
            }
        }
        c->finished = true;
    });
    if (rc == GOSS_OK && out)
    {
        out->windows = c->windows; out->keys = c->keys_total; out->distinct = c->M;
        out->key_words = (uint32_t)c->words; out->reserved = 0;
    }
    return rc;
}

int goss_gpu_result(goss_gpu_ctx* c, const void** d_keys, const uint32_t** d_counts, uint64_t* distinct)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (!c->finished) { c->last_error = "result before finish"; return GOSS_ERR_STATE; }
    if (d_keys) *d_keys = c->res_keys;
    if (d_counts) *d_counts = c->res_counts;
    if (distinct) *distinct = c->M;
    return GOSS_OK;
}

int goss_gpu_result_copy(goss_gpu_ctx* c, uint64_t first, uint64_t n, uint64_t* h_keys, uint32_t* h_counts)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (!c->finished) { c->last_error = "result before finish"; return GOSS_ERR_STATE; }
    if (first > c->M || n > c->M - first) return GOSS_ERR_INVALID_ARG;
    return guarded(c, [&]() {
        const uint64_t ksz = c->words * 8;
        if (h_keys && n) HIP_TRY(hipMemcpyAsync(h_keys, (const uint8_t*)c->res_keys + first * ksz, n * ksz, hipMemcpyDeviceToHost, c->stream));
        if (h_counts && n) HIP_TRY(hipMemcpyAsync(h_counts, c->res_counts + first, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
}

int goss_gpu_emit(goss_gpu_ctx* c)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (!c->finished || c->emitted) { c->last_error = "emit needs exactly one finish before it"; return GOSS_ERR_STATE; }
    int rc = guarded(c, [&]() {
        c->files.clear();
        {
            // file images: low bits (up to 16 B), bitmap, DenseSelect blocks, counts per element
            const uint64_t need = c->M * 40 + (256ULL << 20);
            if (c->arena.avail() < need) grow_arena(c, need);
        }
        if (c->words == 1) emit_object<Key1>(c); else emit_object<Key2>(c);
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
    if (rc == GOSS_OK) c->emitted = true;
    return rc;
}

int goss_gpu_big_counts(goss_gpu_ctx* c, uint64_t* keys, uint64_t* counts, uint32_t cap, uint32_t* n)
{
    if (!c || !n) return GOSS_ERR_INVALID_ARG;
    if (!c->finished) { c->last_error = "big_counts before finish"; return GOSS_ERR_STATE; }
    *n = (uint32_t)c->res_big.size();
    uint32_t i = 0;
    for (auto& kv : c->res_big)
    {
        if (i >= cap) break;
        if (keys) { keys[2 * i] = kv.first.second; keys[2 * i + 1] = kv.first.first; }
        if (counts) counts[i] = kv.second;
        ++i;
    }
    return GOSS_OK;
}

static int emit_part_entry(goss_gpu_ctx* c, uint64_t first_index, uint64_t total, uint64_t estimate, uint64_t prev_high, bool have_prev)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (!c->finished || c->emitted) { c->last_error = "emit_part needs exactly one finish before it"; return GOSS_ERR_STATE; }
    if (first_index + c->M > total) { c->last_error = "emit_part: the range does not fit the total"; return GOSS_ERR_INVALID_ARG; }
    int rc = guarded(c, [&]() {
        c->files.clear();
        {
            const uint64_t need = c->M * 48 + (256ULL << 20);
            if (c->arena.avail() < need) grow_arena(c, need);
        }
        const uint64_t est = estimate ? estimate : total;
        if (c->words == 1) emit_part<Key1>(c, first_index, total, est, prev_high, have_prev);
        else emit_part<Key2>(c, first_index, total, est, prev_high, have_prev);
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
    if (rc == GOSS_OK) c->emitted = true;
    return rc;
}
int goss_gpu_emit_part(goss_gpu_ctx* c, uint64_t first_index, uint64_t total, uint64_t estimate)
{
    return emit_part_entry(c, first_index, total, estimate, 0, false);
}
int goss_gpu_emit_part_ranges(goss_gpu_ctx* c, uint64_t first_index, uint64_t total, uint64_t estimate, uint64_t prev_last_high)
{
    return emit_part_entry(c, first_index, total, estimate, prev_last_high, true);
}
int goss_gpu_emit_last_high(goss_gpu_ctx* c, uint64_t total, uint64_t estimate, uint64_t* high, int* nonempty)
{
    if (!c || !high || !nonempty) return GOSS_ERR_INVALID_ARG;
    if (!c->finished) { c->last_error = "emit_last_high needs a finished range"; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        *high = 0; *nonempty = c->M ? 1 : 0;
        if (!c->M) return;
        uint64_t nlo, nhi; std::string base;
        object_universe(c, &nlo, &nhi, &base);
        const uint32_t D = (uint32_t)sparse_d(nlo, nhi, estimate ? estimate : total);
        if (c->words == 1)
        {
            Key1 k;
            HIP_TRY(hipMemcpyAsync(&k, (const Key1*)c->res_keys + (c->M - 1), sizeof k, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            *high = D >= 128 ? 0 : key_shr64(k, D);
        }
        else
        {
            Key2 k;
            HIP_TRY(hipMemcpyAsync(&k, (const Key2*)c->res_keys + (c->M - 1), sizeof k, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            *high = D >= 128 ? 0 : key_shr64(k, D);
        }
    });
}

int goss_gpu_emit_assemble(goss_gpu_ctx* c, const void* d_spans, uint64_t span_bytes, uint64_t total, uint64_t estimate,
                           const void* h_big, uint64_t nbig, const uint64_t* h_hist, uint64_t nhist)
{
    if (!c || (span_bytes & 7u) || (span_bytes && !d_spans) || (total && !span_bytes) || (nbig && !h_big) || (nhist && !h_hist))
        return GOSS_ERR_INVALID_ARG;
    if (!c->finished && (!c->runs.empty() || c->stage_fill))
    {
        c->last_error = "emit_assemble while a count is in progress";
        return GOSS_ERR_STATE;
    }
    return guarded(c, [&]() {
        ensure_arena(c);
        {
            const uint64_t need = total * 12 + (256ULL << 20);
            if (c->arena.avail() < need) grow_arena(c, need);
        }
        const auto t0 = std::chrono::steady_clock::now();
        emit_assemble(c, d_spans, span_bytes, total, estimate ? estimate : total, (const BigCount*)h_big, nbig, h_hist, nhist);
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->assemble_us = (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    });
}

// ---- several contexts in one process (one per GPU): the exchange and the emission of gossamer_amd/dist.py
// ---- without a process group -- device-to-device copies instead of collectives ---------------------------

extern "C++" {
namespace {

template <class K>
void group_exchange(goss_gpu_ctx* const* ctxs, uint32_t n, uint32_t sample)
{
    const uint64_t ksz = sizeof(K);
    // 1. splitters: quantiles of a pooled sample of every context's sorted distinct keys
    std::vector<K> pool;
    for (uint32_t i = 0; i < n; ++i)
    {
        goss_gpu_ctx* c = ctxs[i];
        const uint64_t m = c->M;
        if (m == 0) continue;
        HIP_TRY(hipSetDevice(c->device));
        const uint64_t take = std::min<uint64_t>(m, sample);
        const uint64_t stride = m / take;
        std::vector<K> h(take);
        HIP_TRY(hipMemcpy2DAsync(h.data(), ksz, (const uint8_t*)c->res_keys + (stride / 2) * ksz, stride * ksz, ksz, take,
                                 hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        pool.insert(pool.end(), h.begin(), h.end());
    }
    std::sort(pool.begin(), pool.end(), [](const K& a, const K& b) { return a < b; });
    std::vector<K> split(n > 1 ? n - 1 : 0);
    for (uint32_t p = 1; p < n; ++p)
        split[p - 1] = pool.empty() ? K{} : pool[std::min<uint64_t>(pool.size() - 1, (uint64_t)p * pool.size() / n)];
    // 2. where every context's run is cut
    std::vector<std::vector<uint64_t>> cut(n, std::vector<uint64_t>(n + 1, 0));
    for (uint32_t i = 0; i < n; ++i)
    {
        goss_gpu_ctx* c = ctxs[i];
        cut[i][n] = c->M;
        if (n == 1 || c->M == 0) { for (uint32_t p = 1; p < n; ++p) cut[i][p] = 0; if (c->M == 0) continue; }
        if (n == 1) continue;
        HIP_TRY(hipSetDevice(c->device));
        uint64_t mark = c->arena.mark();
        K* dq = (K*)c->arena.temp((n - 1) * ksz);
        uint64_t* dout = (uint64_t*)c->arena.temp((n - 1) * 8);
        HIP_TRY(hipMemcpyAsync(dq, split.data(), (n - 1) * ksz, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lower_bound_keys_kernel<K>), dim3((n - 1 + 63) / 64), dim3(64), 0, c->stream,
                           (const K*)c->res_keys, c->M, (const K*)dq, n - 1, dout);
        HIP_TRY(hipMemcpyAsync(&cut[i][1], dout, (n - 1) * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
    }
    // 3. range j of every run -> buffers on context j's device: at the top of context j's own arena where it has the
    //    room (the arenas of a build without --hbm-budget may have grown to nearly all of HBM: device memory outside
    //    them is then scarce), else memory of their own.  The top of an arena is temporary space: it is fenced off
    //    while the context is reset and takes the runs in.
    struct Inbox { uint8_t* keys = nullptr; uint32_t* counts = nullptr; std::vector<uint64_t> off; bool in_arena = false; uint64_t fence = 0; };
    std::vector<Inbox> inbox(n);
    auto free_all = [&]() {
        for (uint32_t j = 0; j < n; ++j)
        {
            if (inbox[j].in_arena) continue;
            (void)hipSetDevice(ctxs[j]->device);
            if (inbox[j].keys) (void)hipFree(inbox[j].keys);
            if (inbox[j].counts) (void)hipFree(inbox[j].counts);
        }
    };
    try
    {
        for (uint32_t j = 0; j < n; ++j)
        {
            goss_gpu_ctx* d = ctxs[j];
            Inbox& in = inbox[j];
            in.off.assign(n + 1, 0);
            for (uint32_t i = 0; i < n; ++i) in.off[i + 1] = in.off[i] + (cut[i][j + 1] - cut[i][j]);
            const uint64_t tot = in.off[n];
            if (tot == 0) continue;
            HIP_TRY(hipSetDevice(d->device));
            // (room for the inbox above the result, and for the runs it becomes plus their merge below it)
            if (d->arena.avail() >= 3 * tot * (ksz + 4) + (512ULL << 20))
            {
                in.counts = (uint32_t*)d->arena.temp(tot * 4);
                in.keys = (uint8_t*)d->arena.temp(tot * ksz);
                in.in_arena = true;
                in.fence = d->arena.hi;
            }
            else
            {
                HIP_TRY(hipMalloc((void**)&in.keys, tot * ksz));
                HIP_TRY(hipMalloc((void**)&in.counts, tot * 4));
            }
            for (uint32_t i = 0; i < n; ++i)
            {
                const uint64_t cnt = cut[i][j + 1] - cut[i][j];
                if (!cnt) continue;
                goss_gpu_ctx* s = ctxs[i];
                HIP_TRY(hipMemcpyPeerAsync(in.keys + in.off[i] * ksz, d->device, (const uint8_t*)s->res_keys + cut[i][j] * ksz, s->device,
                                           cnt * ksz, d->stream));
                HIP_TRY(hipMemcpyPeerAsync(in.counts + in.off[i], d->device, s->res_counts + cut[i][j], s->device, cnt * 4, d->stream));
            }
        }
        for (uint32_t j = 0; j < n; ++j)
        {
            HIP_TRY(hipSetDevice(ctxs[j]->device));
            HIP_TRY(hipStreamSynchronize(ctxs[j]->stream));
        }
        // 4. every context merges what it received (one host thread per device: the merges run side by side)
        std::vector<int> status(n, GOSS_OK);
        std::vector<std::thread> pool_t;
        for (uint32_t j = 0; j < n; ++j)
            pool_t.emplace_back([&, j]() {
                goss_gpu_ctx* d = ctxs[j];
                const uint64_t windows = d->windows, keys_total = d->keys_total;
                int rc = goss_gpu_reset(d);
                if (inbox[j].in_arena) d->arena.hi = inbox[j].fence;          // (the reset has given the whole arena back: the inbox lives on)
                for (uint32_t i = 0; i < n && rc == GOSS_OK; ++i)
                {
                    const uint64_t cnt = inbox[j].off[i + 1] - inbox[j].off[i];
                    if (cnt) rc = goss_gpu_push_run_device(d, inbox[j].keys + inbox[j].off[i] * ksz, inbox[j].counts + inbox[j].off[i], cnt);
                }
                if (inbox[j].in_arena) d->arena.hi = d->arena.size;           // (the runs are copies: the inbox is done)
                d->windows = windows; d->keys_total = keys_total;
                goss_gpu_counts cts;
                if (rc == GOSS_OK) rc = goss_gpu_finish(d, &cts);
                status[j] = rc;
            });
        for (auto& t : pool_t) t.join();
        for (uint32_t j = 0; j < n; ++j)
            if (status[j] != GOSS_OK) throw StatusError{status[j], "merging the received ranges: " + ctxs[j]->last_error};
    }
    catch (...)
    {
        free_all();
        throw;
    }
    free_all();
}

}  // namespace
}  // extern "C++"

int goss_gpu_group_exchange(goss_gpu_ctx* const* ctxs, uint32_t n, uint32_t sample_per_context, uint64_t* range_sizes)
{
    if (!ctxs || n == 0 || n > 64) return GOSS_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) if (!ctxs[i]) return GOSS_ERR_INVALID_ARG;
    goss_gpu_ctx* c0 = ctxs[0];
    for (uint32_t i = 0; i < n; ++i)
    {
        goss_gpu_ctx* c = ctxs[i];
        if (c->k != c0->k || c->mode != c0->mode) { c0->last_error = "group: contexts of different k or mode"; return GOSS_ERR_INVALID_ARG; }
        if (!c->finished || c->emitted) { c0->last_error = "group exchange needs every context finished and not emitted"; return GOSS_ERR_STATE; }
        for (uint32_t j = 0; j < i; ++j) if (ctxs[j] == c) { c0->last_error = "group: the same context twice"; return GOSS_ERR_INVALID_ARG; }
        if (!c->res_big.empty())
        {
            c0->last_error = "group exchange: a multiplicity of 2^32 - 1 or more (exact counts do not travel between contexts)";
            return GOSS_ERR_COUNT_OVERFLOW;
        }
    }
    int rc = guarded(c0, [&]() {
        const uint32_t sample = sample_per_context ? sample_per_context : 1024u;
        if (c0->words == 1) group_exchange<Key1>(ctxs, n, sample); else group_exchange<Key2>(ctxs, n, sample);
    });
    if (rc == GOSS_OK && range_sizes) for (uint32_t i = 0; i < n; ++i) range_sizes[i] = ctxs[i]->M;
    return rc;
}

int goss_gpu_group_emit(goss_gpu_ctx* const* ctxs, uint32_t n, uint64_t estimate)
{
    if (!ctxs || n == 0 || n > 64) return GOSS_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) if (!ctxs[i]) return GOSS_ERR_INVALID_ARG;
    goss_gpu_ctx* c0 = ctxs[0];
    for (uint32_t i = 0; i < n; ++i)
        if (ctxs[i]->k != c0->k || ctxs[i]->mode != c0->mode) { c0->last_error = "group: contexts of different k or mode"; return GOSS_ERR_INVALID_ARG; }
    uint64_t total = 0;
    std::vector<uint64_t> first(n);
    for (uint32_t i = 0; i < n; ++i) { first[i] = total; total += ctxs[i]->M; }
    // the high part of the last key of the ranges below every range: which zeros of the bitmap a range owns
    std::vector<uint64_t> prev(n, 0);
    {
        uint64_t run = 0;
        for (uint32_t i = 0; i < n; ++i)
        {
            prev[i] = run;
            uint64_t h = 0; int ne = 0;
            const int rc = goss_gpu_emit_last_high(ctxs[i], total, estimate, &h, &ne);
            if (rc != GOSS_OK) { if (i) c0->last_error = ctxs[i]->last_error; return rc; }
            if (ne) run = h;
        }
    }
    // every context's own span and blocks, side by side
    {
        // (a thread that cannot be started -- std::system_error out of std::thread's constructor -- must neither cross the
        // C boundary nor leave joinable threads behind: that member's part is then built on the caller's thread, the parts
        // being independent of one another)
        std::vector<int> status(n, GOSS_OK);
        std::vector<std::thread> pool;
        pool.reserve(n);
        for (uint32_t i = 0; i < n; ++i)
        {
            auto work = [&status, ctxs, &first, total, estimate, &prev, i]() {
                status[i] = goss_gpu_emit_part_ranges(ctxs[i], first[i], total, estimate, prev[i]);
            };
            try { pool.emplace_back(work); }
            catch (const std::system_error&) { work(); }
        }
        for (auto& t : pool) if (t.joinable()) t.join();
        for (uint32_t i = 0; i < n; ++i)
            if (status[i] != GOSS_OK) { if (i) c0->last_error = ctxs[i]->last_error; return status[i]; }
    }
    // the compact parts -> context 0
    uint8_t* d_high = nullptr;
    int rc = guarded(c0, [&]() {
        std::vector<uint8_t> big;
        std::vector<uint64_t> hist;
        std::vector<std::pair<const OutFile*, goss_gpu_ctx*>> highs;
        uint64_t high_total = 0;
        for (uint32_t i = 0; i < n; ++i)
        {
            goss_gpu_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->device));
            for (const OutFile& f : c->files)
            {
                if (f.suffix == ".part.span")
                {
                    highs.push_back({&f, c});
                    high_total += f.size;
                }
                else if (f.suffix == ".part.big" || f.suffix == ".part.hist")
                {
                    std::vector<uint8_t> h(f.size);
                    if (f.size)
                    {
                        if (f.dev)
                        {
                            HIP_TRY(hipMemcpyAsync(h.data(), f.dev, f.size, hipMemcpyDeviceToHost, c->stream));
                            HIP_TRY(hipStreamSynchronize(c->stream));
                        }
                        else std::memcpy(h.data(), f.host.data(), f.size);
                    }
                    if (f.suffix == ".part.big") big.insert(big.end(), h.begin(), h.end());
                    else { const size_t o = hist.size(); hist.resize(o + f.size / 8); if (f.size) std::memcpy(hist.data() + o, h.data(), f.size); }
                }
            }
        }
        if (highs.size() != n) throw StatusError{GOSS_ERR_STATE, "group emit: a context without its span of the bitmap"};
        HIP_TRY(hipSetDevice(c0->device));
        if (high_total)
        {
            HIP_TRY(hipMalloc((void**)&d_high, high_total));
            uint64_t at = 0;
            for (auto& hp : highs)
            {
                if (hp.first->size)
                    HIP_TRY(hipMemcpyPeerAsync(d_high + at, c0->device, hp.first->dev, hp.second->device, hp.first->size, c0->stream));
                at += hp.first->size;
            }
            HIP_TRY(hipStreamSynchronize(c0->stream));
        }
        const int arc = goss_gpu_emit_assemble(c0, d_high, high_total, total, estimate, big.empty() ? nullptr : big.data(),
                                               big.size() / 16, hist.empty() ? nullptr : hist.data(), hist.size() / 2);
        if (arc != GOSS_OK) throw StatusError{arc, c0->last_error};
    });
    if (d_high) { (void)hipSetDevice(c0->device); (void)hipFree(d_high); }
    return rc;
}

// ---- the exchange BEFORE counting for one process that owns several GPUs ---------------------------------
// (gossamer_amd/dist.py: route_and_exchange_records does the same between processes)

extern "C++" {
namespace {

// RCCL, loaded at run time (the library links against nothing but HIP): point-to-point sends and receives inside one
// group call are an all-to-all at full xGMI bandwidth.  Absent library, communicators that cannot be made (the same
// device twice) or GOSS_GROUP_TRANSPORT=peer -> hipMemcpyPeerAsync.
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool tried = false;
    std::map<std::vector<int>, std::vector<void*>> comms;          // device list -> one communicator per member
    std::mutex m;
};
Rccl g_rccl;

bool rccl_load()
{
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
    {
        void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        Rccl r;
        r.lib = h;
        r.CommInitAll = (int (*)(void**, int, const int*))dlsym(h, "ncclCommInitAll");
        r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        r.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
        r.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
        r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
        r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
        r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        if (r.CommInitAll && r.GroupStart && r.GroupEnd && r.Send && r.Recv)
        {
            g_rccl.lib = r.lib; g_rccl.CommInitAll = r.CommInitAll; g_rccl.CommDestroy = r.CommDestroy; g_rccl.GroupStart = r.GroupStart;
            g_rccl.GroupEnd = r.GroupEnd; g_rccl.Send = r.Send; g_rccl.Recv = r.Recv; g_rccl.GetErrorString = r.GetErrorString;
            return true;
        }
        dlclose(h);
    }
    return false;
}

// communicators of a device list (made once per list and process); nullptr = not to be had
const std::vector<void*>* rccl_comms(const std::vector<int>& devs)
{
    // (a device twice -- several contexts on one GPU, what a one-GPU box can show -- has no communicators: decided before
    // the library is loaded, which alone takes a second)
    std::vector<int> sorted = devs;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return nullptr;
    std::lock_guard<std::mutex> lk(g_rccl.m);
    if (!rccl_load()) return nullptr;
    auto it = g_rccl.comms.find(devs);
    if (it != g_rccl.comms.end()) return it->second.empty() ? nullptr : &it->second;
    std::vector<void*> cs;
    {
        cs.assign(devs.size(), nullptr);
        if (g_rccl.CommInitAll(cs.data(), (int)devs.size(), devs.data()) != 0) cs.clear();
    }
    auto& slot = g_rccl.comms[devs];
    slot = cs;
    return slot.empty() ? nullptr : &slot;
}

constexpr uint64_t kXferRound = 512ULL << 20;          // bytes per (source, destination) pair and round (RCCL 2.26 drops the second half of segments above 1 GiB)

void group_route_exchange(goss_gpu_ctx* const* ctxs, uint32_t n, int transport, goss_gpu_group_xstats* st)
{
    const uint64_t kRecBytes = rec_bytes(ctxs[0]);          // (all members have one k and mode)
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    // 0. the round before: its records are counted by a thread of every member while the caller stages this round's reads
    // (step 3); that counting must have ended before the inboxes are written again -- what is waited for here is the
    // part of it that the staging did not cover
    const auto tw = std::chrono::steady_clock::now();
    double prev_count_ms = 0;
    {
        int bad = GOSS_OK; std::string why;
        for (uint32_t i = 0; i < n; ++i)
        {
            try { HIP_TRY(hipSetDevice(ctxs[i]->device)); wait_background(ctxs[i]); }
            catch (const StatusError& e) { if (bad == GOSS_OK) { bad = e.status; why = "member " + std::to_string(i) + " counting the records of the round before: " + e.msg; } }
            prev_count_ms = std::max(prev_count_ms, ctxs[i]->grp_count_ms);
            ctxs[i]->grp_count_ms = 0;
        }
        if (bad != GOSS_OK) { for (uint32_t i = 0; i < n; ++i) ctxs[i]->broken = true; throw StatusError{bad, why}; }
    }
    const double wait_ms = ms_since(tw);
    const auto t0 = std::chrono::steady_clock::now();
    // 1. every member routes what it has staged into n parts (side by side: one host thread per device)
    std::vector<std::vector<uint64_t>> recs(n, std::vector<uint64_t>(n, 0)), wins(n, std::vector<uint64_t>(n, 0)), first(n, std::vector<uint64_t>(n, 0));
    std::vector<int> status(n, GOSS_OK);
    {
        // (a thread that cannot be started must not take the process down with the ones that were: join what runs,
        // then report; and once one member has failed, the others' staged reads are in records nobody will count --
        // the whole group is marked and refuses further work)
        std::vector<std::thread> pool;
        auto join_all = [&]() { for (auto& t : pool) if (t.joinable()) t.join(); };
        auto mark_broken = [&]() { for (uint32_t i = 0; i < n; ++i) ctxs[i]->broken = true; };
        auto start = [&](std::function<void()> f) {
            try { pool.emplace_back(std::move(f)); }
            catch (const std::system_error& e) { join_all(); mark_broken(); throw StatusError{GOSS_ERR_OOM, std::string("no thread for a member of the group: ") + e.what()}; }
        };
        for (uint32_t i = 0; i < n; ++i)
            start([&, i]() {
                goss_gpu_ctx* c = ctxs[i];
                status[i] = guarded(c, [&]() {
                    if (!c->stage || c->stage_fill == 0) return;
                    HIP_TRY(hipStreamSynchronize(c->copy_stream));          // (the copies that filled the buffer)
                    const uint64_t nb = c->stage_fill;
                    // room per part: ~7 windows per record, a third of slack; a part that needs more is reported and the
                    // routing is redone with what every part asked for
                    std::vector<uint64_t> cap(n, nb / 5 / n + nb / 15 / n + 4096);
                    for (int attempt = 0;; ++attempt)
                    {
                        uint64_t tot = 0;
                        for (uint32_t p = 0; p < n; ++p) { first[i][p] = tot; tot += cap[p]; }
                        if (tot > c->grp_send_cap)
                        {
                            if (c->grp_send) { HIP_TRY(hipFree(c->grp_send)); c->grp_send = nullptr; c->grp_send_cap = 0; }
                            if (hipMalloc((void**)&c->grp_send, tot * kRecBytes) != hipSuccess)
                            { (void)hipGetLastError(); throw StatusError{GOSS_ERR_OOM, "no device memory for the routed records"}; }
                            c->grp_send_cap = tot;
                        }
                        // (a packed staging buffer is routed as it is: ctx.pk for the call's duration)
                        const bool pk = c->stage_pk[c->stage_cur];
                        if (pk) { c->pk.on = true; c->pk.codes = pk_codes_of(c->stage); c->pk.bad = pk_bad_of(c->stage, c->stage_cap); }
                        const int rc = goss_gpu_route_records_device(c, pk ? (const void*)kPkFake : (const void*)c->stage, nb, n, c->grp_send, first[i].data(), cap.data(),
                                                                     recs[i].data(), wins[i].data());
                        c->pk.on = false;
                        if (rc == GOSS_OK) break;
                        if (rc != GOSS_ERR_BUFFER || attempt) throw StatusError{rc, "routing the staged reads: " + c->last_error};
                        for (uint32_t p = 0; p < n; ++p) cap[p] = recs[i][p] + 1;
                    }
                    c->stage_fill = 0;          // (the windows of these bases now live in the records)
                });
            });
        join_all();
        for (uint32_t i = 0; i < n; ++i)
            if (status[i] != GOSS_OK) { mark_broken(); throw StatusError{status[i], "member " + std::to_string(i) + ": " + ctxs[i]->last_error}; }
    }
    const double route_ms = ms_since(t0);
    // 2. part p of every member -> member p's inbox
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<std::vector<uint64_t>> in_off(n, std::vector<uint64_t>(n + 1, 0));          // in_off[p][i]: where member i's records start in p's inbox
    uint64_t total_recs = 0, total_wins = 0;
    for (uint32_t p = 0; p < n; ++p)
    {
        for (uint32_t i = 0; i < n; ++i) { in_off[p][i + 1] = in_off[p][i] + recs[i][p]; total_wins += wins[i][p]; }
        total_recs += in_off[p][n];
        goss_gpu_ctx* d = ctxs[p];
        HIP_TRY(hipSetDevice(d->device));
        if (!d->xstream) HIP_TRY(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
        if (in_off[p][n] > d->grp_inbox_cap)
        {
            if (d->grp_inbox) { HIP_TRY(hipFree(d->grp_inbox)); d->grp_inbox = nullptr; d->grp_inbox_cap = 0; }
            const uint64_t want = in_off[p][n] + in_off[p][n] / 8 + 4096;
            if (hipMalloc((void**)&d->grp_inbox, want * kRecBytes) != hipSuccess)
            { (void)hipGetLastError(); throw StatusError{GOSS_ERR_OOM, "no device memory for the received records"}; }
            d->grp_inbox_cap = want;
        }
    }
    std::vector<int> devs(n);
    for (uint32_t i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
    const std::vector<void*>* comms = nullptr;
    if (transport != 2) comms = rccl_comms(devs);
    if (transport == 1 && !comms) throw StatusError{GOSS_ERR_STATE, "RCCL transport asked for, but librccl or its communicators for these devices are not to be had"};
    uint32_t rounds = 0;
    if (comms)
    {
        uint64_t largest = 0;
        for (uint32_t i = 0; i < n; ++i) for (uint32_t p = 0; p < n; ++p) largest = std::max(largest, recs[i][p] * kRecBytes);
        for (uint64_t off = 0; off < std::max<uint64_t>(largest, 1); off += kXferRound, ++rounds)
        {
            if (largest == 0) break;
            auto nccl_check = [&](int rc, const char* what) {
                if (rc != 0) throw StatusError{GOSS_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error")};
            };
            nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
            for (uint32_t i = 0; i < n; ++i)
                for (uint32_t p = 0; p < n; ++p)
                {
                    const uint64_t bytes = recs[i][p] * kRecBytes;
                    if (off >= bytes) continue;
                    const uint64_t len = std::min<uint64_t>(kXferRound, bytes - off);
                    nccl_check(g_rccl.Send(ctxs[i]->grp_send + first[i][p] * kRecBytes + off, len, 0 /* ncclInt8 */, (int)p, (*comms)[i], ctxs[i]->xstream), "ncclSend");
                    nccl_check(g_rccl.Recv(ctxs[p]->grp_inbox + in_off[p][i] * kRecBytes + off, len, 0, (int)i, (*comms)[p], ctxs[p]->xstream), "ncclRecv");
                }
            nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
        }
    }
    else
    {
        for (uint32_t p = 0; p < n; ++p)
        {
            goss_gpu_ctx* d = ctxs[p];
            HIP_TRY(hipSetDevice(d->device));
            for (uint32_t i = 0; i < n; ++i)
                if (recs[i][p])
                {
                    // (one device listed several times -- how the tests run a group on one GPU: a plain device-to-device copy,
                    // the peer call takes a slow path between a device and itself)
                    if (ctxs[i]->device == d->device)
                        HIP_TRY(hipMemcpyAsync(d->grp_inbox + in_off[p][i] * kRecBytes, ctxs[i]->grp_send + first[i][p] * kRecBytes, recs[i][p] * kRecBytes,
                                               hipMemcpyDeviceToDevice, d->xstream));
                    else
                        HIP_TRY(hipMemcpyPeerAsync(d->grp_inbox + in_off[p][i] * kRecBytes, d->device, ctxs[i]->grp_send + first[i][p] * kRecBytes, ctxs[i]->device,
                                                   recs[i][p] * kRecBytes, d->xstream));
                }
        }
        rounds = 1;
    }
    for (uint32_t p = 0; p < n; ++p)
    {
        HIP_TRY(hipSetDevice(ctxs[p]->device));
        HIP_TRY(hipStreamSynchronize(ctxs[p]->xstream));
    }
    const double wire_ms = ms_since(t1);
    // 3. every member counts what it received -- on a thread of its own that this call does NOT wait for (round 5): the
    // members' staging buffers are free again (the windows of their bases live in the records), so the caller stages the
    // next round's reads while the devices count this round's.  The thread is the context's background thread: every
    // entry point that works on the arena or the runs (finish, emit, a device push, the next exchange round) waits for
    // it and takes over its failure; host pushes, which only stage, go on beside it.
    for (uint32_t p = 0; p < n; ++p)
    {
        goss_gpu_ctx* c = ctxs[p];
        const uint64_t nrec = in_off[p][n];
        if (!nrec) continue;
        uint64_t w = 0;
        for (uint32_t i = 0; i < n; ++i) w += wins[i][p];
        auto work = [c, nrec, w]() {
            const auto tc = std::chrono::steady_clock::now();
            try
            {
                HIP_TRY(hipSetDevice(c->device));
                (void)hipGetLastError();
                if (c->words == 1) push_records<Key1>(c, c->grp_inbox, nrec, w); else push_records<Key2>(c, c->grp_inbox, nrec, w);
                check_launch("a kernel launch was refused");
            }
            catch (const HipError& e) { c->bg_status = GOSS_ERR_HIP; c->bg_error = std::string(e.what) + ": " + hipGetErrorString(e.e); }
            catch (const StatusError& e) { c->bg_status = e.status; c->bg_error = e.msg; }
            catch (const std::bad_alloc&) { c->bg_status = GOSS_ERR_OOM; c->bg_error = "host allocation failed"; }
            c->grp_count_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc).count();
        };
        c->bg_status = GOSS_OK;
        try { c->bg = std::thread(work); }
        catch (const std::system_error&) { work(); }          // (no thread to be had: this member's share on the caller's)
    }
    if (st)
    {
        st->transport = comms ? 1u : 2u;
        st->rounds = rounds;
        st->records = total_recs; st->windows = total_wins; st->record_bytes = total_recs * kRecBytes;
        st->route_ms = route_ms; st->wire_ms = wire_ms;
        st->count_ms = prev_count_ms;          // (of the round BEFORE this one: this round's counting has only been started)
        st->count_wait_ms = wait_ms;
    }
}

}  // namespace
}  // extern "C++"

int goss_gpu_set_deferred(goss_gpu_ctx* c, int on)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "deferred counting is set before the pushes"; return GOSS_ERR_STATE; }
    c->deferred = on != 0;
    return GOSS_OK;
}

int goss_gpu_stage_room(goss_gpu_ctx* c, uint64_t* free_bytes, uint64_t* capacity)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    return guarded(c, [&]() {
        ensure_stage(c);
        // (a push of n bytes takes n + 1: the separator behind it)
        if (free_bytes) *free_bytes = c->stage_cap > c->stage_fill + 64 ? c->stage_cap - c->stage_fill - 64 : 0;
        if (capacity) *capacity = c->stage_cap;
    }, false);
}

int goss_gpu_group_route_exchange(goss_gpu_ctx* const* ctxs, uint32_t n, int transport, goss_gpu_group_xstats* out)
{
    if (!ctxs || n == 0 || n > (uint32_t)kRouteMaxParts || transport < 0 || transport > 2) return GOSS_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) if (!ctxs[i]) return GOSS_ERR_INVALID_ARG;
    goss_gpu_ctx* c0 = ctxs[0];
    for (uint32_t i = 0; i < n; ++i)
    {
        goss_gpu_ctx* c = ctxs[i];
        if (c->k != c0->k || c->mode != c0->mode) { c0->last_error = "group: contexts of different k or mode"; return GOSS_ERR_INVALID_ARG; }
        if (c->finished) { c0->last_error = "group route exchange after finish"; return GOSS_ERR_STATE; }
        if (c->broken) { c0->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
        if (c->len > 63) { c0->last_error = "records carry windows of at most 63 bases"; return GOSS_ERR_INVALID_ARG; }
        for (uint32_t j = 0; j < i; ++j) if (ctxs[j] == c) { c0->last_error = "group: the same context twice"; return GOSS_ERR_INVALID_ARG; }
    }
    if (out) std::memset(out, 0, sizeof(*out));
    if (const char* e = std::getenv("GOSS_GROUP_TRANSPORT"))
    {
        if (!std::strcmp(e, "peer")) transport = 2;
        else if (!std::strcmp(e, "rccl")) transport = 1;
    }
    return guarded(c0, [&]() { group_route_exchange(ctxs, n, transport, out); });
}

int goss_gpu_file_device(goss_gpu_ctx* c, uint32_t i, const void** d_ptr)
{
    if (!c || !d_ptr || i >= c->files.size()) return GOSS_ERR_INVALID_ARG;
    *d_ptr = c->files[i].dev;          // NULL for a file built on the host
    return GOSS_OK;
}

int goss_gpu_emit_sparse_array(goss_gpu_ctx* c, const void* d_positions, uint32_t key_words, uint64_t n,
                               uint64_t N_lo, uint64_t N_hi, uint64_t M, uint64_t Nend_lo, uint64_t Nend_hi)
{
    if (!c || (key_words != 1 && key_words != 2) || (!d_positions && n)) return GOSS_ERR_INVALID_ARG;
    // its key copy and file images are permanent allocations: between pushes a later merge or an
    // out-of-memory rollback would rewind the permanent end underneath them
    if (!c->finished && (!c->runs.empty() || c->stage_fill))
    {
        c->last_error = "emit_sparse_array while a count is in progress (reset or finish first)";
        return GOSS_ERR_STATE;
    }
    return guarded(c, [&]() {
        ensure_arena(c);
        c->files.clear();
        PhaseTimer t(c, GOSS_T_EMIT, n);
        // private copy so the caller's buffer need not outlive the call
        const uint64_t ksz = key_words * 8;
        void* keys = c->arena.perm(std::max<uint64_t>(n * ksz, 16));
        if (n) HIP_TRY(hipMemcpyAsync(keys, d_positions, n * ksz, hipMemcpyDeviceToDevice, c->stream));
        if (key_words == 1) emit_sparse_array<Key1>(c, (const Key1*)keys, n, N_lo, N_hi, M, Nend_lo, Nend_hi, "");
        else emit_sparse_array<Key2>(c, (const Key2*)keys, n, N_lo, N_hi, M, Nend_lo, Nend_hi, "");
        t.stop();
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
}

int goss_gpu_file_count(goss_gpu_ctx* c, uint32_t* n)
{
    if (!c || !n) return GOSS_ERR_INVALID_ARG;
    *n = (uint32_t)c->files.size();
    return GOSS_OK;
}

int goss_gpu_file_info(goss_gpu_ctx* c, uint32_t i, char* suffix, size_t cap, uint64_t* size)
{
    if (!c || i >= c->files.size()) return GOSS_ERR_INVALID_ARG;
    if (suffix && cap) { std::snprintf(suffix, cap, "%s", c->files[i].suffix.c_str()); }
    if (size) *size = c->files[i].size;
    return GOSS_OK;
}

int goss_gpu_file_read(goss_gpu_ctx* c, uint32_t i, uint64_t offset, void* dst, uint64_t n)
{
    if (!c || i >= c->files.size() || (!dst && n)) return GOSS_ERR_INVALID_ARG;
    const OutFile& f = c->files[i];
    if (offset > f.size || n > f.size - offset) return GOSS_ERR_INVALID_ARG;
    if (n == 0) return GOSS_OK;
    if (!f.dev) { std::memcpy(dst, f.host.data() + offset, n); return GOSS_OK; }
    return guarded(c, [&]() {
        HIP_TRY(hipMemcpyAsync(dst, f.dev + offset, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
}

int goss_gpu_timing_get(goss_gpu_ctx* c, goss_gpu_timing* out)
{
    if (!c || !out) return GOSS_ERR_INVALID_ARG;
    int rc = guarded(c, [&]() { resolve_timing(c); });
    *out = c->timing;
    return rc;
}

int goss_gpu_timing_reset(goss_gpu_ctx* c)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    int rc = guarded(c, [&]() { resolve_timing(c); });
    c->timing = goss_gpu_timing{};
    return rc;
}

int goss_gpu_host_alloc(void** p, size_t bytes)
{
    if (!p || bytes == 0) return GOSS_ERR_INVALID_ARG;
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    return e == hipSuccess ? GOSS_OK : GOSS_ERR_OOM;
}

void goss_gpu_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int goss_gpu_host_register(void* p, size_t bytes)
{
    if (!p || bytes == 0) return GOSS_ERR_INVALID_ARG;
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? GOSS_OK : GOSS_ERR_OOM;
}

void goss_gpu_host_unregister(void* p)
{
    if (p) { if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError(); }
}

int goss_gpu_set_path(goss_gpu_ctx* c, int path)
{
    if (!c || path < 0 || path > 1) return GOSS_ERR_INVALID_ARG;
    c->path = path;
    return GOSS_OK;
}

int goss_gpu_reset(goss_gpu_ctx* c)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    return guarded(c, [&]() {
        if (c->arena_thread.joinable()) ensure_arena(c);      // a background mapping (goss_gpu_prepare) ends first
        HIP_TRY(hipStreamSynchronize(c->stream));
        release_pending(c, true);
        c->runs.clear();
        c->big_maps.clear();
        c->res_big.clear();
        c->files.clear();
        c->windows = c->keys_total = 0;
        c->space_choice = -1;
        c->expect_bases = 0;
        c->finished = c->emitted = false;
        c->broken = false;
        c->res_keys = nullptr; c->res_counts = nullptr; c->M = 0;
        c->arena.lo = 0; c->arena.hi = c->arena.size;
        if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
        c->stage_cur = 0; c->stage = c->stage_buf[0]; c->stage_fill = 0;          // (the staging buffers stay)
        c->dump_live = false;
        HIP_TRY(hipMemsetAsync(c->d_flags, 0, 16, c->stream));
    });
}

int goss_gpu_push_run_device(goss_gpu_ctx* c, const void* d_keys, const uint32_t* d_counts, uint64_t m)
{
    if (!c || (m && (!d_keys || !d_counts))) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    if (m == 0) return GOSS_OK;
    return guarded(c, [&]() {
        ensure_arena(c);
        flush_staging(c);
        const uint64_t ksz = c->words * 8;
        if (c->arena.avail() < m * (ksz + 4) + (64u << 20)) grow_arena(c, m * (ksz + 4) + (64u << 20));
        Run r{nullptr, nullptr, m};
        r.keys = c->arena.perm(m * ksz);
        r.counts = (uint32_t*)c->arena.perm(m * 4);
        HIP_TRY(hipMemcpyAsync(r.keys, d_keys, m * ksz, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(r.counts, d_counts, m * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->runs.push_back(r);
        // keys_total keeps meaning "keys inserted": a run stands for the sum of its counts,
        // which the caller accounts for; windows are not known here.
    });
}

int goss_gpu_push_run_host(goss_gpu_ctx* c, const uint64_t* keys, const uint32_t* counts, uint64_t m)
{
    if (!c || (m && (!keys || !counts))) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    if (m == 0) return GOSS_OK;
    return guarded(c, [&]() {
        ensure_arena(c);
        flush_staging(c);
        const uint64_t ksz = c->words * 8;
        if (c->arena.avail() < m * (ksz + 4) + (64u << 20)) grow_arena(c, m * (ksz + 4) + (64u << 20));
        Run r{nullptr, nullptr, m};
        r.keys = c->arena.perm(m * ksz);
        r.counts = (uint32_t*)c->arena.perm(m * 4);
        HIP_TRY(hipMemcpyAsync(r.keys, keys, m * ksz, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(r.counts, counts, m * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->runs.push_back(r);
    });
}

extern "C++" {
// Raw keys (unsorted, un-normalised) -> normalise (mode 0) -> count -> one run per piece.
template <class K>
static void push_keys(goss_gpu_ctx* c, const void* keys, uint64_t n, bool on_host)
{
    ensure_arena(c);
    flush_staging(c);
    const uint64_t ksz = sizeof(K);
    uint64_t done = 0;
    while (done < n)
    {
        // two key buffers, the sort's tables and the worst-case run of the piece (chunk_capacity's pessimistic sizing)
        auto capacity = [&]() { return (uint64_t)((double)c->arena.avail() * 0.95 / (2.0 * ksz + 1.2 + (ksz + 12.0))) & ~4095ULL; };
        uint64_t cap = capacity();
        if (cap < 4096 && c->runs.size() > 1) { merge_runs<K>(c); cap = capacity(); }
        if (cap < 4096 && grow_arena(c, 0)) cap = capacity();
        if (cap < 4096) throw StatusError{GOSS_ERR_OOM, "HBM budget too small for one piece of keys"};
        const uint64_t m = std::min(cap, n - done);
        const uint64_t mark = c->arena.mark();
        K* ka = (K*)c->arena.temp(m * ksz);
        K* kb = (K*)c->arena.temp(m * ksz);
        const K* src = (const K*)keys + done;
        if (on_host)
        {
            HIP_TRY(hipMemcpyAsync(kb, src, m * ksz, hipMemcpyHostToDevice, c->stream));
            src = kb;
        }
        HIP_TRY(hipMemsetAsync(c->d_flags + 2, 0, 4, c->stream));
        {
            PhaseTimer t(c, GOSS_T_EXTRACT, m);
            if (c->mode == GOSS_MODE_GRAPH)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(normalize_keys_kernel<K, 1>), dim3(grid_for(m, kTB)), dim3(kTB), 0, c->stream, src, ka, m, c->len, c->d_flags);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(normalize_keys_kernel<K, 0>), dim3(grid_for(m, kTB)), dim3(kTB), 0, c->stream, src, ka, m, c->len, c->d_flags);
            t.stop();
        }
        uint32_t* hf = (uint32_t*)c->h_pinned;
        HIP_TRY(hipMemcpyAsync(hf, c->d_flags + 2, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (hf[0]) { c->arena.release(mark); throw StatusError{GOSS_ERR_INVALID_ARG, "a pushed key has bits beyond 2*len"}; }
        c->extract_hist_shift = 0xFFFFFFFFu;          // no extraction kernel histogrammed THESE keys
        Run r = count_keys<K>(c, ka, kb, m);
        c->runs.push_back(r);
        c->arena.release(mark);
        // a k-mer is one window and one key; a graph key is one of the two a rho-mer window contributes
        c->keys_total += m;
        c->windows += c->mode == GOSS_MODE_GRAPH ? m / 2 : m;
        done += m;
        uint64_t run_bytes = 0;
        for (auto& q : c->runs) run_bytes += q.m * (ksz + 4);
        if (c->runs.size() > 1 && run_bytes > c->arena.size / 4 &&
            (c->arena.avail() >= 2 * run_bytes + (512ULL << 20) || grow_arena(c, 2 * run_bytes + (512ULL << 20)))) merge_runs<K>(c);
    }
}
}

static int push_keys_entry(goss_gpu_ctx* c, const void* keys, uint64_t n, bool on_host)
{
    if (!c || (n && !keys)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    if (n == 0) return GOSS_OK;
    return guarded(c, [&]() {
        if (c->words == 1) push_keys<Key1>(c, keys, n, on_host); else push_keys<Key2>(c, keys, n, on_host);
    });
}
int goss_gpu_route_records_device(goss_gpu_ctx* c, const void* d_bases, uint64_t nbytes, uint32_t nparts, void* d_records,
                                  const uint64_t* part_first, const uint64_t* part_cap, uint64_t* part_records, uint64_t* part_windows)
{
    if (!c || (!d_bases && nbytes) || nparts == 0 || nparts > (uint32_t)kRouteMaxParts || !part_first || !part_cap || !part_records)
        return GOSS_ERR_INVALID_ARG;
    if (c->len > 63) { c->last_error = "records carry windows of at most 63 bases"; return GOSS_ERR_INVALID_ARG; }
    return guarded(c, [&]() {
        for (uint32_t p = 0; p < nparts; ++p) { part_records[p] = 0; if (part_windows) part_windows[p] = 0; }
        if (nbytes < c->len) return;
        // counters and the parts' places: a small device block of its own (the arena may not be mapped yet, and need not
        // be), kept for the context's life -- hipFree waits for the whole device, also for a collective of the caller
        // that is still moving the records of the piece before
        const size_t tab = (size_t)kRouteMaxParts * 8;
        if (!c->d_route) HIP_TRY(hipMalloc(&c->d_route, sizeof(RouteCounters) + 2 * tab));
        RouteCounters* rc = (RouteCounters*)c->d_route;
        unsigned long long* dfirst = (unsigned long long*)((uint8_t*)c->d_route + sizeof(RouteCounters));
        unsigned long long* dcap = dfirst + kRouteMaxParts;
        HIP_TRY(hipMemsetAsync(rc, 0, sizeof(RouteCounters), c->stream));
        HIP_TRY(hipMemcpyAsync(dfirst, part_first, nparts * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dcap, part_cap, nparts * 8, hipMemcpyHostToDevice, c->stream));
        const uint64_t nstarts = nbytes - c->len + 1;
        const uint64_t ntiles = (nstarts + kTB * 16 - 1) / (kTB * 16);
        const uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, 256 * GOSS_ROUTE_OCC);
        const uintptr_t addr = (uintptr_t)d_bases;
        const uint32_t mis = (uint32_t)(addr & 15u);
        const uint8_t* aligned = (const uint8_t*)(addr - mis);
        const uint32_t maxwin = 16;          // (the counting side takes records of any length in both modes)
        // slots a workgroup takes ahead per part: ~8 tiles' worth (a tile cuts ~550 records); what it does not use ends as pads
        const uint32_t block = std::max<uint32_t>(32, std::min<uint32_t>(GOSS_ROUTE_BLOCK, 8 * GOSS_ROUTE_BLOCK / nparts));
        {
            PhaseTimer t(c, GOSS_T_EXTRACT, nstarts);
            // (a packed staging buffer -- the group's exchange round sets ctx.pk -- is routed as it is)
            const uint8_t* src = aligned; const uint16_t* pbad = nullptr;
            if (c->pk.on) pk_ptrs(c, aligned, &src, &pbad);
#define GOSS_LAUNCH_ROUTE(W)                                                                                                \
    do {                                                                                                                    \
        if (c->pk.on)                                                                                                       \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(route_records_kernel<W, true>), dim3(grid), dim3(kTB), 0, c->stream, src, mis, nstarts, nbytes, \
                               c->len, maxwin, nparts, block, (SkRec*)d_records, (const unsigned long long*)dfirst, (const unsigned long long*)dcap, rc, ntiles, pbad); \
        else                                                                                                                \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(route_records_kernel<W, false>), dim3(grid), dim3(kTB), 0, c->stream, src, mis, nstarts, nbytes, \
                               c->len, maxwin, nparts, block, (SkRec*)d_records, (const unsigned long long*)dfirst, (const unsigned long long*)dcap, rc, ntiles, pbad); \
    } while (0)
            if (c->words == 2)
            {
                // two-word keys: 20-byte records, the minimizer taken over the central 31 / 30 bases of a window (17 positions)
                const dim3 g2((uint32_t)std::min<uint64_t>(ntiles, 256 * 4));
                if (c->pk.on)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(route_records2_kernel<17, true>), g2, dim3(kTB), 0, c->stream, src, mis,
                                       nstarts, nbytes, c->len, nparts, block, (SkRec2*)d_records, (const unsigned long long*)dfirst,
                                       (const unsigned long long*)dcap, rc, ntiles, pbad);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(route_records2_kernel<17, false>), g2, dim3(kTB), 0, c->stream, src, mis,
                                       nstarts, nbytes, c->len, nparts, block, (SkRec2*)d_records, (const unsigned long long*)dfirst,
                                       (const unsigned long long*)dcap, rc, ntiles, pbad);
            }
            else
            switch (route_positions(c->len))
            {
                case 17: GOSS_LAUNCH_ROUTE(17); break;
                case 13: GOSS_LAUNCH_ROUTE(13); break;
                case 9: GOSS_LAUNCH_ROUTE(9); break;
                case 5: GOSS_LAUNCH_ROUTE(5); break;
                default: GOSS_LAUNCH_ROUTE(1); break;
            }
#undef GOSS_LAUNCH_ROUTE
            t.stop();
        }
        std::vector<unsigned long long> h(sizeof(RouteCounters) / 8);
        HIP_TRY(hipMemcpyAsync(h.data(), rc, sizeof(RouteCounters), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const RouteCounters* hr = (const RouteCounters*)h.data();
        for (uint32_t p = 0; p < nparts; ++p) { part_records[p] = hr->records[p]; if (part_windows) part_windows[p] = hr->windows[p]; }
#if defined(GOSS_STAMPS)
        if (nparts <= 128 && hr->records[128 + 7])
        {
            const double n = (double)hr->records[128 + 7];
            std::fprintf(stderr, "libgossgpu: routing stamps per tile (ticks of wave 0): A %.0f  B %.0f  barrier %.0f  runs+scan %.0f  C %.0f  room %.0f  D %.0f   (%.0f tiles)\n",
                         hr->records[128] / n, hr->records[129] / n, hr->records[130] / n, hr->records[131] / n, hr->records[132] / n, hr->records[133] / n, hr->records[134] / n, n);
        }
#endif
        if (hr->overflow) throw StatusError{GOSS_ERR_BUFFER, "a part's record buffer is too small (part_records holds what every part needs)"};
    });
}

int goss_gpu_push_records_device(goss_gpu_ctx* c, const void* d_records, uint64_t nrecords, uint64_t nwindows)
{
    if (!c || (nrecords && !d_records)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    if (c->len > 63) { c->last_error = "records carry windows of at most 63 bases"; return GOSS_ERR_INVALID_ARG; }
    return guarded(c, [&]() {
        flush_staging(c);
        if (c->words == 1) push_records<Key1>(c, (const uint8_t*)d_records, nrecords, nwindows);
        else push_records<Key2>(c, (const uint8_t*)d_records, nrecords, nwindows);
    });
}

int goss_gpu_push_keys_device(goss_gpu_ctx* c, const void* d_keys, uint64_t n) { return push_keys_entry(c, d_keys, n, false); }
int goss_gpu_push_keys_host(goss_gpu_ctx* c, const uint64_t* keys, uint64_t n) { return push_keys_entry(c, keys, n, true); }

extern "C++" {
// SparseArray in its on-disk form (host pointers) -> m decoded positions on the device
// (SparseArray::LazyIterator, SparseArray.hh:185-224).  Temporaries above `out` in the arena are the caller's to release.
template <class K>
static void decode_sparse(goss_gpu_ctx* c, const goss_gpu_sparse_run* s, K* out)
{
    const uint64_t m = s->count;
    uint64_t* words = (uint64_t*)c->arena.temp(std::max<uint64_t>(s->high_words, 1) * 8);
    uint64_t* prefix = (uint64_t*)c->arena.temp(std::max<uint64_t>(s->high_words, 1) * 8);
    HIP_TRY(hipMemcpyAsync(words, s->high_bits, s->high_words * 8, hipMemcpyHostToDevice, c->stream));
    EfColumnsIn cols{};
    cols.n = s->ncols;
    for (uint32_t i = 0; i < s->ncols; ++i)
    {
        uint8_t* d = (uint8_t*)c->arena.temp(m * s->col_bytes[i] + 16);
        HIP_TRY(hipMemcpyAsync(d, s->col[i], m * s->col_bytes[i], hipMemcpyHostToDevice, c->stream));
        cols.src[i] = d; cols.bytes[i] = s->col_bytes[i]; cols.shift[i] = s->col_shift[i];
    }
    hipLaunchKernelGGL(popc_words_kernel, dim3(grid_for(s->high_words, 256)), dim3(256), 0, c->stream,
                       (const uint64_t*)words, s->high_words, prefix);
    exclusive_scan_u64(c, prefix, s->high_words);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ef_decode_kernel<K>), dim3(grid_for(s->high_words, 256)), dim3(256), 0, c->stream,
                       (const uint64_t*)words, s->high_words, (const uint64_t*)prefix, (uint32_t)s->D, cols, m, out);
}
static bool sparse_run_ok(const goss_gpu_sparse_run* s) { return s && s->ncols >= 1 && s->ncols <= 4 && (!s->count || s->high_bits); }
static uint64_t sparse_run_need(const goss_gpu_sparse_run* s)
{
    uint64_t need = s->high_words * 16 + 4096;
    for (uint32_t i = 0; i < s->ncols; ++i) need += s->count * s->col_bytes[i] + 256;
    return need;
}
}  // extern "C++"

int goss_gpu_push_run_sparse(goss_gpu_ctx* c, const goss_gpu_sparse_run* s)
{
    if (!c || !sparse_run_ok(s)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    if (s->count == 0) return GOSS_OK;
    return guarded(c, [&]() {
        ensure_arena(c);
        flush_staging(c);
        const uint64_t m = s->count, ksz = c->words * 8;
        {
            const uint64_t need = m * (ksz + 4) + sparse_run_need(s) + (64u << 20);
            if (c->arena.avail() < need) grow_arena(c, need);
        }
        Run r{nullptr, nullptr, m};
        r.keys = c->arena.perm(m * ksz);
        r.counts = (uint32_t*)c->arena.perm(m * 4);
        uint64_t mark = c->arena.mark();
        if (c->words == 1) decode_sparse<Key1>(c, s, (Key1*)r.keys); else decode_sparse<Key2>(c, s, (Key2*)r.keys);
        if (s->counts) HIP_TRY(hipMemcpyAsync(r.counts, s->counts, m * 4, hipMemcpyHostToDevice, c->stream));
        else hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, r.counts, m, s->weight ? s->weight : 1u);
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
        c->runs.push_back(r);
    });
}

int goss_gpu_push_run_graph(goss_gpu_ctx* c, const goss_gpu_sparse_run* edges, const goss_gpu_vba* v)
{
    if (!c || !sparse_run_ok(edges) || !v || !sparse_run_ok(&v->ord1p) || !sparse_run_ok(&v->ord2p)) return GOSS_ERR_INVALID_ARG;
    if (c->finished) { c->last_error = "push after finish"; return GOSS_ERR_STATE; }
    if (c->broken) { c->last_error = kBrokenMsg; return GOSS_ERR_STATE; }
    const uint64_t m = edges->count, n1 = v->ord1p.count, n2 = v->ord2p.count;
    if (m == 0) return GOSS_OK;
    if (v->ord0_bytes < m || v->ord1_bytes < n1 || v->ord2_bytes < 2 * n2 || (m && !v->ord0) || (n1 && !v->ord1) || (n2 && !v->ord2))
    {
        c->last_error = "VariableByteArray files shorter than their presence arrays say";
        return GOSS_ERR_INVALID_ARG;
    }
    return guarded(c, [&]() {
        ensure_arena(c);
        flush_staging(c);
        const uint64_t ksz = c->words * 8;
        {
            const uint64_t need = m * (ksz + 5) + sparse_run_need(edges) + sparse_run_need(&v->ord1p) + sparse_run_need(&v->ord2p)
                                  + n1 * 9 + n2 * 10 + (64u << 20);
            if (c->arena.avail() < need) grow_arena(c, need);
        }
        Run r{nullptr, nullptr, m};
        r.keys = c->arena.perm(m * ksz);
        r.counts = (uint32_t*)c->arena.perm(m * 4);
        uint64_t mark = c->arena.mark();
        if (c->words == 1) decode_sparse<Key1>(c, edges, (Key1*)r.keys); else decode_sparse<Key2>(c, edges, (Key2*)r.keys);
        // the multiplicities: VariableByteArray read on the device
        uint8_t* d0 = (uint8_t*)c->arena.temp(m + 16);
        HIP_TRY(hipMemcpyAsync(d0, v->ord0, m, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(vba_read0_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, (const uint8_t*)d0, m, r.counts);
        if (n1)
        {
            Key1* p1 = (Key1*)c->arena.temp(n1 * 8);
            uint8_t* d1 = (uint8_t*)c->arena.temp(n1 + 16);
            decode_sparse<Key1>(c, &v->ord1p, p1);
            HIP_TRY(hipMemcpyAsync(d1, v->ord1, n1, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(vba_read1_kernel, dim3(grid_for(n1, 256)), dim3(256), 0, c->stream, (const Key1*)p1, (const uint8_t*)d1, n1, m, r.counts);
            if (n2)
            {
                Key1* p2 = (Key1*)c->arena.temp(n2 * 8);
                uint16_t* d2 = (uint16_t*)c->arena.temp(n2 * 2 + 16);
                decode_sparse<Key1>(c, &v->ord2p, p2);
                HIP_TRY(hipMemcpyAsync(d2, v->ord2, n2 * 2, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(vba_read2_kernel, dim3(grid_for(n2, 256)), dim3(256), 0, c->stream, (const Key1*)p2, (const uint16_t*)d2, n2,
                                   (const Key1*)p1, n1, m, r.counts);
            }
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
        c->runs.push_back(r);
    });
}

extern "C++" {
template <class K>
static void select_counts(goss_gpu_ctx* c, uint32_t lo, uint32_t hi)
{
    const uint64_t n = c->M;
    if (n == 0) return;
    uint64_t mark = c->arena.mark();
    const uint64_t ntiles = (n + kRedTile - 1) / kRedTile;
    uint64_t* tile_counts = (uint64_t*)c->arena.temp((ntiles + 1) * 8);
    hipLaunchKernelGGL(select_count_kernel, dim3(grid_for(n, kRedTile)), dim3(kTB), 0, c->stream,
                       (const uint32_t*)c->res_counts, n, lo, hi, tile_counts);
    HIP_TRY(hipMemsetAsync(tile_counts + ntiles, 0, 8, c->stream));
    exclusive_scan_u64(c, tile_counts, ntiles + 1);
    uint64_t* h = (uint64_t*)c->h_pinned;
    HIP_TRY(hipMemcpyAsync(h, tile_counts + ntiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t m = h[0];
    K* keys = (K*)c->arena.perm(std::max<uint64_t>(m * sizeof(K), 16));
    uint32_t* counts = (uint32_t*)c->arena.perm(std::max<uint64_t>(m * 4, 16));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(select_write_kernel<K>), dim3(grid_for(n, kRedTile)), dim3(kTB), 0, c->stream,
                       (const K*)c->res_keys, (const uint32_t*)c->res_counts, n, lo, hi, (const uint64_t*)tile_counts, keys, counts);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->arena.release(mark);
    c->res_keys = keys; c->res_counts = counts; c->M = m;
}
}  // extern "C++"

int goss_gpu_select_counts(goss_gpu_ctx* c, uint32_t lo, uint32_t hi)
{
    if (!c || lo > hi) return GOSS_ERR_INVALID_ARG;
    if (!c->finished || c->emitted) { c->last_error = "select_counts belongs between finish and emit"; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        PhaseTimer t(c, GOSS_T_REDUCE, c->M);
        if (c->words == 1) select_counts<Key1>(c, lo, hi); else select_counts<Key2>(c, lo, hi);
        t.stop();
    });
}

int goss_gpu_select_normal(goss_gpu_ctx* c)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (!c->finished || c->emitted) { c->last_error = "select_normal belongs between finish and emit"; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        PhaseTimer t(c, GOSS_T_REDUCE, c->M);
        if (c->M)
        {
            const dim3 grid = unit_grid((c->M + kTB - 1) / kTB);
            if (c->words == 1)
            {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(mark_normal_kernel<Key1>), grid, dim3(kTB), 0, c->stream,
                                   (const Key1*)c->res_keys, c->M, c->len, c->res_counts);
                select_counts<Key1>(c, 1, 1);
            }
            else
            {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(mark_normal_kernel<Key2>), grid, dim3(kTB), 0, c->stream,
                                   (const Key2*)c->res_keys, c->M, c->len, c->res_counts);
                select_counts<Key2>(c, 1, 1);
            }
        }
        t.stop();
    });
}

// DenseSelect::DenseSelect (DenseArray.cc:36-91): header checks, then the device view of the file
static RdDenseSelect open_dense_select(goss_gpu_ctx* c, const void* host, uint64_t size, int invert)
{
    if (!host || size < sizeof(DsHeader)) throw StatusError{GOSS_ERR_INVALID_ARG, "DenseSelect file too short"};
    DsHeader h;
    std::memcpy(&h, host, sizeof h);
    if (h.version != 2012092701ULL) throw StatusError{GOSS_ERR_INVALID_ARG, "DenseSelect version mismatch"};
    if (h.logBlockSize > 40 || h.logSampleRate > h.logBlockSize || (1ULL << h.logBlockSize) != h.blockSize ||
        (1ULL << h.logSampleRate) != h.sampleRate || h.smallBlocks + h.intermediateBlocks + h.largeBlocks != h.numBlocks)
        throw StatusError{GOSS_ERR_INVALID_ARG, "Corrupt DenseSelect index header"};
    if ((int)(h.flags & 1) != invert) throw StatusError{GOSS_ERR_INVALID_ARG, "DenseSelect index does not have the expected sense"};
    if (h.indexArrayOffset + h.numBlocks * 8 > size || h.rankArrayOffset + h.numBlocks * 8 > size)
        throw StatusError{GOSS_ERR_INVALID_ARG, "DenseSelect arrays lie outside the file"};
    uint8_t* d = (uint8_t*)c->arena.temp(size + 16);
    HIP_TRY(hipMemcpyAsync(d, host, size, hipMemcpyHostToDevice, c->stream));
    RdDenseSelect r{};
    r.data = d; r.size = size; r.flags = h.flags; r.indexArrayOffset = h.indexArrayOffset; r.rankArrayOffset = h.rankArrayOffset;
    r.logBlockSize = h.logBlockSize; r.blockSize = h.blockSize; r.logSampleRate = h.logSampleRate; r.sampleRate = h.sampleRate;
    r.numBlocks = h.numBlocks;
    return r;
}

int goss_gpu_check_index(goss_gpu_ctx* c, const goss_gpu_sparse_files* f, goss_gpu_index_report* out)
{
    if (!c || !f || !out || f->ncols == 0 || f->ncols > 4) return GOSS_ERR_INVALID_ARG;
    static_assert(sizeof(goss_gpu_index_report) == sizeof(IndexReport), "index report layout");
    if (!c->finished) { c->last_error = "check_index before finish"; return GOSS_ERR_STATE; }
    std::memset(out, 0, sizeof *out);
    if (c->M != f->count) { c->last_error = "the object's count differs from the number of decoded elements"; return GOSS_ERR_STATE; }
    if (c->M == 0) return GOSS_OK;
    return guarded(c, [&]() {
        uint64_t mark = c->arena.mark();
        RdSparse s{};
        s.D = f->D; s.count = f->count; s.size_lo = f->size_lo; s.size_hi = f->size_hi;
        uint64_t* hw = (uint64_t*)c->arena.temp(f->high_words * 8 + 16);
        HIP_TRY(hipMemcpyAsync(hw, f->high_bits, f->high_words * 8, hipMemcpyHostToDevice, c->stream));
        s.hi.w = hw; s.hi.nwords = f->high_words;
        s.d0 = open_dense_select(c, f->d0, f->d0_bytes, 1);
        s.d1 = open_dense_select(c, f->d1, f->d1_bytes, 0);
        s.ncols = f->ncols;
        for (uint32_t i = 0; i < f->ncols; ++i)
        {
            uint8_t* d = (uint8_t*)c->arena.temp(f->count * f->col_bytes[i] + 16);
            HIP_TRY(hipMemcpyAsync(d, f->col[i], f->count * f->col_bytes[i], hipMemcpyHostToDevice, c->stream));
            s.col[i] = d; s.col_bytes[i] = f->col_bytes[i]; s.col_shift[i] = f->col_shift[i];
        }
        IndexReport* rep = (IndexReport*)c->arena.temp(sizeof(IndexReport));
        HIP_TRY(hipMemsetAsync(rep, 0, sizeof(IndexReport), c->stream));
        if (c->words == 1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(check_index_kernel<Key1>), dim3(grid_for(c->M, 256)), dim3(256), 0, c->stream, s,
                               (const Key1*)c->res_keys, c->M, rep);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(check_index_kernel<Key2>), dim3(grid_for(c->M, 256)), dim3(256), 0, c->stream, s,
                               (const Key2*)c->res_keys, c->M, rep);
        HIP_TRY(hipMemcpyAsync(out, rep, sizeof(IndexReport), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
        if (out->nexamples > 16) out->nexamples = 16;
    });
}

int goss_gpu_set_budget_limit(goss_gpu_ctx* c, uint64_t max_bytes)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    c->budget_limit = max_bytes;
    return GOSS_OK;
}

int goss_gpu_stat(goss_gpu_ctx* c, const char* name, uint64_t* value)
{
    if (!c || !name || !value) return GOSS_ERR_INVALID_ARG;
    const std::string n = name;
    if (n == "fused_chunks") *value = c->fused_chunks;
    else if (n == "rep_chunks") *value = c->rep_chunks;
    else if (n == "canon_chunks") *value = c->canon_chunks;
    else if (n == "rec_chunks") *value = c->rec_chunks;
    else if (n == "flush_wait_us") *value = c->flush_wait_us;
    else if (n == "flush_count_us") *value = c->flush_count_us;
    else if (n == "flushes") *value = c->flushes;
    else if (n == "fused_overflows") *value = c->fused_overflows;
    else if (n == "fused_msd_chunks") *value = c->fused_msd_chunks;
    else if (n == "rem32_chunks") *value = c->rem32_chunks;
    else if (n == "narrow_chunks") *value = c->narrow_chunks;
    else if (n == "overflow_units") *value = c->overflow_units;
    else if (n == "packed_fused_chunks") *value = c->pk_fused_chunks;
    else if (n == "ds_blocks_from_ranges") *value = c->ds_blocks_ranges;
    else if (n == "assemble_us") *value = c->assemble_us;
    else if (n == "ds_blocks_assembled") *value = c->ds_blocks_own;
    else if (n == "packed_unpacked_chunks") *value = c->pk_unpacked_chunks;
    else if (n == "rem32_bits") *value = c->rem32_bits_last;
    else if (n == "rem32_split") *value = c->rem32_split_last;
    else if (n == "big_table_chunks") *value = c->big_table_chunks;
    else if (n == "wide_table_chunks") *value = c->wide_table_chunks;
    else if (n == "table96_chunks") *value = c->table96_chunks;
    else if (n == "valid_sized_chunks") *value = c->valid_sized_chunks;
    else if (n == "valid_resizes") *value = c->valid_resizes;
    else if (n == "seg_merges") *value = c->seg_merges;
    else if (n == "hash_merges") *value = c->hash_merges;
    else if (n == "segment_retries") *value = c->segment_retries;
    else if (n == "lookback_failures") *value = c->lookback_failures;
    else if (n == "runs") *value = c->runs.size();
    else if (n == "arena_ms") *value = c->arena_ms;
    else if (n == "arena_grows") *value = c->arena_grows;
    else if (n == "arena_bytes") *value = c->arena.size;
    else return GOSS_ERR_INVALID_ARG;
    return GOSS_OK;
}

int goss_gpu_emit_dump_range(goss_gpu_ctx* c, uint64_t flags, uint64_t first, uint64_t count)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    if (!c->finished) { c->last_error = "dump before finish"; return GOSS_ERR_STATE; }
    if (first > c->M || count > c->M - first) return GOSS_ERR_INVALID_ARG;
    return guarded(c, [&]() {
        c->files.clear();
        // the previous piece's text is dead: give its room back (nothing else was allocated since)
        if (c->dump_live) { c->arena.lo = c->dump_lo; c->dump_live = false; }
        PhaseTimer t(c, GOSS_T_EMIT, count);
        const uint64_t m = count;
        const uint32_t len = c->len;
        // "#<version>\nK\tcount\n" (GossCmdDumpKmerSet.cc:44-45) / "#<version>\nK\tcount\tflags\n"
        // (GossCmdDumpGraph.cc:50-51): in front of the first piece only
        char head[128];
        int hl = 0;
        if (first == 0)
            hl = c->mode == GOSS_MODE_KMER_SET
                     ? std::snprintf(head, sizeof head, "#%llu\n%u\t%llu\n", 2011101701ULL, c->k, (unsigned long long)c->M)
                     : std::snprintf(head, sizeof head, "#%llu\n%u\t%llu\t%llu\n", 2011101014ULL, c->k, (unsigned long long)c->M,
                                     (unsigned long long)flags);
        const uint64_t ksz = c->words * 8;
        const uint8_t* keys = (const uint8_t*)c->res_keys + first * ksz;
        const uint32_t* counts = c->res_counts + first;
        uint64_t body = 0;
        uint64_t mark = c->arena.mark();
        uint64_t* offs = nullptr;
        if (c->mode == GOSS_MODE_KMER_SET) body = m * (len + 1ULL);
        else if (m)
        {
            offs = (uint64_t*)c->arena.temp((m + 1) * 8);
            hipLaunchKernelGGL(dump_line_len_kernel, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, counts, m, len, offs);
            HIP_TRY(hipMemsetAsync(offs + m, 0, 8, c->stream));
            exclusive_scan_u64(c, offs, m + 1);
            uint64_t* h = (uint64_t*)c->h_pinned;
            HIP_TRY(hipMemcpyAsync(h, offs + m, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            body = h[0];
        }
        c->dump_lo = c->arena.lo;
        uint8_t* text = (uint8_t*)c->arena.perm(hl + body + 16);
        c->dump_live = true;
        if (hl) HIP_TRY(hipMemcpyAsync(text, head, hl, hipMemcpyHostToDevice, c->stream));
        if (m)
        {
            if (c->mode == GOSS_MODE_KMER_SET)
            {
                const uint32_t grid = (uint32_t)std::min<uint64_t>(grid_for(body, 256), 256 * 64);
                if (c->words == 1)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(dump_kmers_kernel<Key1>), dim3(grid), dim3(256), 0, c->stream,
                                       (const Key1*)keys, m, len, text + hl);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(dump_kmers_kernel<Key2>), dim3(grid), dim3(256), 0, c->stream,
                                       (const Key2*)keys, m, len, text + hl);
            }
            else if (c->words == 1)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(dump_edges_kernel<Key1>), dim3(grid_for(m, 256)), dim3(256), 0, c->stream,
                                   (const Key1*)keys, counts, (const uint64_t*)offs, m, len, text + hl);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(dump_edges_kernel<Key2>), dim3(grid_for(m, 256)), dim3(256), 0, c->stream,
                                   (const Key2*)keys, counts, (const uint64_t*)offs, m, len, text + hl);
        }
        t.stop();
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
        OutFile f; f.suffix = ".dump"; f.size = hl + body; f.dev = text;
        c->files.push_back(std::move(f));
    });
}

int goss_gpu_emit_dump(goss_gpu_ctx* c, uint64_t flags)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    return goss_gpu_emit_dump_range(c, flags, 0, c->M);
}

int goss_gpu_lint(goss_gpu_ctx* c, int asymmetric, goss_gpu_lint_report* out)
{
    if (!c || !out) return GOSS_ERR_INVALID_ARG;
    static_assert(sizeof(goss_gpu_lint_report) == sizeof(LintReport), "lint report layout");
    if (!c->finished) { c->last_error = "lint before finish"; return GOSS_ERR_STATE; }
    if (c->mode != GOSS_MODE_GRAPH) { c->last_error = "lint checks a graph"; return GOSS_ERR_STATE; }
    std::memset(out, 0, sizeof *out);
    if (c->M == 0) return GOSS_OK;
    return guarded(c, [&]() {
        uint64_t mark = c->arena.mark();
        LintReport* rep = (LintReport*)c->arena.temp(sizeof(LintReport));
        HIP_TRY(hipMemsetAsync(rep, 0, sizeof(LintReport), c->stream));
        if (c->words == 1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(lint_edges_kernel<Key1>), dim3(grid_for(c->M, 256)), dim3(256), 0, c->stream,
                               (const Key1*)c->res_keys, (const uint32_t*)c->res_counts, c->M, c->len, asymmetric, rep);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(lint_edges_kernel<Key2>), dim3(grid_for(c->M, 256)), dim3(256), 0, c->stream,
                               (const Key2*)c->res_keys, (const uint32_t*)c->res_counts, c->M, c->len, asymmetric, rep);
        HIP_TRY(hipMemcpyAsync(out, rep, sizeof(LintReport), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->arena.release(mark);
        if (out->nexamples > 32) out->nexamples = 32;
    });
}

int goss_gpu_emit_count_bits(goss_gpu_ctx* c, uint32_t mask, const char* suffix)
{
    if (!c || !suffix || !mask) return GOSS_ERR_INVALID_ARG;
    if (!c->emitted) { c->last_error = "emit_count_bits follows emit"; return GOSS_ERR_STATE; }
    return guarded(c, [&]() {
        // WordyBitVector::Builder after M push_backX calls and end(): floor((M-1)/64)+1 words,
        // one (zero) word when nothing was pushed (WordyBitVector.hh:90-116, .cc:18-29)
        const uint64_t m = c->M, nwords = m ? (m - 1) / 64 + 1 : 1;
        uint64_t* words = (uint64_t*)c->arena.perm(nwords * 8);
        if (m == 0) HIP_TRY(hipMemsetAsync(words, 0, 8, c->stream));
        else
            hipLaunchKernelGGL(count_bits_kernel, dim3(grid_for((nwords + 63) / 64 * 64, 256)), dim3(256), 0, c->stream,
                               (const uint32_t*)c->res_counts, m, mask, words, nwords);
        HIP_TRY(hipStreamSynchronize(c->stream));
        OutFile f; f.suffix = suffix; f.size = nwords * 8; f.dev = (const uint8_t*)words;
        c->files.push_back(std::move(f));
    });
}

int goss_gpu_emit_estimate(goss_gpu_ctx* c, uint64_t m_estimate)
{
    if (!c) return GOSS_ERR_INVALID_ARG;
    c->emit_estimate = m_estimate;
    c->has_emit_estimate = true;
    int rc = goss_gpu_emit(c);
    c->has_emit_estimate = false;
    return rc;
}

int goss_gpu_synth_reads(goss_gpu_ctx* c, void* d_out, uint64_t nreads, uint32_t read_len, uint64_t genome_len,
                         uint64_t seed, uint64_t first_read)
{
    if (!c || !d_out || read_len == 0 || genome_len < read_len) return GOSS_ERR_INVALID_ARG;
    return guarded(c, [&]() {
        hipLaunchKernelGGL(synth_reads_kernel, dim3(256 * 8), dim3(256), 0, c->stream, (uint8_t*)d_out, nreads, read_len,
                           genome_len, seed, first_read);
        HIP_TRY(hipStreamSynchronize(c->stream));
    });
}

int goss_synth_reads_host(char* out, uint64_t nreads, uint32_t read_len, uint64_t genome_len, uint64_t seed,
                          uint64_t first_read)
{
    if (!out || read_len == 0 || genome_len < read_len) return GOSS_ERR_INVALID_ARG;
    const uint64_t stride = read_len + 1;
    for (uint64_t r = 0; r < nreads; ++r)
        for (uint32_t j = 0; j <= read_len; ++j)
            out[r * stride + j] = synth_read_byte(seed, genome_len, read_len, first_read + r, j);
    return GOSS_OK;
}

}  // extern "C"
