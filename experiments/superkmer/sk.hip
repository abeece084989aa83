// Super-k-mer extraction, standalone experiment (NOT part of the product: nothing here is built by
// __graft_entry__.build() or loaded by the library).  Purpose: measure, on C2's reads, what DESIGN.md section 7
// item 1 estimates -- records per window, bytes per window, and the time of an extraction kernel that computes
// minimizers and cuts runs of windows that share one -- and keep a verified starting point for the redesign.
//
// Record (16 bytes, the shape of a two-word key so that the partition kernels can take it unchanged):
//   hi = bin:16 | (nwin-1):4 | 0:16 | bases[91:64]:28      lo = bases[63:0]
// bases = the (nwin + k - 1) bases of the run as 2-bit codes, first base most significant, right-aligned.
// bin = 16 bits of a second hash of the run's minimizer (canonical m-mer, the smallest under a first hash), so
// that both strands of a k-mer land in the same bin.  A thread cuts records inside its own 16 windows only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o sk sk.hip     Run (GPU box): ./sk [reads] [genome] [k] [m]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kTB = 256;
constexpr int P = 16;                      // windows per thread
constexpr int T = kTB * P;                 // window starts per tile

struct Rec { uint64_t lo, hi; };

__device__ __forceinline__ uint64_t splitmix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// reads: L bases + '\n', positions from a counter-based generator (one N in every 97th read)
__global__ void synth_kernel(uint8_t* out, uint64_t nreads, uint32_t L, uint64_t genome, uint64_t seed)
{
    // (grid-stride: a grid of gridDim.x * blockDim.x >= 2^32 threads is refused without an error)
    const uint64_t total = nreads * (L + 1), stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        const uint64_t r = i / (L + 1), j = i % (L + 1);
        if (j == L) { out[i] = '\n'; continue; }
        const uint64_t pos = splitmix(seed * 0x51ED27 + r) % (genome - L);
        const bool flip = splitmix(seed ^ (r * 0x9E37)) & 1;
        const uint64_t g = flip ? pos + (L - 1 - j) : pos + j;
        uint32_t c = (uint32_t)(splitmix(seed * 77 + g / 32) >> (2 * (g % 32))) & 3u;
        if (flip) c ^= 3u;
        out[i] = (r % 97 == 96 && j == L / 2) ? 'N' : "ACGT"[c];
    }
}

__device__ __forceinline__ uint64_t rev2(uint64_t x)       // reverse the 2-bit groups of a 64-bit word
{
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
}
__device__ __forceinline__ uint64_t canon(uint64_t f, uint32_t k)   // min of a k-mer and its reverse complement (k <= 31)
{
    const uint64_t r = rev2(~f) >> (64 - 2 * k);
    return f < r ? f : r;
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) { return splitmix(x); }

// order-independent multiset hash of the canonical k-mers + their number: the reference the records are checked against
__global__ void direct_kernel(const uint8_t* bases, uint64_t nbytes, uint32_t k, unsigned long long* acc)
{
    // every thread scans a slice of 4096 window starts, rolling
    const uint64_t s0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 64;
    if (s0 >= nbytes) return;
    const uint64_t kmask = (1ULL << (2 * k)) - 1;
    uint64_t f = 0, sum = 0, cnt = 0;
    uint32_t good = 0;
    for (uint64_t p = s0; p < nbytes && p < s0 + 64 + k - 1; ++p)
    {
        const uint8_t c = bases[p] | 0x20;
        uint32_t code = 4;
        if (c == 'a') code = 0; else if (c == 'c') code = 1; else if (c == 'g') code = 2; else if (c == 't') code = 3;
        if (code > 3) { good = 0; continue; }
        f = ((f << 2) | code) & kmask;
        if (++good >= k && p + 1 - k < s0 + 64) { sum += mix64(canon(f, k)); ++cnt; }
    }
    atomicAdd(&acc[0], (unsigned long long)sum);
    atomicAdd(&acc[1], (unsigned long long)cnt);
}

// ---- extraction -------------------------------------------------------------------------------------------
// COUNT: only the number of records per tile (first pass); else records are written from tile_off[tile].
template <int W, bool COUNT>     // W = k - m + 1 minimizer positions per window; P + W - 1 <= 32
__global__ __launch_bounds__(kTB) void sk_extract_kernel(const uint8_t* __restrict__ bases, uint64_t nbytes, uint32_t k, uint32_t m,
                                                         uint32_t* __restrict__ tile_cnt, const uint64_t* __restrict__ tile_off,
                                                         Rec* __restrict__ out, unsigned long long* __restrict__ stats)
{
    constexpr int NPOS = P + W - 1;
    static_assert(NPOS <= 32, "m-mer positions per thread");
    constexpr int NVEC = T / 16 + 4;
    __shared__ uint32_t pk[NVEC], iv[NVEC];
    __shared__ uint32_t sh_scan[kTB / 64 + 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const uint64_t tb = tile * T;
    // bytes of the tile -> 2-bit codes (16 per word) + non-base flags (16 per word)
    for (uint32_t v = tid; v < NVEC; v += kTB)
    {
        uint32_t codes = 0, bads = 0;
        for (int j = 0; j < 16; ++j)
        {
            const uint64_t p = tb + (uint64_t)v * 16 + j;
            const uint8_t c = p < nbytes ? (bases[p] | 0x20) : '\n';
            uint32_t code = 0, bad = 0;
            if (c == 'a') code = 0; else if (c == 'c') code = 1; else if (c == 'g') code = 2; else if (c == 't') code = 3; else bad = 1;
            codes |= code << (2 * j); bads |= bad << j;
        }
        pk[v] = codes; iv[v] = bads;
    }
    __syncthreads();
    const uint32_t v0 = tid;               // thread tid starts at vector tid (16 windows = one vector)
    const uint64_t inv = (uint64_t)iv[v0] | ((uint64_t)iv[v0 + 1] << 16) | ((uint64_t)iv[v0 + 2] << 32) | ((uint64_t)iv[v0 + 3] << 48);
    const uint64_t blo = (uint64_t)pk[v0] | ((uint64_t)pk[v0 + 1] << 32);       // bases 0..31 (base j at bits 2j)
    const uint64_t bhi = (uint64_t)pk[v0 + 2] | ((uint64_t)pk[v0 + 3] << 32);   // bases 32..63
    // valid windows: bits [i, i+k) of inv zero, by doubling
    uint32_t valid;
    {
        uint64_t run = ~inv, acc = ~0ULL; uint32_t covered = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) { if ((k >> j) & 1u) { acc &= run >> covered; covered += 1u << j; } run &= run >> (1u << j); }
        const uint64_t p0 = tb + (uint64_t)tid * P;
        const uint64_t nstarts = nbytes >= k ? nbytes - k + 1 : 0;
        const uint64_t left = nstarts > p0 ? nstarts - p0 : 0;
        valid = (uint32_t)acc & (left >= P ? 0xFFFFu : ((1u << left) - 1u));
    }
    // m-mers at positions 0 .. NPOS-1: forward and reverse complement rolled, canonical, first hash | position
    const uint32_t mmask = (1u << (2 * m)) - 1u;
    uint32_t fm = 0, rm = 0;
    uint32_t val[NPOS];
    auto base_at = [&](int j) -> uint32_t { return (uint32_t)(j < 32 ? (blo >> (2 * j)) : (bhi >> (2 * (j - 32)))) & 3u; };
#pragma unroll
    for (int j = 0; j < NPOS + 8; ++j)       // m <= 9: the first m-1 bases prime the roll
    {
        if (j >= (int)(NPOS + m - 1)) break;
        const uint32_t nb = base_at(j);
        fm = ((fm << 2) | nb) & mmask;
        rm = (rm >> 2) | ((nb ^ 3u) << (2 * (m - 1)));
        const int pos = j - (int)(m - 1);
        if (pos >= 0 && pos < NPOS)
        {
            const uint32_t c = fm < rm ? fm : rm;
            val[pos] = ((c * 0x9E3779B1u) & ~63u) | (uint32_t)pos;
        }
    }
    // sliding minimum over W positions: suffix minima of [0, W), prefix minima of [W, NPOS)
    uint32_t suf[W];
    suf[W - 1] = val[W - 1];
#pragma unroll
    for (int j = W - 2; j >= 0; --j) suf[j] = min(val[j], suf[j + 1]);
    uint32_t mn[P];
    mn[0] = suf[0];
    uint32_t pre = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 1; i < P; ++i) { pre = min(pre, val[W - 1 + i]); mn[i] = min(suf[i], pre); }
    // a record starts at a valid window whose predecessor is not valid or has another minimizer
    uint32_t starts = 0;
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        const uint32_t ok = (valid >> i) & 1u;
        const uint32_t prev_ok = i ? (valid >> (i - 1)) & 1u : 0u;
        const uint32_t same = i ? (mn[i] == mn[i - 1] ? 1u : 0u) : 0u;
        starts |= (ok & ((prev_ok & same) ^ 1u)) << i;
    }
    const uint32_t nrec = __popc(starts);
    // place of this thread's records in the tile
    uint32_t tot = 0, at = 0;
    {
        uint32_t x = nrec;
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o, 64); if ((tid & 63) >= (uint32_t)o) x += y; }
        if ((tid & 63) == 63) sh_scan[tid >> 6] = x;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t w = 0; w < (tid >> 6); ++w) base += sh_scan[w];
        for (uint32_t w = 0; w < kTB / 64; ++w) tot += sh_scan[w];
        at = base + x - nrec;
    }
    if (COUNT) { if (tid == 0) tile_cnt[tile] = tot; return; }
    const uint64_t obase = tile_off[tile] + at;
    uint32_t todo = starts, r = 0;
    unsigned long long nw = 0;
    while (todo)
    {
        const uint32_t s = __ffs(todo) - 1;
        todo &= todo - 1;
        // length: up to the next start or the first window that is not valid
        const uint32_t stop = (todo | ~valid | (1u << P)) >> s;            // bit j set: window s + j ends the run
        const uint32_t n = __ffs(stop & ~1u) - 1;                          // (bit 0 is the start itself)
        const uint32_t nbases = n + k - 1;
        // bases s .. s + nbases - 1, first base most significant
        unsigned __int128 all = ((unsigned __int128)bhi << 64) | blo;
        unsigned __int128 field = (all >> (2 * s)) & ((((unsigned __int128)1) << (2 * nbases)) - 1);
        // reverse the base order inside the field (base j at bits 2j -> first base most significant)
        const uint64_t flo = (uint64_t)field, fhi = (uint64_t)(field >> 64);
        unsigned __int128 rv = (((unsigned __int128)rev2(flo)) << 64) | rev2(fhi);
        rv >>= (128 - 2 * nbases);
        // bin: a second hash of the run's minimizer (its first-hash value, the upper bits of mn[s])
        uint32_t mv = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) if ((uint32_t)i == s) mv = mn[i];
        const uint32_t bin = (uint32_t)(mix64(mv >> 6) >> 48);
        Rec rec;
        rec.lo = (uint64_t)rv;
        rec.hi = ((uint64_t)bin << 48) | ((uint64_t)(n - 1) << 44) | (uint64_t)(rv >> 64);
        out[obase + r] = rec;
        ++r; nw += n;
    }
    for (int o = 32; o > 0; o >>= 1) nw += __shfl_down(nw, o, 64);
    if ((tid & 63) == 0 && nw) atomicAdd(&stats[0], nw);
}

// records -> canonical k-mers: multiset hash + count (and the number of bin changes between neighbours, for information)
__global__ void sk_expand_kernel(const Rec* __restrict__ recs, uint64_t n, uint32_t k, unsigned long long* acc)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t sum = 0, cnt = 0;
    if (i < n)
    {
        const Rec r = recs[i];
        const uint32_t nwin = (uint32_t)((r.hi >> 44) & 15u) + 1;
        const unsigned __int128 b = (((unsigned __int128)(r.hi & 0xFFFFFFFULL)) << 64) | r.lo;
        const uint64_t kmask = (1ULL << (2 * k)) - 1;
        for (uint32_t j = 0; j < nwin; ++j)
        {
            const uint64_t f = (uint64_t)(b >> (2 * (nwin - 1 - j))) & kmask;
            sum += mix64(canon(f, k)); ++cnt;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o, 64); cnt += __shfl_down(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0 && cnt) { atomicAdd(&acc[0], (unsigned long long)sum); atomicAdd(&acc[1], (unsigned long long)cnt); }
}

int main(int argc, char** argv)
{
    const uint64_t nreads = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ULL;
    const uint64_t genome = argc > 2 ? strtoull(argv[2], nullptr, 10) : 100000000ULL;
    const uint32_t k = argc > 3 ? atoi(argv[3]) : 25, m = argc > 4 ? atoi(argv[4]) : 9, L = 150;
    if (k - m + 1 != 17) { std::fprintf(stderr, "this build is instantiated for k - m + 1 = 17 (k = 25, m = 9)\n"); return 2; }
    const uint64_t nbytes = nreads * (L + 1);
    uint8_t* bases; CHECK(hipMalloc(&bases, nbytes));
    hipLaunchKernelGGL(synth_kernel, dim3(65536), dim3(256), 0, 0, bases, nreads, L, genome, 1ULL);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    unsigned long long* acc; CHECK(hipMalloc(&acc, 64)); CHECK(hipMemset(acc, 0, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;

    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(direct_kernel, dim3((uint32_t)((nbytes / 64 + 255) / 256 + 1)), dim3(256), 0, 0, bases, nbytes, k, acc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long ref[2]; CHECK(hipMemcpy(ref, acc, 16, hipMemcpyDeviceToHost));
    std::printf("direct scan: %llu windows, multiset hash %016llx (%.1f ms)\n", ref[1], ref[0], ms);

    const uint64_t ntiles = (nbytes + T - 1) / T;
    uint32_t* tile_cnt; CHECK(hipMalloc(&tile_cnt, ntiles * 4));
    uint64_t* tile_off; CHECK(hipMalloc(&tile_off, (ntiles + 1) * 8));
    unsigned long long* stats; CHECK(hipMalloc(&stats, 64)); CHECK(hipMemset(stats, 0, 64));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sk_extract_kernel<17, true>), dim3((uint32_t)ntiles), dim3(kTB), 0, 0, bases, nbytes, k, m, tile_cnt,
                       (const uint64_t*)nullptr, (Rec*)nullptr, stats);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> hc(ntiles); CHECK(hipMemcpy(hc.data(), tile_cnt, ntiles * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> ho(ntiles + 1); uint64_t nrec = 0;
    for (uint64_t t = 0; t < ntiles; ++t) { ho[t] = nrec; nrec += hc[t]; }
    ho[ntiles] = nrec;
    CHECK(hipMemcpy(tile_off, ho.data(), (ntiles + 1) * 8, hipMemcpyHostToDevice));
    std::printf("count pass: %llu records (%.1f ms)\n", (unsigned long long)nrec, ms);
    Rec* recs; CHECK(hipMalloc(&recs, nrec * sizeof(Rec)));
    for (int it = 0; it < 2; ++it)
    {
        CHECK(hipMemset(stats, 0, 64));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sk_extract_kernel<17, false>), dim3((uint32_t)ntiles), dim3(kTB), 0, 0, bases, nbytes, k, m, tile_cnt,
                           (const uint64_t*)tile_off, recs, stats);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long st[1]; CHECK(hipMemcpy(st, stats, 8, hipMemcpyDeviceToHost));
    std::printf("extraction: %.1f ms for %llu windows in %llu records: %.2f windows per record, %.2f bytes per window written, %.1f G windows/s\n",
                ms, st[0], (unsigned long long)nrec, (double)st[0] / (double)nrec, 16.0 * (double)nrec / (double)st[0], (double)st[0] / ms / 1e6);
    CHECK(hipMemset(acc, 0, 64));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(sk_expand_kernel, dim3((uint32_t)((nrec + 255) / 256)), dim3(256), 0, 0, (const Rec*)recs, nrec, k, acc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long got[2]; CHECK(hipMemcpy(got, acc, 16, hipMemcpyDeviceToHost));
    std::printf("expansion: %llu windows, multiset hash %016llx (%.1f ms, one thread per record)\n", got[1], got[0], ms);
    const bool ok = got[0] == ref[0] && got[1] == ref[1] && st[0] == ref[1];
    std::printf("%s\n", ok ? "OK: the records hold exactly the k-mers of the reads" : "MISMATCH");
    return ok ? 0 : 1;
}
