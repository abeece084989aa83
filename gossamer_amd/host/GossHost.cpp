// GossHost.cpp -- see GossHost.hpp.  Host C++ only: all device work goes through the C ABI of
// libgossgpu.so; there is no CPU counting path in this program.
#include "GossHost.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstring>
#include <iostream>
#include <map>
#include <set>
#include <sstream>

#include "../../include/goss_gpu.h"

namespace gosshost {

// --------------------------------------------------------------------------------------
// InFile
// --------------------------------------------------------------------------------------

static bool endsWith(const std::string& s, const char* suf)
{
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// bzip2 input (PhysicalFileFactory.cc:262-298 puts a bzip2 filter in front of "*.bz2").  The image has the
// system's libbz2.so.1.0 but no bzlib.h, so the library is loaded at run time and its four documented reading
// functions are declared here; a system without it gets the error the build used to give for every .bz2 file.
namespace {
struct Bz2Api {
    void* (*readOpen)(int*, FILE*, int, int, void*, int) = nullptr;
    int (*read)(int*, void*, void*, int) = nullptr;
    void (*readClose)(int*, void*) = nullptr;
    void (*getUnused)(int*, void*, void**, int*) = nullptr;
    bool ok = false;
};
const Bz2Api& bz2Api()
{
    static Bz2Api api = [] {
        Bz2Api a;
        void* h = dlopen("libbz2.so.1.0", RTLD_NOW);
        if (!h) h = dlopen("libbz2.so.1", RTLD_NOW);
        if (!h) return a;
        a.readOpen = (void* (*)(int*, FILE*, int, int, void*, int))dlsym(h, "BZ2_bzReadOpen");
        a.read = (int (*)(int*, void*, void*, int))dlsym(h, "BZ2_bzRead");
        a.readClose = (void (*)(int*, void*))dlsym(h, "BZ2_bzReadClose");
        a.getUnused = (void (*)(int*, void*, void**, int*))dlsym(h, "BZ2_bzReadGetUnused");
        a.ok = a.readOpen && a.read && a.readClose && a.getUnused;
        return a;
    }();
    return api;
}
constexpr int kBzOk = 0, kBzStreamEnd = 4, kBzBadMagic = -5;
}  // namespace

InFile::InFile(const std::string& name) : mName(name)
{
    if (name == "-") { mStdin = true; mFd = 0; return; }
    if (endsWith(name, ".bz2"))
    {
        if (!bz2Api().ok) throw Error::General("bzip2 input needs libbz2.so.1.0, which this system lacks: '" + name + "'\n");
        FILE* f = fopen(name.c_str(), "rb");
        if (!f) throw Error::Errno(name, errno ? errno : ENOENT);
        int err = 0;
        mBzFile = f;
        mBz = bz2Api().readOpen(&err, f, 0, 0, nullptr, 0);
        if (!mBz || err != kBzOk) { fclose(f); mBzFile = nullptr; throw Error::General("cannot read bzip2 stream '" + name + "'\n"); }
        return;
    }
    if (endsWith(name, ".gz"))
    {
        mGz = gzopen(name.c_str(), "rb");
        if (!mGz) throw Error::Errno(name, errno ? errno : ENOENT);
        gzbuffer((gzFile)mGz, 1 << 20);
        return;
    }
    mFd = ::open(name.c_str(), O_RDONLY);
    if (mFd < 0) throw Error::Errno(name, errno);
}

InFile::~InFile()
{
    if (mBz) { int err = 0; bz2Api().readClose(&err, mBz); }
    if (mBzFile) fclose((FILE*)mBzFile);
    if (mGz) gzclose((gzFile)mGz);
    else if (mFd > 0) ::close(mFd);
}

size_t InFile::read(char* dst, size_t cap)
{
    if (mBzFile)
    {
        // a .bz2 file may hold several streams one after the other (pbzip2, cat a.bz2 b.bz2): at the end of one,
        // the bytes the decoder read ahead start the next
        const Bz2Api& bz = bz2Api();
        size_t got = 0;
        while (got < cap && !mBzEnd)
        {
            int err = 0;
            const int n = bz.read(&err, mBz, dst + got, (int)std::min<size_t>(cap - got, 1u << 30));
            if (err != kBzOk && err != kBzStreamEnd)
            {
                // bytes behind the last stream that are not a stream (padding, a tape block's fill): the bzip2 tool
                // warns "trailing garbage after EOF ignored" and so does this reader
                if (err == kBzBadMagic && mBzStreams > 0 && mBzFresh)
                {
                    std::fprintf(stderr, "goss: %s: trailing garbage after the last bzip2 stream ignored\n", mName.c_str());
                    mBzEnd = true;
                    break;
                }
                throw Error::General("corrupt bzip2 stream '" + mName + "'\n");
            }
            got += (size_t)std::max(n, 0);
            if (n > 0) mBzFresh = false;
            if (err == kBzStreamEnd)
            {
                void* unused = nullptr; int nUnused = 0;
                bz.getUnused(&err, mBz, &unused, &nUnused);
                std::vector<char> rest((char*)unused, (char*)unused + std::max(nUnused, 0));
                bz.readClose(&err, mBz);
                mBz = nullptr;
                ++mBzStreams;
                mBzFresh = true;
                FILE* f = (FILE*)mBzFile;
                int c = EOF;
                if (rest.empty() && (c = fgetc(f)) == EOF) { mBzEnd = true; break; }
                if (rest.empty()) ungetc(c, f);
                mBz = bz.readOpen(&err, f, 0, 0, rest.empty() ? nullptr : rest.data(), (int)rest.size());
                if (!mBz || err != kBzOk) throw Error::General("corrupt bzip2 stream '" + mName + "'\n");
            }
        }
        return got;
    }
    if (mGz)
    {
        int n = gzread((gzFile)mGz, dst, (unsigned)std::min<size_t>(cap, 1u << 30));
        if (n < 0) throw Error::Errno(mName, EIO);
        return (size_t)n;
    }
    size_t got = 0;
    while (got < cap)
    {
        ssize_t n = ::read(mFd, dst + got, cap - got);
        if (n < 0) { if (errno == EINTR) continue; throw Error::Errno(mName, errno); }
        if (n == 0) break;
        got += (size_t)n;
        if (mStdin) break;
    }
    return got;
}

bool InFile::readable(const std::string& name)
{
    if (name == "-") return true;
    struct stat st;
    if (::stat(name.c_str(), &st) != 0 || S_ISDIR(st.st_mode)) return false;
    return ::access(name.c_str(), R_OK) == 0;
}

// --------------------------------------------------------------------------------------
// LineReader
// --------------------------------------------------------------------------------------

LineReader::LineReader(const std::string& name, size_t bufBytes) : mFile(name), mBuf(bufBytes)
{
    // PlainLineSource ctor: if (good) getline  (LineSource.cc:40-48)
    getline();
}

bool LineReader::refill()
{
    if (mEof) return false;
    if (mBeg > 0 && mBeg < mEnd) memmove(mBuf.data(), mBuf.data() + mBeg, mEnd - mBeg);
    mEnd -= mBeg; mBeg = 0;
    if (mEnd == mBuf.size()) mBuf.resize(mBuf.size() * 2);
    size_t n = mFile.read(mBuf.data() + mEnd, mBuf.size() - mEnd);
    if (n == 0) { mEof = true; return false; }
    mEnd += n;
    return true;
}

void LineReader::getline()
{
    mLine = mBuf.data() + mBeg; mLen = 0;
    if (!mGood) return;
    size_t scanned = 0;
    for (;;)
    {
        const char* p = mBuf.data() + mBeg;
        const char* nl = (const char*)memchr(p + scanned, '\n', mEnd - mBeg - scanned);
        if (nl)
        {
            mLine = p; mLen = (size_t)(nl - p);
            mBeg += mLen + 1;
            return;
        }
        scanned = mEnd - mBeg;
        if (!refill())
        {
            // end of file: whatever is left is the last line; the stream is no longer good
            mLine = mBuf.data() + mBeg; mLen = mEnd - mBeg;
            mBeg = mEnd;
            mGood = false;
            return;
        }
    }
}

void LineReader::next() { getline(); }

// --------------------------------------------------------------------------------------
// parsers
// --------------------------------------------------------------------------------------

static std::string num(uint64_t v) { return std::to_string(v); }

// A parse failure inside the shared FASTQ record loop: message text up to (not including) the
// line number, and the line number relative to the first line the loop saw (1-based).
struct FastqFail { const char* what; uint64_t line; };

// FastqParser::next (FastqParser.hh:78-176); getLine strips one trailing '\r' (:62-75).
// Src offers valid()/line()/len()/next()/offset() with PlainLineSource semantics.  Records are
// parsed while the offset of their first line is < limit.  Returns false and fills *fail on a
// parse error.  *lines = lines consumed.
template <class Src, class Sink>
bool fastqLoop(Src& src, uint64_t limit, Sink&& sink, uint64_t* reads, uint64_t* lines, FastqFail* fail)
{
    uint64_t lineNum = 1;
    std::string label, seq;
    auto cur = [&](const char*& l, size_t& n) {
        l = src.line(); n = src.len();
        if (n > 0 && l[n - 1] == '\r') --n;
    };
    auto bad = [&](const char* what) { fail->what = what; fail->line = lineNum; *lines = lineNum - 1; return false; };
    for (;;)
    {
        if (!src.valid() || src.offset() >= limit) break;
        const char* l; size_t n;
        cur(l, n);
        if (!(n > 0 && l[0] == '@')) return bad("expected '@' at beginning of line ");
        // (the title is only ever compared with a repeated one on the '+' line: a source whose lines stay where they are
        // -- a chunk of the parallel parser -- keeps a pointer; copying 100 M titles was a sixth of the framer's time)
        const char* lab = l + 1; size_t labn = n - 1;
        if (!Src::kStableLines) { label.assign(l + 1, n - 1); lab = label.data(); }
        // the common record has its sequence on one line: no copy then
        const char* seq1 = nullptr; size_t seq1n = 0;
        bool multi = false;
        seq.clear();
        for (;;)
        {
            src.next(); ++lineNum;
            if (!src.valid()) return bad("expected sequence data or quality header at line ");
            cur(l, n);
            if (n > 0 && (l[0] == '@' || l[0] == '+')) break;
            if (!Src::kStableLines && !multi) { seq.assign(l, n); multi = true; }   // line() dies at next()
            else if (!seq1 && !multi) { seq1 = l; seq1n = n; }
            else
            {
                if (!multi) { seq.assign(seq1, seq1n); multi = true; }
                seq.append(l, n);
            }
        }
        if (!(n > 0 && l[0] == '+')) return bad("expected '+' at beginning of line ");
        if (n > 1 && !(n - 1 == labn && memcmp(l + 1, lab, n - 1) == 0))
            return bad("quality title does not match sequence title at line ");
        const char* sp = multi ? seq.data() : seq1;
        const size_t sn = multi ? seq.size() : seq1n;
        size_t qual = 0;
        for (;;)
        {
            src.next(); ++lineNum;
            if (!src.valid()) break;
            cur(l, n);
            if (n > 0 && (l[0] == '@' || l[0] == '+'))
            {
                // '@' may start a quality line: only a record boundary once enough quality
                // has been seen
                if (qual >= sn) break;
            }
            qual += n;
        }
        if (sn != qual) return bad("length mistmatch between sequence and quality data just before line ");
        ++*reads;
        sink(sp ? sp : "", sn);
    }
    *lines = lineNum - 1;
    return true;
}

uint64_t parseFastq(const std::string& name, const ReadSink& sink)
{
    LineReader src(name);
    uint64_t reads = 0, lines = 0;
    FastqFail f{};
    if (!fastqLoop(src, ~0ULL, sink, &reads, &lines, &f)) throw Error::Parse(name, f.what + num(f.line));
    return reads;
}

// ---- parallel FASTQ parsing of a plain file -------------------------------------------------
//
// The file is mapped and cut into chunks; workers parse chunks concurrently with the SAME record
// loop as the serial parser.  A worker must start at a record boundary, which it guesses (a line
// starting with '@', a '+' line two lines on, equal sequence/quality lengths, '@' or end of file
// after that).  Guesses are verified, not trusted: chunk i is accepted only if it started exactly
// where chunk i-1 ended; the first mismatch makes the caller parse the rest of the file serially
// from the last verified boundary, so the framing is always the reference's.

namespace {

struct MemLines {
    static constexpr bool kStableLines = true;
    const char* p; size_t n, at; bool good; const char* l; size_t ln; size_t off;
    MemLines(const char* base, size_t size, size_t start) : p(base), n(size), at(start), good(true), l(base), ln(0), off(start) { next(); }
    bool valid() const { return good || ln; }
    const char* line() const { return l; }
    size_t len() const { return ln; }
    uint64_t offset() const { return off; }
    void next()
    {
        l = p + at; ln = 0; off = at;
        if (!good) return;
        if (at >= n) { good = false; return; }
        const char* nl = (const char*)memchr(p + at, '\n', n - at);
        if (nl) { ln = (size_t)(nl - (p + at)); at += ln + 1; }
        else { ln = n - at; at = n; good = false; }
    }
};

// First plausible record start at or after `from` (a line start), or npos.
size_t guessRecordStart(const char* p, size_t n, size_t from)
{
    size_t at = from;
    if (at > 0)
    {
        const char* nl = (const char*)memchr(p + at - 1, '\n', n - (at - 1));
        if (!nl) return (size_t)-1;
        at = (size_t)(nl - p) + 1;
    }
    for (int tries = 0; tries < 256 && at < n; ++tries)
    {
        size_t ls[5]; size_t ll[5]; size_t q = at; int got = 0;
        for (; got < 5 && q <= n; ++got)
        {
            ls[got] = q;
            if (q >= n) { ll[got] = 0; ++got; break; }
            const char* nl = (const char*)memchr(p + q, '\n', n - q);
            size_t e = nl ? (size_t)(nl - p) : n;
            ll[got] = e - q; if (ll[got] && p[e - 1] == '\r') --ll[got];
            q = nl ? e + 1 : n + 1;
        }
        if (got >= 4 && ll[0] > 0 && p[ls[0]] == '@' && ll[2] > 0 && p[ls[2]] == '+' && ll[1] == ll[3]
            && !(ll[1] > 0 && (p[ls[1]] == '@' || p[ls[1]] == '+'))
            && (got == 4 || ls[4] >= n || p[ls[4]] == '@'))
            return at;
        const char* nl = (const char*)memchr(p + at, '\n', n - at);
        if (!nl) break;
        at = (size_t)(nl - p) + 1;
    }
    return (size_t)-1;
}

// ---- 2-bit packing of parsed bases (goss_gpu_push_packed_host: one u32 of codes + one u16 of non-base flags per 16
// positions).  GossReadBaseString.hh:133-188 is the reference's per-base encoder; here the parser threads pack
// whole chunks so that 3 bits per base cross PCIe instead of 8.  Two forms with the same result: a table-driven
// loop, and an AVX2 + BMI2 one (32 positions per step) chosen at run time where the CPU has both.
void packBasesScalar(const char* p, size_t n, uint32_t* codes, uint16_t* bad)
{
    static const struct Lut { uint8_t v[256]; Lut() { memset(v, 4, 256); v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } } lut;
    const size_t groups = (n + 15) / 16;
    for (size_t g = 0; g < groups; ++g)
    {
        uint32_t c = 0, b = 0;
        const size_t lim = std::min<size_t>(16, n - g * 16);
        for (size_t j = 0; j < lim; ++j)
        {
            const uint32_t v = lut.v[(uint8_t)p[g * 16 + j]];
            c |= (v & 3u) << (2 * j);
            b |= (v >> 2) << j;
        }
        if (lim < 16) b |= 0xFFFFu << lim;               // positions behind the end are no bases
        codes[g] = c; bad[g] = (uint16_t)b;
    }
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) void packBasesAvx2(const char* p, size_t n, uint32_t* codes, uint16_t* bad)
{
    const size_t whole = n / 32;                         // steps of two groups
    const __m256i fold = _mm256_set1_epi8(0x20);
    const __m256i ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'), cg = _mm256_set1_epi8('g'), ct = _mm256_set1_epi8('t');
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
    for (size_t i = 0; i < whole; ++i)
    {
        const __m256i raw = _mm256_loadu_si256((const __m256i*)(p + 32 * i));
        const __m256i l = _mm256_or_si256(raw, fold);
        const __m256i good = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(l, ca), _mm256_cmpeq_epi8(l, cc)),
                                             _mm256_or_si256(_mm256_cmpeq_epi8(l, cg), _mm256_cmpeq_epi8(l, ct)));
        const uint32_t gm = (uint32_t)_mm256_movemask_epi8(good);
        // code = ((l >> 1) & 3) ^ (((l >> 1) & 3) >> 1): a 0 c 1 g 3->2 t 2->3
        __m256i x = _mm256_and_si256(_mm256_srli_epi16(l, 1), three);
        x = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), one));
        alignas(32) uint64_t w[4];
        _mm256_store_si256((__m256i*)w, x);
        const uint64_t m = 0x0303030303030303ULL;
        const uint32_t c0 = (uint32_t)_pext_u64(w[0], m) | ((uint32_t)_pext_u64(w[1], m) << 16);
        const uint32_t c1 = (uint32_t)_pext_u64(w[2], m) | ((uint32_t)_pext_u64(w[3], m) << 16);
        codes[2 * i] = c0; codes[2 * i + 1] = c1;
        bad[2 * i] = (uint16_t)~gm; bad[2 * i + 1] = (uint16_t)~(gm >> 16);
    }
    if (whole * 32 < n) packBasesScalar(p + whole * 32, n - whole * 32, codes + 2 * whole, bad + 2 * whole);
}
#endif

void packBases(const char* p, size_t n, uint32_t* codes, uint16_t* bad)
{
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !std::getenv("GOSS_PACK_SCALAR");
    if (fast) { packBasesAvx2(p, n, codes, bad); return; }
#endif
    packBasesScalar(p, n, codes, bad);
}

struct ChunkResult {
    char* buf = nullptr;               // from the buffer pool (pinned when a GPU is attached)
    size_t len = 0;
    uint64_t reads = 0, lines = 0;
    size_t start = (size_t)-1, end = 0;
    bool ok = false;
    FastqFail fail{};
    bool done = false;
    uint32_t* codes = nullptr;         // packed form of buf[0, len) (behind the bytes in the same buffer), when asked for
    uint16_t* bad = nullptr;
};

}  // namespace

// FastaParser::next (FastaParser.hh:51-87); no '\r' stripping; line numbers start at 0.
uint64_t parseFasta(const std::string& name, const ReadSink& sink)
{
    LineReader src(name);
    uint64_t lineNum = 0, reads = 0;
    std::string seq;
    for (;;)
    {
        if (!src.valid()) break;
        if (!(src.len() > 0 && src.line()[0] == '>'))
            throw Error::Parse(name, "expected '>' at beginning of line " + num(lineNum));
        seq.clear();
        for (;;)
        {
            src.next(); ++lineNum;
            if (!src.valid()) break;
            if (src.len() > 0 && src.line()[0] == '>') break;
            seq.append(src.line(), src.len());
        }
        ++reads;
        sink(seq.data(), seq.size());
    }
    return reads;
}

// LineParser::next (LineParser.hh:71-82): every line is a read.
uint64_t parseLines(const std::string& name, const ReadSink& sink)
{
    LineReader src(name);
    uint64_t reads = 0;
    while (src.valid())
    {
        ++reads;
        sink(src.line(), src.len());
        src.next();
    }
    return reads;
}

// --------------------------------------------------------------------------------------
// the build commands
// --------------------------------------------------------------------------------------

namespace {

struct GpuCtx {
    goss_gpu_ctx* h = nullptr;
    ~GpuCtx() { if (h) goss_gpu_destroy(h); }
    void check(int rc, const char* what)
    {
        if (rc == GOSS_OK) return;
        std::string msg = std::string(what) + ": " + goss_gpu_strerror(rc);
        const char* d = h ? goss_gpu_last_error(h) : "";
        if (d && *d) msg += std::string(" (") + d + ")";
        throw Error::General(msg + "\n");
    }
};

}  // namespace

// Every file of the emitted object to "<out><suffix>".  The calling thread (the context's) copies
// 32 MB pieces from the device into a ring of page-locked buffers; a pool of writer threads pwrites
// them, so the copies into the page cache run on several cores while the next pieces arrive.
namespace {
// What the parallel parser parks between files (HostAlloc::keep): page-locked buffers by size, and the slabs of its
// two-step pools -- plain pages locked in place.  A parked slab is idle while the object's files are written: the
// writer below borrows one for its output buffers instead of page-locking 192 MB of its own (0.3 s/GB).
std::mutex gKeptMutex;
std::map<size_t, std::vector<void*>> gKeptBuffers;          // by size
struct KeptSlab { void* p; bool pinned; };
std::map<size_t, std::vector<KeptSlab>> gKeptSlabs;         // two-step pools, by size
}  // namespace

void writeObjectFiles(goss_gpu_ctx* h, const std::string& out)
{
    writeObjectFiles(std::vector<goss_gpu_ctx*>{h}, out);
}

// The same for a group of contexts after goss_gpu_group_emit: context 0 lists every file of the object; the
// low-bits column files and "-counts.ord0" are slices -- context j's bytes follow those of the contexts
// before it; ".part.*" files are transport between the contexts and are not written.
void writeObjectFiles(const std::vector<goss_gpu_ctx*>& hs, const std::string& out)
{
    goss_gpu_ctx* h = hs.at(0);
    auto check = [&](int rc, const char* what) {
        if (rc == GOSS_OK) return;
        std::string msg = std::string(what) + ": " + goss_gpu_strerror(rc);
        const char* d = goss_gpu_last_error(h);
        if (d && *d) msg += std::string(" (") + d + ")";
        throw Error::General(msg + "\n");
    };
    constexpr size_t kPiece = 32u << 20;
    constexpr int kBuffers = 6, kWriters = 4;
    struct Job { int fd; uint64_t off; size_t n; int buf; };
    std::vector<void*> bufs(kBuffers, nullptr);
    std::vector<bool> pinned(kBuffers, false);
    std::mutex m;
    std::condition_variable cvJob, cvFree;
    std::deque<Job> jobs;
    std::vector<int> freeBufs;
    bool done = false, failed = false;
    // (a page-locked slab the parser has parked, when there is one that is large enough: borrowed, parked again at the end)
    KeptSlab borrowed{nullptr, false};
    size_t borrowedBytes = 0;
    {
        std::lock_guard<std::mutex> lk(gKeptMutex);
        for (auto& kv : gKeptSlabs)
            if (kv.first >= (size_t)kBuffers * kPiece)
                for (size_t q = 0; q < kv.second.size() && !borrowed.p; ++q)
                    if (kv.second[q].pinned) { borrowed = kv.second[q]; borrowedBytes = kv.first; kv.second.erase(kv.second.begin() + (long)q); }
    }
    for (int i = 0; i < kBuffers; ++i)
    {
        if (borrowed.p) bufs[i] = (char*)borrowed.p + (size_t)i * kPiece;
        else if (goss_gpu_host_alloc(&bufs[i], kPiece) == GOSS_OK) pinned[i] = true;
        else bufs[i] = malloc(kPiece);
        if (!bufs[i]) throw Error::General("out of host memory for the output buffers\n");
        freeBufs.push_back(i);
    }
    auto writer = [&]() {
        for (;;)
        {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m);
                cvJob.wait(lk, [&] { return !jobs.empty() || done; });
                if (jobs.empty()) return;
                j = jobs.front(); jobs.pop_front();
            }
            size_t w = 0;
            while (w < j.n)
            {
                ssize_t r = pwrite(j.fd, (const char*)bufs[j.buf] + w, j.n - w, (off_t)(j.off + w));
                if (r <= 0) { std::lock_guard<std::mutex> lk(m); failed = true; break; }
                w += (size_t)r;
            }
            { std::lock_guard<std::mutex> lk(m); freeBufs.push_back(j.buf); }
            cvFree.notify_one();
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < kWriters; ++i) pool.emplace_back(writer);
    std::vector<int> fds;
    auto finishAll = [&]() {
        { std::lock_guard<std::mutex> lk(m); done = true; }
        cvJob.notify_all();
        for (auto& t : pool) t.join();
        for (int fd : fds) if (::close(fd) != 0) failed = true;
        if (borrowed.p) { std::lock_guard<std::mutex> lk(gKeptMutex); gKeptSlabs[borrowedBytes].push_back(borrowed); }
        else for (int i = 0; i < kBuffers; ++i) { if (pinned[i]) goss_gpu_host_free(bufs[i]); else free(bufs[i]); }
    };
    try
    {
        auto isSlice = [](const std::string& n) {
            return (n.find(".low-bits") != std::string::npos && n.find(".ord") == std::string::npos) || n == "-counts.ord0";
        };
        auto listFiles = [&](goss_gpu_ctx* c) {
            std::vector<std::pair<std::string, uint64_t>> v;
            uint32_t n = 0;
            check(goss_gpu_file_count(c, &n), "listing output files");
            for (uint32_t i = 0; i < n; ++i)
            {
                char suffix[256]; uint64_t size = 0;
                check(goss_gpu_file_info(c, i, suffix, sizeof suffix, &size), "listing output files");
                v.emplace_back(suffix, size);
            }
            return v;
        };
        std::vector<std::vector<std::pair<std::string, uint64_t>>> lists;
        for (goss_gpu_ctx* c : hs) lists.push_back(listFiles(c));
        for (uint32_t i = 0; i < lists[0].size(); ++i)
        {
            const std::string& suffix = lists[0][i].first;
            if (suffix.compare(0, 6, ".part.") == 0) continue;
            int fd = ::open((out + suffix).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            if (fd < 0) throw Error::Write(out);
            fds.push_back(fd);
            uint64_t base = 0;
            for (size_t j = 0; j < hs.size(); ++j)
            {
                uint32_t idx = i;
                if (j > 0)
                {
                    if (!isSlice(suffix)) break;
                    idx = ~0u;
                    for (uint32_t q = 0; q < lists[j].size(); ++q) if (lists[j][q].first == suffix) idx = q;
                    if (idx == ~0u) throw Error::General("a context of the group lacks its slice of " + suffix + "\n");
                }
                const uint64_t size = lists[j][idx].second;
                for (uint64_t off = 0; off < size; off += kPiece)
                {
                    const size_t n = (size_t)std::min<uint64_t>(kPiece, size - off);
                    int b;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cvFree.wait(lk, [&] { return !freeBufs.empty(); });
                        b = freeBufs.back(); freeBufs.pop_back();
                    }
                    check(goss_gpu_file_read(hs[j], idx, off, bufs[b], n), "reading device file");
                    { std::lock_guard<std::mutex> lk(m); jobs.push_back(Job{fd, base + off, n, b}); }
                    cvJob.notify_one();
                }
                base += size;
            }
        }
    }
    catch (...)
    {
        finishAll();
        throw;
    }
    finishAll();
    if (failed) throw Error::Write(out);
}

namespace {

// Bytes of file per parser work item (GOSS_PARSE_CHUNK overrides: tests use small chunks).
size_t parseChunkBytes()
{
    const char* e = std::getenv("GOSS_PARSE_CHUNK");
    if (e && *e) { long v = atol(e); if (v >= 256) return (size_t)v; }
    return 8u << 20;      // small enough that page-locking the buffer pool (threads + 4 buffers of half this size) takes ~0.1 s
}

// worker threads of the parallel FASTQ parser, whatever -T says beyond it (GOSS_PARSE_MAX_THREADS overrides: experiments)
uint64_t maxFramers()
{
    static const uint64_t n = [] { const char* e = std::getenv("GOSS_PARSE_MAX_THREADS"); const long v = e ? atol(e) : 0; return v >= 1 ? (uint64_t)v : (uint64_t)32; }();
    return n;
}

// Bytes read behind a chunk for the record that crosses its end (GOSS_PARSE_SLACK overrides: the tests make it tiny so
// that the re-reads with a larger window happen).
size_t parseSlackBytes()
{
    const char* e = std::getenv("GOSS_PARSE_SLACK");
    if (e && *e) { long v = atol(e); if (v >= 1) return (size_t)v; }
    return 1u << 20;
}

// Parse a plain FASTQ file with `threads` workers (see "parallel FASTQ parsing" above).  Batches of
// bases go to `push` in file order.  Returns the number of reads, or ~0 if the file is not eligible
// (too small, not a regular file) and the caller should use the serial parser.
struct HostAlloc {                       // how the chunk buffers are obtained (pinned under a GPU)
    std::function<void*(size_t)> alloc;
    std::function<void(void*)> release;
    // keep: buffers are not given back to the system when a file is done but kept for the next one (page-locked memory
    // costs ~0.3 s/GB to obtain and milliseconds per buffer to return: a build ends faster without the returns)
    bool keep = false;
    // Two-step form (pin set): `alloc` is asked ONCE, for plain pages for the whole pool -- the workers parse into them
    // while the device runtime is still starting -- and the pool is page-locked in place later, in one call, by the
    // parser's allocator thread, once `ready` has returned true (it blocks until the runtime is up; false: it will not
    // come up).  What is pushed before that is copied through the driver's own bounce buffers.
    std::function<bool()> ready;
    std::function<bool(void*, size_t)> pin;
    std::function<void(void*)> unpin;
};

}  // namespace

// What the parser parked between files (HostAlloc::keep: only the page-locked allocator of the build commands parks
// anything -- its one-step buffers come from goss_gpu_host_alloc, its two-step slabs are plain pages registered in
// place) is given back on an orderly exit: registered pages unregistered before they are freed.  `goss` ends with
// _exit once its files are closed and never gets here; GOSS_FULL_EXIT=1 and runs under a profiler do.
void releaseKeptBuffers()
{
    std::lock_guard<std::mutex> lk(gKeptMutex);
    for (auto& kv : gKeptSlabs)
        for (KeptSlab& s : kv.second) { if (s.pinned) goss_gpu_host_unregister(s.p); std::free(s.p); }
    gKeptSlabs.clear();
    for (auto& kv : gKeptBuffers)
        for (void* b : kv.second) goss_gpu_host_free(b);
    gKeptBuffers.clear();
}

namespace {

// pushOwned (optional): the consumer keeps the buffer until it calls `release` -- several devices then copy
// from several buffers at once; without it `push` returns when the bytes are on their way.
typedef std::function<void(const char*, size_t, std::function<void()>)> OwnedPush;
// pushPacked (optional, one device): the workers also pack their bases (packBases) and the consumer hands the packed
// arrays over without waiting -- `release` comes back later, on the consumer's own thread from inside a library call
// (goss_gpu_push_packed_host_async); `drain` makes every outstanding release happen (goss_gpu_flush).
struct PackedPush {
    std::function<void(const uint32_t*, const uint16_t*, size_t, std::function<void()>)> push;
    std::function<void()> drain;
};

uint64_t parseFastqParallel(const std::string& name, unsigned threads, size_t chunkBytes,
                            const std::function<void(const char*, size_t)>& push, const HostAlloc& ha,
                            const OwnedPush* pushOwned = nullptr, const PackedPush* pushPacked = nullptr)
{
    // (compressed inputs are framed by the serial parser behind InFile's decompressor)
    if (threads < 2 || name == "-" || endsWith(name, ".gz") || endsWith(name, ".bz2")) return ~0ULL;
    const auto tEnter = std::chrono::steady_clock::now();
    struct Whole { std::chrono::steady_clock::time_point t0; bool on;
                   ~Whole() { if (on) std::fprintf(stderr, "goss: parallel parser: %.3f s from entry to return\n",
                                                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } }
        whole{tEnter, std::getenv("GOSS_PARSE_STATS") != nullptr};
    int fd = ::open(name.c_str(), O_RDONLY);
    if (fd < 0) throw Error::Errno(name, errno);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < 2 * chunkBytes) { ::close(fd); return ~0ULL; }
    const size_t size = (size_t)st.st_size;
    // (the mapping serves the serial path below, should a chunk boundary not line up: untouched otherwise)
    void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    const int rfd = fd;
    struct CloseFd { int f; ~CloseFd() { ::close(f); } } closeFd{fd};
    if (map == MAP_FAILED) return ~0ULL;
    madvise(map, size, MADV_SEQUENTIAL);
    const char* p = (const char*)map;
    struct Unmap { void* m; size_t n; ~Unmap() { munmap(m, n); } } unmap{map, size};

    const size_t nchunks = (size + chunkBytes - 1) / chunkBytes;
    std::vector<ChunkResult> res(nchunks);
    std::unique_ptr<std::atomic<uint8_t>[]> doneFlag(new std::atomic<uint8_t>[nchunks]);          // (what the consumer spins on before it sleeps)
    for (size_t i = 0; i < nchunks; ++i) doneFlag[i].store(0, std::memory_order_relaxed);
    std::atomic<size_t> nextChunk{0};
    std::atomic<bool> abortAll{false};
    std::mutex m;
    std::condition_variable cv;
    // a bounded pool of reusable output buffers (a chunk of file yields at most as many bases):
    // bounds the memory in flight and keeps the pages warm / pinned for the copy to the device
    // (the bases are at most half of a FASTQ chunk's bytes: as many quality values as bases, plus
    // titles and separators).  Page-locking memory costs ~0.3 s/GB, so the pool is filled by a thread
    // of its own while the workers already parse: one buffer after the other (one-step allocators: a
    // short file never pays for buffers it does not use), or as ONE slab of plain pages that is handed
    // out at once and page-locked in place when the device runtime is up (two-step allocators).
    const size_t bufCap = chunkBytes / 2 + (1u << 16);
    // (packed pushes: codes and flags behind the bytes -- 6 bytes per 16 positions)
    // Packed pushes: a pool buffer holds only what travels -- codes, then flags: 6 bytes per 16 positions -- and the
    // bytes a chunk's bases are framed into before they are packed are the worker's own.  The same page-locked memory
    // then makes four times as many buffers, and it takes many: the consumer takes the chunks in file order, a worker
    // that is held up (the box is shared) holds up everything behind it, and the others need buffers to go on meanwhile
    // (with 32 + 32 buffers of bytes + codes the workers waited for a buffer as long as they worked).
    const size_t codesBytes = (bufCap / 16 + 2) * 4;
    const size_t bufBytes = pushPacked ? codesBytes + (bufCap / 16 + 2) * 2 + 64 : bufCap;
    // (packed pushes keep their buffer until its copy has completed, and the consumer is away for ~25 ms whenever the
    // staging buffer is counted: more buffers than workers, so that the workers go on meanwhile)
    size_t extraBufs = pushPacked ? 160 : 4;
    if (const char* e = std::getenv("GOSS_PARSE_POOL")) { const long v = atol(e); if (v >= 1) extraBufs = (size_t)v; }
    // (Tried in round 6 and not kept: "run-ahead" buffers behind the locked ones, never page-locked, for the workers to parse
    // into while the device runtime is still starting -- the pageable copies out of them held the pusher thread 0.55 s
    // instead of 0.22 and their first-touch page faults doubled the packing time: 0.62-0.68 s became 0.77-0.83,
    // profiles/r06/e2e_steps.txt.)
    const size_t nbuf = std::min<size_t>((size_t)threads + extraBufs, nchunks + 1);
    // Two condition variables on the one mutex: workers wait for a free buffer (cvFree, one of them woken per buffer that
    // comes back), the in-order consumer for its next chunk (cvDone, woken by the worker that finishes a chunk).  With
    // one for both every event woke all 33 threads, and the workers spent more time waiting than working.
    std::condition_variable cvFree;
    std::condition_variable& cvDone = cv;
    auto wakeAll = [&]() { cvFree.notify_all(); cvDone.notify_all(); };
    std::vector<char*> freeBufs;
    std::vector<void*> allBufs;
    KeptSlab slab{nullptr, false};
    const size_t stride = (bufBytes + 4095) & ~(size_t)4095, slabBytes = stride * nbuf;
    auto putFree = [&](char* b) { freeBufs.push_back(b); };          // (under `m`)
    std::atomic<bool> allocFailed{false};
    struct FreeAll { std::vector<void*>& v; const HostAlloc& h; size_t bytes; KeptSlab& slab; size_t slabBytes;
                     ~FreeAll() {
                         if (slab.p)
                         {
                             if (h.keep) { std::lock_guard<std::mutex> lk(gKeptMutex); gKeptSlabs[slabBytes].push_back(slab); }
                             else { if (slab.pinned) h.unpin(slab.p); h.release(slab.p); }
                         }
                         if (h.keep) { std::lock_guard<std::mutex> lk(gKeptMutex); auto& k = gKeptBuffers[bytes]; k.insert(k.end(), v.begin(), v.end()); }
                         else for (void* b : v) h.release(b);
                     } } freeAll{allBufs, ha, bufBytes, slab, slabBytes};
    std::thread allocator([&]() {
        auto fail = [&]() {
            // (under the lock: a worker or the consumer between its predicate and its wait must not miss this)
            { std::lock_guard<std::mutex> lk(m); allocFailed.store(true); abortAll.store(true); }
            wakeAll();
        };
        if (ha.pin)
        {
            // two-step form: the whole pool at once, plain; page-locked when the device runtime is up
            KeptSlab got{nullptr, false};
            if (ha.keep)
            {
                std::lock_guard<std::mutex> lk(gKeptMutex);
                auto it = gKeptSlabs.find(slabBytes);
                if (it != gKeptSlabs.end() && !it->second.empty()) { got = it->second.back(); it->second.pop_back(); }
            }
            if (!got.p) got.p = ha.alloc(slabBytes);
            if (!got.p) { fail(); return; }
            {
                std::lock_guard<std::mutex> lk(m);
                slab = got;
                for (size_t i = 0; i < nbuf; ++i) freeBufs.push_back((char*)got.p + i * stride);
            }
            cvFree.notify_all();
            if (!got.pinned && ha.ready && ha.ready() && !abortAll.load() && nextChunk.load() < nchunks)
            {
                const bool ok = ha.pin(got.p, slabBytes);
                std::lock_guard<std::mutex> lk(m);
                slab.pinned = ok;
            }
            return;
        }
        for (size_t i = 0; i < nbuf && !abortAll.load() && nextChunk.load() < nchunks; ++i)
        {
            // (GOSS_TEST_FAIL_PARSER_ALLOC: fault injection for the test of this path -- the second buffer cannot be had)
            void* b = nullptr;
            if (ha.keep)
            {
                std::lock_guard<std::mutex> lk(gKeptMutex);
                auto it = gKeptBuffers.find(bufBytes);
                if (it != gKeptBuffers.end() && !it->second.empty()) { b = it->second.back(); it->second.pop_back(); }
            }
            if (!b) b = (i >= 1 && std::getenv("GOSS_TEST_FAIL_PARSER_ALLOC")) ? nullptr : ha.alloc(bufBytes);
            if (!b) { fail(); return; }
            { std::lock_guard<std::mutex> lk(m); allBufs.push_back(b); freeBufs.push_back((char*)b); }
            cvFree.notify_one();
        }
    });
    struct JoinAlloc { std::thread& t; ~JoinAlloc() { if (t.joinable()) t.join(); } } joinAlloc{allocator};
    // buffers handed to pushOwned and not yet released: nothing here may be torn down before they are back
    size_t lent = 0;
    struct WaitLent { std::mutex& m; std::condition_variable& cv; size_t& lent;
                      ~WaitLent() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return lent == 0; }); } } waitLent{m, cvFree, lent};
    // (packed pushes give their buffers back from inside library calls of THIS thread: make them all happen before waiting)
    struct Drain { const PackedPush* p; ~Drain() { if (p && p->drain) { try { p->drain(); } catch (...) {} } } } drainLent{pushPacked};

    // GOSS_PARSE_STATS=1: where the workers' time goes, summed over all of them (ns)
    const bool wstats = std::getenv("GOSS_PARSE_STATS") != nullptr;
    // How a worker gets at its chunk's bytes.  Reads (pread into a buffer of its own) are the faster way through pages
    // that have been read before: 55-63 GB/s for the 32 workers against 40 through page faults on the mapping.  A tmpfs
    // file that was JUST WRITTEN is another matter: the kernel's read path hands out its new pages at 13-15 GB/s
    // whatever the number of readers (43 s of system time for 31.5 GB), page faults on the mapping at 34-40 GB/s
    // (profiles/r05/cold_threads.txt, cold_mmap.txt: first pass 2.8 s against 0.8-0.9).  So the chunks are read, every
    // worker says how fast its reads went (not its first), and as soon as three of four of sixteen reads in a row ran
    // below 1.5 GB/s on a tmpfs file the rest of the file is framed where it is mapped.  (Files on disks stay with reads: page faults there mean
    // many small requests.)  GOSS_PARSE_MMAP=1 / 0: the mapping / reads whatever the file.
    std::atomic<int> readMode{0};            // 0: reads, undecided; 1: the mapping; 2: reads, decided
    std::atomic<uint32_t> slowReads{0}, timedReads{0}, slowSeen{0};
    {
        struct statfs sfs;
        const bool tmpfs = fstatfs(rfd, &sfs) == 0 && (unsigned long)sfs.f_type == 0x01021994UL;          // TMPFS_MAGIC
        if (!tmpfs) readMode.store(2);
        if (const char* e = std::getenv("GOSS_PARSE_MMAP")) readMode.store(*e == '1' ? 1 : 2);
    }
    std::atomic<uint64_t> wBufNs{0}, wReadNs{0}, wParseNs{0}, wPackNs{0};
    auto wnow = [] { return std::chrono::steady_clock::now(); };
    auto wadd = [&](std::atomic<uint64_t>& a, std::chrono::steady_clock::time_point t) {
        if (wstats) a.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(wnow() - t).count(), std::memory_order_relaxed);
    };
    auto worker = [&]() {
        std::vector<char> raw;                  // the chunk's bytes as read from the file
        std::vector<char> bytes;                // (packed pushes) the chunk's bases before they are packed
        if (pushPacked) bytes.resize(bufCap);
        bool firstRead = true;
        for (;;)
        {
            ChunkResult r;
            size_t i;
            char* mybuf;
            const auto tBuf = wnow();
            {
                // buffer first, chunk number second (both under the lock): every outstanding chunk
                // then owns a buffer and the in-order consumer can always make progress
                std::unique_lock<std::mutex> lk(m);
                cvFree.wait(lk, [&] { return abortAll.load() || !freeBufs.empty(); });
                if (abortAll.load()) return;
                i = nextChunk.fetch_add(1);
                if (i >= nchunks) return;
                mybuf = freeBufs.back(); freeBufs.pop_back();
            }
            wadd(wBufNs, tBuf);
            const size_t begin = i * chunkBytes, limit = std::min(size, begin + chunkBytes);
            // The chunk's bytes are READ into a buffer of the worker's own (pread), with room behind `limit` for the
            // record that crosses it: 64 threads faulting the pages of one mapping in serialise on the address space's
            // lock (24 GB/s for the whole pool on the bench's box); copies out of the page cache do not.  A record
            // that runs past the room is read again with eight times as much.
            const size_t lo = begin ? begin - 1 : 0;              // (guessRecordStart looks at the byte in front)
            for (size_t slack = parseSlackBytes();; slack *= 8)
            {
                const size_t hi = std::min(size, limit + slack);

                size_t got = 0;
                const auto tRead = wnow();
                const int mode = readMode.load(std::memory_order_relaxed);
                const bool fromMapping = mode == 1;
                if (fromMapping) got = hi - lo;          // (the chunk is framed where the file is mapped)
                else
                {
                    if (raw.size() < hi - lo) raw.resize(hi - lo);
                    while (got < hi - lo)
                    {
                        const ssize_t k = pread(rfd, raw.data() + got, hi - lo - got, (off_t)(lo + got));
                        if (k < 0) { if (errno == EINTR) continue; break; }
                        if (k == 0) break;
                        got += (size_t)k;
                    }
                    // (a worker's first read also faults in the pages of its buffer: its later ones are timed)
                    const bool timeIt = mode == 0 && !firstRead && slack == parseSlackBytes() && got >= (1u << 20);
                    firstRead = false;
                    if (timeIt)
                    {
                        // (every sixteen timed reads: twelve or more below 1.5 GB/s -> the mapping from here on.  The reads
                        // are watched to the end of the file: pages that were last touched through a mapping read fast
                        // for a few dozen chunks and then as slowly as new ones)
                        const double secs = std::chrono::duration<double>(wnow() - tRead).count();
                        if ((double)got < 1.5e9 * secs) slowReads.fetch_add(1);
                        const uint32_t n = timedReads.fetch_add(1) + 1;
                        if (n % 16 == 0)
                        {
                            const uint32_t slow = slowReads.exchange(0);
                            slowSeen.fetch_add(slow);
                            if (slow >= 12) { int expect = 0; readMode.compare_exchange_strong(expect, 1); }
                        }
                    }
                }
                wadd(wReadNs, tRead);
                const auto tParse = wnow();
                struct ParseDone { decltype(wadd)& add; std::atomic<uint64_t>& a; std::chrono::steady_clock::time_point t;
                                   ~ParseDone() { add(a, t); } } parseDone{wadd, wParseNs, tParse};
                const char* lp = fromMapping ? p + lo : raw.data();
                r = ChunkResult{};
                r.buf = mybuf;
                if (got != hi - lo) { r.start = begin; r.len = bufCap + 1; r.ok = true; break; }      // (a short read: the serial path reports it)
                const size_t s0 = i == 0 ? 0 : guessRecordStart(lp, got, begin - lo);
                if (s0 == (size_t)-1 || lo + s0 >= limit)
                {
                    // no record start in the chunk (fine when the record before runs past it) -- unless the window was short
                    if (s0 == (size_t)-1 && hi < size && slack < size) continue;
                    r.end = r.start; r.ok = true; r.start = (size_t)-1;
                    break;
                }
                r.start = lo + s0;
                MemLines src(lp, got, s0);
                char* const to = pushPacked ? bytes.data() : r.buf;
                auto sink = [&](const char* seq, size_t len) {
                    // a chunk's reads are shorter than the chunk's bytes (titles, '+', qualities)
                    if (r.len + len + 1 <= bufCap) { memcpy(to + r.len, seq, len); r.len += len; to[r.len++] = '\n'; }
                    else r.len = bufCap + 1;                 // cannot happen for a record-aligned chunk
                };
                r.ok = fastqLoop(src, limit - lo, sink, &r.reads, &r.lines, &r.fail);
                // the parse must have ended inside the window (or at the end of the file): else a record was cut off
                if (!src.good && hi < size) continue;
                r.end = src.valid() ? lo + (size_t)src.offset() : size;
                break;
            }
            if (pushPacked && r.ok && r.start != (size_t)-1 && r.len && r.len <= bufCap)
            {
                // (flags right behind the codes: the library then moves both with one copy)
                r.codes = (uint32_t*)r.buf;
                r.bad = (uint16_t*)(r.codes + (r.len + 15) / 16);
                const auto tPack = wnow();
                packBases(bytes.data(), r.len, r.codes, r.bad);
                wadd(wPackNs, tPack);
            }
            // (a chunk framed from the mapping: its page-table entries go now, by this worker -- left to the munmap at
            // the end, one thread takes a quarter of a second for 31.5 GB of them)
            if (readMode.load(std::memory_order_relaxed) == 1 && limit > begin + 2 * 4096)
            {
                const size_t a = (begin + 4095) & ~(size_t)4095, e = limit & ~(size_t)4095;
                if (e > a) madvise((void*)(p + a), e - a, MADV_DONTNEED);
            }
            r.done = true;
            {
                std::lock_guard<std::mutex> lk(m);
                res[i] = r;
            }
            doneFlag[i].store(1, std::memory_order_release);
            cvDone.notify_one();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker);
    struct Join { std::vector<std::thread>& p; std::atomic<bool>& a; std::condition_variable& c; std::condition_variable& c2; std::mutex& m;
                  ~Join() { { std::lock_guard<std::mutex> lk(m); a.store(true); } c.notify_all(); c2.notify_all(); for (auto& t : p) if (t.joinable()) t.join(); } }
        join{pool, abortAll, cv, cvFree, m};

    uint64_t reads = 0, baseLine = 1;
    size_t expected = 0;
    bool serialRest = false;
    double waitSeconds = 0, pushSeconds = 0;         // GOSS_PARSE_STATS=1: where the in-order consumer spends its time
    uint64_t drains = 0;                             // ... and how often it had to ask the library for its buffers (all lent, none free)
    const bool stats = std::getenv("GOSS_PARSE_STATS") != nullptr;
    struct Report { bool on; double& w; double& p; uint64_t& d; const std::string& n; std::chrono::steady_clock::time_point t0;
                    ~Report() { if (on) std::fprintf(stderr, "goss: %s: consumer waited %.3f s for parsed chunks, %.3f s in pushes; loop started %.3f s after entry, ended at %.3f s; every buffer lent and none free: %llu times\n",
                                                     n.c_str(), w, p, start, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), (unsigned long long)d); }
                    double start; }
        report{stats, waitSeconds, pushSeconds, drains, name, tEnter, std::chrono::duration<double>(std::chrono::steady_clock::now() - tEnter).count()};
    struct WorkerReport { bool on; unsigned th; std::atomic<uint64_t>& b; std::atomic<uint64_t>& r; std::atomic<uint64_t>& p; std::atomic<uint64_t>& k;
                          ~WorkerReport() { if (on) std::fprintf(stderr, "goss: parser workers (%u), seconds summed over them: waiting for a buffer %.3f, reading the file %.3f, framing %.3f, packing %.3f\n",
                                                                 th, b.load() * 1e-9, r.load() * 1e-9, p.load() * 1e-9, k.load() * 1e-9); } }
        workerReport{stats, threads, wBufNs, wReadNs, wParseNs, wPackNs};
    struct ModeReport { bool on; std::atomic<int>& m; std::atomic<uint32_t>& slow; std::atomic<uint32_t>& timed;
                        ~ModeReport() { if (on) std::fprintf(stderr, "goss: parser: chunks %s (%u of %u timed reads below 1.5 GB/s)\n",
                                                             m.load() == 1 ? "framed where the file is mapped at the end" : "read into the workers' buffers", slow.load(), timed.load()); } }
        modeReport{stats, readMode, slowSeen, timedReads};
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (size_t i = 0; i < nchunks && !serialRest; ++i)
    {
        ChunkResult r;
        {
            const auto a = now();
            // (a chunk that is about to be done is not worth a sleep: a look every few dozen nanoseconds for ~50 us first)
            for (int spin = 0; spin < 1500 && !doneFlag[i].load(std::memory_order_acquire); ++spin)
            {
#if defined(__x86_64__)
                _mm_pause();
#endif
            }
            std::unique_lock<std::mutex> lk(m);
            // Buffers lent to the library (packed pushes) come back only from inside a library call of THIS thread.  When
            // all of them are lent -- a burst of parsed chunks pushed back to back, their copies still queued -- the
            // workers wait for a buffer, chunk i is nobody's yet, and waiting here without calling the library would
            // wait for ever (one build in four hung that way once the device side had got faster): while buffers are
            // out and none is free, the wait is bounded and the library is asked to hand them back.
#ifdef GOSS_TSAN_BUILD
            // (tools/host_tsan.sh: gcc 11's ThreadSanitizer does not know pthread_cond_clockwait -- behind a wait_for its
            //  picture of the mutex is wrong and every later report with it; the same wait by sleeping)
            auto waitDone = [&] {
                if (res[i].done || allocFailed.load()) return true;
                lk.unlock();
                std::this_thread::sleep_for(std::chrono::microseconds(200));
                lk.lock();
                return res[i].done || allocFailed.load();
            };
            while (!waitDone())
#else
            while (!cvDone.wait_for(lk, std::chrono::milliseconds(2), [&] { return res[i].done || allocFailed.load(); }))
#endif
            {
                if (lent > 0 && freeBufs.empty() && pushPacked && pushPacked->drain)
                {
                    lk.unlock();
                    pushPacked->drain();
                    ++drains;
                    lk.lock();
                }
            }
            // no buffer could be had: the workers have left and nobody will parse this chunk
            if (!res[i].done) throw Error::General("cannot allocate parser buffers\n");
            r = res[i];
            waitSeconds += std::chrono::duration<double>(now() - a).count();
        }
        const size_t limit = std::min(size, (i + 1) * chunkBytes);
        if (r.start == (size_t)-1)
        {
            // the chunk found no record start: fine only if the previous record ran past it
            { std::lock_guard<std::mutex> lk(m); putFree(r.buf); }
            cvFree.notify_one();
            if (expected < limit) serialRest = true;
            continue;
        }
        auto giveBack = [&]() {
            if (!r.buf) return;
            { std::lock_guard<std::mutex> lk(m); putFree(r.buf); }
            r.buf = nullptr;
            cvFree.notify_one();
        };
        if (r.start != expected || r.len > bufCap) { giveBack(); serialRest = true; break; }
        if (!r.ok) throw Error::Parse(name, r.fail.what + num(baseLine + r.fail.line - 1));
        if (r.len && pushPacked && r.codes)
        {
            char* b = r.buf;
            r.buf = nullptr;
            { std::lock_guard<std::mutex> lk(m); ++lent; }
            const auto a = now();
            pushPacked->push(r.codes, r.bad, r.len, [&m, &cvFree, &putFree, &lent, b]() {
                bool last;
                { std::lock_guard<std::mutex> lk(m); putFree(b); last = --lent == 0; }
                if (last) cvFree.notify_all(); else cvFree.notify_one();          // (all: whoever waits for `lent` to reach zero is among them)
            });
            pushSeconds += std::chrono::duration<double>(now() - a).count();
        }
        else if (r.len && pushOwned)
        {
            char* b = r.buf;
            r.buf = nullptr;
            { std::lock_guard<std::mutex> lk(m); ++lent; }
            (*pushOwned)(b, r.len, [&m, &cvFree, &putFree, &lent, b]() {
                bool last;
                { std::lock_guard<std::mutex> lk(m); putFree(b); last = --lent == 0; }
                if (last) cvFree.notify_all(); else cvFree.notify_one();
            });
        }
        else if (r.len)
        {
            const auto a = now();
            push(r.buf, r.len);
            pushSeconds += std::chrono::duration<double>(now() - a).count();
        }
        giveBack();
        reads += r.reads;
        baseLine += r.lines;
        expected = r.end;
    }
    if (allocFailed.load()) throw Error::General("cannot allocate parser buffers\n");
    if (pushPacked && pushPacked->drain) pushPacked->drain();
    if (serialRest || expected < size)
    {
        // a boundary guess did not line up (wrapped records, '@' starting quality lines, ...):
        // the rest of the file is framed serially from the last verified record boundary
        { std::lock_guard<std::mutex> lk(m); abortAll.store(true); }
        wakeAll();
        std::vector<char> batch;
        batch.reserve(chunkBytes);
        MemLines src(p, size, expected);
        auto sink = [&](const char* seq, size_t len) {
            batch.insert(batch.end(), seq, seq + len);
            batch.push_back('\n');
            if (batch.size() >= chunkBytes) { push(batch.data(), batch.size()); batch.clear(); }
        };
        uint64_t lines = 0; FastqFail f{};
        if (!fastqLoop(src, ~0ULL, sink, &reads, &lines, &f)) throw Error::Parse(name, f.what + num(baseLine + f.line - 1));
        if (!batch.empty()) push(batch.data(), batch.size());
    }
    return reads;
}

void runBuild(const GossCmdContext& cxt, uint64_t K, int mode, const std::string& out, const strings& fastas,
              const strings& fastqs, const strings& lines, uint64_t threads, BuildStats& stats)
{
    auto t0 = std::chrono::steady_clock::now();
    Logger& log = cxt.log;
    const uint64_t maxK = mode == GOSS_MODE_GRAPH ? 62 : 63;
    if (K > maxK || K == 0)
        throw Error::General("unable to build a graph with k=" + num(K));      // KmerSet.hh:89-95, Graph.cc:152-158

    // HBM budget: mapping the whole device takes seconds, so size the arena from the input when
    // the caller gave no budget: ~1 base per FASTQ byte/2 (FASTA, lines: per byte), two key
    // buffers + tables per key, compressed input expands ~4x
    uint64_t budget = cxt.hbmBudget;
    uint64_t inputBases = 0;          // bases of all inputs, estimated from the files' sizes (0: unknown) -- goss_gpu_expect_bases
    {
        uint64_t bases = 0; bool unknown = false;
        auto add = [&](const std::string& f, double basesPerByte) {
            struct stat st;
            if (f == "-" || ::stat(f.c_str(), &st) != 0) { unknown = true; return; }
            double b = (double)st.st_size * basesPerByte;
            if (endsWith(f, ".gz")) b *= 4.5;
            if (endsWith(f, ".bz2")) b *= 5.0;
            bases += (uint64_t)b;
        };
        for (auto& f : lines) add(f, 1.0);
        for (auto& f : fastas) add(f, 1.0);
        for (auto& f : fastqs) add(f, 0.5);
        if (!unknown) inputBases = bases;
        if (!unknown && budget == 0)
        {
            const uint64_t keyBytes = (2 * (K + (mode == GOSS_MODE_GRAPH ? 1 : 0)) <= 62) ? 8 : 16;
            const uint64_t perBase = (mode == GOSS_MODE_GRAPH ? 2 : 1) * (2 * keyBytes + 2) + 1;
            // (with --devices every context sees its share of the input)
            budget = bases / std::max<size_t>(1, cxt.devices.size()) * perBase + (6ULL << 30);      // the library clamps to the free memory
        }
        // Mapping HBM costs up to 30 ms/GB when the driver has to clear pages a previous process
        // left behind (measured: 264 GB in 7.7 s, or 0.2 s when clean), while counting in
        // ~2 G-window chunks that overlap the parser costs nothing measurable (100 M reads:
        // 1.5 s with 32 GB and with 264 GB): stay small unless told otherwise.
        const bool wide = 2 * (K + (mode == GOSS_MODE_GRAPH ? 1 : 0)) > 62;      // two-word keys
        // (24 GB start as fast as 48 GB on clean pages -- C2 from FASTQ 2.05 s against 2.1 s, 20 M reads 0.72 s
        // against 0.77 s -- and cost half as much on used ones; 12 GB is slower: 2.8 s, the chunks get too small)
        uint64_t kDefaultCap = (wide ? 48ULL : 24ULL) << 30;
        if (const char* e = std::getenv("GOSS_ARENA_START_GB")) { const long v = atol(e); if (v >= 1) kDefaultCap = (uint64_t)v << 30; }
        if (cxt.hbmBudget == 0 && (budget == 0 || budget > kDefaultCap)) budget = kDefaultCap;
    }
    // One context per device.  With several (--devices) every context has a feeder thread with a short queue:
    // batches go round the devices, each device copies and counts its own while the others do the same.
    std::vector<int> devs = cxt.devices.empty() ? std::vector<int>{cxt.device} : cxt.devices;
    const size_t P = devs.size();
    std::vector<std::unique_ptr<GpuCtx>> gs;
    for (size_t d = 0; d < P; ++d) gs.emplace_back(new GpuCtx);
    // The contexts are created by a thread of their own: loading the device runtime and its code takes 0.07 to 0.2 s, and
    // with one device the parser's workers read and frame the first chunks meanwhile (into plain pages that are
    // page-locked once the runtime is up: HostAlloc's two-step form).  Whoever needs a context waits for it (needCtx).
    struct CtxState { std::mutex m; std::condition_variable cv; bool done = false; std::exception_ptr err; double readyAt = 0; } cs;
    std::thread ctxThread([&]() {
        std::exception_ptr err;
        try
        {
            for (size_t d = 0; d < P; ++d)
            {
                GpuCtx& g = *gs[d];
                g.check(goss_gpu_create(&g.h, devs[d], (uint32_t)K, mode, budget, nullptr), "creating the GPU context");
                // a budget the user did not ask for is a starting size: inputs with little duplication (a
                // genome in FASTA: every k-mer once) need room for runs that do not shrink
                if (cxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
                // what the build will push in all (every context its share): the first chunks of a build of many then
                // choose the key space they count in for the whole build (goss_gpu_expect_bases)
                if (inputBases) g.check(goss_gpu_expect_bases(g.h, inputBases / P), "announcing the input's size");
                // the arena is mapped while the first buffers are read and parsed
                g.check(goss_gpu_prepare(g.h), "mapping HBM");
            }
        }
        catch (...) { err = std::current_exception(); }
        { std::lock_guard<std::mutex> lk(cs.m); cs.done = true; cs.err = err; cs.readyAt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
        cs.cv.notify_all();
    });
    struct JoinCtx { std::thread& t; ~JoinCtx() { if (t.joinable()) t.join(); } } joinCtx{ctxThread};          // (before `gs` goes)
    std::atomic<bool> ctxSeen{false};
    auto needCtx = [&]() {
        if (ctxSeen.load(std::memory_order_acquire)) return;
        std::unique_lock<std::mutex> lk(cs.m);
        cs.cv.wait(lk, [&] { return cs.done; });
        if (cs.err) std::rethrow_exception(cs.err);
        ctxSeen.store(true, std::memory_order_release);
    };
    auto ctxUp = [&]() -> bool {              // (for the parser's allocator thread: blocks; false = no context will come)
        std::unique_lock<std::mutex> lk(cs.m);
        cs.cv.wait(lk, [&] { return cs.done; });
        return !cs.err;
    };
    // (several devices: feeders, staging sizes and the exchange all start from the contexts)
    const bool beside = P == 1 && !std::getenv("GOSS_CONTEXT_FIRST");
    if (!beside) needCtx();
    GpuCtx& g = *gs[0];
    struct Feeder {
        struct Job { const char* p; size_t n; std::function<void()> done; };
        std::mutex m;
        std::condition_variable cv;
        std::deque<Job> q;
        bool stop = false, busy = false;
        std::string error;
        std::thread t;
    };
    std::vector<std::unique_ptr<Feeder>> feeders;
    std::atomic<bool> feedFailed{false};
    const bool fed = P > 1;           // (one device: no feeders -- the parallel parser's packed chunks go through a pusher thread of their own, see below)
    if (fed)
        for (size_t d = 0; d < P; ++d)
        {
            feeders.emplace_back(new Feeder);
            Feeder* f = feeders.back().get();
            goss_gpu_ctx* h = gs[d]->h;
            f->t = std::thread([f, h, &feedFailed]() {
                for (;;)
                {
                    Feeder::Job j;
                    {
                        std::unique_lock<std::mutex> lk(f->m);
                        f->cv.wait(lk, [&] { return f->stop || !f->q.empty(); });
                        if (f->q.empty()) return;
                        j = std::move(f->q.front()); f->q.pop_front();
                        f->busy = true;
                    }
                    if (f->error.empty())
                    {
                        const int rc = goss_gpu_push_bases_host(h, j.p, j.n);
                        if (rc != GOSS_OK)
                        {
                            std::lock_guard<std::mutex> lk(f->m);
                            f->error = std::string("counting k-mers: ") + goss_gpu_strerror(rc) + " (" + goss_gpu_last_error(h) + ")";
                            feedFailed.store(true);
                        }
                    }
                    if (j.done) j.done();
                    { std::lock_guard<std::mutex> lk(f->m); f->busy = false; }
                    f->cv.notify_all();
                }
            });
        }
    struct StopFeeders { std::vector<std::unique_ptr<Feeder>>& fs;
                         ~StopFeeders() { for (auto& f : fs) { { std::lock_guard<std::mutex> lk(f->m); f->stop = true; } f->cv.notify_all(); if (f->t.joinable()) f->t.join(); } } } stopFeeders{feeders};
    size_t nextDev = 0;
    auto feedError = [&]() {
        for (auto& f : feeders) { std::lock_guard<std::mutex> lk(f->m); if (!f->error.empty()) throw Error::General(f->error + "\n"); }
    };
    // Several devices: the exchange BEFORE counting (goss_gpu_group_route_exchange) from four devices on -- every
    // context only stages its batches; when a staging buffer is full, and once at the end, every context's
    // reads are cut into super-k-mer records routed by minimizer, part p goes to device p (RCCL or peer copies) and is
    // counted there: the work per device stays what one device's share of the windows costs, however many devices
    // there are.  With two or three devices every device counts its own reads and only the counted ranges are
    // exchanged (GOSS_GROUP_EXCHANGE=records|counted overrides).
    bool useRecords = fed && P >= 4;
    if (const char* e = std::getenv("GOSS_GROUP_EXCHANGE"))
    {
        if (!std::strcmp(e, "records")) useRecords = fed;
        else if (!std::strcmp(e, "counted")) useRecords = false;
    }
    std::vector<uint64_t> stagedBytes(P, 0), stageCap(P, 0);
    uint64_t xRounds = 0, xRecords = 0, xWindows = 0;
    double xRouteMs = 0, xWireMs = 0, xCountMs = 0, xCountWaitMs = 0;
    uint32_t xTransport = 0;
    if (useRecords)
        for (size_t d = 0; d < P; ++d) gs[d]->check(goss_gpu_set_deferred(gs[d]->h, 1), "deferring the count");
    std::function<void()> drainForRound;          // (drainFeeders, defined below)
    auto exchangeRound = [&]() {
        drainForRound();
        std::vector<goss_gpu_ctx*> hs;
        for (auto& x : gs) hs.push_back(x->h);
        goss_gpu_group_xstats st;
        gs[0]->check(goss_gpu_group_route_exchange(hs.data(), (uint32_t)P, 0, &st), "exchanging the reads' records");
        ++xRounds; xRecords += st.records; xWindows += st.windows; xRouteMs += st.route_ms; xWireMs += st.wire_ms; xCountMs += st.count_ms;
        xCountWaitMs += st.count_wait_ms;
        xTransport = st.transport;
        std::fill(stagedBytes.begin(), stagedBytes.end(), 0);
    };
    // hand a batch to the next device; `done` runs when the device has taken the bytes (wait = until then)
    std::function<void(const char*, size_t, std::function<void()>, bool)> feed;
    feed = [&](const char* p, size_t n, std::function<void()> done, bool wait) {
        if (feedFailed.load()) { if (done) done(); feedError(); }
        if (useRecords && stageCap[nextDev] && n + 1 > stageCap[nextDev])
        {
            // a batch larger than a staging buffer (tiny buffers only: a buffer is 1/24 of the arena): handed over in
            // pieces cut behind read separators, each taken before the next is offered, the batch released at the end
            size_t at = 0;
            while (at < n)
            {
                size_t len = std::min(n - at, (size_t)stageCap[nextDev] - 1);
                if (at + len < n)
                {
                    size_t cut = len;
                    while (cut > 0 && p[at + cut - 1] != '\n') --cut;
                    if (cut == 0) { if (done) done(); throw Error::General("a read longer than a device's staging buffer (--hbm-budget too small for --devices)\n"); }
                    len = cut;
                }
                feed(p + at, len, nullptr, true);
                at += len;
            }
            if (done) done();
            return;
        }
        if (useRecords)
        {
            // (what a device has staged is tracked here: the feeders run behind, and a push that does not fit is refused)
            const size_t d = nextDev;
            if (stageCap[d] == 0)
            {
                uint64_t room = 0, cap = 0;
                gs[d]->check(goss_gpu_stage_room(gs[d]->h, &room, &cap), "sizing the staging buffer");
                stageCap[d] = cap > 4096 ? cap - 4096 : cap;
            }
            if (n + 1 > stageCap[d]) { feed(p, n, std::move(done), wait); return; }          // (now that the size is known: in pieces)
            if (stagedBytes[d] + n + 1 > stageCap[d]) exchangeRound();
            stagedBytes[d] += n + 1;
        }
        Feeder* f = feeders[nextDev].get();
        nextDev = (nextDev + 1) % P;
        {
            std::unique_lock<std::mutex> lk(f->m);
            f->cv.wait(lk, [&] { return f->q.size() < 2; });
            f->q.push_back(Feeder::Job{p, n, std::move(done)});
        }
        f->cv.notify_all();
        if (wait)
        {
            std::unique_lock<std::mutex> lk(f->m);
            f->cv.wait(lk, [&] { return f->q.empty() && !f->busy; });
        }
    };
    auto drainFeeders = [&]() {
        for (auto& f : feeders) { std::unique_lock<std::mutex> lk(f->m); f->cv.wait(lk, [&] { return f->q.empty() && !f->busy; }); }
        feedError();
    };
    drainForRound = drainFeeders;

    std::vector<char> batch;
    batch.reserve(cxt.batchBytes + (1u << 20));
    uint64_t reads = 0;
    HostAlloc pinned{[](size_t n) { void* p = nullptr; return goss_gpu_host_alloc(&p, n) == GOSS_OK ? p : nullptr; },
                     [](void* p) { goss_gpu_host_free(p); }, true, nullptr, nullptr, nullptr};
    if (beside)
    {
        pinned.alloc = [](size_t n) -> void* { void* p = nullptr; return posix_memalign(&p, 4096, (n + 4095) & ~(size_t)4095) == 0 ? p : nullptr; };
        pinned.release = [](void* p) { std::free(p); };
        pinned.ready = ctxUp;
        pinned.pin = [](void* p, size_t n) { return goss_gpu_host_register(p, (n + 4095) & ~(size_t)4095) == GOSS_OK; };
        pinned.unpin = [](void* p) { goss_gpu_host_unregister(p); };
    }
    double pushSeconds = 0;
    auto timedPush = [&](const char* p, size_t n) {
        needCtx();
        auto a = std::chrono::steady_clock::now();
        if (fed) feed(p, n, nullptr, true);
        else g.check(goss_gpu_push_bases_host(g.h, p, n), "counting k-mers");
        pushSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    };
    const OwnedPush ownedPush = [&](const char* p, size_t n, std::function<void()> release) { feed(p, n, std::move(release), false); };
    auto flush = [&]() {
        if (batch.empty()) return;
        timedPush(batch.data(), batch.size());
        batch.clear();
    };
    uint64_t progress = 0;
    ReadSink sink = [&](const char* seq, size_t len) {
        batch.insert(batch.end(), seq, seq + len);
        batch.push_back('\n');
        if (batch.size() >= cxt.batchBytes) flush();
        if (++progress % 100000 == 0) log(info, num(progress) + " reads");   // UnboundedProgressMonitor
    };
    // same order as the reference: line, fasta, fastq (GossCmdBuildKmerSet.cc:118-141)
    for (auto& f : lines) { log(info, "parsing sequences from " + f); reads += parseLines(f, sink); }
    for (auto& f : fastas) { log(info, "parsing sequences from " + f); reads += parseFasta(f, sink); }
    // Compressed FASTQ (.gz, .bz2) cannot be cut into chunks (one stream per file), but several such
    // files -- lanes, read pairs -- can be decompressed and framed side by side: one worker per file,
    // each with its own batch, pushes serialised.  The k-mer multiset does not depend on the
    // order in which reads arrive, so the output is the reference's.
    std::vector<std::string> gzFiles;
    auto compressed = [](const std::string& f) { return endsWith(f, ".gz") || endsWith(f, ".bz2"); };
    for (auto& f : fastqs) if (compressed(f)) gzFiles.push_back(f);
    if (gzFiles.size() > 1 && threads > 1)
    {
        flush();
        std::mutex pushMutex, errMutex;
        std::unique_ptr<Error> firstError;
        std::atomic<size_t> nextFile{0};
        std::atomic<uint64_t> gzReads{0};
        auto worker = [&]() {
            std::vector<char> mine;
            mine.reserve(cxt.batchBytes / 4 + (1u << 20));
            auto mflush = [&]() {
                if (mine.empty()) return;
                std::lock_guard<std::mutex> lk(pushMutex);
                timedPush(mine.data(), mine.size());
                mine.clear();
            };
            for (;;)
            {
                const size_t i = nextFile.fetch_add(1);
                if (i >= gzFiles.size()) break;
                try
                {
                    ReadSink s = [&](const char* seq, size_t len) {
                        mine.insert(mine.end(), seq, seq + len);
                        mine.push_back('\n');
                        if (mine.size() >= cxt.batchBytes / 4) mflush();
                    };
                    gzReads += parseFastq(gzFiles[i], s);
                    mflush();
                }
                catch (const Error& e)
                {
                    std::lock_guard<std::mutex> lk(errMutex);
                    if (!firstError) firstError.reset(new Error(e));
                }
            }
        };
        for (auto& f : gzFiles) log(info, "parsing sequences from " + f);
        std::vector<std::thread> pool;
        const size_t nw = std::min<size_t>(gzFiles.size(), (size_t)std::min<uint64_t>(threads, 64));
        for (size_t w = 0; w < nw; ++w) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
        if (firstError) throw *firstError;
        reads += gzReads;
    }
    for (auto& f : fastqs)
    {
        if (gzFiles.size() > 1 && threads > 1 && compressed(f)) continue;      // done above
        if (ctxSeen.load())
        { std::ostringstream o; o << "parsing sequences from " << f << " (contexts ready at "
            << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "s)"; log(info, o.str()); }
        else log(info, "parsing sequences from " + f + " (the context is created beside the parser)");
        flush();
        // one device: the workers pack their bases to 2 bits + a flag and the packed arrays are handed over without
        // waiting for the copy (GOSS_HOST_ASCII=1: the byte form, one synchronous push per chunk, as before)
        // The library calls are made by a thread of their own, fed through a queue: a push costs 50 to 80 us of driver
        // calls (two copies queued, an event, the buffers that have come back), 0.2 to 0.3 s for C2's 3 750 chunks -- on
        // the consumer's own thread that time came on top of its waiting for the workers.  All calls on the context
        // between here and the join below are this thread's, so the buffers are released on it.
        struct Cb { std::function<void()> fn; };
        struct PushJob { const uint32_t* codes; const uint16_t* bad; size_t n; std::function<void()> release; bool flush; };
        struct Pusher {
            std::mutex m; std::condition_variable cv, cvIdle;
            std::deque<PushJob> q; bool stop = false, busy = false; Error error; std::atomic<bool> failed{false};
            double seconds = 0;
            std::thread t;
        } pu;
        PackedPush packed;
        const bool usePacked = !fed && !std::getenv("GOSS_HOST_ASCII");
        auto pusherBody = [&]() {
            try { needCtx(); }
            catch (const Error& e) { std::lock_guard<std::mutex> lk(pu.m); pu.error = e; pu.failed.store(true); }
            catch (...) { std::lock_guard<std::mutex> lk(pu.m); pu.error = Error::General("creating the GPU context failed\n"); pu.failed.store(true); }
            for (;;)
            {
                PushJob j;
                {
                    std::unique_lock<std::mutex> lk(pu.m);
                    pu.busy = false;
                    if (pu.q.empty()) pu.cvIdle.notify_all();
                    pu.cv.wait(lk, [&] { return pu.stop || !pu.q.empty(); });
                    if (pu.q.empty()) return;
                    j = std::move(pu.q.front()); pu.q.pop_front();
                    pu.busy = true;
                }
                if (pu.failed.load()) { if (j.release) j.release(); continue; }          // (nothing is pushed any more: the buffers go back)
                const auto a = std::chrono::steady_clock::now();
                int rc;
                if (j.flush) rc = goss_gpu_flush(g.h);
                else
                {
                    Cb* cb = new Cb{std::move(j.release)};
                    rc = goss_gpu_push_packed_host_async(g.h, j.codes, j.bad, j.n, [](void* u) { Cb* c = (Cb*)u; c->fn(); delete c; }, cb);
                    if (rc != GOSS_OK) { cb->fn(); delete cb; }          // (a failed push: the buffer is ours again, goss_gpu.h)
                }
                pu.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
                if (rc != GOSS_OK)
                {
                    std::lock_guard<std::mutex> lk(pu.m);
                    pu.error = Error::General(std::string("counting k-mers: ") + goss_gpu_strerror(rc) + " (" + goss_gpu_last_error(g.h) + ")\n");
                    pu.failed.store(true);
                }
            }
        };
        auto pusherError = [&]() {
            if (!pu.failed.load()) return;
            std::lock_guard<std::mutex> lk(pu.m);
            throw pu.error;
        };
        packed.push = [&](const uint32_t* codes, const uint16_t* bad, size_t n, std::function<void()> release) {
            if (pu.failed.load()) { release(); pusherError(); }
            { std::lock_guard<std::mutex> lk(pu.m); pu.q.push_back(PushJob{codes, bad, n, std::move(release), false}); }
            pu.cv.notify_one();
        };
        // (every buffer handed over so far comes back before this returns)
        packed.drain = [&]() {
            {
                std::unique_lock<std::mutex> lk(pu.m);
                pu.q.push_back(PushJob{nullptr, nullptr, 0, nullptr, true});
                pu.cv.notify_one();
                pu.cvIdle.wait(lk, [&] { return pu.q.empty() && !pu.busy; });
            }
            pusherError();
        };
        if (usePacked) pu.t = std::thread(pusherBody);
        struct StopPusher { Pusher& p; ~StopPusher() { { std::lock_guard<std::mutex> lk(p.m); p.stop = true; } p.cv.notify_all(); if (p.t.joinable()) p.t.join(); } } stopPusher{pu};
        // (at most 32 framing threads: measured on the bench's box -- 256 cores shared with other jobs -- the 31.5 GB of
        // C2 take 0.5 s with 32 and 1.2 to 3.5 s with 64, whose system time is four to ten times higher; the pushes'
        // driver calls slow down with them: 1.34 / 1.43 s for the build with -T 32 against 1.64 / 1.94 s with -T 64)
        uint64_t r = parseFastqParallel(f, (unsigned)std::min<uint64_t>(threads, maxFramers()), parseChunkBytes(), timedPush, pinned,
                                        fed ? &ownedPush : nullptr, usePacked ? &packed : nullptr);
        // (the pusher is idle -- the parser has drained it -- and ends here: the calls that follow are this thread's again)
        { std::lock_guard<std::mutex> lk(pu.m); pu.stop = true; }
        pu.cv.notify_all();
        if (pu.t.joinable()) pu.t.join();
        pusherError();
        pushSeconds += pu.seconds;
        if (r == ~0ULL) r = parseFastq(f, sink);
        reads += r;
    }
    if (reads == 0) throw Error::General("No valid reads.");                  // KmerizingAdapter.hh:70-78
    flush();
    needCtx();
    if (beside) { std::ostringstream o; o << "contexts ready at " << cs.readyAt << "s (beside the parser)"; log(info, o.str()); }
    if (fed) drainFeeders();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (useRecords)
    {
        exchangeRound();          // (what is still staged)
        std::ostringstream o;
        o << "records routed by minimizer: " << xRecords << " records (" << xWindows << " windows) exchanged in " << xRounds << " round(s) over "
          << (xTransport == 1 ? "RCCL" : "peer copies") << ": routing " << xRouteMs / 1e3 << "s, transfer " << xWireMs / 1e3 << "s, counting (beside the staging of the next round; the last round's is the finish's) "
          << xCountMs / 1e3 << "s, of which waited for " << xCountWaitMs / 1e3 << "s";
        log(info, o.str());
    }
    { std::ostringstream o; o << "parsed and counted " << reads << " reads at " << secs() << "s (device time in pushes "
        << pushSeconds << "s)"; log(info, o.str()); }
    for (size_t d = 0; d < P; ++d)
    {
        uint64_t ms = 0, bytes = 0;
        goss_gpu_stat(gs[d]->h, "arena_ms", &ms); goss_gpu_stat(gs[d]->h, "arena_bytes", &bytes);
        std::ostringstream o; o << "HBM arena";
        if (P > 1) o << " of device " << devs[d];
        o << ": " << (bytes >> 30) << " GB mapped in " << ms / 1000.0 << "s";
        log(info, o.str());
        uint64_t fw = 0, fc = 0, fn = 0;
        goss_gpu_stat(gs[d]->h, "flush_wait_us", &fw); goss_gpu_stat(gs[d]->h, "flush_count_us", &fc); goss_gpu_stat(gs[d]->h, "flushes", &fn);
        std::ostringstream o2; o2 << "staging buffer counted " << fn << " times: " << fw / 1e6 << "s waiting for queued copies, " << fc / 1e6 << "s counting";
        log(info, o2.str());
    }

    log(info, "sorting the hashtable...");
    goss_gpu_counts counts;
    if (P == 1) g.check(goss_gpu_finish(g.h, &counts), "sorting");
    else
    {
        // every device finishes its own count (side by side), then the ranges are exchanged and merged
        std::vector<goss_gpu_counts> each(P);
        std::vector<int> rcs(P, GOSS_OK);
        std::vector<std::thread> pool;
        for (size_t d = 0; d < P; ++d) pool.emplace_back([&, d]() { rcs[d] = goss_gpu_finish(gs[d]->h, &each[d]); });
        for (auto& t : pool) t.join();
        for (size_t d = 0; d < P; ++d) gs[d]->check(rcs[d], "sorting");
        counts = each[0];
        for (size_t d = 1; d < P; ++d) { counts.windows += each[d].windows; counts.keys += each[d].keys; }
        { std::ostringstream o; o << "counted on " << P << " devices at " << secs() << "s"; log(info, o.str()); }
        std::vector<goss_gpu_ctx*> hs;
        for (auto& x : gs) hs.push_back(x->h);
        std::vector<uint64_t> sizes(P);
        g.check(goss_gpu_group_exchange(hs.data(), (uint32_t)P, 0, sizes.data()), "exchanging the key ranges");
        counts.distinct = 0;
        for (uint64_t v : sizes) counts.distinct += v;
    }
    log(info, "sorting done.");
    { std::ostringstream o; o << "merged at " << secs() << "s"; log(info, o.str()); }
    log(info, "writing out graph.");
    if (P == 1)
    {
        g.check(goss_gpu_emit(g.h), "building the on-disk arrays");
        writeObjectFiles(g.h, out);
    }
    else
    {
        std::vector<goss_gpu_ctx*> hs;
        for (auto& x : gs) hs.push_back(x->h);
        g.check(goss_gpu_group_emit(hs.data(), (uint32_t)P, 0), "building the on-disk arrays");
        writeObjectFiles(hs, out);
    }
    { std::ostringstream o; o << "written at " << secs() << "s"; log(info, o.str()); }
    stats.reads = reads; stats.windows = counts.windows; stats.keys = counts.keys; stats.distinct = counts.distinct;
    stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::ostringstream os;
    os << "total build time: " << stats.seconds << "s";
    log(info, os.str());
    os.str("");
    os << "k-mer windows: " << counts.windows << ", keys: " << counts.keys << ", distinct: " << counts.distinct;
    log(info, os.str());
}

}  // namespace

void GossCmdBuildKmerSet::operator()(const GossCmdContext& pCxt)
{
    runBuild(pCxt, mK, GOSS_MODE_KMER_SET, mKmerSetName, mFastaNames, mFastqNames, mLineNames, mT, mStats);
}

void GossCmdBuildGraph::operator()(const GossCmdContext& pCxt)
{
    runBuild(pCxt, mK, GOSS_MODE_GRAPH, mGraphName, mFastaNames, mFastqNames, mLineNames, mT, mStats);
}

// --------------------------------------------------------------------------------------
// command line (App.cc:176-417, GossApp.cc:145-202, GossOptionChecker.hh)
// --------------------------------------------------------------------------------------

namespace {

enum OptKind { kFlag, kU64, kString, kStrings };
struct OptDef { const char* lng; const char* sht; OptKind kind; const char* help; };

const OptDef kGlobal[] = {
    {"debug", "D", kStrings, "enable particular debugging output"},
    {"help", "h", kFlag, "show a help message"},
    {"log-file", "l", kString, "place to write messages"},
    {"tmp-dir", "", kStrings, "a directory to use for temporary files (default /tmp)"},
    {"num-threads", "T", kU64, "maximum number of worker threads to use, where possible"},
    {"verbose", "v", kFlag, "show progress messages"},
    {"version", "V", kFlag, "show the software version"},
};
const OptDef kCommon[] = {
    {"buffer-size", "B", kU64, "maximum size (in GB) for in-memory buffers (default: 2)"},
    {"fasta-in", "I", kStrings, "input file in FASTA format"},
    {"fastas-in", "F", kStrings, "input file containing filenames in FASTA format"},
    {"fastq-in", "i", kStrings, "input file in FASTQ format"},
    {"fastqs-in", "f", kStrings, "input file containing filenames in FASTQ format"},
    {"line-in", "", kStrings, "input file with one sequence per line"},
    {"graph-out", "O", kString, "name of the output graph object"},
    {"kmer-size", "k", kU64, "kmer size to use"},
};
const OptDef kKmerSetSpecific[] = {
    {"log-hash-slots", "S", kU64, "log2 of the number of hash slots to use (default 24)"},
};
// not in the reference: where and how much HBM to use
const OptDef kGpuSpecific[] = {
    {"device", "", kU64, "HIP device ordinal (default 0)"},
    {"devices", "", kString, "build commands: comma-separated HIP device ordinals to count on side by side (one context each)"},
    {"hbm-budget", "", kU64, "HBM budget in GB for keys and sort workspace (default: 80% of free HBM)"},
};

// --tmp-dir: the last one given, which must be a directory (the reference fails when it first creates a temporary file
// there: "PhysicalFileFactory::tmpName"; here the check comes first, with the name in the message)
std::string tmpDirOption(const std::map<std::string, std::vector<std::string>>& vals)
{
    auto it = vals.find("tmp-dir");
    if (it == vals.end() || it->second.empty()) return std::string();
    const std::string& d = it->second.back();
    struct stat st;
    if (::stat(d.c_str(), &st) != 0) throw Error::Errno(d, errno);
    if (!S_ISDIR(st.st_mode)) throw Error::General("--tmp-dir: " + d + " is not a directory\n");
    return d;
}

struct Parsed {
    std::map<std::string, std::vector<std::string>> vals;
    size_t count(const std::string& k) const { auto i = vals.find(k); return i == vals.end() ? 0 : i->second.size(); }
    const std::string& str(const std::string& k) const { return vals.at(k).back(); }
    const std::vector<std::string>& strs(const std::string& k) const { return vals.at(k); }
};

struct OptTable {
    std::vector<OptDef> defs;
    const OptDef* findLong(const std::string& n, bool& ambiguous) const
    {
        ambiguous = false;
        const OptDef* hit = nullptr;
        for (auto& d : defs) if (n == d.lng) return &d;
        for (auto& d : defs)                      // unique-prefix guessing, as program_options does
            if (strncmp(d.lng, n.c_str(), n.size()) == 0) { if (hit) { ambiguous = true; return nullptr; } hit = &d; }
        return hit;
    }
    const OptDef* findShort(char c) const
    {
        for (auto& d : defs) if (d.sht[0] == c && d.sht[0]) return &d;
        return nullptr;
    }
    std::string describe() const
    {
        std::ostringstream os;
        for (auto& d : defs)
        {
            std::string l = "  ";
            if (d.sht[0]) l += std::string("-") + d.sht + " [ --" + d.lng + " ]"; else l += std::string("--") + d.lng;
            if (d.kind != kFlag) l += " arg";
            if (l.size() < 40) l.resize(40, ' '); else l += " ";
            os << l << d.help << "\n";
        }
        return os.str();
    }
};

uint64_t toU64(const std::string& opt, const std::string& v)
{
    if (v.empty() || v[0] == '-') throw Error::Usage("the argument ('" + v + "') for option '--" + opt + "' is invalid\n");
    char* end = nullptr;
    errno = 0;
    unsigned long long x = strtoull(v.c_str(), &end, 10);
    if (errno || *end) throw Error::Usage("the argument ('" + v + "') for option '--" + opt + "' is invalid\n");
    return x;
}

// Parses argv[first..] against the table; unknown tokens are collected like
// command_line_parser(...).allow_unregistered() + the check at App.cc:255-262.
void parseArgs(int argc, char** argv, int first, const OptTable& t, Parsed& out, std::string& badMsg)
{
    for (int i = first; i < argc; ++i)
    {
        std::string tok = argv[i];
        const OptDef* d = nullptr;
        std::string inlineVal; bool hasInline = false;
        if (tok.size() > 2 && tok[0] == '-' && tok[1] == '-')
        {
            std::string name = tok.substr(2);
            size_t eq = name.find('=');
            if (eq != std::string::npos) { inlineVal = name.substr(eq + 1); name = name.substr(0, eq); hasInline = true; }
            bool amb;
            d = t.findLong(name, amb);
        }
        else if (tok.size() >= 2 && tok[0] == '-' && tok[1] != '-')
        {
            d = t.findShort(tok[1]);
            if (d && tok.size() > 2)
            {
                if (d->kind == kFlag) d = nullptr;      // grouped flags are not used by goss
                else { inlineVal = tok.substr(2); hasInline = true; }
            }
        }
        if (!d)
        {
            badMsg += "unknown option '" + tok + "'\n";
            continue;
        }
        if (d->kind == kFlag)
        {
            out.vals[d->lng].push_back("1");
            continue;
        }
        std::string v;
        if (hasInline) v = inlineVal;
        else if (i + 1 < argc) v = argv[++i];
        else throw Error::Usage(std::string("the required argument for option '--") + d->lng + "' is missing\n");
        if (d->kind == kU64) toU64(d->lng, v);
        out.vals[d->lng].push_back(v);
    }
}

// GossOptionChecker: accumulates error text, decides usage vs general error.
struct Checker {
    const Parsed& o;
    std::string errors;
    bool suggestUsage = false;
    bool mandatoryU64(const std::string& n, uint64_t& v, uint64_t maxv)
    {
        if (!o.count(n)) { errors += "mandatory option " + n + " was not given.\n"; suggestUsage = true; return false; }
        v = toU64(n, o.str(n));
        if (v > maxv)
        {
            errors += "The given value of the option " + n + " was invalid.\n";
            errors += "\tvalue " + num(v) + " is out of range.\n\tthe maximum allowed value is " + num(maxv) + ".\n";
            suggestUsage = true;
            return false;
        }
        return true;
    }
    bool optionalU64(const std::string& n, uint64_t& v)
    {
        if (!o.count(n)) return false;
        v = toU64(n, o.str(n));
        return true;
    }
    // FileCreateCheck(fac, true): "<name>.test" must be creatable (GossOptionChecker.hh:80-122)
    bool mandatoryOut(const std::string& n, std::string& v)
    {
        if (!o.count(n)) { errors += "mandatory option " + n + " was not given.\n"; suggestUsage = true; return false; }
        v = o.str(n);
        std::string probe = v + ".test";
        FILE* fp = fopen(probe.c_str(), "wb");
        if (!fp)
        {
            errors += "The given value of the option " + n + " was invalid.\n";
            errors += "\tcannot create filenames with prefix '" + v + "'\n";
            return false;
        }
        fclose(fp);
        ::remove(probe.c_str());
        return true;
    }
    // FileReadCheck (GossOptionChecker.hh:125-156)
    void repeatingIn(const std::string& n, strings& vals)
    {
        if (!o.count(n)) return;
        vals = o.strs(n);
        for (auto& f : vals)
            if (!InFile::readable(f))
            {
                errors += "The given value of the option " + n + " was invalid.\n";
                errors += "\tcannot open file '" + f + "' for reading\n";
            }
    }
    // expandFilenames (GossOptionChecker.hh:405-430): one name per '\n'-terminated line
    void expand(const std::string& n, strings& names)
    {
        if (!o.count(n)) return;
        for (auto& lf : o.strs(n))
        {
            InFile in(lf);
            std::string all; char buf[65536]; size_t got;
            while ((got = in.read(buf, sizeof buf)) > 0) all.append(buf, got);
            size_t b = 0;
            for (;;)
            {
                size_t e = all.find('\n', b);
                if (e == std::string::npos) break;      // a last line without '\n' is dropped
                names.push_back(all.substr(b, e - b));
                b = e + 1;
            }
        }
    }
    void throwIfNecessary()
    {
        if (errors.empty()) return;
        if (suggestUsage) throw Error::Usage(errors);
        throw Error::General(errors);
    }
};

const char* kVersion = "1.3.0-mi355x";

void printHelp(const std::string& cmdName, const OptTable* t)
{
    std::cerr << "goss <command> [options]\n\ncommands implemented by this build:\n"
              << "  build-graph      create a new graph\n"
              << "  build-kmer-set   create a new graph\n"
              << "  dump-graph       write out the graph in a robust text representation.\n"
              << "  dump-kmer-set    write out the graph in a robust text representation.\n"
              << "  help             print a summary of all the commands.\n"
              << "  lint-graph       verify that a graph structure is internally consistent\n"
              << "  restore-graph    read in a graph from a robust text representation.\n"
              << "  intersect-kmer-sets  generate the intersection of the given k-mer sets\n"
              << "  merge-and-annotate-kmer-sets  Decorate a graph with an assignment of kmers to graphs.\n"
              << "  merge-graphs     create a new graph by merging zero or more existing graphs\n"
              << "  merge-kmer-sets  create a new graph by merging zero or more existing graphs\n"
              << "  subtract-kmer-set  subtract the second k-mer set from the first\n"
              << "  graph-to-kmer-set  generate a graph's k-mer set\n";
    if (t)
    {
        std::cerr << "\n" << cmdName << "\n" << t->describe() << std::endl;
    }
}

}  // namespace

int gossMain(int argc, char* argv[])
{
    std::string cmdName;
    try
    {
        for (int i = 1; i < argc; ++i)
            if (!strcmp(argv[i], "--version")) { std::cout << "goss version " << kVersion << std::endl; return 0; }

        int argsToSkip = 0;
        if (argc < 2) cmdName = "help";
        else if (argc == 2 && (!strcmp(argv[1], "--help") || !strcmp(argv[1], "-h"))) cmdName = "help";
        else { cmdName = argv[1]; if (argv[1][0] != '-') argsToSkip = 1; }

        const bool isKmerSet = cmdName == "build-kmer-set", isGraph = cmdName == "build-graph";
        const bool isMerge = cmdName == "merge-kmer-sets" || cmdName == "merge-graphs";
        const bool isIntersect = cmdName == "intersect-kmer-sets", isSubtract = cmdName == "subtract-kmer-set";
        const bool isAnnotate = cmdName == "merge-and-annotate-kmer-sets";
        const bool isToKmerSet = cmdName == "graph-to-kmer-set";
        const bool isDump = cmdName == "dump-kmer-set" || cmdName == "dump-graph";
        const bool isRestore = cmdName == "restore-graph", isLint = cmdName == "lint-graph";
        if (isMerge || isIntersect || isSubtract || isAnnotate || isDump || isRestore || isLint || isToKmerSet)
        {
            // GossCmdFactoryDumpKmerSet/DumpGraph::create (GossCmdDumpKmerSet.cc:58-73,
            // GossCmdDumpGraph.cc:64-79), GossCmdFactoryRestoreGraph::create (GossCmdRestoreGraph.cc:138-152),
            // GossCmdFactoryLintGraph::create (GossCmdLintGraph.cc:278-292),
            // GossCmdFactoryIntersectKmerSets::create (GossCmdIntersectKmerSets.cc:131-150),
            // GossCmdFactorySubtractKmerSet::create (GossCmdSubtractKmerSet.cc:88-113),
            // GossCmdFactoryMergeAndAnnotateKmerSets::create (GossCmdMergeAndAnnotateKmerSets.cc:209-224),
            // GossCmdFactoryMerge<T>::create (GossCmdMerge.tcc:329-378)
            static const OptDef kMerge[] = {
                {"graph-in", "G", kStrings, "name of the input graph object"},
                {"graphs-in", "", kStrings, "read graph names (one per line) from the given file."},
                {"graph-out", "O", kString, "name of the output graph object"},
                {"max-merge", "", kU64, "The maximum number of graphs to merge at once."},
                {"input-file", "f", kString, "input file name ('-' for standard input)"},
                {"output-file", "o", kString, "output file name ('-' for standard output)"},
                {"dump-properties", "", kFlag, "show the internal properties of the graph"},
            };
            OptTable t;
            for (auto& d : kGlobal) t.defs.push_back(d);
            for (auto& d : kMerge) t.defs.push_back(d);
            for (auto& d : kGpuSpecific) t.defs.push_back(d);
            Parsed opts; std::string bad;
            parseArgs(argc, argv, 2, t, opts, bad);
            if (!bad.empty())
            {
                if (opts.count("help")) { std::cerr << bad; printHelp(cmdName, &t); return 1; }
                throw Error::Usage(bad);
            }
            Severity sev = opts.count("verbose") ? info : warning;
            std::unique_ptr<Logger> logger;
            if (opts.count("log-file"))
            {
                FILE* fp = fopen(opts.str("log-file").c_str(), "w");
                if (!fp) throw Error::Errno(opts.str("log-file"), errno);
                logger.reset(new Logger(fp, sev, true));
            }
            else logger.reset(new Logger(stderr, sev));
            Checker chk{opts, std::string(), false};
            strings ins;
            uint64_t maxMerge = 8;
            std::string outName;
            std::string textName = "-";
            if (isDump || isLint)
            {
                // getRepeatingOnce("graph-in") (GossOptionChecker.hh:235-254)
                if (!opts.count("graph-in")) { chk.errors += "mandatory option graph-in was not given.\n"; chk.suggestUsage = true; }
                else if (opts.strs("graph-in").size() != 1)
                { chk.errors += "mandatory option graph-in must be supplied exactly once.\n"; chk.suggestUsage = true; }
                else ins = opts.strs("graph-in");
                if (isDump && opts.count("output-file"))
                {
                    textName = opts.str("output-file");
                    if (textName != "-")
                    {
                        FILE* fp = fopen(textName.c_str(), "wb");      // FileCreateCheck(fac, false)
                        if (!fp)
                        {
                            chk.errors += "The given value of the option output-file was invalid.\n";
                            chk.errors += "\tcannot create file '" + textName + "'\n";
                        }
                        else fclose(fp);
                    }
                }
            }
            else if (isRestore)
            {
                if (opts.count("input-file")) textName = opts.str("input-file");
                chk.mandatoryOut("graph-out", outName);
            }
            else if (isToKmerSet)
            {
                // GossCmdFactoryGraphToKmerSet::create (GossCmdGraphToKmerSet.cc:62-77)
                if (!opts.count("graph-in")) { chk.errors += "mandatory option graph-in was not given.\n"; chk.suggestUsage = true; }
                else if (opts.strs("graph-in").size() != 1)
                { chk.errors += "mandatory option graph-in must be supplied exactly once.\n"; chk.suggestUsage = true; }
                else ins = opts.strs("graph-in");
                if (!opts.count("graph-out")) { chk.errors += "mandatory option graph-out was not given.\n"; chk.suggestUsage = true; }
                else outName = opts.str("graph-out");
            }
            else if (isAnnotate)
            {
                if (!opts.count("graph-in")) { chk.errors += "mandatory option graph-in was not given.\n"; chk.suggestUsage = true; }
                else if (opts.strs("graph-in").size() != 2)
                { chk.errors += "mandatory option graph-in must be supplied exactly twice.\n"; chk.suggestUsage = true; }
                else ins = opts.strs("graph-in");
                if (!opts.count("graph-out")) { chk.errors += "mandatory option graph-out was not given.\n"; chk.suggestUsage = true; }
                else outName = opts.str("graph-out");
            }
            else
            {
                if (opts.count("graph-in")) ins = opts.strs("graph-in");
                chk.expand("graphs-in", ins);
                if (isMerge && ins.empty())
                    chk.errors += "At least one input graph must be supplied either using --graph-in or --graphs-in.\n";
                if (isMerge) chk.optionalU64("max-merge", maxMerge);
                chk.mandatoryOut("graph-out", outName);
                if (isSubtract && ins.size() != 2) chk.errors += "Exactly two input k-mer sets required!";
            }
            if (opts.count("help")) { printHelp(cmdName, &t); return 1; }
            chk.throwIfNecessary();
            GossCmdContext cxt{*logger, cmdName};
            uint64_t dev = 0, budgetGb = 0;
            if (chk.optionalU64("device", dev)) cxt.device = (int)dev;
            if (chk.optionalU64("hbm-budget", budgetGb)) cxt.hbmBudget = budgetGb << 30;
            cxt.tmpDir = tmpDirOption(opts.vals);
            try
            {
                if (cmdName == "merge-kmer-sets") { GossCmdMergeKmerSets cmd(ins, maxMerge, outName); cmd(cxt); }
                else if (isMerge) { GossCmdMergeGraphs cmd(ins, maxMerge, outName); cmd(cxt); }
                else if (isIntersect) { GossCmdIntersectKmerSets cmd(ins, outName); cmd(cxt); }
                else if (isSubtract) { GossCmdSubtractKmerSet cmd(ins, outName); cmd(cxt); }
                else if (cmdName == "dump-kmer-set") { GossCmdDumpKmerSet cmd(ins[0], textName); cmd(cxt); }
                else if (cmdName == "dump-graph") { GossCmdDumpGraph cmd(ins[0], textName); cmd(cxt); }
                else if (isRestore) { GossCmdRestoreGraph cmd(textName, outName); cmd(cxt); }
                else if (isLint) { GossCmdLintGraph cmd(ins[0], opts.count("dump-properties") != 0); cmd(cxt); }
                else if (isToKmerSet) { GossCmdGraphToKmerSet cmd(ins[0], outName); cmd(cxt); }
                else { GossCmdMergeAndAnnotateKmerSets cmd(ins[0], ins[1], outName); cmd(cxt); }
            }
            catch (Error& e) { e.cmd = cmdName; throw; }
            return 0;
        }
        if (cmdName == "synth-reads")
        {
            // goss synth-reads <nreads> <read_len> <genome_len> <seed> <out.fq>: the bench's
            // deterministic synthetic read set as 4-line FASTQ
            if (argc != 7) throw Error::Usage("usage: goss synth-reads <nreads> <read-len> <genome-len> <seed> <out.fq>\n");
            uint64_t n = toU64("nreads", argv[2]), L = toU64("read-len", argv[3]), G = toU64("genome-len", argv[4]);
            uint64_t seed = toU64("seed", argv[5]);
            FILE* fp = fopen(argv[6], "wb");
            if (!fp) throw Error::Errno(argv[6], errno);
            const uint64_t per = 65536;
            std::vector<char> bases(per * (L + 1));
            std::string out;
            std::string qual(L, 'I');
            for (uint64_t r0 = 0; r0 < n; r0 += per)
            {
                uint64_t m = std::min(per, n - r0);
                if (goss_synth_reads_host(bases.data(), m, (uint32_t)L, G, seed, r0) != GOSS_OK)
                    throw Error::General("synth-reads: invalid arguments\n");
                out.clear();
                for (uint64_t i = 0; i < m; ++i)
                {
                    out += "@r"; out += std::to_string(r0 + i); out += '\n';
                    out.append(bases.data() + i * (L + 1), L + 1);
                    out += "+\n"; out += qual; out += '\n';
                }
                if (fwrite(out.data(), 1, out.size(), fp) != out.size()) { fclose(fp); throw Error::Write(argv[6]); }
            }
            fclose(fp);
            return 0;
        }
        if (cmdName == "dump-bases")
        {
            // diagnostic: print exactly the byte stream the build commands hand to the device
            // (each read's bases followed by '\n'), inputs in the order line, fasta, fastq
            OptTable t;
            for (auto& d : kGlobal) t.defs.push_back(d);
            for (auto& d : kCommon) t.defs.push_back(d);
            Parsed opts; std::string bad;
            parseArgs(argc, argv, 2, t, opts, bad);
            if (!bad.empty()) throw Error::Usage(bad);
            uint64_t reads = 0;
            ReadSink sink = [&](const char* seq, size_t len) { fwrite(seq, 1, len, stdout); fputc('\n', stdout); };
            strings fastas, fastqs, lines;
            Checker chk{opts, std::string(), false};
            chk.repeatingIn("fasta-in", fastas);
            chk.expand("fastas-in", fastas);
            chk.repeatingIn("fastq-in", fastqs);
            chk.expand("fastqs-in", fastqs);
            chk.repeatingIn("line-in", lines);
            chk.throwIfNecessary();
            try
            {
                for (auto& f : lines) reads += parseLines(f, sink);
                for (auto& f : fastas) reads += parseFasta(f, sink);
                uint64_t T = 1;
                chk.optionalU64("num-threads", T);
                for (auto& f : fastqs)
                {
                    // -T > 1 exercises the parallel parser (same byte stream, file order)
                    HostAlloc heap{[](size_t n) { return malloc(n); }, [](void* p) { free(p); }, false, nullptr, nullptr, nullptr};
                    // GOSS_DUMP_PACKED=1: the chunks go through the workers' 2-bit packer and are unpacked here the
                    // way the device unpacks them (a base letter in upper case, a newline for every non-base)
                    PackedPush viaPacked;
                    viaPacked.push = [&](const uint32_t* codes, const uint16_t* bad, size_t n, std::function<void()> release) {
                        std::string out(n, '\n');
                        for (size_t i = 0; i < n; ++i)
                            if (!((bad[i / 16] >> (i % 16)) & 1u)) out[i] = "ACGT"[(codes[i / 16] >> (2 * (i % 16))) & 3u];
                        fwrite(out.data(), 1, n, stdout);
                        release();
                    };
                    uint64_t r = parseFastqParallel(f, (unsigned)std::min<uint64_t>(T, maxFramers()), parseChunkBytes(),
                                                    [&](const char* p, size_t n) { fwrite(p, 1, n, stdout); }, heap, nullptr,
                                                    std::getenv("GOSS_DUMP_PACKED") ? &viaPacked : nullptr);
                    if (r == ~0ULL) r = parseFastq(f, sink);
                    reads += r;
                }
            }
            catch (Error& e) { e.cmd = cmdName; throw; }
            fflush(stdout);
            if (reads == 0) { Error e = Error::General("No valid reads."); e.cmd = cmdName; throw e; }
            return 0;
        }
        if (!isKmerSet && !isGraph)
        {
            if (cmdName != "help") std::cerr << "unknown command '" << cmdName << "'" << std::endl;
            printHelp(cmdName, nullptr);
            return 0;
        }

        OptTable t;
        for (auto& d : kGlobal) t.defs.push_back(d);
        for (auto& d : kCommon) t.defs.push_back(d);
        if (isKmerSet) for (auto& d : kKmerSetSpecific) t.defs.push_back(d);
        for (auto& d : kGpuSpecific) t.defs.push_back(d);

        Parsed opts;
        std::string bad;
        parseArgs(argc, argv, 1 + argsToSkip, t, opts, bad);
        if (!bad.empty())
        {
            if (opts.count("help")) { std::cerr << bad; printHelp(cmdName, &t); return 1; }
            throw Error::Usage(bad);
        }

        // logging (App.cc:291-301)
        Severity sev = opts.count("verbose") ? info : warning;
        std::unique_ptr<Logger> logger;
        if (opts.count("log-file"))
        {
            FILE* fp = fopen(opts.str("log-file").c_str(), "w");
            if (!fp) throw Error::Errno(opts.str("log-file"), errno);
            logger.reset(new Logger(fp, sev, true));
        }
        else logger.reset(new Logger(stderr, sev));
        if (opts.count("debug"))
            for (auto& d : opts.strs("debug")) (*logger)(warning, "unknown debug: " + d);

        // the factory (GossCmdBuildKmerSet.cc:151-198, GossCmdBuildGraph.cc:428-476)
        Checker chk{opts, std::string(), false};
        uint64_t K = 0;
        chk.mandatoryU64("kmer-size", K, isGraph ? 62 : 63);
        uint64_t B = 2;
        chk.optionalU64("buffer-size", B);
        if (isGraph && B > 24)
        {
            (*logger)(warning, "Unsupported --buffer-size " + num(B) + ", truncating to 24.");
            B = 24;
        }
        // BackyardHash::maxSlotBits(B << 30) = floor(log2((B<<30)/22)) (BackyardHash.hh:414-418)
        uint64_t S = 0;
        { uint64_t slots = (uint64_t)((double)(B << 30) / (1.5 * 4 + 16)); S = (uint64_t)std::log2((double)slots); }
        if (isKmerSet) chk.optionalU64("log-hash-slots", S);
        uint64_t N = (uint64_t)((double)(B << 30) / (1.5 * 4 + 16));
        uint64_t T = 4;
        chk.optionalU64("num-threads", T);
        std::string outName;
        chk.mandatoryOut("graph-out", outName);
        strings fastas, fastqs, lines;
        chk.repeatingIn("fasta-in", fastas);
        chk.expand("fastas-in", fastas);
        chk.repeatingIn("fastq-in", fastqs);
        chk.expand("fastqs-in", fastqs);
        chk.repeatingIn("line-in", lines);
        if (opts.count("help")) { printHelp(cmdName, &t); return 1; }
        chk.throwIfNecessary();

        GossCmdContext cxt{*logger, cmdName};
        uint64_t dev = 0, budgetGb = 0;
        if (chk.optionalU64("device", dev)) cxt.device = (int)dev;
        if (chk.optionalU64("hbm-budget", budgetGb)) cxt.hbmBudget = budgetGb << 30;
        cxt.tmpDir = tmpDirOption(opts.vals);
        if (!cxt.tmpDir.empty())
            (*logger)(info, "--tmp-dir " + cxt.tmpDir + ": not needed by this build (the runs of its chunks are held in HBM and merged there; no temporary file is written)");
        if (opts.count("devices"))
        {
            // "0,1,2,3": one context per listed device (an ordinal may appear twice: two contexts share that GPU)
            const std::string& v = opts.str("devices");
            size_t at = 0;
            while (at <= v.size())
            {
                const size_t comma = std::min(v.find(',', at), v.size());
                const std::string item = v.substr(at, comma - at);
                if (item.empty() || item.find_first_not_of("0123456789") != std::string::npos || item.size() > 3)
                    throw Error::Usage("--devices takes a comma-separated list of device ordinals, e.g. 0,1,2,3\n");
                cxt.devices.push_back(atoi(item.c_str()));
                at = comma + 1;
            }
            if (cxt.devices.size() > 64) throw Error::Usage("--devices: at most 64 devices\n");
        }
        try
        {
            if (isKmerSet) { GossCmdBuildKmerSet cmd(K, S, N, T, outName, fastas, fastqs, lines); cmd(cxt); }
            else { GossCmdBuildGraph cmd(K, S, N, T, outName, fastas, fastqs, lines); cmd(cxt); }
        }
        catch (Error& e) { e.cmd = cmdName; throw; }
    }
    catch (const Error& e)
    {
        // the printing order of App.cc:328-417
        if (!e.cmd.empty()) std::cerr << "error performing " << e.cmd << ":" << std::endl;
        if (!e.parse.empty()) std::cerr << "\t'" << e.file << "': " << e.parse << std::endl;
        if (!e.write_name.empty()) std::cerr << "\tcannot write to '" << e.write_name << "'" << std::endl;
        if (!e.general.empty()) std::cerr << e.general;
        if (e.err_no)
        {
            std::cerr << "\t";
            if (!e.file.empty()) std::cerr << "'" << e.file << "': ";
            std::cerr << strerror(e.err_no) << std::endl;
        }
        if (!e.usage.empty())
        {
            std::cerr << e.usage;
            std::cerr << "use\n\tgoss " << cmdName << " -h\nfor more usage information." << std::endl;
        }
        return 1;
    }
    catch (std::exception& e)
    {
        std::cerr << "caught unexpected exception: " << e.what() << std::endl;
        return 1;
    }
    catch (...)
    {
        std::cerr << "caught unknown exception" << std::endl;
        return 1;
    }
    return 0;
}

}  // namespace gosshost
