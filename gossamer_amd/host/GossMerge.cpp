// GossMerge.cpp -- `goss merge-kmer-sets` / `goss merge-graphs` (GossCmdMerge.tcc:151-326,
// GossCmdMergeKmerSets.hh, GossCmdMergeGraphs.hh): k-way merge of existing objects, equal keys'
// counts summed, the result built with the estimate M = sum of the inputs' counts.
//
// The inputs are read in their on-disk form and decoded on the device
// (goss_gpu_push_run_sparse); the merge itself is the library's merge of sorted runs; the output
// files are the device-built images, as for the build commands.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <deque>
#include <map>
#include <sstream>

#include "../../include/goss_gpu.h"
#include "GossHost.hpp"

namespace gosshost {

namespace {

std::string num(uint64_t v) { return std::to_string(v); }

// A whole file mapped read-only.
struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    Mapped() = default;
    Mapped(const Mapped&) = delete;
    Mapped& operator=(const Mapped&) = delete;
    Mapped(Mapped&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ~Mapped() { if (p && n) munmap((void*)p, n); }
    void open(const std::string& name)
    {
        int fd = ::open(name.c_str(), O_RDONLY);
        if (fd < 0) throw Error::Errno(name, errno);
        struct stat st;
        if (fstat(fd, &st) != 0) { int e = errno; ::close(fd); throw Error::Errno(name, e); }
        n = (size_t)st.st_size;
        if (n)
        {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { int e = errno; ::close(fd); throw Error::Errno(name, e); }
            p = (const uint8_t*)m;
        }
        ::close(fd);
    }
};

struct Col { std::string suffix; uint32_t bytes; uint32_t shift; };

// IntegerArray::builder column layout (IntegerArray.cc:259-357, StackedArray.hh:152-178).
bool layout(uint32_t bits, const std::string& prefix, uint32_t shift, std::vector<Col>& out)
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
    return layout(ub, prefix + ".upr", shift + lb, out) && layout(lb, prefix + ".lwr", shift, out);
}

struct SparseFiles {
    uint64_t D = 0, qD = 0, count = 0;
    Mapped high;
    std::vector<Col> cols;
    std::vector<Mapped> colFiles;
};

// SparseArray(base, fac) (SparseArray.cc:175-194): header version check, files of the array.
void openSparse(const std::string& base, SparseFiles& s)
{
    Mapped hdr; hdr.open(base + ".header");
    if (hdr.n < 64) throw Error::General("\tfile '" + base + ".header' is too short to be a SparseArray header\n");
    uint64_t h[8]; memcpy(h, hdr.p, 64);
    if (h[0] != 2012030501ULL)
        throw Error::General("\ta version mismatch was detected. goss expected 2012030501 but found " + num(h[0]) + ".\n");
    s.D = h[1]; s.qD = h[2]; s.count = h[7];
    s.high.open(base + ".high-bits");
    if (!layout((uint32_t)s.qD, "", 0, s.cols))
        throw Error::General("IntegerArray::create: unsupported integer width " + num(s.qD));
    s.colFiles.resize(s.cols.size());
    for (size_t i = 0; i < s.cols.size(); ++i)
    {
        s.colFiles[i].open(base + ".low-bits" + s.cols[i].suffix);
        if (s.colFiles[i].n < s.count * s.cols[i].bytes)
            throw Error::General("\tfile '" + base + ".low-bits" + s.cols[i].suffix + "' is shorter than its header says\n");
    }
}

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

struct ObjectInfo { uint64_t K = 0, count = 0; bool asymmetric = false; };

// Header of an object: T::LazyIterator(name, fac).K() / .count() / .asymmetric().
ObjectInfo objectInfo(const std::string& name, bool graph)
{
    ObjectInfo o;
    Mapped hdr; hdr.open(name + ".header");
    if (hdr.n < 24) throw Error::General("\tunable to open graph '" + name + "'\n");
    uint64_t h[3]; memcpy(h, hdr.p, 24);
    const uint64_t want = graph ? 2011101014ULL : 2011101701ULL;
    if (h[0] != want)
        throw Error::General("\ta version mismatch was detected. goss expected " + num(want) + " but found " + num(h[0]) + ".\n");
    o.K = h[1];
    if (!graph) { o.count = h[2]; return o; }
    o.asymmetric = h[2] & 1;
    // Graph::LazyIterator: count = sum of the frequencies in <name>-counts-hist.txt (Graph.cc:195-216)
    Mapped hist; hist.open(name + "-counts-hist.txt");
    const char* p = (const char*)hist.p; const char* e = p + hist.n;
    std::string text(p ? p : "", hist.n);
    std::istringstream in(text);
    uint64_t m, c;
    while (in >> m >> c) o.count += c;
    (void)e;
    return o;
}

// Decode one object into the context as a run.
void pushObject(GpuCtx& g, const std::string& name, bool graph, uint32_t weight = 1)
{
    auto describe = [](const SparseFiles& s) {
        goss_gpu_sparse_run r{};
        r.D = s.D; r.count = s.count;
        r.high_bits = (const uint64_t*)s.high.p; r.high_words = s.high.n / 8;
        r.ncols = (uint32_t)s.cols.size();
        for (size_t i = 0; i < s.cols.size(); ++i)
        {
            r.col[i] = s.colFiles[i].p; r.col_bytes[i] = s.cols[i].bytes; r.col_shift[i] = s.cols[i].shift;
        }
        return r;
    };
    SparseFiles s;
    openSparse(graph ? name + "-edges" : name + ".kmers", s);
    goss_gpu_sparse_run r = describe(s);
    r.weight = weight;
    if (!graph)
    {
        g.check(goss_gpu_push_run_sparse(g.h, &r), "reading an input object");
        return;
    }
    // the multiplicities stay in their VariableByteArray files: the library reads them on the device
    // (VariableByteArray::GeneralIterator, VariableByteArray.hh:120-195)
    const std::string cb = name + "-counts";
    Mapped o0, o1, o2;
    o0.open(cb + ".ord0"); o1.open(cb + ".ord1"); o2.open(cb + ".ord2");
    if (o0.n < s.count) throw Error::General("\tfile '" + cb + ".ord0' is shorter than the edge count\n");
    SparseFiles p1, p2;
    openSparse(cb + ".ord1p", p1);
    openSparse(cb + ".ord2p", p2);
    goss_gpu_vba v{};
    v.ord0 = (const uint8_t*)o0.p; v.ord0_bytes = o0.n;
    v.ord1p = describe(p1); v.ord1 = (const uint8_t*)o1.p; v.ord1_bytes = o1.n;
    v.ord2p = describe(p2); v.ord2 = (const uint16_t*)o2.p; v.ord2_bytes = o2.n;
    // a damaged object may list more items than its byte files hold: read what is there
    v.ord1p.count = std::min<uint64_t>(v.ord1p.count, o1.n);
    v.ord2p.count = std::min<uint64_t>(v.ord2p.count, o2.n / 2);
    g.check(goss_gpu_push_run_graph(g.h, &r, &v), "reading an input object");
}

struct HostRun { std::vector<uint64_t> keys; std::vector<uint32_t> counts; uint64_t m = 0; };

struct Item { std::string name; bool temp = false; HostRun run; uint64_t count = 0; std::string spill; };

// A partial merge as a temporary file under --tmp-dir (the role of the reference's temporary objects, GossCmdMerge.tcc:
// 176-208 with PhysicalFileFactory::tmpName): `m` keys of `words` 64-bit words, then `m` 32-bit counts.  Written once,
// mapped once when its group is merged, removed by the command -- also when the command fails in between.
struct SpillFiles {
    std::vector<std::string> names;
    ~SpillFiles() { for (auto& n : names) ::unlink(n.c_str()); }
};
std::string writeSpill(const std::string& dir, uint64_t serial, const HostRun& run, size_t words)
{
    const std::string name = dir + "/goss-merge-" + std::to_string((long long)::getpid()) + "-" + std::to_string(serial) + ".run";
    FILE* fp = std::fopen(name.c_str(), "wb");
    if (!fp) throw Error::Errno(name, errno);
    const size_t nk = run.m * words;
    const bool ok = (nk == 0 || std::fwrite(run.keys.data(), 8, nk, fp) == nk) && (run.m == 0 || std::fwrite(run.counts.data(), 4, run.m, fp) == run.m);
    const int werr = errno;
    if (std::fclose(fp) != 0 || !ok) { ::unlink(name.c_str()); throw Error::Errno(name, ok ? errno : werr); }
    return name;
}

void writeOut(GpuCtx& g, const std::string& out)
{
    writeObjectFiles(g.h, out);
}

void runMerge(const GossCmdContext& cxt, bool graph, const strings& ins, uint64_t maxMerge, const std::string& out)
{
    auto t0 = std::chrono::steady_clock::now();
    Logger& log = cxt.log;
    if (ins.empty()) throw Error::General("At least one input graph must be supplied either using --graph-in or --graphs-in.\n");
    if (maxMerge < 2) maxMerge = 2;

    // all inputs must agree on k (and on sense): GossCmdMerge.tcc:219-262.  The reference checks
    // inside every merge() call; the groups are disjoint, so checking all pairs against the first
    // member of their group is what it does -- done up front here, with the same text.
    std::vector<ObjectInfo> infos;
    uint64_t bytes = 0;
    for (auto& n : ins)
    {
        infos.push_back(objectInfo(n, graph));
        struct stat st;
        for (const char* suf : {".kmers.high-bits", "-edges.high-bits"})
            if (::stat((n + suf).c_str(), &st) == 0) bytes += (uint64_t)st.st_size;
        bytes += infos.back().count * 24;
    }
    const uint64_t K = infos[0].K;
    const uint64_t maxK = graph ? 62 : 63;
    if (K > maxK || K == 0) throw Error::General("unable to build a graph with k=" + num(K));

    uint64_t budget = cxt.hbmBudget ? cxt.hbmBudget : bytes * 6 + (4ULL << 30);
    GpuCtx g;
    g.check(goss_gpu_create(&g.h, cxt.device, (uint32_t)K, graph ? GOSS_MODE_GRAPH : GOSS_MODE_KMER_SET, budget, nullptr),
            "creating the GPU context");
    if (cxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    const size_t words = (2 * (K + (graph ? 1 : 0)) <= 62) ? 1 : 2;

    std::deque<Item> todo;
    for (size_t i = 0; i < ins.size(); ++i) { Item it; it.name = ins[i]; it.count = infos[i].count; todo.push_back(std::move(it)); }
    std::deque<ObjectInfo> tinfo(infos.begin(), infos.end());

    // one GossCmdMerge::merge(): returns the estimate `tot` it would give the Builder
    auto mergeGroup = [&](std::vector<Item>& group, std::vector<ObjectInfo>& ginfo) -> uint64_t {
        uint64_t tot = 0;
        for (size_t i = 0; i < group.size(); ++i)
        {
            if (ginfo[i].K != ginfo[0].K)
                throw Error::General("all graphs involved in a merge must have the same kmer-size.\n" + group[0].name + " has k="
                                     + num(ginfo[0].K) + ".\n" + group[i].name + " has k=" + num(ginfo[i].K) + ".\n");
            if (ginfo[i].asymmetric != ginfo[0].asymmetric)
                throw Error::General("graphs involved in a merge must either all preserve sense or not.\n" + group[0].name
                                     + (ginfo[0].asymmetric ? " preserves sense" : " does not preserve sense") + ".\n" + group[i].name
                                     + (ginfo[i].asymmetric ? " preserves sense" : " does not preserve sense") + ".\n");
            log(info, " " + group[i].name + " " + num(group[i].count));
            tot += group[i].count;
        }
        g.check(goss_gpu_reset(g.h), "resetting the GPU context");
        for (auto& it : group)
        {
            if (it.temp && !it.spill.empty())
            {
                // a partial merge that was written under --tmp-dir: mapped, handed over, unmapped (the file goes when the command ends)
                if (it.run.m)
                {
                    const size_t bytes = (size_t)it.run.m * (words * 8 + 4);
                    const int fd = ::open(it.spill.c_str(), O_RDONLY);
                    if (fd < 0) throw Error::Errno(it.spill, errno);
                    void* mp = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
                    const int merr = errno;
                    ::close(fd);
                    if (mp == MAP_FAILED) throw Error::Errno(it.spill, merr);
                    const int rc = goss_gpu_push_run_host(g.h, (const uint64_t*)mp, (const uint32_t*)((const char*)mp + (size_t)it.run.m * words * 8), it.run.m);
                    munmap(mp, bytes);
                    g.check(rc, "re-reading a partial merge");
                }
            }
            else if (it.temp) g.check(goss_gpu_push_run_host(g.h, it.run.keys.data(), it.run.counts.data(), it.run.m), "re-reading a partial merge");
            else pushObject(g, it.name, graph);
        }
        log(info, "starting graph merge");
        goss_gpu_counts counts;
        g.check(goss_gpu_finish(g.h, &counts), "merging");
        return tot;
    };

    uint64_t serial = 0;
    SpillFiles spills;
    while (todo.size() > maxMerge)
    {
        std::vector<Item> group; std::vector<ObjectInfo> ginfo;
        for (uint64_t i = 0; i < maxMerge; ++i)
        {
            group.push_back(std::move(todo.front())); todo.pop_front();
            ginfo.push_back(tinfo.front()); tinfo.pop_front();
        }
        mergeGroup(group, ginfo);
        // the temporary object of the reference stays in host memory here
        Item t; t.temp = true; t.name = "<temporary " + num(serial++) + ">";
        const void* dk; const uint32_t* dc; uint64_t m = 0;
        g.check(goss_gpu_result(g.h, &dk, &dc, &m), "reading a partial merge");
        t.run.m = m; t.run.keys.resize(m * words); t.run.counts.resize(m);
        g.check(goss_gpu_result_copy(g.h, 0, m, t.run.keys.data(), t.run.counts.data()), "reading a partial merge");
        t.count = m;
        if (!cxt.tmpDir.empty())
        {
            // --tmp-dir given: the partial result leaves host memory as well
            t.spill = writeSpill(cxt.tmpDir, serial, t.run, words);
            spills.names.push_back(t.spill);
            log(info, "partial merge " + t.name + " written to " + t.spill);
            std::vector<uint64_t>().swap(t.run.keys);
            std::vector<uint32_t>().swap(t.run.counts);
        }
        ObjectInfo ti = ginfo[0]; ti.count = m;
        todo.push_back(std::move(t)); tinfo.push_back(ti);
        log(info, "finishing graph merge");
    }
    std::vector<Item> group; std::vector<ObjectInfo> ginfo;
    while (!todo.empty())
    {
        group.push_back(std::move(todo.front())); todo.pop_front();
        ginfo.push_back(tinfo.front()); tinfo.pop_front();
    }
    const uint64_t tot = mergeGroup(group, ginfo);
    g.check(goss_gpu_emit_estimate(g.h, tot), "building the on-disk arrays");
    writeOut(g, out);
    log(info, "finishing graph merge");
    std::ostringstream os;
    os << "total build time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    log(info, os.str());
}

// The estimate of HBM needed to hold the decoded inputs and merge them.
uint64_t setBudget(const GossCmdContext& cxt, const strings& ins, std::vector<ObjectInfo>& infos)
{
    uint64_t bytes = 0;
    for (auto& n : ins)
    {
        infos.push_back(objectInfo(n, false));
        struct stat st;
        if (::stat((n + ".kmers.high-bits").c_str(), &st) == 0) bytes += (uint64_t)st.st_size;
        bytes += infos.back().count * 24;
    }
    return cxt.hbmBudget ? cxt.hbmBudget : bytes * 6 + (4ULL << 30);
}

std::string elapsed(std::chrono::steady_clock::time_point t0)
{
    std::ostringstream os;
    os << "total elapsed time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return os.str();
}

// Every input must fit the key width chosen from the first input's K.  The reference compares
// the raw values whatever their K (GossCmdIntersectKmerSets.cc:112-116 takes K from the first
// set only); mixing k-mer sizes is meaningless, so it is an error here.
void requireSameK(const strings& ins, const std::vector<ObjectInfo>& infos)
{
    for (size_t i = 1; i < ins.size(); ++i)
        if (infos[i].K != infos[0].K)
            throw Error::General("all k-mer sets must have the same kmer-size.\n" + ins[0] + " has k=" + num(infos[0].K) + ".\n"
                                 + ins[i] + " has k=" + num(infos[i].K) + ".\n");
}

}  // namespace

// GossCmdIntersectKmerSets::operator() (GossCmdIntersectKmerSets.cc:96-128): the k-mers present
// in every non-empty input, built with the exact count.  Each set is one run of weight 1; after
// the merge the k-mers whose count equals the number of sets are kept.
void GossCmdIntersectKmerSets::operator()(const GossCmdContext& pCxt)
{
    auto t0 = std::chrono::steady_clock::now();
    Logger& log = pCxt.log;
    if (mIns.empty())
    {
        log(info, "no input k-mer sets!");
        log(info, elapsed(t0));
        return;
    }
    std::vector<ObjectInfo> infos;
    const uint64_t budget = setBudget(pCxt, mIns, infos);
    requireSameK(mIns, infos);
    const uint64_t K = infos[0].K;
    if (K > 63 || K == 0) throw Error::General("unable to build a graph with k=" + num(K));
    GpuCtx g;
    g.check(goss_gpu_create(&g.h, pCxt.device, (uint32_t)K, GOSS_MODE_KMER_SET, budget, nullptr), "creating the GPU context");
    if (pCxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    log(info, "counting k-mers");
    uint32_t sets = 0;
    for (size_t i = 0; i < mIns.size(); ++i)
    {
        if (infos[i].count == 0) continue;           // an invalid iterator is not kept (:38-41)
        pushObject(g, mIns[i], false, 1);
        ++sets;
    }
    // every input empty: the reference's loop dereferences an empty vector; here the result is
    // the empty set
    goss_gpu_counts counts;
    g.check(goss_gpu_finish(g.h, &counts), "merging");
    g.check(goss_gpu_select_counts(g.h, sets ? sets : 1, sets ? sets : 1), "selecting the common k-mers");
    uint64_t m = 0;
    g.check(goss_gpu_result(g.h, nullptr, nullptr, &m), "counting");
    log(info, "found " + num(m) + " k-mers");
    log(info, "building intersection");
    g.check(goss_gpu_emit(g.h), "building the on-disk arrays");
    writeOut(g, mOut);
    log(info, elapsed(t0));
}

// GossCmdSubtractKmerSet::operator() (GossCmdSubtractKmerSet.cc:32-85): lhs weight 1, rhs
// weight 2; a merged count of exactly 1 means "in lhs only".
void GossCmdSubtractKmerSet::operator()(const GossCmdContext& pCxt)
{
    auto t0 = std::chrono::steady_clock::now();
    Logger& log = pCxt.log;
    if (mIns.size() != 2) throw Error::General("Exactly two input k-mer sets required!");
    std::vector<ObjectInfo> infos;
    const uint64_t budget = setBudget(pCxt, mIns, infos);
    requireSameK(mIns, infos);
    const uint64_t K = infos[0].K;
    if (K > 63 || K == 0) throw Error::General("unable to build a graph with k=" + num(K));
    GpuCtx g;
    g.check(goss_gpu_create(&g.h, pCxt.device, (uint32_t)K, GOSS_MODE_KMER_SET, budget, nullptr), "creating the GPU context");
    if (pCxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    log(info, "calculating difference");
    pushObject(g, mIns[0], false, 1);
    pushObject(g, mIns[1], false, 2);
    goss_gpu_counts counts;
    g.check(goss_gpu_finish(g.h, &counts), "merging");
    g.check(goss_gpu_select_counts(g.h, 1, 1), "selecting the difference");
    uint64_t m = 0;
    g.check(goss_gpu_result(g.h, nullptr, nullptr, &m), "counting");
    log(info, "found " + num(m) + " k-mers in difference");
    log(info, "building difference set");
    g.check(goss_gpu_emit(g.h), "building the on-disk arrays");
    writeOut(g, mOut);
    log(info, elapsed(t0));
}

// GossCmdGraphToKmerSet::operator() (GossCmdGraphToKmerSet.cc:30-59): every edge that is its own
// normal form goes into a KmerSet::Builder(K + 1, out, fac, <edge count of the graph>).
void GossCmdGraphToKmerSet::operator()(const GossCmdContext& pCxt)
{
    auto t0 = std::chrono::steady_clock::now();
    Logger& log = pCxt.log;
    const ObjectInfo gi = objectInfo(mIn, true);
    const uint64_t rho = gi.K + 1;
    if (rho > 63) throw Error::General("unable to build a k-mer set with k=" + num(rho));
    SparseFiles s;
    openSparse(mIn + "-edges", s);
    const uint64_t budget = pCxt.hbmBudget ? pCxt.hbmBudget : s.count * 24 * 6 + s.high.n + (4ULL << 30);
    GpuCtx g;
    g.check(goss_gpu_create(&g.h, pCxt.device, (uint32_t)rho, GOSS_MODE_KMER_SET, budget, nullptr), "creating the GPU context");
    if (pCxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    log(info, "building (k+1)-mer set");
    goss_gpu_sparse_run r{};
    r.D = s.D; r.count = s.count;
    r.high_bits = (const uint64_t*)s.high.p; r.high_words = s.high.n / 8;
    r.ncols = (uint32_t)s.cols.size();
    for (size_t i = 0; i < s.cols.size(); ++i)
    {
        r.col[i] = s.colFiles[i].p; r.col_bytes[i] = s.cols[i].bytes; r.col_shift[i] = s.cols[i].shift;
    }
    r.counts = nullptr;
    r.weight = 1;
    g.check(goss_gpu_push_run_sparse(g.h, &r), "reading the graph's edges");
    goss_gpu_counts counts;
    g.check(goss_gpu_finish(g.h, &counts), "reading the graph's edges");
    g.check(goss_gpu_select_normal(g.h), "selecting the normal edges");
    g.check(goss_gpu_emit_estimate(g.h, gi.count), "building the on-disk arrays");
    writeOut(g, mOut);
    log(info, elapsed(t0));
}

// GossCmdMergeAndAnnotateKmerSets::operator() (GossCmdMergeAndAnnotateKmerSets.cc:30-206): the
// union of two k-mer sets built with the exact count, plus one membership bitmap per side.
void GossCmdMergeAndAnnotateKmerSets::operator()(const GossCmdContext& pCxt)
{
    auto t0 = std::chrono::steady_clock::now();
    Logger& log = pCxt.log;
    strings ins{mLhs, mRhs};
    std::vector<ObjectInfo> infos;
    const uint64_t budget = setBudget(pCxt, ins, infos);
    if (infos[0].count == 0 || infos[1].count == 0) throw "nonsense";      // :40-43
    if (infos[0].K != infos[1].K) throw "nonsense";                        // :45-48
    const uint64_t K = infos[0].K;
    if (K > 63 || K == 0) throw Error::General("unable to build a graph with k=" + num(K));
    GpuCtx g;
    g.check(goss_gpu_create(&g.h, pCxt.device, (uint32_t)K, GOSS_MODE_KMER_SET, budget, nullptr), "creating the GPU context");
    if (pCxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    log(info, "counting kmers.");
    pushObject(g, mLhs, false, 1);
    pushObject(g, mRhs, false, 2);
    goss_gpu_counts counts;
    g.check(goss_gpu_finish(g.h, &counts), "merging");
    uint64_t n = 0;
    g.check(goss_gpu_result(g.h, nullptr, nullptr, &n), "counting");
    const uint64_t common = infos[0].count + infos[1].count - n;
    log(info, "writing out " + num(n) + " kmers.");
    log(info, "of which " + num(common) + " are common.");
    g.check(goss_gpu_emit(g.h), "building the on-disk arrays");
    g.check(goss_gpu_emit_count_bits(g.h, 1, ".lhs-bits"), "building the annotation");
    g.check(goss_gpu_emit_count_bits(g.h, 2, ".rhs-bits"), "building the annotation");
    writeOut(g, mOut);
    printf("%llu\t%llu\t%llu\n", (unsigned long long)infos[0].count, (unsigned long long)infos[1].count,
           (unsigned long long)common);
    fflush(stdout);
    log(info, elapsed(t0));
}

namespace {

// Decode one object into a fresh context and finish: the sorted (key,count) list in HBM.
void loadObject(const GossCmdContext& cxt, GpuCtx& g, const std::string& name, bool graph, ObjectInfo& o)
{
    o = objectInfo(name, graph);
    const uint64_t maxK = graph ? 62 : 63;
    if (o.K > maxK || o.K == 0) throw Error::General("unable to build a graph with k=" + num(o.K));
    struct stat st;
    uint64_t bytes = 0;
    if (::stat((name + (graph ? "-edges.high-bits" : ".kmers.high-bits")).c_str(), &st) == 0) bytes = (uint64_t)st.st_size * 4;
    if (::stat((name + (graph ? "-counts.ord0" : ".kmers.low-bits")).c_str(), &st) == 0) bytes += (uint64_t)st.st_size * 64;
    const uint64_t budget = cxt.hbmBudget ? cxt.hbmBudget : bytes + (4ULL << 30);
    g.check(goss_gpu_create(&g.h, cxt.device, (uint32_t)o.K, graph ? GOSS_MODE_GRAPH : GOSS_MODE_KMER_SET, budget, nullptr),
            "creating the GPU context");
    if (cxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    pushObject(g, name, graph);
    goss_gpu_counts counts;
    g.check(goss_gpu_finish(g.h, &counts), "decoding");
}

// FileFactory::out(name): "-" is standard output (PhysicalFileFactory.cc:279-298).  The text is
// formatted on the device 32 M elements at a time (about 1 GB) and written as it comes.
void writeText(GpuCtx& g, const std::string& out, uint64_t flags)
{
    FILE* fp = out == "-" ? stdout : fopen(out.c_str(), "wb");
    if (!fp) throw Error::Errno(out, errno);
    uint64_t m = 0;
    g.check(goss_gpu_result(g.h, nullptr, nullptr, &m), "counting");
    const uint64_t step = 32u << 20, piece = 64u << 20;
    std::vector<char> buf((size_t)piece);
    for (uint64_t first = 0; first == 0 || first < m; first += step)
    {
        g.check(goss_gpu_emit_dump_range(g.h, flags, first, std::min(step, m - first)), "formatting");
        uint64_t size = 0; char suffix[32];
        g.check(goss_gpu_file_info(g.h, 0, suffix, sizeof suffix, &size), "listing output files");
        for (uint64_t off = 0; off < size; off += piece)
        {
            uint64_t n = std::min(piece, size - off);
            g.check(goss_gpu_file_read(g.h, 0, off, buf.data(), n), "reading device file");
            if (fwrite(buf.data(), 1, (size_t)n, fp) != n) { if (fp != stdout) fclose(fp); throw Error::Write(out); }
        }
    }
    if (fp == stdout) fflush(stdout);
    else if (fclose(fp) != 0) throw Error::Write(out);
}

std::string edgeText(const uint64_t* w, size_t words, uint64_t len)
{
    std::string s;
    for (uint64_t i = 0; i < len; ++i)
    {
        uint64_t sh = 2 * (len - 1 - i);
        uint64_t v = sh < 64 ? (w[0] >> sh) : (words > 1 ? (w[1] >> (sh - 64)) : 0);
        s += "ACGT"[v & 3];
    }
    return s;
}

}  // namespace

// GossCmdDumpKmerSet::operator() (GossCmdDumpKmerSet.cc:31-55)
void GossCmdDumpKmerSet::operator()(const GossCmdContext& pCxt)
{
    auto t0 = std::chrono::steady_clock::now();
    GpuCtx g; ObjectInfo o;
    loadObject(pCxt, g, mIn, false, o);
    writeText(g, mOut, 0);
    pCxt.log(info, elapsed(t0));
}

// GossCmdDumpGraph::operator() (GossCmdDumpGraph.cc:31-61)
void GossCmdDumpGraph::operator()(const GossCmdContext& pCxt)
{
    GpuCtx g; ObjectInfo o;
    loadObject(pCxt, g, mIn, true, o);
    writeText(g, mOut, o.asymmetric ? 1 : 0);
}

// GossCmdRestoreGraph::operator() (GossCmdRestoreGraph.cc:72-135): "#version" line, then
// "K n flags", then "<edge> <count>" pairs until one does not parse; the graph is built with the
// estimate n of the text's header.
void GossCmdRestoreGraph::operator()(const GossCmdContext& pCxt)
{
    std::string all;
    {
        InFile in(mIn);
        std::vector<char> buf(16u << 20);
        size_t got;
        while ((got = in.read(buf.data(), buf.size())) > 0) all.append(buf.data(), got);
    }
    size_t p = all.find('\n');
    if (p == std::string::npos) { Error e = Error::General("unexpected end of file"); e.file = mIn; throw e; }
    ++p;
    auto skipWs = [&]() { while (p < all.size() && isspace((unsigned char)all[p])) ++p; };
    auto readU64 = [&](uint64_t& v) -> bool {
        skipWs();
        size_t b = p; v = 0;
        while (p < all.size() && isdigit((unsigned char)all[p])) { v = v * 10 + (uint64_t)(all[p] - '0'); ++p; }
        return p > b;
    };
    uint64_t k = 0, n = 0, flags = 0;
    // `in >> k >> n >> flags; if (!in.good())`: the stream must still be good after the flags,
    // i.e. something (at least the newline) follows them
    if (!readU64(k) || !readU64(n) || !readU64(flags) || p >= all.size())
    { Error e = Error::General("unexpected end of file"); e.file = mIn; throw e; }
    if (k > 62 || k == 0) throw Error::General("unable to build a graph with k=" + num(k));
    const bool asymmetric = flags & 1;
    const size_t words = 2 * (k + 1) <= 62 ? 1 : 2;
    std::vector<uint64_t> keys; std::vector<uint32_t> counts;
    for (;;)
    {
        skipWs();
        size_t b = p;
        while (p < all.size() && !isspace((unsigned char)all[p])) ++p;
        if (p == b) break;
        const size_t xl = p - b;
        skipWs();
        size_t cb = p; uint64_t c = 0;
        while (p < all.size() && isdigit((unsigned char)all[p]) && c <= 0xFFFFFFFFULL) { c = c * 10 + (uint64_t)(all[p] - '0'); ++p; }
        // `in >> x >> c; if (!in.good()) break;` -- a count that does not parse, overflows u32,
        // or is the very last token of the stream (eof reached while reading it) ends the loop
        if (p == cb || c > 0xFFFFFFFFULL || p >= all.size()) break;
        if (xl != k + 1) { Error e = Error::General("sequence " + all.substr(b, xl) + " has wrong length"); e.file = mIn; throw e; }
        uint64_t w[2] = {0, 0};
        for (size_t i = 0; i < xl; ++i)
        {
            uint64_t code;
            switch (all[b + i])
            {
                case 'A': case 'a': code = 0; break;
                case 'C': case 'c': code = 1; break;
                case 'G': case 'g': code = 2; break;
                case 'T': case 't': code = 3; break;
                default: throw Error::General("invalid sequence " + all.substr(b, xl));
            }
            w[1] = (w[1] << 2) | (w[0] >> 62);
            w[0] = (w[0] << 2) | code;
        }
        if (!keys.empty())
        {
            const uint64_t* prev = &keys[keys.size() - words];
            bool inc = words == 1 ? prev[0] < w[0] : (prev[1] < w[1] || (prev[1] == w[1] && prev[0] < w[0]));
            if (!inc) throw Error::General("edges must be in strictly increasing order: " + all.substr(b, xl) + "\n");
        }
        keys.push_back(w[0]); if (words == 2) keys.push_back(w[1]);
        counts.push_back((uint32_t)c);
    }
    GpuCtx g;
    const uint64_t budget = pCxt.hbmBudget ? pCxt.hbmBudget : counts.size() * 64 + (4ULL << 30);
    g.check(goss_gpu_create(&g.h, pCxt.device, (uint32_t)k, GOSS_MODE_GRAPH, budget, nullptr), "creating the GPU context");
    if (pCxt.hbmBudget == 0) g.check(goss_gpu_set_budget_limit(g.h, ~0ULL), "setting the HBM limit");
    g.check(goss_gpu_push_run_host(g.h, keys.data(), counts.data(), counts.size()), "loading the edges");
    goss_gpu_counts gc;
    g.check(goss_gpu_finish(g.h, &gc), "loading the edges");
    g.check(goss_gpu_emit_estimate(g.h, n), "building the on-disk arrays");
    writeOut(g, mOut);
    if (asymmetric)
    {
        // Graph::Builder(..., pAsymmetric) sets the flag in the 24-byte header (Graph.cc:162)
        FILE* fp = fopen((mOut + ".header").c_str(), "r+b");
        if (!fp) throw Error::Write(mOut);
        uint64_t f = 1;
        if (fseek(fp, 16, SEEK_SET) != 0 || fwrite(&f, 8, 1, fp) != 1) { fclose(fp); throw Error::Write(mOut); }
        fclose(fp);
    }
}

// ---- `lint-graph --dump-properties`: Graph::stat().print(out, 1) (GossCmdLintGraph.cc:120-133) ----
// The reference's containers describe themselves as a PropertyTree (Properties.hh:31-104: properties first, then the
// sub-trees, both in the order of std::map<std::string,...>, one space of indentation per level, "name<TAB>value").
// Everything in it comes from the headers and the sizes of the object's files:
//   Graph::stat (Graph.hh:588-603), SparseArray::stat (SparseArray.cc:142-162), DenseSelect::stat
//   (DenseArray.cc:94-132), WordyBitVector::stat (WordyBitVector.hh:265-270), IntegerArrayBasic / Stacked::stat
//   (IntegerArray.cc:111-116, 182-187), VariableByteArray::stat (VariableByteArray.hh:249-266).
// One figure cannot be restated: for low-bits widths of 56, 88, 104, 112 and 120 bits the reference's "storage" adds
// sizeof() of a C++ ARRAY OBJECT (StackedArray<...> as the lower type, IntegerArray.cc:374-382) per element instead of
// its bytes; here the bytes of the column files are counted for every width.
struct PropTree {
    std::map<std::string, std::string> props;
    std::map<std::string, PropTree> kids;
    void put(const std::string& k, uint64_t v) { props[k] = num(v); }
    uint64_t get(const std::string& k) const { auto i = props.find(k); return i == props.end() ? 0 : std::strtoull(i->second.c_str(), nullptr, 10); }
    void print(std::vector<std::string>& out, size_t ind) const
    {
        for (auto& kv : props) out.push_back(std::string(ind, ' ') + kv.first + "\t" + kv.second);
        for (auto& kv : kids) { out.push_back(std::string(ind, ' ') + kv.first); kv.second.print(out, ind + 1); }
    }
};

std::string decimal128(uint64_t lo, uint64_t hi)          // BigIntegerBase::writeDecimal (BigInteger.cc:41-110)
{
    unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    if (v == 0) return "0";
    std::string r;
    while (v) { r.push_back((char)('0' + (int)(v % 10))); v /= 10; }
    std::reverse(r.begin(), r.end());
    return r;
}

uint64_t fileBytes(const std::string& name)
{
    struct stat st;
    if (::stat(name.c_str(), &st) != 0) throw Error::Errno(name, errno);
    return (uint64_t)st.st_size;
}

PropTree denseSelectStat(const std::string& name)
{
    Mapped f; f.open(name);
    if (f.n < 128) throw Error::General("\tfile '" + name + "' is too short to be a DenseSelect index\n");
    uint64_t h[16]; memcpy(h, f.p, 128);
    PropTree t;
    t.put("storage", 128 + (uint64_t)f.n);
    t.put("invertSense", h[1] & 1);
    PropTree index, small, inter, large;
    index.put("entries", h[8]); index.put("size", h[9]);
    small.put("entries", h[10]); small.put("size", h[11]);
    inter.put("entries", h[12]); inter.put("size", h[13]);
    large.put("entries", h[14]); large.put("size", h[15]);
    t.kids["index"] = index; t.kids["smallBlocks"] = small; t.kids["intermediateBlocks"] = inter; t.kids["largeBlocks"] = large;
    return t;
}

PropTree sparseStat(const std::string& base)
{
    Mapped hdr; hdr.open(base + ".header");
    if (hdr.n < 64) throw Error::General("\tfile '" + base + ".header' is too short to be a SparseArray header\n");
    uint64_t h[8]; memcpy(h, hdr.p, 64);
    PropTree t, hb, lb;
    hb.put("storage", fileBytes(base + ".high-bits") / 8 * 8);
    std::vector<Col> cols;
    if (!layout((uint32_t)h[2], "", 0, cols)) throw Error::General("IntegerArray::create: unsupported integer width " + num(h[2]));
    uint64_t low = 0;
    for (auto& c : cols) low += fileBytes(base + ".low-bits" + c.suffix) / c.bytes * c.bytes;
    lb.put("storage", low);
    t.kids["high-bits"] = hb; t.kids["low-bits"] = lb;
    t.kids["D0"] = denseSelectStat(base + "-d0");
    t.kids["D1"] = denseSelectStat(base + "-d1");
    t.props["size"] = decimal128(h[5], h[6]);
    t.put("count", h[7]);
    t.put("storage", 64 + hb.get("storage") + lb.get("storage") + t.kids["D0"].get("storage") + t.kids["D1"].get("storage"));
    return t;
}

PropTree graphStat(const std::string& name, uint64_t K)
{
    PropTree t, counts;
    t.kids["edges"] = sparseStat(name + "-edges");
    counts.kids["ord1-pred"] = sparseStat(name + "-counts.ord1p");
    counts.kids["ord2-pred"] = sparseStat(name + "-counts.ord2p");
    const uint64_t o0 = fileBytes(name + "-counts.ord0"), o1 = fileBytes(name + "-counts.ord1"), o2 = fileBytes(name + "-counts.ord2") / 2;
    counts.put("size", o0);
    counts.put("storage", o0 + counts.kids["ord1-pred"].get("storage") + o1 + counts.kids["ord2-pred"].get("storage") + 2 * o2);
    t.kids["counts"] = counts;
    t.put("count", t.kids["edges"].get("count"));
    t.put("K", K);
    t.put("storage", 24 + t.kids["edges"].get("storage") + counts.get("storage"));
    return t;
}

// GossCmdLintGraph::operator() (GossCmdLintGraph.cc:110-275).  Pass 1 runs on the device over the
// decoded edge list; pass 2 (iterator against select / rank) becomes the strict-ordering check of
// the same list.  At most 32 offending edges are printed, in edge order.
void GossCmdLintGraph::operator()(const GossCmdContext& pCxt)
{
    Logger& log = pCxt.log;
    GpuCtx g; ObjectInfo o;
    loadObject(pCxt, g, mIn, true, o);
    uint64_t m = 0;
    g.check(goss_gpu_result(g.h, nullptr, nullptr, &m), "counting");
    if (mDumpProperties)
    {
        // (the reference prints the tree into a string and logs it line by line until eof: one empty line at the end)
        std::vector<std::string> lines;
        graphStat(mIn, o.K).print(lines, 1);
        log(info, "Graph properties:");
        for (auto& l : lines) log(info, l);
        log(info, "");
    }
    log(info, "Pass 1: Checking counts are sane.");
    goss_gpu_lint_report rep;
    g.check(goss_gpu_lint(g.h, o.asymmetric ? 1 : 0, &rep), "checking");
    const size_t words = 2 * (o.K + 1) <= 62 ? 1 : 2;
    std::vector<uint32_t> order(rep.nexamples);
    for (uint32_t i = 0; i < rep.nexamples; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rep.ex_index[a] < rep.ex_index[b]; });
    auto edgeAt = [&](uint64_t idx, uint32_t& mult) {
        uint64_t w[2] = {0, 0};
        g.check(goss_gpu_result_copy(g.h, idx, 1, w, &mult), "reading an edge");
        return edgeText(w, words, o.K + 1);
    };
    for (uint32_t j : order)
    {
        const uint64_t i = rep.ex_index[j];
        uint32_t mult = 0, mp = 0;
        std::string fwd = edgeAt(i, mult);
        switch (rep.ex_kind[j])
        {
            case 1:
                log(warning, "No reverse complement for the following edge exists:");
                log(warning, "  edge number " + num(i));
                log(warning, "  fwd edge    " + fwd + " " + num(mult));
                break;
            case 2:
            {
                std::string rev = edgeAt(rep.ex_other[j], mp);
                log(warning, o.asymmetric ? "neither fwd nor rev edge has a nonzero multiplicity:" : "counts on fwd and rev edges are not equal:");
                log(warning, "  edge number " + num(i));
                log(warning, "  fwd edge    " + fwd + " " + num(mult));
                log(warning, "  rev edge    " + rev + " " + num(mp));
                if (!o.asymmetric && mult == 0) log(warning, num(mult) + " !> 0");
                break;
            }
            case 3:
                log(warning, num(mult) + " !> 0");
                break;
            default: break;
        }
    }
    log(info, "Pass 2: Checking traversal is sane.");
    for (uint32_t j : order)
        if (rep.ex_kind[j] == 4)
        {
            uint32_t mult = 0;
            log(warning, "iterator and rank conflict.");
            log(warning, "    " + edgeAt(rep.ex_index[j], mult));
            log(warning, "iterator: " + num(rep.ex_index[j]));
        }
    // the object's own select / rank structures against the decoded list (GossCmdLintGraph.cc:201-243)
    goss_gpu_index_report irep{};
    {
        SparseFiles s;
        openSparse(mIn + "-edges", s);
        Mapped hdr, d0, d1;
        hdr.open(mIn + "-edges.header"); d0.open(mIn + "-edges-d0"); d1.open(mIn + "-edges-d1");
        uint64_t h[8]; memcpy(h, hdr.p, 64);
        goss_gpu_sparse_files f{};
        f.D = s.D; f.count = s.count; f.size_lo = h[5]; f.size_hi = h[6];
        f.high_bits = (const uint64_t*)s.high.p; f.high_words = s.high.n / 8;
        f.d0 = d0.p; f.d0_bytes = d0.n; f.d1 = d1.p; f.d1_bytes = d1.n;
        f.ncols = (uint32_t)s.cols.size();
        for (size_t i = 0; i < s.cols.size(); ++i)
        { f.col[i] = s.colFiles[i].p; f.col_bytes[i] = s.cols[i].bytes; f.col_shift[i] = s.cols[i].shift; }
        g.check(goss_gpu_check_index(g.h, &f, &irep), "checking the select / rank indexes");
    }
    {
        std::vector<uint32_t> io(irep.nexamples);
        for (uint32_t i = 0; i < irep.nexamples; ++i) io[i] = i;
        std::sort(io.begin(), io.end(), [&](uint32_t a, uint32_t b) { return irep.ex_index[a] < irep.ex_index[b]; });
        for (uint32_t j : io)
        {
            uint32_t mult = 0;
            const std::string e = edgeAt(irep.ex_index[j], mult);
            if (irep.ex_kind[j] == 1) { log(warning, "iterator and select conflict."); log(warning, "    " + e); }
            else
            {
                log(warning, "iterator and rank conflict.");
                log(warning, "    " + e);
                log(warning, "iterator: " + num(irep.ex_index[j]));
            }
        }
    }
    const uint64_t indexProblems = irep.select_mismatch + irep.rank_mismatch + irep.access_miss + irep.failures;
    if (indexProblems > irep.nexamples)
        log(warning, num(indexProblems) + " index problems in total (" + num(irep.select_mismatch) + " select, "
                     + num(irep.rank_mismatch) + " rank, " + num(irep.access_miss) + " lookups that miss, "
                     + num(irep.failures) + " walks off the index)");
    mProblems = rep.missing_rc + rep.count_mismatch + rep.zero_count + rep.order_violation + indexProblems;
    if (mProblems > rep.nexamples)
        log(warning, num(mProblems) + " problems in total (" + num(rep.missing_rc) + " missing reverse complements, "
                     + num(rep.count_mismatch) + " unequal counts, " + num(rep.zero_count) + " zero counts, "
                     + num(rep.order_violation) + " out of order); the first " + num(rep.nexamples) + " found are shown");
}

void GossCmdMergeKmerSets::operator()(const GossCmdContext& pCxt) { runMerge(pCxt, false, mIns, mMaxMerge, mOut); }
void GossCmdMergeGraphs::operator()(const GossCmdContext& pCxt) { runMerge(pCxt, true, mIns, mMaxMerge, mOut); }

}  // namespace gosshost
