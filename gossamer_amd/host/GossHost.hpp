// GossHost.hpp -- host-side C++ for the `goss build-kmer-set` / `goss build-graph` commands:
// errors, logger, input files, the three read parsers, and the command classes that mirror
// the reference's operator interface (GossCmdBuildKmerSet.hh:25-38, GossCmdBuildGraph.hh:23-29).
//
// The parsers frame records exactly like the reference (FastqParser.hh:78-176,
// FastaParser.hh:51-87, LineParser.hh:71-82 over PlainLineSource, LineSource.cc:17-48) but
// emit only what the device needs: the read's bases followed by '\n', appended to a batch
// buffer that is handed to libgossgpu.so (include/goss_gpu.h).
#pragma once

#include <cstdint>
#include <cstdio>
#include <ctime>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace gosshost {

// ---- errors (role of Gossamer::error + boost::error_info tags, GossamerException.hh:27-40) ----
struct Error {
    std::string general;       // general_error_info
    std::string usage;         // usage_info
    std::string parse;         // parse_error_info
    std::string file;          // errinfo_file_name
    std::string write_name;    // write_error_info
    std::string cmd;           // cmd_name_info
    int err_no = 0;            // errinfo_errno

    static Error General(const std::string& m) { Error e; e.general = m; return e; }
    static Error Usage(const std::string& m) { Error e; e.usage = m; return e; }
    static Error Parse(const std::string& file, const std::string& m) { Error e; e.file = file; e.parse = m; return e; }
    static Error Write(const std::string& name) { Error e; e.write_name = name; return e; }
    static Error Errno(const std::string& file, int eno) { Error e; e.file = file; e.err_no = eno; return e; }
};

// ---- logger (Logger.hh:62-96: "<asctime>\t<severity>\t<message>") ----
enum Severity { info = 0, warning = 1, error = 2 };

class Logger {
public:
    Logger(FILE* out, Severity sev, bool owns = false) : mOut(out), mSev(sev), mOwns(owns) {}
    ~Logger() { if (mOwns && mOut) fclose(mOut); }
    void operator()(Severity sev, const std::string& msg)
    {
        if (mSev > sev) return;
        time_t rt; time(&rt);
        std::string now = asctime(localtime(&rt));
        now.erase(now.size() - 1);
        static const char* names[] = {"info", "warning", "error"};
        fprintf(mOut, "%s\t%s\t%s\n", now.c_str(), names[sev], msg.c_str());
        fflush(mOut);
    }
    Severity sev() const { return mSev; }
private:
    FILE* mOut; Severity mSev; bool mOwns;
};

// ---- input files: plain, .gz (zlib), "-" = stdin (PhysicalFileFactory.cc:262-298) ----
class InFile {
public:
    explicit InFile(const std::string& name);      // throws Error
    ~InFile();
    // read up to cap bytes; 0 at end of file
    size_t read(char* dst, size_t cap);
    const std::string& name() const { return mName; }
    static bool readable(const std::string& name);
private:
    std::string mName;
    void* mGz = nullptr;
    void* mBz = nullptr;            // BZFILE* of the stream being read (libbz2, loaded at run time)
    void* mBzFile = nullptr;        // FILE* under it
    bool mBzEnd = false;
    unsigned mBzStreams = 0;        // complete streams decoded so far
    bool mBzFresh = true;           // nothing decoded from the current stream yet
    int mFd = -1;
    bool mStdin = false;
};

// ---- line source: std::getline semantics over a large refillable buffer ----
class LineReader {
public:
    explicit LineReader(const std::string& name, size_t bufBytes = 16u << 20);
    // PlainLineSource::valid(): stream good, or a non-empty last line without '\n'
    bool valid() const { return mGood || mLen; }
    const char* line() const { return mLine; }
    size_t len() const { return mLen; }
    void next();                                  // operator++
    const std::string& name() const { return mFile.name(); }
    uint64_t offset() const { return 0; }         // streams have no record-boundary limit
    static constexpr bool kStableLines = false;   // line() is only valid until next()
private:
    void getline();
    bool refill();
    InFile mFile;
    std::vector<char> mBuf;
    size_t mBeg = 0, mEnd = 0;
    bool mEof = false, mGood = true;
    const char* mLine = nullptr;
    size_t mLen = 0;
    std::string mSpill;                           // a line that straddled a refill
};

// A sink for reads: called with the bases of one read (no terminator).
using ReadSink = std::function<void(const char* seq, size_t len)>;

// Parse a whole file; every read goes to `sink`.  Returns the number of reads.
uint64_t parseFastq(const std::string& name, const ReadSink& sink);
uint64_t parseFasta(const std::string& name, const ReadSink& sink);
uint64_t parseLines(const std::string& name, const ReadSink& sink);

// ---- command context (GossCmdContext.hh:25-39, minus the FileFactory: files are real) ----
struct GossCmdContext {
    Logger& log;
    std::string cmdName;
    int device = 0;                 // HIP device ordinal
    std::vector<int> devices = {};  // --devices a,b,..: the build commands count on all of them (one context each)
    uint64_t hbmBudget = 0;         // bytes; 0 = library default (80% of free HBM)
    size_t batchBytes = 256u << 20; // bases handed to the device per push
    // --tmp-dir (App.cc:233-240, PhysicalFileFactory::tmpName): where temporary files go.  Given: the partial results of a
    // merge with more inputs than --max-merge are written there as runs and removed by the command (the reference's
    // temporary objects, GossCmdMerge.tcc:176-208); not given: they stay in host memory.  The build commands need no
    // temporary file: the runs of their chunks are held in HBM.
    std::string tmpDir = {};
};

typedef std::vector<std::string> strings;

struct BuildStats { uint64_t reads = 0, windows = 0, keys = 0, distinct = 0; double seconds = 0; };

// Same constructor arguments as the reference (K, S, N, T, out, fastas, fastqs, lines).
// S and N size the reference's BackyardHash, whose layout never reaches disk: they are
// accepted and ignored.  T = host parser threads.
class GossCmdBuildKmerSet {
public:
    GossCmdBuildKmerSet(const uint64_t& pK, const uint64_t& pS, const uint64_t& pN, const uint64_t& pT,
                        const std::string& pKmerSetName, const strings& pFastaNames,
                        const strings& pFastqNames, const strings& pLineNames)
        : mK(pK), mS(pS), mN(pN), mT(pT), mKmerSetName(pKmerSetName),
          mFastaNames(pFastaNames), mFastqNames(pFastqNames), mLineNames(pLineNames) {}
    void operator()(const GossCmdContext& pCxt);
    const BuildStats& stats() const { return mStats; }
private:
    const uint64_t mK, mS, mN, mT;
    const std::string mKmerSetName;
    const strings mFastaNames, mFastqNames, mLineNames;
    BuildStats mStats;
};

class GossCmdBuildGraph {
public:
    GossCmdBuildGraph(const uint64_t& pK, const uint64_t& pS, const uint64_t& pN, const uint64_t& pT,
                      const std::string& pGraphName, const strings& pFastaNames,
                      const strings& pFastqNames, const strings& pLineNames)
        : mK(pK), mS(pS), mN(pN), mT(pT), mGraphName(pGraphName),
          mFastaNames(pFastaNames), mFastqNames(pFastqNames), mLineNames(pLineNames) {}
    void operator()(const GossCmdContext& pCxt);
    const BuildStats& stats() const { return mStats; }
private:
    const uint64_t mK, mS, mN, mT;
    const std::string mGraphName;
    const strings mFastaNames, mFastqNames, mLineNames;
    BuildStats mStats;
};

// merge-kmer-sets / merge-graphs: GossCmdMerge<T>(ins, maxMerge, out) (GossCmdMerge.hh:22-37).
class GossCmdMergeKmerSets {
public:
    GossCmdMergeKmerSets(const strings& pIns, const uint64_t& pMaxMerge, const std::string& pOut)
        : mIns(pIns), mMaxMerge(pMaxMerge), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const strings mIns; const uint64_t mMaxMerge; const std::string mOut;
};

class GossCmdMergeGraphs {
public:
    GossCmdMergeGraphs(const strings& pIns, const uint64_t& pMaxMerge, const std::string& pOut)
        : mIns(pIns), mMaxMerge(pMaxMerge), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const strings mIns; const uint64_t mMaxMerge; const std::string mOut;
};

// Set algebra on existing k-mer sets: GossCmdIntersectKmerSets.hh:23, GossCmdSubtractKmerSet.hh:23,
// GossCmdMergeAndAnnotateKmerSets.hh:21 (same constructor arguments).
class GossCmdIntersectKmerSets {
public:
    GossCmdIntersectKmerSets(const strings& pIns, const std::string& pOut) : mIns(pIns), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const strings mIns; const std::string mOut;
};

class GossCmdSubtractKmerSet {
public:
    GossCmdSubtractKmerSet(const strings& pIns, const std::string& pOut) : mIns(pIns), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const strings mIns; const std::string mOut;
};

// GossCmdGraphToKmerSet (GossCmdGraphToKmerSet.{hh,cc}): the normal-form edges of a graph as a
// k-mer set of k = K + 1.
class GossCmdGraphToKmerSet {
public:
    GossCmdGraphToKmerSet(const std::string& pIn, const std::string& pOut) : mIn(pIn), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const std::string mIn, mOut;
};

class GossCmdMergeAndAnnotateKmerSets {
public:
    GossCmdMergeAndAnnotateKmerSets(const std::string& pLhs, const std::string& pRhs, const std::string& pOut)
        : mLhs(pLhs), mRhs(pRhs), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const std::string mLhs, mRhs, mOut;
};

// Text form and self-check of existing objects: GossCmdDumpKmerSet.hh, GossCmdDumpGraph.hh,
// GossCmdRestoreGraph.hh (in = text file, out = graph), GossCmdLintGraph.hh.  "-" = stdin/stdout.
class GossCmdDumpKmerSet {
public:
    GossCmdDumpKmerSet(const std::string& pIn, const std::string& pOut) : mIn(pIn), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const std::string mIn, mOut;
};

class GossCmdDumpGraph {
public:
    GossCmdDumpGraph(const std::string& pIn, const std::string& pOut) : mIn(pIn), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const std::string mIn, mOut;
};

class GossCmdRestoreGraph {
public:
    GossCmdRestoreGraph(const std::string& pIn, const std::string& pOut) : mIn(pIn), mOut(pOut) {}
    void operator()(const GossCmdContext& pCxt);
private:
    const std::string mIn, mOut;
};

class GossCmdLintGraph {
public:
    GossCmdLintGraph(const std::string& pIn, bool pDumpProperties) : mIn(pIn), mDumpProperties(pDumpProperties) {}
    void operator()(const GossCmdContext& pCxt);
    uint64_t problems() const { return mProblems; }
private:
    const std::string mIn; const bool mDumpProperties; uint64_t mProblems = 0;
};

}  // namespace gosshost
struct goss_gpu_ctx;
namespace gosshost {

// Writes every file image of the emitted object to "<out><suffix>" (the write half of the
// reference's FileFactory for KmerSet::Builder / Graph::Builder::end()).
void writeObjectFiles(goss_gpu_ctx* h, const std::string& out);
void writeObjectFiles(const std::vector<goss_gpu_ctx*>& hs, const std::string& out);      // a group after goss_gpu_group_emit

// App::main for the commands of this build (App.cc:176-417).
int gossMain(int argc, char* argv[]);

// Gives back what the parser parked between files (page-locked buffers; registered slabs are unregistered first): the
// orderly exit's share of the tear-down (`goss` itself ends with _exit once its files are closed).
void releaseKeptBuffers();

}  // namespace gosshost
