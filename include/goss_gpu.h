/*
 * goss_gpu.h -- C ABI of libgossgpu.so: the MI355X (gfx950) implementation of gossamer's
 * k-mer counting / de Bruijn edge-set construction hot path.
 *
 * The reference (data61/gossamer) has no FFI; this is the seam a maintainer would bind from
 * GossCmdBuildKmerSet / GossCmdBuildGraph.  Each entry point names the
 * reference code whose role it takes over (paths relative to the reference's src/).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 (GOSS_OK) or a
 * negative goss_status; no exceptions cross the boundary; one host thread per context
 * (the library itself runs one more: a full staging buffer of host pushes is counted by a
 * thread of its own while the caller fills a second buffer -- every other entry point waits
 * for that thread first and reports its failure as its own);
 * the library fails loudly (GOSS_ERR_NO_DEVICE) when no gfx950 device is usable -- there is
 * no CPU fallback behind this ABI.
 */
#ifndef GOSS_GPU_H
#define GOSS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct goss_gpu_ctx goss_gpu_ctx;

typedef enum {
    GOSS_OK               = 0,
    GOSS_ERR_INVALID_ARG  = -1,
    GOSS_ERR_NO_DEVICE    = -2,   /* no HIP device / not gfx950 / runtime failure at create */
    GOSS_ERR_OOM          = -3,   /* HBM budget exceeded */
    GOSS_ERR_HIP          = -4,   /* a HIP call failed; see goss_gpu_last_error */
    GOSS_ERR_STATE        = -5,   /* call out of order (e.g. push after finish) */
    GOSS_ERR_K_RANGE      = -6,   /* "unable to build a graph with k=<K>" (KmerSet.hh:89-95, Graph.cc:152-158) */
    GOSS_ERR_COUNT_OVERFLOW = -7, /* graph mode: more than 256 keys occurred >= 2^32 - 1 times each (up to that many are
                                     kept exactly, goss_gpu_big_counts) */
    GOSS_ERR_TOO_LARGE    = -8,   /* high-bits value does not fit 64 bits (SparseArray.hh:91-95) */
    GOSS_ERR_BUFFER       = -9    /* a buffer supplied by the caller is too small; the sizes needed were returned */
} goss_status;

/* mode: which reference command's key stream is produced. */
enum {
    GOSS_MODE_KMER_SET = 0,  /* canonical k-mers: KmerizingAdapter + position_type::normalize
                                (KmerizingAdapter.hh:20-86, RankSelect.hh:126-140) */
    GOSS_MODE_GRAPH    = 1   /* every (k+1)-mer and its reverse complement, un-normalised:
                                ReverseComplementAdapter (ReverseComplementAdapter.hh:20-93) */
};

/* Human readable text for a status code. */
const char* goss_gpu_strerror(int status);

/* Detail of the last failure on this context (HIP error string etc.). */
const char* goss_gpu_last_error(const goss_gpu_ctx* ctx);

/* ABI version (bumped on incompatible change). */
uint32_t goss_gpu_abi_version(void);

/*
 * Create a counting context on HIP device `device`.
 * Takes over: BackyardHash h(S, 2K, N) + the consumer threads
 * (GossCmdBuildKmerSet.tcc:226-233, GossCmdBuildGraph.cc:315-322).
 *   k           the command's -k (KmerSet: 1..63, Graph: 1..62; Graph keys are (k+1)-mers)
 *   mode        GOSS_MODE_*
 *   hbm_budget  bytes of HBM the context may use for keys/sort workspace (0 = 80% of free)
 *   stream      a hipStream_t to run on, or NULL for a stream owned by the context
 */
int goss_gpu_create(goss_gpu_ctx** ctx, int device, uint32_t k, int mode,
                    uint64_t hbm_budget, void* stream);

void goss_gpu_destroy(goss_gpu_ctx* ctx);

/*
 * Start mapping the context's HBM arena in the background and return at once; the first call that
 * needs the arena waits for it (and reports its failure).  Mapping HBM costs up to 30 ms/GB when the
 * driver has to clear pages a previous process left behind: a caller that still has input to read
 * and parse -- the goss commands -- hides that time behind its own start-up.  Without this call the
 * arena is mapped by the first push.  No reference counterpart (BackyardHash allocates its table in
 * its constructor, BackyardHash.cc:273-285).
 */
int goss_gpu_prepare(goss_gpu_ctx* ctx);

/*
 * Feed read bases.  `bases` is a byte string in which A/C/G/T (either case) are bases and
 * ANY other byte ends the current run of k-windows -- the caller separates reads with one
 * such byte (the host parsers emit '\n').  Windows never span two push calls.
 * Takes over: GossRead::Iterator / GossReadBaseString::{firstKmer,nextKmer,getBase,getEdge}
 * (GossRead.hh:57-114, GossReadBaseString.hh:52-188) and the insert loop
 * (GossCmdBuildKmerSet.tcc:246-256 + BackyardHash::insert BackyardHash.cc:115-242).
 * _host: bytes in host memory (pinned memory makes the copy asynchronous).  Host pushes are gathered in one of two
 *        staging buffers in device memory beside the arena (1/24 of the arena each) by copies on a stream of the
 *        context's own; a full buffer is counted in the background while the next pushes fill the other;
 * _device: bytes already resident in HBM on the context's device (not modified).  The context
 *          works on its own stream and does not wait for any other: whatever produced the bytes
 *          (a copy, a kernel on another stream) must have completed before the call.
 */
int goss_gpu_push_bases_host(goss_gpu_ctx* ctx, const char* bases, uint64_t nbytes);
int goss_gpu_push_bases_device(goss_gpu_ctx* ctx, const void* d_bases, uint64_t nbytes);

/*
 * Asynchronous form of goss_gpu_push_bases_host: returns as soon as the copy is queued.  The buffer belongs to the
 * library until release(user) is called -- by the CALLER'S OWN THREAD, from inside a later call on this context
 * (another push, goss_gpu_flush, finish, reset, destroy), once the bytes have left it: release never runs on a thread
 * of the library (the library does keep ONE thread of its own per context, which counts a full staging buffer while the
 * caller fills the other; it touches no buffer of the caller).  OWNERSHIP ON FAILURE: when the call returns anything but
 * GOSS_OK the library does not call release for this push, now or later -- the buffer is the caller's again when the
 * call returns (copies already queued from it have completed by then); buffers of EARLIER pushes are still released
 * by later calls, and by goss_gpu_destroy at the latest.  The
 * buffer should be page-locked (goss_gpu_host_alloc), else the copy is synchronous inside the driver.  With a
 * pool of such buffers the parser threads keep filling while earlier batches cross PCIe: the per-push wait of the
 * synchronous form is what bounded `goss build-kmer-set` on a large FASTQ file.  release may be NULL (the caller
 * then finds out through goss_gpu_flush).  Same role as the synchronous form (BackgroundMultiConsumer<KmerBlockPtr>
 * ::push_back, GossCmdBuildKmerSet.tcc:246-256, is likewise a hand-over of a block the producer does not touch again).
 */
typedef void (*goss_gpu_release_fn)(void* user);
int goss_gpu_push_bases_host_async(goss_gpu_ctx* ctx, const char* bases, uint64_t nbytes, goss_gpu_release_fn release, void* user);

/*
 * 2-bit packed bases: codes = one u32 per 16 positions, position j of a group at bits [2j, 2j+2), A=0 C=1 G=2 T=3;
 * nonbase = one u16 per 16 positions, bit j set where position j is NOT a base (read separators, N, ...: it ends the
 * run of windows like any non-ACGT byte of the byte form).  nbases positions; the upper bits of a partial last
 * group are ignored.  Packed by the host's parser threads this moves 3 bits per base over PCIe instead of 8
 * (north_star: "2-bit read encoding", "coalesced HBM loads of packed reads"; the per-base encoder it stands for is
 * GossReadBaseString.hh:133-188); on the device the groups stay as they are -- the staging buffer keeps codes and flags,
 * and the kernels that read bases (the fused extraction, its sample, the routing kernels of a group's exchange) load
 * them straight: 3 bits per base in HBM too, nothing is unpacked (the plain kernels that small inputs and the fallback
 * sequence go through read bytes: such a chunk is unpacked for them; goss_gpu_stat "packed_fused_chunks" /
 * "packed_unpacked_chunks" count the two).  A staging buffer holds bytes or packed groups: a push of the other form
 * counts what is staged first (a deferred context refuses it with GOSS_ERR_BUFFER, like a push that does not fit).
 * Windows never span two pushes.  _async: as goss_gpu_push_bases_host_async,
 * both arrays belong to the library until release(user).  A caller that recycles a bounded pool of such buffers must not
 * wait for a free one (or for the thread that would fill it) without calling the library: release only ever runs inside a
 * call on this context -- goss_gpu_flush is the one that makes every outstanding release happen (the `goss` parser's
 * consumer does so while buffers are out and none is free; waiting without it hung one build in four).
 */
int goss_gpu_push_packed_host(goss_gpu_ctx* ctx, const uint32_t* codes, const uint16_t* nonbase, uint64_t nbases);
int goss_gpu_push_packed_host_async(goss_gpu_ctx* ctx, const uint32_t* codes, const uint16_t* nonbase, uint64_t nbases,
                                    goss_gpu_release_fn release, void* user);
/*
 * Packed bases already resident in HBM on the context's device (not modified; ceil(nbases / 16) elements per array,
 * positions at or beyond nbases in the last group are ignored): counted where they lie, as goss_gpu_push_bases_device
 * counts bytes -- the same kernels, fetching 3 bits per base instead of 8 and encoding nothing.  Whatever produced the
 * arrays must have completed before the call.  288 GB of HBM hold 2.7 x the reads in this form; bench.py reports the
 * headline workload in both forms.  Same role as goss_gpu_push_bases_device (GossRead::Iterator + the insert loop,
 * GossRead.hh:57-114, GossCmdBuildKmerSet.tcc:246-256).
 * goss_gpu_pack_bases_device: the byte form (any byte that is not ACGTacgt a non-base) -> the packed form in the
 * caller's two arrays, on the context's stream; returns when they are written.  GossReadBaseString's per-base encoder
 * (GossReadBaseString.hh:133-188) as an entry point of its own.
 */
int goss_gpu_push_packed_device(goss_gpu_ctx* ctx, const uint32_t* d_codes, const uint16_t* d_nonbase, uint64_t nbases);
int goss_gpu_pack_bases_device(goss_gpu_ctx* ctx, const void* d_bases, uint64_t nbytes, uint32_t* d_codes, uint16_t* d_nonbase);
/*
 * How many bases the caller means to push in all before goss_gpu_finish (0 = unknown; forgotten by goss_gpu_reset).
 * A hint, never needed for the result: a build that is counted in several chunks merges their runs before it re-orders
 * them into the reference's canonical order, and which key space the chunks count in (strand representatives, or the
 * FNV-ordered form per window: DESIGN.md section 3) is then a question of the WHOLE build's distinct keys per window,
 * which its first chunk cannot see (thirteen chunks of C2 each hold all 10^8 k-mers of the genome).  The reference
 * sizes its hash table from -B / -S the same way (GossCmdBuildKmerSet.cc:151-215): from what the caller says.
 */
int goss_gpu_expect_bases(goss_gpu_ctx* ctx, uint64_t total_bases);
/* Wait for every queued copy and hand all buffers of asynchronous pushes back (release is called for each). */
int goss_gpu_flush(goss_gpu_ctx* ctx);

typedef struct {
    uint64_t windows;    /* valid k-windows seen (kmer-set: k-mers; graph: rho-mer windows) */
    uint64_t keys;       /* keys inserted (= windows, or 2*windows in graph mode) */
    uint64_t distinct;   /* M: distinct keys */
    uint32_t key_words;  /* 1: keys are u64; 2: keys are {lo,hi} u64 pairs */
    uint32_t reserved;
} goss_gpu_counts;

/*
 * Sort + merge equal keys, summing counts.
 * Takes over: BackyardHash::sort + BlendedSort (BackyardHash.cc:244-271, BlendedSort.hh:58-167)
 * and the run-summing walk of flush() (GossCmdBuildKmerSet.tcc:183-202,
 * GossCmdBuildGraph.cc:238-257).  After this call no more bases may be pushed.
 */
int goss_gpu_finish(goss_gpu_ctx* ctx, goss_gpu_counts* out);

/* Device-resident result: sorted distinct keys (key_words u64 per key) and u32 counts.
 * The pointers are valid until the next emit, select, reset, push or destroy on this context
 * (an emit may move the arena when goss_gpu_set_budget_limit allows it to grow; a reset
 * recycles the memory).  Either pointer may be NULL. */
int goss_gpu_result(goss_gpu_ctx* ctx, const void** d_keys, const uint32_t** d_counts,
                    uint64_t* distinct);

/*
 * Graph mode: the keys whose count does not fit the u32 array (2^32 - 1 occurrences or more), with their
 * exact counts.  In the result (and in the emitted VariableByteArray) such a key carries its count modulo
 * 2^32 -- what the reference stores when Graph::Builder::push_back narrows its u64 count
 * (Graph.hh:101-106, VariableByteArray.hh:72,81) -- while "-counts-hist.txt" is keyed by the exact count,
 * as the reference's histogram is.  keys: 2 u64 (lo, hi) per entry; at most `cap` entries are written,
 * *n = how many there are.  Always 0 for a k-mer set (no counts are stored).
 */
int goss_gpu_big_counts(goss_gpu_ctx* ctx, uint64_t* keys, uint64_t* counts, uint32_t cap, uint32_t* n);

/* Copy a slice [first, first+n) of the result to host memory (keys: n*key_words u64). */
int goss_gpu_result_copy(goss_gpu_ctx* ctx, uint64_t first, uint64_t n,
                         uint64_t* h_keys, uint32_t* h_counts);

/*
 * Build every on-disk array of the output object on the device.
 * kmer-set: KmerSet::Builder (KmerSet.hh:64-103) = SparseArray at "<out>.kmers".
 * graph:    Graph::Builder (Graph.hh:101-127, Graph.cc:115-167) = SparseArray "<out>-edges",
 *           VariableByteArray "<out>-counts", "<out>-counts-hist.txt".
 * Covers SparseArray::Builder (SparseArray.hh:87-118, SparseArray.cc:47-131),
 * WordyBitVector::Builder (WordyBitVector.hh:54-134), DenseSelect::Builder
 * (DenseArray.cc:446-694), IntegerArray::builder (IntegerArray.cc:259-357) and
 * VariableByteArray::Builder (VariableByteArray.hh:76-118).
 * The SparseArray estimate M is the exact distinct count (single-pass reference result).
 */
int goss_gpu_emit(goss_gpu_ctx* ctx);

/* The emitted object as a list of files: name suffix (appended to the -O prefix), size. */
int goss_gpu_file_count(goss_gpu_ctx* ctx, uint32_t* n);
int goss_gpu_file_info(goss_gpu_ctx* ctx, uint32_t i, char* suffix, size_t suffix_cap,
                       uint64_t* size);
/* Copy bytes [offset, offset+n) of file i into host memory. */
int goss_gpu_file_read(goss_gpu_ctx* ctx, uint32_t i, uint64_t offset, void* dst, uint64_t n);

/*
 * Stand-alone SparseArray build from caller-supplied sorted positions on the device
 * (SparseArray::Builder(base, fac, N, M) + push_back* + end(N_end)).  Positions are
 * key_words u64 each, strictly increasing.  N is given as {lo,hi}.  Produces the files
 * ".header", ".high-bits", "-d0", "-d1", ".low-bits*" retrievable through goss_gpu_file_*.
 * Not while a count is in progress (bases or runs pushed and not finished): GOSS_ERR_STATE.
 */
int goss_gpu_emit_sparse_array(goss_gpu_ctx* ctx, const void* d_positions, uint32_t key_words,
                               uint64_t n, uint64_t N_lo, uint64_t N_hi, uint64_t M,
                               uint64_t Nend_lo, uint64_t Nend_hi);

/* Device time per kernel class on this context since the last reset, measured with HIP events
 * recorded on the context's stream around every launch (for bench.py's roofline block).
 * units[] = items processed by the launches of that class (keys for the sort kernels,
 * window starts for extraction, distinct keys for emit). */
enum {
    GOSS_T_EXTRACT = 0,   /* extraction: extract1/2_kernel, and extract1/2_part_kernel (extraction fused
                             with the first partition level) */
    GOSS_T_HIST    = 1,   /* digit histograms: global_hist_kernel, radix_hist_kernel */
    GOSS_T_SCAN    = 2,   /* scans over digit tables */
    GOSS_T_SCATTER = 3,   /* partition passes: radix_onesweep_kernel (look-back, cursor and sub-region
                             forms), radix_scatter_kernel */
    GOSS_T_REDUCE  = 4,   /* counting: seg_hash_reduce*_kernel + seg_gather_kernel, seg_merge_kernel,
                             heads_* / run_* compaction after a full sort */
    GOSS_T_EMIT    = 5,   /* Elias-Fano / DenseSelect / VariableByteArray image build */
    GOSS_T_ORDER   = 6,   /* strand representatives -> canonical forms + re-ordering of the distinct keys
                             (canonical_map_kernel and the radix passes over the (key,count) pairs) */
    GOSS_T_CLASSES = 8
};
typedef struct {
    float    ms[GOSS_T_CLASSES];
    uint32_t launches[GOSS_T_CLASSES];
    uint64_t units[GOSS_T_CLASSES];
} goss_gpu_timing;
int goss_gpu_timing_get(goss_gpu_ctx* ctx, goss_gpu_timing* out);
int goss_gpu_timing_reset(goss_gpu_ctx* ctx);

/*
 * Forget everything counted so far but keep the HBM arena, the stream and (k, mode):
 * the context can then be used for a new build.  Role of BackyardHash::clear()
 * (GossCmdBuildKmerSet.tcc:288).
 */
int goss_gpu_reset(goss_gpu_ctx* ctx);

/*
 * Feed an existing object as a run: a SparseArray in its on-disk form (host pointers to the
 * high-bits words and the low-bits column files, D and count from its 64-byte header), decoded on
 * the device the way SparseArray::LazyIterator walks it (SparseArray.hh:185-224), with one u32
 * count per key or NULL for all ones (KmerSet::LazyIterator, KmerSet.hh:147-150).  This is how
 * merge-kmer-sets / merge-graphs read their inputs (GossCmdMerge.tcc:52-69).
 */
typedef struct {
    uint64_t D;
    uint64_t count;
    const uint64_t* high_bits;
    uint64_t high_words;
    uint32_t ncols;
    uint32_t weight;             /* count given to every key when counts is NULL; 0 means 1 */
    const void* col[4];          /* low-bits column files */
    uint32_t col_bytes[4];       /* element size of each column file */
    uint32_t col_shift[4];       /* bit position of the column inside the low bits */
    const uint32_t* counts;      /* count entries, or NULL */
} goss_gpu_sparse_run;
int goss_gpu_push_run_sparse(goss_gpu_ctx* ctx, const goss_gpu_sparse_run* run);

/*
 * Feed an existing GRAPH as a run: its edge SparseArray as above and its multiplicities as the
 * VariableByteArray files they are stored in -- read on the device the way VariableByteArray::operator[] /
 * GeneralIterator read them (VariableByteArray.hh:120-247): ord0 holds byte 0 of every value, the items that
 * the ord1p SparseArray lists take bits 8..15 from ord1, the entries of that list which ord2p names take bits
 * 16..31 from ord2.  How merge-graphs reads its inputs (GossCmdMerge.tcc:52-69, Graph::LazyIterator).
 */
typedef struct {
    const uint8_t* ord0;  uint64_t ord0_bytes;
    goss_gpu_sparse_run ord1p;                 /* positions (universe = item count) of the items with a second byte */
    const uint8_t* ord1;  uint64_t ord1_bytes;
    goss_gpu_sparse_run ord2p;                 /* indices into the ord1 list of the items with an upper half */
    const uint16_t* ord2; uint64_t ord2_bytes;
} goss_gpu_vba;
int goss_gpu_push_run_graph(goss_gpu_ctx* ctx, const goss_gpu_sparse_run* edges, const goss_gpu_vba* counts);

/* A run in host memory (key_words u64 per key, strictly increasing) with u32 counts. */
int goss_gpu_push_run_host(goss_gpu_ctx* ctx, const uint64_t* keys, const uint32_t* counts, uint64_t m);

/* goss_gpu_emit with the SparseArray size estimate M given by the caller instead of the exact
 * distinct count: GossCmdMerge builds with M = sum of the inputs' counts
 * (GossCmdMerge.tcc:256-258,296), which changes D. */
int goss_gpu_emit_estimate(goss_gpu_ctx* ctx, uint64_t m_estimate);

/*
 * Between finish and emit: keep only the result items whose count lies in [lo, hi] (order
 * preserved).  With one run per input set pushed with weights, this is the set algebra of
 * intersect-kmer-sets (every set weight 1, keep count == number of sets;
 * GossCmdIntersectKmerSets.cc:29-79) and subtract-kmer-set (weights 1 and 2, keep count == 1;
 * GossCmdSubtractKmerSet.cc:47-66).
 */
int goss_gpu_select_counts(goss_gpu_ctx* ctx, uint32_t lo, uint32_t hi);

/*
 * Between finish and emit: keep only the result items that are their own canonical form
 * (edge_type::isNormal, RankSelect.hh:117-124); their counts become 1.  With the edges of a graph
 * pushed into a k-mer-set context of k = K + 1 and goss_gpu_emit_estimate(<edge count>) this is
 * graph-to-kmer-set (GossCmdGraphToKmerSet.cc:30-59).
 */
int goss_gpu_select_normal(goss_gpu_ctx* ctx);

/*
 * After emit: append a file `suffix` holding one bit per result item, set where
 * (count & mask) != 0, in WordyBitVector layout -- the <out>.lhs-bits / <out>.rhs-bits
 * annotation of GossCmdMergeAndAnnotateKmerSets.cc:126-201 when the two sets were pushed with
 * weights 1 and 2.
 */
int goss_gpu_emit_count_bits(goss_gpu_ctx* ctx, uint32_t mask, const char* suffix);

/*
 * After finish, with the context holding the decoded elements of an object (pushed through
 * goss_gpu_push_run_sparse): evaluate the object's OWN index structures on the device -- for
 * every i, SparseArray::select(i) through the -d1 DenseSelect must give element i, and
 * SparseArray::rank / access of element i through the -d0 DenseSelect must give i / true
 * (SparseArray.hh:246-364, DenseArray.cc:134-258, WordyBitVector.tcc:17-54).  This is the
 * iterator-against-select-and-rank pass of lint-graph (GossCmdLintGraph.cc:201-243).
 * Example kinds: 1 select differs, 2 rank differs, 3 element not found, 4 index walk failed.
 */
typedef struct {
    uint64_t D, count, size_lo, size_hi;            /* SparseArray header fields */
    const uint64_t* high_bits; uint64_t high_words;
    const void* d0; uint64_t d0_bytes;               /* "<base>-d0" file image */
    const void* d1; uint64_t d1_bytes;               /* "<base>-d1" file image */
    uint32_t ncols, pad;
    const void* col[4]; uint32_t col_bytes[4]; uint32_t col_shift[4];
} goss_gpu_sparse_files;
typedef struct {
    uint64_t select_mismatch, rank_mismatch, access_miss, failures;
    uint32_t nexamples, pad;
    uint64_t ex_index[16];
    uint32_t ex_kind[16];
} goss_gpu_index_report;
int goss_gpu_check_index(goss_gpu_ctx* ctx, const goss_gpu_sparse_files* files, goss_gpu_index_report* out);

/*
 * Distributed emission (one context per GPU, each holding one range of the globally sorted result; no
 * reference counterpart -- the reference is single-node; the files are those of SparseArray::Builder
 * (SparseArray.hh:87-118, SparseArray.cc:47-131) and VariableByteArray::Builder (VariableByteArray.hh:76-118)
 * fed the concatenation of the ranges).
 *
 * goss_gpu_emit_part, after finish, on every range's owner: builds what the range alone determines --
 *   "<base>.low-bits*"   this range's slice of every low-bits column file (base ".kmers" / "-edges"),
 *   "-counts.ord0"        (graph) this range's slice of the ord0 byte file,
 *   ".part.span"          this range's span of the high-bits bitmap (SparseArray.hh:87-118: one i of the whole array is
 *                         bit (key_i >> D) + i), built from its own keys: a record {1, first word, words, bytes} + the words --
 *                         about 2.4 bits per key for the assembler (round 3 sent key >> D of every key: 4 or 8 bytes) --
 *                         followed by the range's DenseSelect blocks (below),
 *   ".part.big"           (graph) the entries with count > 255: {u64 global index, u32 count, u32 0},
 *   ".part.hist"          (graph) {u64 count, u64 frequency} pairs of this range, ascending.
 * first_index = number of result items in the ranges below this one, total = items in all ranges,
 * estimate = the SparseArray estimate M (0 = total: the single-pass build).  The slices of range p belong at
 * element offset first_index of the whole file.
 *
 * goss_gpu_emit_assemble, on one context of the same (k, mode): from the concatenation (in range order)
 * of every ".part.span" on the device (span_bytes in all) and of the ".part.big" / ".part.hist" records in host
 * memory, builds the files that need all ranges: ".header", "<base>.header", "<base>.high-bits" (the spans ORed
 * together: neighbours share their boundary words), "<base>-d0", "<base>-d1" (DenseArray.cc:446-675: the ranges' blocks
 * where they sent them, the others from the positions of the assembled bitmap's ones / zeros)
 * and for graphs "-counts.ord1p.*", "-counts.ord1", "-counts.ord2p.*", "-counts.ord2", "-counts-hist.txt".
 * The files are appended to the context's list (a context may hold its own part and the assembly).
 */
int goss_gpu_emit_part(goss_gpu_ctx* ctx, uint64_t first_index, uint64_t total, uint64_t estimate);
/*
 * DenseSelect blocks per range ("GPU p owns blocks fully inside its rank span and the host patches
 * the straddling blocks"; the blocks are DenseSelect::Builder's, DenseArray.cc:446-647).  A block of "-d1" indexes 8192
 * consecutive ones of the high-bits bitmap, a block of "-d0" 8192 consecutive zeros, and is a function of their positions
 * alone: the owner of a range builds the blocks that lie wholly inside its ones [first_index, first_index + M) and -- told
 * where the ranges below it end -- inside its zeros, and appends them to its ".part.span" file as records of their own
 * ({kind, a, b, body bytes} + body: kind 1 the span {first word, words}, kind 2 blocks {sense, first block} with
 * {blocks, body bytes}, bytes[], first positions[], types[], bodies); goss_gpu_emit_assemble takes the ranges' blocks as
 * they are, builds the few that straddle two ranges from the assembled bitmap, and writes master index, rank array and
 * header -- its work no longer grows with the number of keys.
 *   goss_gpu_emit_last_high   the high part (key >> D) of the range's last key, *nonempty = 0 for a range without keys;
 *                             the callers gather these (P numbers) and hand every range the last one below it.
 *   goss_gpu_emit_part_ranges goss_gpu_emit_part with that number (prev_last_high: 0 for the first range): "-d0" blocks too
 *                             (goss_gpu_emit_part, which does not know it, builds "-d1" blocks only).
 */
int goss_gpu_emit_last_high(goss_gpu_ctx* ctx, uint64_t total, uint64_t estimate, uint64_t* high, int* nonempty);
int goss_gpu_emit_part_ranges(goss_gpu_ctx* ctx, uint64_t first_index, uint64_t total, uint64_t estimate, uint64_t prev_last_high);
int goss_gpu_emit_assemble(goss_gpu_ctx* ctx, const void* d_spans, uint64_t span_bytes, uint64_t total,
                           uint64_t estimate, const void* h_big, uint64_t nbig, const uint64_t* h_hist, uint64_t nhist);

/*
 * Several contexts in ONE process, one per GPU (what `goss build-kmer-set --devices 0,1,..` drives; the
 * process-per-GPU form of the same steps is gossamer_amd/dist.py over RCCL).  No reference counterpart.
 *
 * goss_gpu_group_exchange: every context has counted its share of the input and is finished (not emitted).
 * Splitters are quantiles of a pooled sample (sample_per_context keys of every result, 0 = 1024); range j
 * of every context's result is copied device to device into context j, which merges what it received:
 * afterwards context j holds range j of the union (sorted, distinct, counts added), finished, and
 * range_sizes[j] (may be NULL) its number of keys.  windows / keys of a context keep describing the input it
 * counted.  A multiplicity of 2^32 - 1 or more in any context: GOSS_ERR_COUNT_OVERFLOW (exact counts do
 * not travel).  The received ranges pass through device memory outside the arenas (12 or 20 bytes per key).
 *
 * goss_gpu_group_emit: goss_gpu_emit_part on every context (range order = array order), the compact parts
 * copied to contexts[0], goss_gpu_emit_assemble there.  Afterwards contexts[0] lists every file of the
 * object -- its own slices of the low-bits columns / "-counts.ord0" plus the assembled files -- and every
 * other context its slices (same suffixes), which belong behind those of the contexts before it; the
 * ".part.*" files are transport only.  estimate as in goss_gpu_emit_part.
 */
int goss_gpu_group_exchange(goss_gpu_ctx* const* contexts, uint32_t n, uint32_t sample_per_context, uint64_t* range_sizes);
int goss_gpu_group_emit(goss_gpu_ctx* const* contexts, uint32_t n, uint64_t estimate);

/*
 * The exchange BEFORE counting for the same group (one process, one context per GPU): what keeps the work per GPU
 * constant as GPUs are added -- with the exchange after counting every GPU counts a set that approaches the whole
 * k-mer set.  Replaces nothing in the reference (its scale-out is build-parts-then-merge, docs/goss.md:314-321); it
 * stands where GossCmdBuildKmerSet::operator() hands its reads to ONE BackyardHash (GossCmdBuildKmerSet.cc:112-149,
 * .tcc:226-256) and keeps that command's result: the windows are the reference's (KmerizingAdapter.hh:20-86 /
 * ReverseComplementAdapter.hh:20-93), only WHICH device counts a window is decided by its minimizer.
 *
 * goss_gpu_set_deferred(ctx, 1): host pushes of that context only stage their bases; a push that does not fit the
 * staging buffer returns GOSS_ERR_BUFFER (nothing of it was taken) instead of counting the buffer.
 * Call it BEFORE the context's first push or goss_gpu_stage_room: the staging buffers are sized when they are first
 * needed, and those of a deferred context leave room beside the arena for the records it will route and receive (0.27
 * records per staged byte each way: 7 staging buffers' worth for one-word keys, 11 for two-word keys).
 * goss_gpu_stage_room: bytes the staging buffer still takes (a push of n bytes needs n + 1) and its capacity.
 * finish / emit / a device push on a deferred context count what is staged locally, as ever: the result of the
 * group is the same (goss_gpu_group_exchange merges equal keys), only the balance is lost.
 *
 * goss_gpu_group_route_exchange(contexts, n, transport, stats): every context cuts the windows of its staged bases
 * into super-k-mer records routed by minimizer into n parts (goss_gpu_route_records_device), part p of every context
 * is moved to context p's device, and every context counts what it received (as goss_gpu_push_records_device does)
 * -- on a thread of its own that the call does not wait for: when it returns, the records have arrived, the staging
 * buffers are empty and take the next round's reads WHILE the devices count this round's (every entry point that needs
 * the counted runs -- the next round, finish, emit, a device push -- waits for that thread and takes over its failure,
 * as for the thread that counts a full staging buffer of a single context).  All copies of a key -- either strand --
 * reach ONE context, so the contexts' counted sets are disjoint.  Call it whenever a buffer is full and once before
 * the finishes, then goss_gpu_group_exchange (range partition of the counted sets) and goss_gpu_group_emit as above.
 *   transport   0 = RCCL when librccl.so can be loaded at run time and communicators for the contexts' devices can
 *               be made (ncclCommInitAll; not with one device twice): ncclSend / ncclRecv of all n x n parts inside
 *               one ncclGroupStart / ncclGroupEnd, at most 512 MiB per pair and round -- else hipMemcpyPeerAsync;
 *               1 = RCCL or fail (GOSS_ERR_STATE); 2 = peer copies.  GOSS_GROUP_TRANSPORT=rccl|peer overrides.
 *   stats       (may be NULL) what the call moved, how long routing and transfer took on the host's clock, and what
 *               the counting of the round before cost beside the caller's staging (count_ms) and beyond it (count_wait_ms).
 * Both key widths: 12-byte records for one-word keys, 20-byte records for two-word keys (below).
 */
typedef struct goss_gpu_group_xstats {
    uint32_t transport;            /* 1 = RCCL send / recv, 2 = peer copies */
    uint32_t rounds;               /* transfer rounds (<= 512 MiB per pair each) */
    uint64_t records, windows;     /* moved by this call, all members */
    uint64_t record_bytes;
    double route_ms, wire_ms;
    double count_ms;               /* counting of the round BEFORE this call's (the slowest member): it ran on the members' own
                                      threads while the caller staged this round's reads */
    double count_wait_ms;          /* ... and how long this call had to wait for it to end: the part the staging did not cover */
} goss_gpu_group_xstats;
int goss_gpu_set_deferred(goss_gpu_ctx* ctx, int on);
int goss_gpu_stage_room(goss_gpu_ctx* ctx, uint64_t* free_bytes, uint64_t* capacity);
int goss_gpu_group_route_exchange(goss_gpu_ctx* const* contexts, uint32_t n, int transport, goss_gpu_group_xstats* stats);
/* Device address of file i's image (NULL when the file was built on the host): for moving slices
 * between GPUs without a host copy.  Valid until the next emit, reset, push or destroy. */
int goss_gpu_file_device(goss_gpu_ctx* ctx, uint32_t i, const void** d_ptr);

/*
 * Let the context's HBM arena grow beyond hbm_budget, up to max_bytes (0 = fixed, the default):
 * mapping HBM costs time, so a caller may start small and let inputs with little duplication --
 * whose sorted runs do not shrink (the case the reference spills to disk for, AsyncMerge.tcc) --
 * enlarge it when a chunk or a merge needs the room.
 */
int goss_gpu_set_budget_limit(goss_gpu_ctx* ctx, uint64_t max_bytes);

/*
 * Diagnostic counters of the context, by name: "fused_chunks" (chunks counted by the extraction
 * that partitions), "fused_overflows" (chunks redone unfused because a bucket region was too
 * small), "segment_retries", "lookback_failures", "runs".
 */
int goss_gpu_stat(goss_gpu_ctx* ctx, const char* name, uint64_t* value);

/*
 * After finish: the text form of the object as one device-built file (suffix ".dump") --
 * "#<version>\nK\tcount\n" and one k-mer per line for a k-mer set (GossCmdDumpKmerSet.cc:31-55),
 * "#<version>\nK\tcount\tflags\n" and "<edge>\t<multiplicity>" per line for a graph
 * (GossCmdDumpGraph.cc:31-61).  Replaces the file list; read it with goss_gpu_file_read.
 */
int goss_gpu_emit_dump(goss_gpu_ctx* ctx, uint64_t flags);
/* The same for elements [first, first + count) only (the header lines go in front of the piece
 * that starts at 0): the text of a large object is produced and written piece by piece; each call
 * releases the previous piece. */
int goss_gpu_emit_dump_range(goss_gpu_ctx* ctx, uint64_t flags, uint64_t first, uint64_t count);

/*
 * After finish, graph mode: the checks of lint-graph's first pass (GossCmdLintGraph.cc:131-199)
 * over the edge list held by the context -- reverse complement present, equal multiplicities
 * (asymmetric: not both zero), positive multiplicities -- plus strict ordering of the list.
 * Up to 32 offending edges are returned: kind 1 = no reverse complement, 2 = multiplicities
 * differ, 3 = zero multiplicity, 4 = out of order; `other` = index of the reverse complement
 * or ~0.
 */
typedef struct {
    uint64_t missing_rc, count_mismatch, zero_count, order_violation;
    uint32_t nexamples, pad;
    uint64_t ex_index[32], ex_other[32];
    uint32_t ex_kind[32];
} goss_gpu_lint_report;
int goss_gpu_lint(goss_gpu_ctx* ctx, int asymmetric, goss_gpu_lint_report* out);

/*
 * Page-locked host memory for the buffers handed to goss_gpu_push_bases_host (the copy to the
 * device then runs at PCIe speed instead of going through the driver's bounce buffers).
 */
int goss_gpu_host_alloc(void** p, size_t bytes);
void goss_gpu_host_free(void* p);
/*
 * The same for memory the caller already has (page-aligned, e.g. from mmap or posix_memalign): page-locks [p, p + bytes)
 * in place.  A caller that wants to fill its buffers BEFORE the device runtime is up -- `goss` starts parsing while its
 * context is created -- allocates them plainly, registers each once the context exists and while nothing is being
 * copied from it, and unregisters it before freeing.  A buffer that is not (yet) registered may still be pushed: the
 * copy is then synchronous inside the driver.  Returns GOSS_ERR_OOM when the pages cannot be locked.
 */
int goss_gpu_host_register(void* p, size_t bytes);
void goss_gpu_host_unregister(void* p);

/*
 * Counting strategy.  0 (default): partition on the top 16 key bits, then count every segment
 * in an LDS hash table, falling back to the full LSD radix sort when a segment holds too many
 * distinct keys.  1: always the full LSD radix sort + run compaction.  The result is the same;
 * BackyardHash's layout never reaches disk either.
 */
int goss_gpu_set_path(goss_gpu_ctx* ctx, int path);

/*
 * Feed an already counted run: m distinct keys (key_words u64 each, strictly increasing)
 * with their u32 counts, both resident in HBM.  Runs are merged with everything else at
 * finish (equal keys summed).  Used to combine partial results -- per-GPU ranges after the
 * all-to-all exchange, or previously built objects (the role AsyncMerge::merge plays for the
 * reference's spill runs, AsyncMerge.tcc:174-324).
 */
int goss_gpu_push_run_device(goss_gpu_ctx* ctx, const void* d_keys, const uint32_t* d_counts,
                             uint64_t m);

/*
 * Feed k-mers the caller has already cut out of its input: n keys of key_words u64 each ({lo,hi} for two
 * words), first base in the most significant used bits, in any order, with repeats, NOT normalised.
 * Takes over the loop of the templated GossCmdBuildKmerSet::operator()(cxt, KmerSrc&)
 * (GossCmdBuildKmerSet.hh:23-30, GossCmdBuildKmerSet.tcc:246-256: `kmer = *pKmerSrc; kmer.normalize(mK);
 * blk->push_back(kmer)` + BackyardHash::insert), which electus drives with a GossRead::Iterator or a
 * KmerizingAdapter of its own (ElectApp.cc:183-215).  kmer-set mode: every key is replaced by its canonical
 * form (position_type::normalize, RankSelect.hh:126-140) and counted; graph mode: the keys are counted as they
 * are (GossCmdBuildGraph.cc:276,307 inserts what ReverseComplementAdapter yields -- the caller pushes both
 * strands).  A key with bits at or above 2*len (len = k, or k+1 in graph mode): GOSS_ERR_INVALID_ARG and
 * nothing of this call is counted.  Keys and bases may be mixed in one build; counts add up at finish.
 * `windows` grows by n (n/2 in graph mode).  _device: resident in HBM, complete before the call (see
 * goss_gpu_push_bases_device), not modified.
 */
int goss_gpu_push_keys_host(goss_gpu_ctx* ctx, const uint64_t* keys, uint64_t n);
int goss_gpu_push_keys_device(goss_gpu_ctx* ctx, const void* d_keys, uint64_t n);

/*
 * The exchange BEFORE counting of a multi-GPU build (no reference counterpart: the reference is single-node; what
 * must be preserved is the adapters' key stream -- every valid window of every read once, KmerizingAdapter.hh:20-86 /
 * ReverseComplementAdapter.hh:20-93 -- and that all copies of a key, from either strand, are counted in ONE place,
 * which is what position_type::normalize, RankSelect.hh:126-140, exists for).
 *
 * goss_gpu_route_records_device cuts the windows of a base string (as goss_gpu_push_bases_device takes it) into
 * SUPER-K-MER RECORDS and appends each to one of nparts buffers: a record is a run of up to 16 consecutive windows
 * that have the same destination, with the run's bases stored once -- 12 bytes (GOSS_RECORD_BYTES) for one-word
 * keys (2*len <= 62, len = k or k + 1) instead of 8 bytes per window, 20 bytes (GOSS_RECORD2_BYTES) for two-word keys
 * (32 <= len <= 63) instead of 16 per window.  The destination of a window is a hash of its MINIMIZER (the smallest
 * canonical m-mer inside it, m = 7..15 depending on the window length; for len >= 32 inside its central 31 -- odd
 * len -- or 30 bases, whose reverse complement is the central part of the window's reverse complement), scaled to
 * [0, nparts): a window and its reverse complement have the same minimizer, so every occurrence of a k-mer -- and of
 * a graph edge and its reverse complement (ReverseComplementAdapter.hh:34-55) -- reaches the same part, and the counts
 * of a part are final.
 *   d_records           device memory for the records of all parts
 *   part_first[p]       first record slot of part p inside d_records, part_cap[p] the slots it may use
 *   part_records[p]     (out) record slots of part p that were filled, pads included (below) -- what it NEEDS when the
 *                       call returns GOSS_ERR_BUFFER because some part_cap was too small (nothing usable was written
 *                       for that part: call again with room; the need of an input is the same in every call)
 *   part_windows[p]     (out, may be NULL) windows routed to part p
 * The context only lends its device, stream and (k, mode): nothing is counted and no state changes.
 *
 * goss_gpu_push_records_device counts the windows of records (its own or received from other ranks) exactly as
 * goss_gpu_push_bases_* counts the windows of bases; nwindows (0 = unknown) = the sum of the parts' part_windows,
 * which sizes the key buffers.  Records and bases may be mixed in one build.
 *
 * Record layout (little endian, three u32): bits 0..91 the run's nwin + len - 1 <= 46 bases as 2-bit codes
 * (A=0 C=1 G=2 T=3), base j at bits [2j, 2j+2), zero above them; bits 92..95 nwin - 1.
 * A PAD is the record {0, 0, 1 << 27} -- no window: the routing kernel takes room in blocks (<= 512 slots per
 * workgroup and part) and fills what it does not use with pads; they are slots of the part like any record, travel
 * with it and are dropped by goss_gpu_push_records_device (no record of windows has bit 91 set with bits 92..95 zero:
 * a single window's bases end below bit 64).  0.1-0.3 % of the slots of a large input.
 * Two-word keys (five u32): bits 0..155 the run's nwin + len - 1 <= 78 bases, bits 156..159 nwin - 1; PAD =
 * {0, 0, 0, 0, 1 << 27}.
 */
#define GOSS_RECORD_BYTES 12
#define GOSS_RECORD2_BYTES 20
int goss_gpu_route_records_device(goss_gpu_ctx* ctx, const void* d_bases, uint64_t nbytes, uint32_t nparts, void* d_records,
                                  const uint64_t* part_first, const uint64_t* part_cap, uint64_t* part_records,
                                  uint64_t* part_windows);
int goss_gpu_push_records_device(goss_gpu_ctx* ctx, const void* d_records, uint64_t nrecords, uint64_t nwindows);

/*
 * Deterministic synthetic read generator: fills d_out (device) with
 * nreads reads of read_len bases sampled from an i.i.d. uniform genome of genome_len bases,
 * each followed by '\n'; strand flipped with p = 1/2; one 'N' in every 97th read.
 * Bytes written = nreads * (read_len + 1).  The same generator exists on the host
 * (goss_synth_reads_host) so CPU and GPU runs see identical input.
 */
int goss_gpu_synth_reads(goss_gpu_ctx* ctx, void* d_out, uint64_t nreads, uint32_t read_len,
                         uint64_t genome_len, uint64_t seed, uint64_t first_read);
int goss_synth_reads_host(char* out, uint64_t nreads, uint32_t read_len,
                          uint64_t genome_len, uint64_t seed, uint64_t first_read);

#ifdef __cplusplus
}
#endif
#endif /* GOSS_GPU_H */
