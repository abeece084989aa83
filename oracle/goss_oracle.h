/*
 * goss_oracle.h -- CPU restatement (plain C) of data61/gossamer's k-mer counting /
 * de Bruijn edge-set build path, of the reference's readers of the objects it writes, and of
 * the commands around it that the product also implements (merge-*, intersect / subtract /
 * merge-and-annotate, dump-*, restore-graph).  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (libgossgpu.so, the goss CLI) never links or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference's src/ directory).
 *
 * PINNING STATUS: the reference itself cannot be built in this image (it needs Boost,
 * which is absent, and writing stand-in headers is not allowed), so this restatement is
 * pinned against (i) the reference's seeded unit tests for the structures on this path,
 * replayed on regenerated inputs (std::mt19937 + libstdc++ distributions with the tests'
 * seeds: tests/golden/gen_reference_inputs.cpp, digests and file:line in
 * tests/golden/reference_kat.json) -- testSparseArray.cc, testDenseArray.cc,
 * testWordyBitVector.cc, testVariableByteArray.cc, testGraph.cc, testBigInteger.cc
 * (tests/test_reference_vectors.py); (ii) the known answers held by the reference's other
 * tests for this path (testGossCmdBuildGraph.cc, testReverseComplementAdapter.cc,
 * testUtils.cc, testVByteCodec.cc, testFastqParser.cc); (iii) the known-answer vectors
 * recorded in SURVEY.md App. C.  No reference test compares output *bytes*: byte parity is
 * pinned through the reference's reader semantics, not through files the reference wrote
 * (see DESIGN.md section 5).
 */
#ifndef GOSS_ORACLE_H
#define GOSS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 128-bit key: lo = word0 (low 64 bits), hi = word1.  BigInteger.hh:565 */
typedef struct { uint64_t lo, hi; } go_key;

/* ---- key arithmetic ---- */
uint64_t go_rev64(uint64_t x);                                  /* Utils.hh:377-396 */
go_key   go_revcomp(go_key x, unsigned k);                      /* BigInteger.hh:204-217 */
uint64_t go_hash(go_key x);                                     /* BigInteger.hh:528-536,572-582 */
go_key   go_normalize(go_key x, unsigned k);                    /* RankSelect.hh:126-140 */
uint64_t go_select1(uint64_t word, uint64_t rank);              /* Utils.hh:334 */
uint64_t go_log2(uint64_t x);                                   /* Utils.hh:340-344 */

/* ---- read -> k-mers ---- */
/* All valid k-windows of seq in order; returns how many; writes at most cap.
 * GossReadBaseString.hh:52-103,133-188 */
size_t go_kmerize(const char* seq, size_t len, unsigned k, go_key* out, size_t cap);

/* ---- in-memory file set (role of StringFileFactory) ---- */
typedef struct go_fs go_fs;
go_fs*  go_fs_new(void);
void    go_fs_free(go_fs*);
size_t  go_fs_count(const go_fs*);
const char* go_fs_name(const go_fs*, size_t i);
size_t  go_fs_size(const go_fs*, size_t i);
const uint8_t* go_fs_data(const go_fs*, size_t i);
int     go_fs_find(const go_fs*, const char* name);             /* index or -1 */
int     go_fs_add(go_fs*, const char* name, const void* data, size_t n);
int     go_fs_write_dir(const go_fs*, const char* dir);         /* 0 ok */
int     go_fs_read_dir(go_fs*, const char* dir, const char* prefix);

/* ---- parsing: kinds ---- */
enum { GO_IN_LINE = 0, GO_IN_FASTA = 1, GO_IN_FASTQ = 2 };

typedef struct {
    int         kind;
    const char* name;    /* for error text */
    const char* data;
    size_t      size;
} go_input;

/* Collect keys from the inputs in the reference's order (line, fasta, fastq is the
 * caller's job).  mode 0: canonical k-mers (build-kmer-set); mode 1: every rho-mer then
 * its reverse complement, rho = k+1 is passed as `len` by the caller (build-graph).
 * Returns 0, or -1 with err filled.  *nreads counts parsed reads. */
typedef struct {
    go_key*  keys;
    size_t   n, cap;
    uint64_t nreads;
    uint64_t nwindows;
} go_keys;
void go_keys_free(go_keys*);
int  go_collect(const go_input* in, size_t nin, unsigned len, int mode, go_keys* out, char* err, size_t errcap);

/* sort + sum equal neighbours (role of BackyardHash::sort + flush). counts u64. */
size_t go_sort_count(go_key* keys, size_t n, uint64_t* counts /* size n */);

/* ---- builders (on-disk formats) ---- */
uint64_t go_sparse_d(go_key N, uint64_t M);                     /* SparseArray.cc:47-72 */
int go_write_kmer_set(go_fs* fs, const char* base, unsigned K, const go_key* keys, size_t n, uint64_t M);
int go_write_graph(go_fs* fs, const char* base, unsigned K, const go_key* keys, const uint64_t* counts, size_t n, uint64_t M);
int go_write_sparse_array(go_fs* fs, const char* base, go_key N_ctor, uint64_t M, const go_key* pos, size_t n, go_key N_end);

/* whole commands; return 0 or -1 with err.  GossCmdBuildKmerSet.tcc:213-332 (single-pass
 * branch), GossCmdBuildGraph.cc:270-426 */
int go_build_kmer_set(go_fs* fs, const char* out, unsigned K, const go_input* in, size_t nin, uint64_t* nwindows, char* err, size_t errcap);
int go_build_graph(go_fs* fs, const char* out, unsigned K, const go_input* in, size_t nin, uint64_t* nwindows, char* err, size_t errcap);
/* go_build_kmer_set over one line-kind input with T worker threads (parse / canonicalise / sort-count per
 * shard, parallel merge by key ranges, serial write -- the thread structure of GossCmdBuildKmerSet.tcc:226-256
 * and BackyardHash.cc:244-271); same files. */
int go_build_kmer_set_mt(go_fs* fs, const char* out, unsigned K, const char* reads, size_t size, unsigned T,
                         uint64_t* nwindows, char* err, size_t errcap);

/* ---- readers (restated from the reference's read side; used to round-trip files) ---- */
typedef struct go_sparse go_sparse;
go_sparse* go_sparse_open(const go_fs* fs, const char* base, char* err, size_t errcap);
void     go_sparse_close(go_sparse*);
uint64_t go_sparse_count(const go_sparse*);
go_key   go_sparse_size(const go_sparse*);
go_key   go_sparse_select(const go_sparse*, uint64_t rnk);      /* SparseArray.hh:311-325 */
uint64_t go_sparse_rank(const go_sparse*, go_key pos);          /* SparseArray.hh:296-309 */
int      go_sparse_access(const go_sparse*, go_key pos);        /* SparseArray.hh:246-260 */
uint64_t go_sparse_d0_select(const go_sparse*, uint64_t i);
uint64_t go_sparse_d1_select(const go_sparse*, uint64_t i);
/* VariableByteArray read (VariableByteArray.hh operator[]) */
int      go_vba_get(const go_fs* fs, const char* base, uint64_t i, uint32_t* out, char* err, size_t errcap);
/* Graph::open-like checks: header version, K, count. */
int      go_kmer_set_header(const go_fs* fs, const char* base, uint64_t* K, uint64_t* count);
int      go_graph_header(const go_fs* fs, const char* base, uint64_t* K, uint64_t* flags);

/* merge-kmer-sets (kind 0) / merge-graphs (kind 1): GossCmdMerge.tcc:151-326.  Inputs are
 * objects in `in`, the result is written to `out`; the estimate M is the sum of the inputs'
 * counts, groups of max_merge are merged first when there are more inputs than that. */
int go_merge(const go_fs* in, const char* const* names, size_t nin, int kind, uint64_t max_merge,
             go_fs* out, const char* out_name, char* err, size_t errcap);

/* intersect-kmer-sets (GossCmdIntersectKmerSets.cc:29-128), subtract-kmer-set
 * (GossCmdSubtractKmerSet.cc:32-85), merge-and-annotate-kmer-sets
 * (GossCmdMergeAndAnnotateKmerSets.cc:30-206; stats = lhs count, rhs count, common). */
int go_intersect_kmer_sets(const go_fs* in, const char* const* names, size_t nin, go_fs* out, const char* out_name,
                           char* err, size_t errcap);
int go_subtract_kmer_set(const go_fs* in, const char* lhs, const char* rhs, go_fs* out, const char* out_name,
                         char* err, size_t errcap);
/* GossCmdGraphToKmerSet.cc:30-59 */
int go_graph_to_kmer_set(const go_fs* in, const char* graph, go_fs* outfs, const char* out_name, char* err, size_t errcap);
int go_merge_and_annotate(const go_fs* in, const char* lhs, const char* rhs, go_fs* out, const char* out_name,
                          uint64_t stats[3], char* err, size_t errcap);

/* dump-kmer-set (kind 0, GossCmdDumpKmerSet.cc:31-55) / dump-graph (kind 1, GossCmdDumpGraph.cc:31-61):
 * malloc'ed text; restore-graph (GossCmdRestoreGraph.cc:72-135). */
int go_dump(const go_fs* fs, const char* name, int kind, char** text, size_t* len, char* err, size_t errcap);
int go_restore_graph(const char* text, size_t len, go_fs* out, const char* out_name, char* err, size_t errcap);

/* ---- stand-alone structures in the shape of the reference's unit tests ----
 * testDenseArray.cc:142-167 etc.: WordyBitVector `vname` over nbits positions + DenseSelect `xname`
 * of the given sense; testWordyBitVector.cc:44-58: sparse push; their readers; and
 * testVariableByteArray.cc: VariableByteArray builder. */
int      go_write_bits_and_select(go_fs* fs, const char* vname, const char* xname, const uint64_t* ones, size_t n,
                                  uint64_t nbits, int invert);
int      go_write_bits_sparse(go_fs* fs, const char* vname, const uint64_t* ones, size_t n);
int      go_bits_get(const go_fs* fs, const char* vname, uint64_t pos);                 /* WordyBitVector.hh:173-179 */
uint64_t go_bits_words(const go_fs* fs, const char* vname);
uint64_t go_bits_select(const go_fs* fs, const char* vname, int invert, uint64_t from, uint64_t count);  /* WordyBitVector.tcc:17-54 */
uint64_t go_bits_popcount_range(const go_fs* fs, const char* vname, uint64_t begin, uint64_t end);
uint64_t go_dense_select(const go_fs* fs, const char* vname, const char* xname, int invert, uint64_t i); /* DenseArray.cc:185-258 */
int      go_write_vba(go_fs* fs, const char* base, const uint32_t* values, size_t n, uint64_t numItems);

/* The assertion loops of testDenseArray.cc / testSparseArray.cc / testVariableByteArray.cc replayed
 * over a file set (written by this oracle or by the product): number of failed checks, ~0 = cannot open. */
uint64_t go_replay_dense_select(const go_fs* fs, const char* vname, const char* xname, int invert,
                                const uint64_t* ones, size_t n, uint64_t nbits);
uint64_t go_replay_sparse(const go_fs* fs, const char* base, const go_key* pos, size_t n, uint64_t universe);
uint64_t go_replay_sparse_highbits(const go_fs* fs, const char* base, const uint64_t* ones, size_t n, uint64_t nbits);
uint64_t go_replay_vba(const go_fs* fs, const char* base, const uint32_t* values, size_t n);

/* VByte (spill-run private format; golden bytes in testVByteCodec.cc) */
size_t   go_vbyte_encode(uint64_t x, uint8_t* out /* >= 9 */);  /* VByteCodec.hh:24-104 */
uint64_t go_vbyte_decode(const uint8_t* in, size_t* used);

#ifdef __cplusplus
}
#endif
#endif
