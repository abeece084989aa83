/*
 * goss_oracle.c -- CPU restatement (plain C11) of data61/gossamer's build-kmer-set /
 * build-graph path.  TEST INFRASTRUCTURE ONLY: see goss_oracle.h for who may use it and
 * for the pinning status ("byte parity unpinned by the reference's tests").
 *
 * The code is sequential and mirrors the reference's streaming builders one to one so that
 * it can be audited against the cited lines; it is deliberately unlike the product's
 * parallel device algorithms.
 */
#define _GNU_SOURCE
#include "goss_oracle.h"

#include <ctype.h>
#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

typedef unsigned __int128 u128;

static inline u128 k2u(go_key k) { return ((u128)k.hi << 64) | k.lo; }
static inline go_key u2k(u128 v) { go_key k; k.lo = (uint64_t)v; k.hi = (uint64_t)(v >> 64); return k; }

/* ------------------------------------------------------------------------------------ */
/* key arithmetic                                                                        */
/* ------------------------------------------------------------------------------------ */

/* Base-4 reverse of a 64-bit word.  Utils.hh:377-396 */
uint64_t go_rev64(uint64_t x)
{
    const uint64_t m2 = 0x3333333333333333ULL, m4 = 0x0F0F0F0F0F0F0F0FULL;
    const uint64_t m8 = 0x00FF00FF00FF00FFULL, m16 = 0x0000FFFF0000FFFFULL;
    const uint64_t m32 = 0x00000000FFFFFFFFULL;
    x = ((x & m2) << 2) | ((x & (m2 << 2)) >> 2);
    x = ((x & m4) << 4) | ((x & (m4 << 4)) >> 4);
    x = ((x & m8) << 8) | ((x & (m8 << 8)) >> 8);
    x = ((x & m16) << 16) | ((x & (m16 << 16)) >> 16);
    x = ((x & m32) << 32) | ((x & (m32 << 32)) >> 32);
    return x;
}

/* BigInteger<2>::reverseComplement.  BigInteger.hh:204-217: swap the words, rev(~w) each,
 * then shift right by 128 - 2k. */
go_key go_revcomp(go_key x, unsigned k)
{
    go_key r;
    uint64_t tmp = go_rev64(~x.lo);
    r.lo = go_rev64(~x.hi);
    r.hi = tmp;
    unsigned sh = 128 - 2 * k;
    u128 v = k2u(r);
    v = sh >= 128 ? 0 : (v >> sh);
    return u2k(v);
}

/* FNV-1a over the 16 bytes, word0 LSB first then word1.  BigInteger.hh:528-536,572-582 */
static uint64_t word_hash(uint64_t w, uint64_t seed)
{
    uint64_t r = seed;
    for (unsigned i = 0; i < 8; ++i)
    {
        r ^= w & 0xFFULL;
        w >>= 8;
        r *= 1099511628211ULL;
    }
    return r;
}

uint64_t go_hash(go_key x)
{
    uint64_t seed = 14695981039346656037ULL;
    seed = word_hash(x.lo, seed);
    seed = word_hash(x.hi, seed);
    return seed;
}

/* position_type::normalize.  RankSelect.hh:126-140 */
go_key go_normalize(go_key x, unsigned k)
{
    go_key rc = go_revcomp(x, k);
    uint64_t h0 = go_hash(x), h1 = go_hash(rc);
    if (h0 > h1) return rc;
    if (h0 == h1 && k2u(rc) < k2u(x)) return rc;
    return x;
}

/* select the pRank'th (0-based) set bit of a word.  Utils.hh:334 (semantics; the
 * reference uses Vigna's broadword form, the answers are pinned by testUtils.cc:36-57). */
uint64_t go_select1(uint64_t w, uint64_t rank)
{
    for (uint64_t i = 0; i < 64; ++i)
    {
        if ((w >> i) & 1)
        {
            if (rank == 0) return i;
            --rank;
        }
    }
    return 64;
}

/* Utils.hh:340-344: ceil(log2(x)) with log2(1) = 0. */
uint64_t go_log2(uint64_t x)
{
    if (x == 1) return 0;
    return 64 - (uint64_t)__builtin_clzll(x - 1);
}

/* ------------------------------------------------------------------------------------ */
/* reads -> k-mers                                                                       */
/* ------------------------------------------------------------------------------------ */

/* GossReadBaseString::getBase.  GossReadBaseString.hh:133-170 */
static int get_base(char c, unsigned* out)
{
    switch (c)
    {
        case 'A': case 'a': *out = 0; return 1;
        case 'C': case 'c': *out = 1; return 1;
        case 'G': case 'g': *out = 2; return 1;
        case 'T': case 't': *out = 3; return 1;
        default: return 0;
    }
}

/* getEdge: build the k-mer at [off, off+len).  GossReadBaseString.hh:172-188 */
static int get_edge(const char* s, size_t off, size_t len, size_t* fail, u128* res)
{
    u128 r = 0;
    unsigned x;
    for (size_t i = off; i < off + len; ++i)
    {
        if (!get_base(s[i], &x)) { *fail = i; return 0; }
        r = (r << 2) | x;
    }
    *res = r;
    return 1;
}

/* firstKmer / nextKmer driven exactly like GossRead::Iterator (GossRead.hh:57-114,
 * GossReadBaseString.hh:52-103).  Emits windows in read order. */
size_t go_kmerize(const char* seq, size_t len, unsigned k, go_key* out, size_t cap)
{
    size_t n = 0;
    if (len < k) return 0;
    u128 mask = (k >= 64) ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1);
    u128 kmer = 0;
    size_t offset = 0;
    int have = 0;
    /* firstKmer */
    for (size_t i = 0; i + k <= len; ++i)
    {
        size_t fail;
        if (!get_edge(seq, i, k, &fail, &kmer)) { i = fail; continue; }
        offset = i; have = 1; break;
    }
    while (have)
    {
        if (n < cap) out[n] = u2k(kmer);
        ++n;
        /* nextKmer */
        if (offset + k >= len) break;
        unsigned x;
        if (get_base(seq[offset + k], &x))
        {
            kmer = ((kmer << 2) | x) & mask;
            ++offset;
            continue;
        }
        have = 0;
        for (size_t i = offset + k; i + k <= len; ++i)
        {
            size_t fail;
            if (!get_edge(seq, i, k, &fail, &kmer)) { i = fail; continue; }
            offset = i; have = 1; break;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* in-memory file set (role of StringFileFactory.hh:25-82)                               */
/* ------------------------------------------------------------------------------------ */

typedef struct { char* name; uint8_t* data; size_t size, cap, pos; } go_file;
struct go_fs { go_file* files; size_t n, cap; };

go_fs* go_fs_new(void) { return (go_fs*)calloc(1, sizeof(go_fs)); }

void go_fs_free(go_fs* fs)
{
    if (!fs) return;
    for (size_t i = 0; i < fs->n; ++i) { free(fs->files[i].name); free(fs->files[i].data); }
    free(fs->files);
    free(fs);
}

size_t go_fs_count(const go_fs* fs) { return fs->n; }
const char* go_fs_name(const go_fs* fs, size_t i) { return fs->files[i].name; }
size_t go_fs_size(const go_fs* fs, size_t i) { return fs->files[i].size; }
const uint8_t* go_fs_data(const go_fs* fs, size_t i) { return fs->files[i].data; }

int go_fs_find(const go_fs* fs, const char* name)
{
    for (size_t i = 0; i < fs->n; ++i) if (!strcmp(fs->files[i].name, name)) return (int)i;
    return -1;
}

/* FileFactory::out : create or truncate */
static go_file* fs_out(go_fs* fs, const char* name)
{
    int i = go_fs_find(fs, name);
    if (i >= 0)
    {
        fs->files[i].size = 0; fs->files[i].pos = 0;
        return &fs->files[i];
    }
    if (fs->n == fs->cap)
    {
        fs->cap = fs->cap ? fs->cap * 2 : 32;
        fs->files = (go_file*)realloc(fs->files, fs->cap * sizeof(go_file));
    }
    go_file* f = &fs->files[fs->n++];
    memset(f, 0, sizeof(*f));
    f->name = strdup(name);
    return f;
}

/* NB: builders keep indices, not pointers, because fs->files may be reallocated. */
static size_t fs_out_idx(go_fs* fs, const char* name) { go_file* f = fs_out(fs, name); return (size_t)(f - fs->files); }

static void f_write(go_fs* fs, size_t idx, const void* p, size_t n)
{
    go_file* f = &fs->files[idx];
    if (f->pos + n > f->cap)
    {
        size_t nc = f->cap ? f->cap : 256;
        while (nc < f->pos + n) nc *= 2;
        f->data = (uint8_t*)realloc(f->data, nc);
        f->cap = nc;
    }
    memcpy(f->data + f->pos, p, n);
    f->pos += n;
    if (f->pos > f->size) f->size = f->pos;
}
static size_t f_tell(const go_fs* fs, size_t idx) { return fs->files[idx].pos; }
static void f_seek(go_fs* fs, size_t idx, size_t pos) { fs->files[idx].pos = pos; }

int go_fs_add(go_fs* fs, const char* name, const void* data, size_t n)
{
    size_t idx = fs_out_idx(fs, name);
    if (n) f_write(fs, idx, data, n);
    return 0;
}

int go_fs_write_dir(const go_fs* fs, const char* dir)
{
    char path[4096];
    for (size_t i = 0; i < fs->n; ++i)
    {
        snprintf(path, sizeof path, "%s/%s", dir, fs->files[i].name);
        FILE* fp = fopen(path, "wb");
        if (!fp) return -1;
        if (fs->files[i].size && fwrite(fs->files[i].data, 1, fs->files[i].size, fp) != fs->files[i].size) { fclose(fp); return -1; }
        fclose(fp);
    }
    return 0;
}

int go_fs_read_dir(go_fs* fs, const char* dir, const char* prefix)
{
    DIR* d = opendir(dir);
    if (!d) return -1;
    struct dirent* e;
    char path[4096];
    size_t pl = strlen(prefix);
    while ((e = readdir(d)))
    {
        if (strncmp(e->d_name, prefix, pl)) continue;
        snprintf(path, sizeof path, "%s/%s", dir, e->d_name);
        struct stat st;
        if (stat(path, &st) || !S_ISREG(st.st_mode)) continue;
        FILE* fp = fopen(path, "rb");
        if (!fp) { closedir(d); return -1; }
        uint8_t* buf = (uint8_t*)malloc(st.st_size ? st.st_size : 1);
        size_t got = fread(buf, 1, st.st_size, fp);
        fclose(fp);
        go_fs_add(fs, e->d_name, buf, got);
        free(buf);
    }
    closedir(d);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* line source + parsers                                                                 */
/* ------------------------------------------------------------------------------------ */

/* PlainLineSource over a memory buffer.  LineSource.cc:17-48: valid() is
 * good() || !line.empty(); operator++ clears the line then std::getline()s. */
typedef struct {
    const char* p; size_t n, at;
    int good;               /* stream.good() */
    const char* line; size_t len;
} line_src;

static void ls_getline(line_src* s)
{
    s->line = s->p + s->at; s->len = 0;
    if (!s->good) return;                 /* getline on a non-good stream extracts nothing */
    if (s->at >= s->n) { s->good = 0; return; }  /* eof|fail, empty line */
    const char* nl = (const char*)memchr(s->p + s->at, '\n', s->n - s->at);
    if (nl) { s->len = (size_t)(nl - (s->p + s->at)); s->at += s->len + 1; }
    else    { s->len = s->n - s->at; s->at = s->n; s->good = 0; /* eofbit */ }
}
static void ls_init(line_src* s, const char* p, size_t n)
{
    s->p = p; s->n = n; s->at = 0; s->good = 1; s->line = p; s->len = 0;
    /* an empty std::istringstream is still good() before the first read */
    ls_getline(s);
}
static int ls_valid(const line_src* s) { return s->good || s->len; }
static void ls_next(line_src* s) { ls_getline(s); }

/* growable char buffer */
typedef struct { char* p; size_t n, cap; } sbuf;
static void sb_clear(sbuf* b) { b->n = 0; }
static void sb_append(sbuf* b, const char* s, size_t n)
{
    if (b->n + n + 1 > b->cap)
    {
        size_t nc = b->cap ? b->cap : 256;
        while (nc < b->n + n + 1) nc *= 2;
        b->p = (char*)realloc(b->p, nc); b->cap = nc;
    }
    if (n) memcpy(b->p + b->n, s, n);
    b->n += n; b->p[b->n] = 0;
}

typedef void (*read_cb)(void* ctx, const char* seq, size_t len);

/* FastqParser::getLine strips one trailing '\r'.  FastqParser.hh:62-75 */
static void fq_line(const line_src* s, const char** l, size_t* n)
{
    *l = s->line; *n = s->len;
    if (*n > 0 && (*l)[*n - 1] == '\r') --*n;
}

/* FastqParser::next loop.  FastqParser.hh:78-176 (mLineNum starts at 1). */
static int parse_fastq(const go_input* in, read_cb cb, void* ctx, uint64_t* nreads, char* err, size_t errcap)
{
    line_src s; ls_init(&s, in->data, in->size);
    sbuf seq = {0}, qual = {0}, label = {0};
    uint64_t lineNum = 1;
    int rc = 0;
    for (;;)
    {
        if (!ls_valid(&s)) break;
        const char* l; size_t n;
        fq_line(&s, &l, &n);
        if (!(n > 0 && l[0] == '@'))
        {
            snprintf(err, errcap, "%s: expected '@' at beginning of line %llu", in->name, (unsigned long long)lineNum);
            rc = -1; break;
        }
        sb_clear(&label); sb_append(&label, l + 1, n - 1);
        sb_clear(&seq); sb_append(&seq, "", 0);
        for (;;)
        {
            ls_next(&s); ++lineNum;
            if (!ls_valid(&s))
            {
                snprintf(err, errcap, "%s: expected sequence data or quality header at line %llu", in->name, (unsigned long long)lineNum);
                rc = -1; goto done;
            }
            fq_line(&s, &l, &n);
            if (n > 0 && (l[0] == '@' || l[0] == '+')) break;
            sb_append(&seq, l, n);
        }
        if (!(n > 0 && l[0] == '+'))
        {
            snprintf(err, errcap, "%s: expected '+' at beginning of line %llu", in->name, (unsigned long long)lineNum);
            rc = -1; break;
        }
        if (n - 1 > 0 && !(n - 1 == label.n && !memcmp(l + 1, label.p, label.n)))
        {
            snprintf(err, errcap, "%s: quality title does not match sequence title at line %llu", in->name, (unsigned long long)lineNum);
            rc = -1; break;
        }
        sb_clear(&qual); sb_append(&qual, "", 0);
        for (;;)
        {
            ls_next(&s); ++lineNum;
            if (!ls_valid(&s)) break;
            fq_line(&s, &l, &n);
            if (n > 0 && (l[0] == '@' || l[0] == '+'))
            {
                if (qual.n >= seq.n) break;
            }
            sb_append(&qual, l, n);
        }
        if (seq.n != qual.n)
        {
            snprintf(err, errcap, "%s: length mistmatch between sequence and quality data just before line %llu", in->name, (unsigned long long)lineNum);
            rc = -1; break;
        }
        ++*nreads;
        cb(ctx, seq.p, seq.n);
    }
done:
    free(seq.p); free(qual.p); free(label.p);
    return rc;
}

/* FastaParser::next loop.  FastaParser.hh:51-87 (mLineNum starts at 0; no '\r' strip). */
static int parse_fasta(const go_input* in, read_cb cb, void* ctx, uint64_t* nreads, char* err, size_t errcap)
{
    line_src s; ls_init(&s, in->data, in->size);
    sbuf seq = {0};
    uint64_t lineNum = 0;
    int rc = 0;
    for (;;)
    {
        if (!ls_valid(&s)) break;
        if (!(s.len > 0 && s.line[0] == '>'))
        {
            snprintf(err, errcap, "%s: expected '>' at beginning of line %llu", in->name, (unsigned long long)lineNum);
            rc = -1; break;
        }
        sb_clear(&seq); sb_append(&seq, "", 0);
        for (;;)
        {
            ls_next(&s); ++lineNum;
            if (!ls_valid(&s)) break;
            if (s.len > 0 && s.line[0] == '>') break;
            sb_append(&seq, s.line, s.len);
        }
        ++*nreads;
        cb(ctx, seq.p, seq.n);
    }
    free(seq.p);
    return rc;
}

/* LineParser::next: every line is a read.  LineParser.hh:71-82 */
static int parse_lines(const go_input* in, read_cb cb, void* ctx, uint64_t* nreads)
{
    line_src s; ls_init(&s, in->data, in->size);
    while (ls_valid(&s))
    {
        ++*nreads;
        cb(ctx, s.line, s.len);
        ls_next(&s);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* key collection (role of KmerizingAdapter / ReverseComplementAdapter + the hot loop)   */
/* ------------------------------------------------------------------------------------ */

typedef struct { go_keys* out; unsigned len; int mode; go_key* tmp; size_t tmpcap; } collect_ctx;

static void keys_push(go_keys* ks, go_key k)
{
    if (ks->n == ks->cap)
    {
        ks->cap = ks->cap ? ks->cap * 2 : 4096;
        ks->keys = (go_key*)realloc(ks->keys, ks->cap * sizeof(go_key));
    }
    ks->keys[ks->n++] = k;
}

static void collect_read(void* vctx, const char* seq, size_t len)
{
    collect_ctx* c = (collect_ctx*)vctx;
    if (len < c->len) return;
    size_t maxw = len - c->len + 1;
    if (maxw > c->tmpcap)
    {
        c->tmpcap = maxw * 2;
        c->tmp = (go_key*)realloc(c->tmp, c->tmpcap * sizeof(go_key));
    }
    size_t n = go_kmerize(seq, len, c->len, c->tmp, c->tmpcap);
    c->out->nwindows += n;
    for (size_t i = 0; i < n; ++i)
    {
        if (c->mode == 0)
        {
            /* KmerizingAdapter + normalize: GossCmdBuildKmerSet.tcc:246-249 */
            keys_push(c->out, go_normalize(c->tmp[i], c->len));
        }
        else
        {
            /* ReverseComplementAdapter: the rho-mer, then its reverse complement
             * (ReverseComplementAdapter.hh:34-55); no normalisation. */
            keys_push(c->out, c->tmp[i]);
            keys_push(c->out, go_revcomp(c->tmp[i], c->len));
        }
    }
}

void go_keys_free(go_keys* k) { free(k->keys); memset(k, 0, sizeof *k); }

int go_collect(const go_input* in, size_t nin, unsigned len, int mode, go_keys* out, char* err, size_t errcap)
{
    collect_ctx c; memset(&c, 0, sizeof c);
    c.out = out; c.len = len; c.mode = mode;
    int rc = 0;
    for (size_t i = 0; i < nin && rc == 0; ++i)
    {
        switch (in[i].kind)
        {
            case GO_IN_LINE:  rc = parse_lines(&in[i], collect_read, &c, &out->nreads); break;
            case GO_IN_FASTA: rc = parse_fasta(&in[i], collect_read, &c, &out->nreads, err, errcap); break;
            case GO_IN_FASTQ: rc = parse_fastq(&in[i], collect_read, &c, &out->nreads, err, errcap); break;
            default: snprintf(err, errcap, "bad input kind"); rc = -1;
        }
    }
    free(c.tmp);
    /* KmerizingAdapter::checkValid: no read at all in any input.  KmerizingAdapter.hh:70-78 */
    if (rc == 0 && out->nreads == 0)
    {
        snprintf(err, errcap, "No valid reads.");
        rc = -1;
    }
    return rc;
}

/* LSD radix sort on 128-bit keys, 8-bit digits, skipping digits that are constant. */
static void radix_sort_keys(go_key* a, size_t n)
{
    if (n < 2) return;
    go_key* b = (go_key*)malloc(n * sizeof(go_key));
    size_t* cnt = (size_t*)malloc(256 * sizeof(size_t));
    for (unsigned d = 0; d < 16; ++d)
    {
        memset(cnt, 0, 256 * sizeof(size_t));
        unsigned sh = (d & 7) * 8;
        for (size_t i = 0; i < n; ++i)
        {
            uint64_t w = d < 8 ? a[i].lo : a[i].hi;
            ++cnt[(w >> sh) & 0xFF];
        }
        int constant = 0;
        for (unsigned v = 0; v < 256; ++v) if (cnt[v] == n) constant = 1;
        if (constant) continue;
        size_t sum = 0;
        for (unsigned v = 0; v < 256; ++v) { size_t c = cnt[v]; cnt[v] = sum; sum += c; }
        for (size_t i = 0; i < n; ++i)
        {
            uint64_t w = d < 8 ? a[i].lo : a[i].hi;
            b[cnt[(w >> sh) & 0xFF]++] = a[i];
        }
        memcpy(a, b, n * sizeof(go_key));
    }
    free(cnt); free(b);
}

/* sort, then merge equal neighbours summing counts: what BackyardHash::sort + flush deliver
 * to the Builder (GossCmdBuildKmerSet.tcc:167-210, GossCmdBuildGraph.cc:222-266). */
size_t go_sort_count(go_key* keys, size_t n, uint64_t* counts)
{
    radix_sort_keys(keys, n);
    size_t m = 0;
    for (size_t i = 0; i < n; )
    {
        size_t j = i + 1;
        while (j < n && keys[j].lo == keys[i].lo && keys[j].hi == keys[i].hi) ++j;
        keys[m] = keys[i];
        counts[m] = (uint64_t)(j - i);
        ++m; i = j;
    }
    return m;
}

/* ------------------------------------------------------------------------------------ */
/* builders                                                                              */
/* ------------------------------------------------------------------------------------ */

/* WordyBitVector::Builder.  WordyBitVector.hh:54-134, WordyBitVector.cc:18-29 */
typedef struct { go_fs* fs; size_t f; uint64_t currPos, fileWordNum, currWordNum, currWord; } wbv_builder;

static void wbv_init(wbv_builder* b, go_fs* fs, const char* name)
{
    memset(b, 0, sizeof *b); b->fs = fs; b->f = fs_out_idx(fs, name);
}
static void wbv_flush(wbv_builder* b)
{
    while (b->fileWordNum < b->currWordNum)
    {
        uint64_t zero = 0;
        f_write(b->fs, b->f, &zero, 8);
        ++b->fileWordNum;
    }
    f_write(b->fs, b->f, &b->currWord, 8);
    ++b->fileWordNum;
}
static void wbv_pad_to(wbv_builder* b, uint64_t pos)
{
    uint64_t dest = pos / 64;
    if (b->currWordNum < dest)
    {
        wbv_flush(b);
        b->currWordNum = dest;
        b->currWord = 0;
    }
    b->currPos = pos;
}
static void wbv_push_backx(wbv_builder* b, int bit)
{
    uint64_t w = b->currPos / 64, bb = b->currPos % 64;
    if (w != b->currWordNum)
    {
        wbv_flush(b);
        b->currWordNum = w;
        b->currWord = 0;
    }
    if (bit) b->currWord |= 1ULL << bb;
    ++b->currPos;
}
static void wbv_push(wbv_builder* b, uint64_t pos) { wbv_pad_to(b, pos); wbv_push_backx(b, 1); }
static void wbv_pad(wbv_builder* b, uint64_t pos) { wbv_pad_to(b, pos + 1); }
static void wbv_end(wbv_builder* b) { wbv_flush(b); }

/* DenseSelect::Header.  DenseArray.hh:98-136 (16 x u64 = 128 bytes) */
typedef struct {
    uint64_t version, flags, indexArrayOffset, rankArrayOffset;
    uint64_t logBlockSize, blockSize, logSampleRate, sampleRate;
    uint64_t numBlocks, indexSize, smallBlocks, smallBlocksSize;
    uint64_t intermediateBlocks, intermediateBlocksSize, largeBlocks, largeBlocksSize;
} ds_header;

enum { tSmall = 0, tFullSpill64 = 1, tFullSpill32 = 2, tFullSpill16 = 3, tFullSpill8 = 4, tIntermediate = 5 };
#define DS_VERSION 2012092701ULL
#define DS_TYPEMASK 7ULL

typedef struct {
    go_fs* fs; size_t f;
    ds_header h;
    uint64_t* cur; size_t ncur;           /* mCurrBlock */
    uint64_t* index; uint64_t* rank; size_t nblk, capblk;
} ds_builder;

static void ds_align(ds_builder* b, uint64_t mask)
{
    /* DenseIndexBuilderBase::alignFilePos.  DenseArray.cc:436-443 */
    uint8_t z = 0;
    for (uint64_t pos = f_tell(b->fs, b->f); (pos & mask) != 0; ++pos) f_write(b->fs, b->f, &z, 1);
}

/* DenseSelect::Builder::Builder.  DenseArray.cc:678-694 + Header ctor :20-33 */
static void ds_init(ds_builder* b, go_fs* fs, const char* name, int invert)
{
    memset(b, 0, sizeof *b);
    b->fs = fs; b->f = fs_out_idx(fs, name);
    b->h.version = DS_VERSION;
    b->h.flags = invert ? 1 : 0;
    b->h.logBlockSize = 13; b->h.blockSize = 1ULL << 13;
    b->h.logSampleRate = 6; b->h.sampleRate = 1ULL << 6;
    f_write(fs, b->f, &b->h, sizeof b->h);
    ds_align(b, 4096 - 1);
    b->cur = (uint64_t*)malloc(sizeof(uint64_t) << 13);
}

/* DenseSelect::Builder::flush.  DenseArray.cc:446-647 */
static void ds_flush(ds_builder* b)
{
    if (b->ncur == 0) return;
    if (b->nblk == b->capblk)
    {
        b->capblk = b->capblk ? b->capblk * 2 : 64;
        b->index = (uint64_t*)realloc(b->index, b->capblk * 8);
        b->rank = (uint64_t*)realloc(b->rank, b->capblk * 8);
    }
    uint64_t fileposition = f_tell(b->fs, b->f);
    uint64_t pp = b->cur[0], p = b->cur[b->ncur - 1];
    uint64_t span = p - pp;
    const uint64_t sampleRate = b->h.sampleRate;
    b->rank[b->nblk] = pp;
    if (span >= (1ULL << 24) || b->ncur < b->h.blockSize)
    {
        /* Large block, or last block. */
        if (span < (1ULL << 32))
        {
            for (size_t i = 0; i < b->ncur; ++i)
            {
                uint32_t pos = (uint32_t)(b->cur[i] - pp);
                f_write(b->fs, b->f, &pos, 4);
                b->h.largeBlocksSize += 4;
            }
            b->index[b->nblk] = fileposition | tFullSpill32;
        }
        else
        {
            for (size_t i = 0; i < b->ncur; ++i)
            {
                uint64_t pos = b->cur[i];         /* absolute, not relative */
                f_write(b->fs, b->f, &pos, 8);
                b->h.largeBlocksSize += 8;
            }
            b->index[b->nblk] = fileposition | tFullSpill64;
        }
        ++b->h.largeBlocks;
    }
    else if (span >= (1ULL << 16))
    {
        /* Intermediate block. */
        uint64_t subRankStart[128]; uint32_t subBlockRange[128]; uint16_t internalPtr[128];
        size_t nsub = 0;
        for (size_t is = 0; is < b->ncur; is += sampleRate)
        {
            subRankStart[nsub] = b->cur[is];
            subBlockRange[nsub] = (uint32_t)(b->cur[is + sampleRate - 1] - b->cur[is]);
            ++nsub;
            uint32_t s = (uint32_t)(b->cur[is] - pp);
            f_write(b->fs, b->f, &s, 4);
            b->h.intermediateBlocksSize += 4;
        }
        uint64_t subBlockBase = nsub * (4 + 2);
        subBlockBase = (subBlockBase + DS_TYPEMASK) & ~DS_TYPEMASK;
        for (size_t i = 0; i < nsub; ++i)
        {
            uint16_t ip = 0;
            if (subBlockRange[i] <= (b->h.blockSize >> b->h.logSampleRate))
            {
                ip = tSmall;            /* bit scan */
            }
            else if (subBlockRange[i] < (1ULL << 8))
            {
                ip = (uint16_t)subBlockBase | tFullSpill8;
                subBlockBase += sampleRate * 1;
            }
            else if (subBlockRange[i] < (1ULL << 16))
            {
                ip = (uint16_t)subBlockBase | tFullSpill16;
                subBlockBase += sampleRate * 2;
            }
            else
            {
                ip = (uint16_t)subBlockBase | tFullSpill32;
                subBlockBase += sampleRate * 4;
            }
            internalPtr[i] = ip;
            f_write(b->fs, b->f, &ip, 2);
            b->h.intermediateBlocksSize += 2;
            subBlockBase = (subBlockBase + DS_TYPEMASK) & ~DS_TYPEMASK;
        }
        for (size_t i = 0; i < nsub; ++i)
        {
            if (!internalPtr[i]) continue;
            uint64_t startRank = subRankStart[i];
            ds_align(b, DS_TYPEMASK);
            for (size_t j = i << 6; j < (i + 1) << 6; ++j)
            {
                switch (internalPtr[i] & DS_TYPEMASK)
                {
                    case tFullSpill8:  { uint8_t s = (uint8_t)(b->cur[j] - startRank);  f_write(b->fs, b->f, &s, 1); b->h.intermediateBlocksSize += 1; break; }
                    case tFullSpill16: { uint16_t s = (uint16_t)(b->cur[j] - startRank); f_write(b->fs, b->f, &s, 2); b->h.intermediateBlocksSize += 2; break; }
                    case tFullSpill32: { uint32_t s = (uint32_t)(b->cur[j] - startRank); f_write(b->fs, b->f, &s, 4); b->h.intermediateBlocksSize += 4; break; }
                }
            }
        }
        b->index[b->nblk] = fileposition | tIntermediate;
        ++b->h.intermediateBlocks;
    }
    else
    {
        /* Small block. */
        for (size_t is = 0; is < b->ncur; is += sampleRate)
        {
            uint16_t s = (uint16_t)(b->cur[is] - pp);
            f_write(b->fs, b->f, &s, 2);
            b->h.smallBlocksSize += 2;
        }
        b->index[b->nblk] = fileposition | tSmall;
        ++b->h.smallBlocks;
    }
    ++b->nblk;
    b->ncur = 0;
    ds_align(b, DS_TYPEMASK);
    ++b->h.numBlocks;
}

static void ds_push(ds_builder* b, uint64_t pos)
{
    b->cur[b->ncur++] = pos;
    if (b->ncur == b->h.blockSize) ds_flush(b);
}

/* DenseSelect::Builder::end.  DenseArray.cc:650-675 */
static void ds_end(ds_builder* b)
{
    ds_flush(b);
    ds_align(b, 15);
    b->h.indexArrayOffset = f_tell(b->fs, b->f);
    if (b->nblk) { f_write(b->fs, b->f, b->index, b->nblk * 8); b->h.indexSize += b->nblk * 8; }
    b->h.rankArrayOffset = f_tell(b->fs, b->f);
    if (b->nblk) { f_write(b->fs, b->f, b->rank, b->nblk * 8); b->h.indexSize += b->nblk * 8; }
    f_seek(b->fs, b->f, 0);
    f_write(b->fs, b->f, &b->h, sizeof b->h);
    free(b->cur); free(b->index); free(b->rank);
    b->cur = b->index = b->rank = NULL;
}

/* IntegerArray::builder column layout.  IntegerArray.cc:259-357, StackedArray.hh:152-178:
 * Stacked<U,L> stores value>>bits(L) (truncated to U) in <base>.upr and the low bits(L)
 * in <base>.lwr, recursively. */
typedef struct { char suffix[24]; unsigned bytes; unsigned shift; } ia_col;

static size_t ia_layout(unsigned bits, const char* prefix, unsigned shift, ia_col* cols, size_t n)
{
    unsigned ub = 0, lb = 0;
    switch (bits)
    {
        case 8: case 16: case 32: case 64:
            snprintf(cols[n].suffix, sizeof cols[n].suffix, "%s", prefix);
            cols[n].bytes = bits / 8; cols[n].shift = shift;
            return n + 1;
        case 24:  ub = 8;  lb = 16; break;
        case 40:  ub = 8;  lb = 32; break;
        case 48:  ub = 16; lb = 32; break;
        case 56:  ub = 8;  lb = 48; break;
        case 72:  ub = 8;  lb = 64; break;
        case 80:  ub = 16; lb = 64; break;
        case 88:  ub = 8;  lb = 80; break;
        case 96:  ub = 32; lb = 64; break;
        case 104: ub = 8;  lb = 96; break;
        case 112: ub = 16; lb = 96; break;
        case 120: ub = 24; lb = 96; break;
        case 128: ub = 64; lb = 64; break;
        default: return (size_t)-1;
    }
    char p[24];
    snprintf(p, sizeof p, "%s.upr", prefix);
    n = ia_layout(ub, p, shift + lb, cols, n);
    snprintf(p, sizeof p, "%s.lwr", prefix);
    n = ia_layout(lb, p, shift, cols, n);
    return n;
}

typedef struct { go_fs* fs; size_t f[4]; ia_col cols[4]; size_t ncols; } ia_builder;

static int ia_init(ia_builder* b, go_fs* fs, unsigned bits, const char* base)
{
    b->fs = fs;
    b->ncols = ia_layout(bits, "", 0, b->cols, 0);
    if (b->ncols == (size_t)-1) return -1;
    char name[4096];
    for (size_t i = 0; i < b->ncols; ++i)
    {
        snprintf(name, sizeof name, "%s%s", base, b->cols[i].suffix);
        b->f[i] = fs_out_idx(fs, name);
    }
    return 0;
}
static void ia_push(ia_builder* b, u128 v)
{
    for (size_t i = 0; i < b->ncols; ++i)
    {
        uint64_t w = (uint64_t)(b->cols[i].shift >= 128 ? 0 : (v >> b->cols[i].shift));
        f_write(b->fs, b->f[i], &w, b->cols[i].bytes);      /* little endian truncation */
    }
}

/* SparseArray::Header.  SparseArray.hh:60-72 (64 bytes) */
typedef struct { uint64_t version, D, quantizedD; go_key DMask; go_key size; uint64_t count; } sa_header;
#define SA_VERSION 2012030501ULL

/* SparseArray::Builder::d.  SparseArray.cc:47-72; n via BigInteger::asDouble
 * (BigInteger.hh:181-190). */
uint64_t go_sparse_d(go_key N, uint64_t M)
{
    double scale = (double)UINT64_MAX + 1;
    double n = 0;
    n = n * scale + (double)N.hi;
    n = n * scale + (double)N.lo;
    double m = (double)M;
    double d0 = log2(n / ((1 + m) * 1.4426950408889634));
    uint64_t d = (uint64_t)ceil(d0);
    if (d < 8) d = 8;
    else if (d > 128) d = 128;
    return d;
}

typedef struct {
    go_fs* fs;
    sa_header h;
    uint64_t bitNum, lastHighBit;
    wbv_builder hi;
    ds_builder d0, d1;
    ia_builder lo;
    size_t fhdr;
} sa_builder;

/* SparseArray::Builder::Builder(base, fac, N, M).  SparseArray.cc:106-117 + Header :11-15 */
static int sa_init_d(sa_builder* b, go_fs* fs, const char* base, uint64_t D)
{
    char name[4096];
    memset(b, 0, sizeof *b);
    b->fs = fs;
    b->h.version = SA_VERSION;
    b->h.D = D;
    b->h.quantizedD = 8 * ((D + 7) / 8);
    b->h.DMask = u2k(D >= 128 ? ~(u128)0 : ((((u128)1) << D) - 1));
    snprintf(name, sizeof name, "%s.high-bits", base); wbv_init(&b->hi, fs, name);
    snprintf(name, sizeof name, "%s-d0", base); ds_init(&b->d0, fs, name, 1);
    snprintf(name, sizeof name, "%s-d1", base); ds_init(&b->d1, fs, name, 0);
    snprintf(name, sizeof name, "%s.low-bits", base);
    if (ia_init(&b->lo, fs, (unsigned)b->h.quantizedD, name)) return -1;
    snprintf(name, sizeof name, "%s.header", base); b->fhdr = fs_out_idx(fs, name);
    return 0;
}

/* SparseArray::Builder::push_back.  SparseArray.hh:87-118 */
static int sa_push(sa_builder* b, go_key pos)
{
    u128 p = k2u(pos);
    u128 nd = b->h.D >= 128 ? 0 : (p >> b->h.D);
    if (nd >> 64) return -1;
    uint64_t h = (uint64_t)nd;
    h += b->bitNum;
    ++b->bitNum;
    wbv_push(&b->hi, h);
    while (b->lastHighBit < h)
    {
        ds_push(&b->d0, b->lastHighBit);
        ++b->lastHighBit;
    }
    ds_push(&b->d1, h);
    b->lastHighBit = ++h;
    ia_push(&b->lo, p & k2u(b->h.DMask));
    b->h.size = u2k(p + 1);
    ++b->h.count;
    return 0;
}

/* SparseArray::Builder::end.  SparseArray.cc:75-103 */
static int sa_end(sa_builder* b, go_key N)
{
    b->h.size = N;
    u128 nd = b->h.D >= 128 ? 0 : (k2u(N) >> b->h.D);
    if (nd >> 64) return -1;
    uint64_t h = (uint64_t)nd + b->h.count + 2;
    while (b->lastHighBit < h)
    {
        ds_push(&b->d0, b->lastHighBit);
        ++b->lastHighBit;
    }
    wbv_pad(&b->hi, b->lastHighBit);
    wbv_end(&b->hi);
    ds_end(&b->d0);
    ds_end(&b->d1);
    f_write(b->fs, b->fhdr, &b->h, sizeof b->h);
    return 0;
}

int go_write_sparse_array(go_fs* fs, const char* base, go_key N_ctor, uint64_t M, const go_key* pos, size_t n, go_key N_end)
{
    sa_builder b;
    if (sa_init_d(&b, fs, base, go_sparse_d(N_ctor, M))) return -1;
    for (size_t i = 0; i < n; ++i) if (sa_push(&b, pos[i])) return -1;
    return sa_end(&b, N_end);
}

/* KmerSet::Builder.  KmerSet.hh:32-103: SparseArray at <base>.kmers with N = 4^K, M;
 * 24-byte header {version, K, count} written by end(). */
int go_write_kmer_set(go_fs* fs, const char* base, unsigned K, const go_key* keys, size_t n, uint64_t M)
{
    if (K > 63) return -2;                         /* KmerSet::MaxK.  KmerSet.hh:30,89-95 */
    char name[4096];
    snprintf(name, sizeof name, "%s.kmers", base);
    go_key N = u2k(((u128)1) << (2 * K));
    if (go_write_sparse_array(fs, name, N, M, keys, n, N)) return -1;
    uint64_t hdr[3] = { 2011101701ULL, K, (uint64_t)n };
    snprintf(name, sizeof name, "%s.header", base);
    size_t f = fs_out_idx(fs, name);
    f_write(fs, f, hdr, sizeof hdr);
    return 0;
}

/* VariableByteArray::Builder.  VariableByteArray.hh:76-118, VariableByteArray.cc:21-43 */
typedef struct {
    go_fs* fs;
    uint64_t pos0, pos1;
    size_t f0, f1, f2;
    sa_builder p1, p2;
} vba_builder;

static int vba_init(vba_builder* b, go_fs* fs, const char* base, uint64_t numItems)
{
    char name[4096];
    memset(b, 0, sizeof *b);
    b->fs = fs;
    snprintf(name, sizeof name, "%s.ord0", base); b->f0 = fs_out_idx(fs, name);
    go_key N = { numItems, 0 };
    uint64_t M = (uint64_t)((double)numItems * 0.001);
    snprintf(name, sizeof name, "%s.ord1p", base);
    if (sa_init_d(&b->p1, fs, name, go_sparse_d(N, M))) return -1;
    snprintf(name, sizeof name, "%s.ord1", base); b->f1 = fs_out_idx(fs, name);
    snprintf(name, sizeof name, "%s.ord2p", base);
    if (sa_init_d(&b->p2, fs, name, go_sparse_d(N, M))) return -1;
    snprintf(name, sizeof name, "%s.ord2", base); b->f2 = fs_out_idx(fs, name);
    return 0;
}
static void vba_push(vba_builder* b, uint32_t v)
{
    uint64_t pos = b->pos0++;
    uint8_t b0 = (uint8_t)(v & 0xff);
    f_write(b->fs, b->f0, &b0, 1);
    if (!(v >>= 8)) return;
    go_key p = { pos, 0 };
    sa_push(&b->p1, p);
    pos = b->pos1++;
    uint8_t b1 = (uint8_t)(v & 0xff);
    f_write(b->fs, b->f1, &b1, 1);
    if (!(v >>= 8)) return;
    go_key q = { pos, 0 };
    sa_push(&b->p2, q);
    uint16_t b2 = (uint16_t)(v & 0xffff);
    f_write(b->fs, b->f2, &b2, 2);
}
static void vba_end(vba_builder* b)
{
    go_key n0 = { b->pos0, 0 }, n1 = { b->pos1, 0 };
    sa_end(&b->p1, n0);
    sa_end(&b->p2, n1);
}

/* Graph::Builder.  Graph.hh:73-127, Graph.cc:115-167: 24-byte header {version, K, flags}
 * at construction; SparseArray <base>-edges (N = 4^(K+1), M); VariableByteArray
 * <base>-counts; <base>-counts-hist.txt "count\tfreq\n" ascending (keys are u64 counts,
 * values pushed to the VariableByteArray are truncated to u32). */
typedef struct { uint64_t c, f; } hist_ent;
static int hist_cmp(const void* a, const void* b)
{
    uint64_t x = ((const hist_ent*)a)->c, y = ((const hist_ent*)b)->c;
    return x < y ? -1 : x > y;
}

int go_write_graph(go_fs* fs, const char* base, unsigned K, const go_key* keys, const uint64_t* counts, size_t n, uint64_t M)
{
    if (K > 62) return -2;                         /* Graph::MaxK.  Graph.hh:89, Graph.cc:152-158 */
    char name[4096];
    uint64_t hdr[3] = { 2011101014ULL, K, 0 };
    snprintf(name, sizeof name, "%s.header", base);
    size_t fh = fs_out_idx(fs, name);
    f_write(fs, fh, hdr, sizeof hdr);

    sa_builder eb;
    snprintf(name, sizeof name, "%s-edges", base);
    go_key N = u2k(((u128)1) << (2 * K + 2));
    if (sa_init_d(&eb, fs, name, go_sparse_d(N, M))) return -1;
    vba_builder cb;
    snprintf(name, sizeof name, "%s-counts", base);
    if (vba_init(&cb, fs, name, M)) return -1;

    uint64_t* cs = (uint64_t*)malloc((n ? n : 1) * 8);
    for (size_t i = 0; i < n; ++i)
    {
        if (sa_push(&eb, keys[i])) { free(cs); return -1; }
        vba_push(&cb, (uint32_t)counts[i]);
        cs[i] = counts[i];
    }
    if (sa_end(&eb, N)) { free(cs); return -1; }
    vba_end(&cb);

    /* histogram (std::map<uint64_t,uint64_t> iteration order = ascending count) */
    hist_ent* he = (hist_ent*)malloc((n ? n : 1) * sizeof(hist_ent));
    for (size_t i = 0; i < n; ++i) { he[i].c = cs[i]; he[i].f = 1; }
    qsort(he, n, sizeof(hist_ent), hist_cmp);
    snprintf(name, sizeof name, "%s-counts-hist.txt", base);
    size_t ft = fs_out_idx(fs, name);
    for (size_t i = 0; i < n; )
    {
        size_t j = i;
        while (j < n && he[j].c == he[i].c) ++j;
        char line[64];
        int l = snprintf(line, sizeof line, "%llu\t%llu\n", (unsigned long long)he[i].c, (unsigned long long)(j - i));
        f_write(fs, ft, line, (size_t)l);
        i = j;
    }
    free(he); free(cs);
    return 0;
}

/* GossCmdBuildKmerSet::operator() single-pass branch.  GossCmdBuildKmerSet.tcc:213-306,
 * flush :167-210 (M = number of table entries = distinct keys). */
int go_build_kmer_set(go_fs* fs, const char* out, unsigned K, const go_input* in, size_t nin, uint64_t* nwindows, char* err, size_t errcap)
{
    go_keys ks; memset(&ks, 0, sizeof ks);
    if (go_collect(in, nin, K, 0, &ks, err, errcap)) { go_keys_free(&ks); return -1; }
    uint64_t* counts = (uint64_t*)malloc((ks.n ? ks.n : 1) * 8);
    size_t m = go_sort_count(ks.keys, ks.n, counts);
    if (nwindows) *nwindows = ks.nwindows;
    int rc = go_write_kmer_set(fs, out, K, ks.keys, m, m);
    if (rc == -2) snprintf(err, errcap, "unable to build a graph with k=%u", K);
    else if (rc) snprintf(err, errcap, "write error");
    free(counts); go_keys_free(&ks);
    return rc ? -1 : 0;
}

/* The same command with T worker threads, for the CPU side-by-side figure of bench.py on a
 * many-core host.  The reference feeds one hash table from T consumer threads
 * (GossCmdBuildKmerSet.tcc:226-256: one producer of k-mers, T BackgroundMultiConsumers inserting
 * into the shared BackyardHash) and then sorts it with T threads (BackyardHash::sort,
 * BackyardHash.cc:244-271); its flush into the Builder is serial (:167-210).  This restatement
 * keeps that shape with the sort-based counting of go_build_kmer_set: the reads (one line-kind
 * input) are cut into T shards at line ends, every thread parses, canonicalises, sorts and counts
 * its shard, the T sorted runs are merged range by range in parallel (equal keys summed), and the
 * KmerSet is written by one thread.  The files are those of go_build_kmer_set. */
#include <pthread.h>

typedef struct {
    const char* data; size_t size; unsigned K;
    go_keys ks; uint64_t* counts; size_t m; int rc;
} mt_shard;
static void* mt_count_shard(void* v)
{
    mt_shard* s = (mt_shard*)v;
    char err[256];
    go_input in = { GO_IN_LINE, "shard", s->data, s->size };
    memset(&s->ks, 0, sizeof s->ks);
    s->rc = go_collect(&in, 1, s->K, 0, &s->ks, err, sizeof err);
    if (s->rc) return NULL;
    s->counts = (uint64_t*)malloc((s->ks.n ? s->ks.n : 1) * 8);
    s->m = go_sort_count(s->ks.keys, s->ks.n, s->counts);
    return NULL;
}
static size_t key_lower_bound(const go_key* a, size_t n, u128 v)
{
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (k2u(a[mid]) < v) lo = mid + 1; else hi = mid; }
    return lo;
}
typedef struct { mt_shard* sh; unsigned T; u128 lo, hi; go_key* out; size_t m; } mt_range;
static void* mt_merge_range(void* v)
{
    mt_range* r = (mt_range*)v;
    size_t at[64], end[64], total = 0;
    for (unsigned t = 0; t < r->T; ++t)
    {
        at[t] = key_lower_bound(r->sh[t].ks.keys, r->sh[t].m, r->lo);
        end[t] = key_lower_bound(r->sh[t].ks.keys, r->sh[t].m, r->hi);
        total += end[t] - at[t];
    }
    r->out = (go_key*)malloc((total ? total : 1) * sizeof(go_key));
    size_t m = 0;
    for (;;)
    {
        int best = -1; u128 bv = 0;
        for (unsigned t = 0; t < r->T; ++t)
            if (at[t] < end[t]) { u128 x = k2u(r->sh[t].ks.keys[at[t]]); if (best < 0 || x < bv) { best = (int)t; bv = x; } }
        if (best < 0) break;
        for (unsigned t = 0; t < r->T; ++t)          /* equal keys of every run are one entry (a k-mer set keeps no counts) */
            if (at[t] < end[t] && k2u(r->sh[t].ks.keys[at[t]]) == bv) ++at[t];
        r->out[m++] = u2k(bv);
    }
    r->m = m;
    return NULL;
}
int go_build_kmer_set_mt(go_fs* fs, const char* out, unsigned K, const char* reads, size_t size, unsigned T,
                         uint64_t* nwindows, char* err, size_t errcap)
{
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    mt_shard sh[64]; pthread_t th[64];
    size_t begin = 0;
    for (unsigned t = 0; t < T; ++t)
    {
        size_t end = t + 1 == T ? size : size / T * (t + 1);
        if (end < begin) end = begin;
        while (end < size && end > begin && reads[end - 1] != '\n') ++end;
        sh[t].data = reads + begin; sh[t].size = end - begin; sh[t].K = K; sh[t].counts = NULL; sh[t].m = 0; sh[t].rc = 0;
        begin = end;
    }
    for (unsigned t = 0; t < T; ++t) pthread_create(&th[t], NULL, mt_count_shard, &sh[t]);
    for (unsigned t = 0; t < T; ++t) pthread_join(th[t], NULL);
    int rc = 0; uint64_t nw = 0, nreads = 0;
    for (unsigned t = 0; t < T; ++t) { if (sh[t].rc) rc = -1; nw += sh[t].ks.nwindows; nreads += sh[t].ks.nreads; }
    mt_range rg[64];
    size_t M = 0;
    if (!rc)
    {
        /* canonical k-mers are uniform on their leading bits: equal slices of the key space */
        const u128 space = 2 * K >= 128 ? ~(u128)0 : ((u128)1 << (2 * K));
        for (unsigned t = 0; t < T; ++t)
        {
            rg[t].sh = sh; rg[t].T = T; rg[t].out = NULL; rg[t].m = 0;
            rg[t].lo = space / T * t;
            rg[t].hi = t + 1 == T ? ~(u128)0 : space / T * (t + 1);
        }
        /* (the last range is closed above by the largest value: no key equals ~0) */
        for (unsigned t = 0; t < T; ++t) pthread_create(&th[t], NULL, mt_merge_range, &rg[t]);
        for (unsigned t = 0; t < T; ++t) { pthread_join(th[t], NULL); M += rg[t].m; }
    }
    for (unsigned t = 0; t < T; ++t) { free(sh[t].counts); go_keys_free(&sh[t].ks); }
    if (rc) { snprintf(err, errcap, "parse error in a shard"); return -1; }
    if (nreads == 0) { for (unsigned t = 0; t < T; ++t) free(rg[t].out); snprintf(err, errcap, "No valid reads."); return -1; }
    go_key* all = (go_key*)malloc((M ? M : 1) * sizeof(go_key));
    size_t off = 0;
    for (unsigned t = 0; t < T; ++t) { memcpy(all + off, rg[t].out, rg[t].m * sizeof(go_key)); off += rg[t].m; free(rg[t].out); }
    if (nwindows) *nwindows = nw;
    rc = go_write_kmer_set(fs, out, K, all, M, M);
    free(all);
    if (rc == -2) snprintf(err, errcap, "unable to build a graph with k=%u", K);
    else if (rc) snprintf(err, errcap, "write error");
    return rc ? -1 : 0;
}

/* GossCmdBuildGraph::operator().  GossCmdBuildGraph.cc:270-426 (rho = K+1, both strands),
 * flush :222-266. */
int go_build_graph(go_fs* fs, const char* out, unsigned K, const go_input* in, size_t nin, uint64_t* nwindows, char* err, size_t errcap)
{
    go_keys ks; memset(&ks, 0, sizeof ks);
    if (go_collect(in, nin, K + 1, 1, &ks, err, errcap)) { go_keys_free(&ks); return -1; }
    uint64_t* counts = (uint64_t*)malloc((ks.n ? ks.n : 1) * 8);
    size_t m = go_sort_count(ks.keys, ks.n, counts);
    if (nwindows) *nwindows = ks.nwindows;
    int rc = go_write_graph(fs, out, K, ks.keys, counts, m, m);
    if (rc == -2) snprintf(err, errcap, "unable to build a graph with k=%u", K);
    else if (rc) snprintf(err, errcap, "write error");
    free(counts); go_keys_free(&ks);
    return rc ? -1 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* readers (restated from the reference's read side)                                     */
/* ------------------------------------------------------------------------------------ */

typedef struct { const uint64_t* w; uint64_t nwords; } wbv;

/* WordyBitVector::select<Sense>.  WordyBitVector.tcc:17-54 */
static int wbv_select(const wbv* v, int invert, uint64_t from, uint64_t count, uint64_t* out)
{
    uint64_t w = from / 64, b = from % 64;
    if (w >= v->nwords) return -1;
    uint64_t c = count;
    uint64_t x = (invert ? ~v->w[w] : v->w[w]) >> b;
    uint64_t p = (uint64_t)__builtin_popcountll(x);
    while (c >= p)
    {
        c -= p; ++w; b = 0;
        if (w >= v->nwords) return -1;
        x = invert ? ~v->w[w] : v->w[w];
        p = (uint64_t)__builtin_popcountll(x);
    }
    *out = w * 64 + b + go_select1(x, c);
    return 0;
}

typedef struct { const uint8_t* data; size_t size; ds_header h; const uint64_t* index; const uint64_t* rank; const wbv* bits; } dsel;

/* DenseSelect::DenseSelect.  DenseArray.cc:36-91 */
static int dsel_open(dsel* d, const uint8_t* data, size_t size, const wbv* bits, int invert, char* err, size_t errcap)
{
    if (size < sizeof(ds_header)) { snprintf(err, errcap, "DenseSelect file too short"); return -1; }
    d->data = data; d->size = size; d->bits = bits;
    memcpy(&d->h, data, sizeof d->h);
    if (d->h.version != DS_VERSION) { snprintf(err, errcap, "DenseSelect version mismatch"); return -1; }
    if ((1ULL << d->h.logBlockSize) != d->h.blockSize || (1ULL << d->h.logSampleRate) != d->h.sampleRate
        || d->h.smallBlocks + d->h.intermediateBlocks + d->h.largeBlocks != d->h.numBlocks)
    { snprintf(err, errcap, "Corrupt DenseSelect index header"); return -1; }
    if ((int)(d->h.flags & 1) != invert) { snprintf(err, errcap, "DenseSelect index does not have the expected sense"); return -1; }
    if (d->h.flags >> 1) { snprintf(err, errcap, "Reserved DenseSelect flag set"); return -1; }
    d->index = (const uint64_t*)(data + d->h.indexArrayOffset);
    d->rank = (const uint64_t*)(data + d->h.rankArrayOffset);
    return 0;
}

/* DenseSelect::lookupSubBlock.  DenseArray.cc:134-182 */
static int dsel_sub(const dsel* d, const uint8_t* blockStart, uint64_t startRank, uint16_t sub, uint64_t i, uint64_t* out)
{
    blockStart += sub & ~DS_TYPEMASK;
    uint64_t r = i & (d->h.sampleRate - 1);
    if (!sub) return wbv_select(d->bits, (int)(d->h.flags & 1), startRank, r, out);
    switch (sub & DS_TYPEMASK)
    {
        case tFullSpill32: *out = startRank + ((const uint32_t*)blockStart)[r]; return 0;
        case tFullSpill16: *out = startRank + ((const uint16_t*)blockStart)[r]; return 0;
        case tFullSpill8:  *out = startRank + ((const uint8_t*)blockStart)[r]; return 0;
        default: return -1;
    }
}

/* DenseSelect::select(i).  DenseArray.cc:185-258 */
static int dsel_select(const dsel* d, uint64_t i, uint64_t* out)
{
    uint64_t blockNum = i >> d->h.logBlockSize;
    if (blockNum >= d->h.numBlocks) return -1;
    uint64_t startRank = d->rank[blockNum];
    uint64_t il = d->index[blockNum];
    const uint8_t* block = d->data + (il & ~DS_TYPEMASK);
    uint64_t subBlockOffset = (i & (d->h.blockSize - 1)) >> d->h.logSampleRate;
    switch (il & DS_TYPEMASK)
    {
        case tSmall:
        {
            startRank += ((const uint16_t*)block)[subBlockOffset];
            uint64_t r = i & (d->h.sampleRate - 1);
            return wbv_select(d->bits, (int)(d->h.flags & 1), startRank, r, out);
        }
        case tFullSpill64: *out = ((const uint64_t*)block)[i & (d->h.blockSize - 1)]; return 0;
        case tFullSpill32: *out = startRank + ((const uint32_t*)block)[i & (d->h.blockSize - 1)]; return 0;
        case tFullSpill16: *out = startRank + ((const uint16_t*)block)[i & (d->h.blockSize - 1)]; return 0;
        case tFullSpill8:  *out = startRank + ((const uint8_t*)block)[i & (d->h.blockSize - 1)]; return 0;
        case tIntermediate:
        {
            const uint32_t* b = (const uint32_t*)block;
            const uint16_t* sbs = (const uint16_t*)(block + (4u << (d->h.logBlockSize - d->h.logSampleRate)));
            return dsel_sub(d, block, startRank + b[subBlockOffset], sbs[subBlockOffset], i, out);
        }
        default: return -1;
    }
}

struct go_sparse {
    sa_header h;
    wbv hi;
    dsel d0, d1;
    ia_col cols[4]; size_t ncols; const uint8_t* col[4]; uint64_t nlow;
};

static const uint8_t* fs_get(const go_fs* fs, const char* name, size_t* n, char* err, size_t errcap)
{
    int i = go_fs_find(fs, name);
    if (i < 0) { snprintf(err, errcap, "missing file %s", name); return NULL; }
    *n = fs->files[i].size;
    return fs->files[i].data ? fs->files[i].data : (const uint8_t*)"";
}

/* SparseArray::SparseArray(base, fac).  SparseArray.cc:175-194 */
go_sparse* go_sparse_open(const go_fs* fs, const char* base, char* err, size_t errcap)
{
    char name[4096]; size_t n;
    go_sparse* s = (go_sparse*)calloc(1, sizeof *s);
    snprintf(name, sizeof name, "%s.header", base);
    const uint8_t* p = fs_get(fs, name, &n, err, errcap);
    if (!p || n < sizeof(sa_header)) { if (p) snprintf(err, errcap, "short header %s", name); free(s); return NULL; }
    memcpy(&s->h, p, sizeof s->h);
    if (s->h.version != SA_VERSION) { snprintf(err, errcap, "SparseArray version mismatch"); free(s); return NULL; }
    snprintf(name, sizeof name, "%s.high-bits", base);
    p = fs_get(fs, name, &n, err, errcap);
    if (!p) { free(s); return NULL; }
    s->hi.w = (const uint64_t*)p; s->hi.nwords = n / 8;
    snprintf(name, sizeof name, "%s-d0", base);
    p = fs_get(fs, name, &n, err, errcap);
    if (!p || dsel_open(&s->d0, p, n, &s->hi, 1, err, errcap)) { free(s); return NULL; }
    snprintf(name, sizeof name, "%s-d1", base);
    p = fs_get(fs, name, &n, err, errcap);
    if (!p || dsel_open(&s->d1, p, n, &s->hi, 0, err, errcap)) { free(s); return NULL; }
    s->ncols = ia_layout((unsigned)s->h.quantizedD, "", 0, s->cols, 0);
    if (s->ncols == (size_t)-1) { snprintf(err, errcap, "bad quantizedD"); free(s); return NULL; }
    for (size_t i = 0; i < s->ncols; ++i)
    {
        snprintf(name, sizeof name, "%s.low-bits%s", base, s->cols[i].suffix);
        s->col[i] = fs_get(fs, name, &n, err, errcap);
        if (!s->col[i]) { free(s); return NULL; }
        s->nlow = n / s->cols[i].bytes;
    }
    /* the DenseSelect structs hold a pointer to s->hi: fix after the struct settled */
    s->d0.bits = &s->hi; s->d1.bits = &s->hi;
    return s;
}
void go_sparse_close(go_sparse* s) { free(s); }
uint64_t go_sparse_count(const go_sparse* s) { return s->h.count; }
go_key go_sparse_size(const go_sparse* s) { return s->h.size; }

static u128 sa_low(const go_sparse* s, uint64_t i)
{
    u128 v = 0;
    for (size_t c = 0; c < s->ncols; ++c)
    {
        uint64_t w = 0;
        memcpy(&w, s->col[c] + i * s->cols[c].bytes, s->cols[c].bytes);
        v |= (u128)w << s->cols[c].shift;
    }
    return v;
}

uint64_t go_sparse_d0_select(const go_sparse* s, uint64_t i) { uint64_t o = ~0ULL; dsel_select(&s->d0, i, &o); return o; }
uint64_t go_sparse_d1_select(const go_sparse* s, uint64_t i) { uint64_t o = ~0ULL; dsel_select(&s->d1, i, &o); return o; }

/* SparseArray::select.  SparseArray.hh:311-325 */
go_key go_sparse_select(const go_sparse* s, uint64_t rnk)
{
    u128 pos = 0;
    if (s->h.D < 128)
    {
        pos |= go_sparse_d1_select(s, rnk);
        pos -= rnk;
        pos <<= s->h.D;
    }
    pos |= sa_low(s, rnk);
    return u2k(pos);
}

/* SparseArray::findLowOrderGroup.  SparseArray.hh:345-364 */
static void sa_group(const go_sparse* s, uint64_t posD, uint64_t* b, uint64_t* e)
{
    if (s->h.D >= 128) { *b = 0; *e = s->nlow; return; }
    if (!posD) { *b = 0; *e = go_sparse_d0_select(s, 0); return; }
    uint64_t r1 = go_sparse_d0_select(s, posD - 1) + 1, r2 = go_sparse_d0_select(s, posD);
    *b = r1 >= posD ? r1 - posD : 0;
    *e = r2 >= posD ? r2 - posD : 0;
}
static uint64_t sa_lower_bound(const go_sparse* s, uint64_t b, uint64_t e, u128 v)
{
    while (b < e)
    {
        uint64_t m = b + (e - b) / 2;
        if (sa_low(s, m) < v) b = m + 1; else e = m;
    }
    return b;
}

/* SparseArray::rank.  SparseArray.hh:296-309 */
uint64_t go_sparse_rank(const go_sparse* s, go_key pos)
{
    u128 p = k2u(pos);
    if (p >= k2u(s->h.size)) return s->h.count;
    uint64_t posD = (uint64_t)(s->h.D >= 128 ? 0 : (p >> s->h.D));
    uint64_t b, e;
    sa_group(s, posD, &b, &e);
    return sa_lower_bound(s, b, e, p & k2u(s->h.DMask));
}

/* SparseArray::access.  SparseArray.hh:246-260 */
int go_sparse_access(const go_sparse* s, go_key pos)
{
    u128 p = k2u(pos);
    uint64_t posD = (uint64_t)(s->h.D >= 128 ? 0 : (p >> s->h.D));
    uint64_t b, e;
    sa_group(s, posD, &b, &e);
    u128 j = p & k2u(s->h.DMask);
    uint64_t r = sa_lower_bound(s, b, e, j);
    if (r >= e) return 0;
    return sa_low(s, r) == j;
}

/* VariableByteArray::operator[].  VariableByteArray.hh:227-247 */
int go_vba_get(const go_fs* fs, const char* base, uint64_t i, uint32_t* out, char* err, size_t errcap)
{
    char name[4096]; size_t n0, n1, n2;
    snprintf(name, sizeof name, "%s.ord0", base);
    const uint8_t* o0 = fs_get(fs, name, &n0, err, errcap); if (!o0) return -1;
    snprintf(name, sizeof name, "%s.ord1", base);
    const uint8_t* o1 = fs_get(fs, name, &n1, err, errcap); if (!o1) return -1;
    snprintf(name, sizeof name, "%s.ord2", base);
    const uint8_t* o2 = fs_get(fs, name, &n2, err, errcap); if (!o2) return -1;
    if (i >= n0) { snprintf(err, errcap, "index out of range"); return -1; }
    uint32_t result = o0[i];
    snprintf(name, sizeof name, "%s.ord1p", base);
    go_sparse* p1 = go_sparse_open(fs, name, err, errcap); if (!p1) return -1;
    go_key ki = { i, 0 };
    int rc = 0;
    if (go_sparse_access(p1, ki))
    {
        uint64_t r1 = go_sparse_rank(p1, ki);
        result |= (uint32_t)o1[r1] << 8;
        snprintf(name, sizeof name, "%s.ord2p", base);
        go_sparse* p2 = go_sparse_open(fs, name, err, errcap);
        if (!p2) rc = -1;
        else
        {
            go_key kr = { r1, 0 };
            if (go_sparse_access(p2, kr))
            {
                uint64_t r2 = go_sparse_rank(p2, kr);
                uint16_t w; memcpy(&w, o2 + 2 * r2, 2);
                result |= (uint32_t)w << 16;
            }
            go_sparse_close(p2);
        }
    }
    go_sparse_close(p1);
    *out = result;
    return rc;
}

int go_kmer_set_header(const go_fs* fs, const char* base, uint64_t* K, uint64_t* count)
{
    char name[4096]; snprintf(name, sizeof name, "%s.header", base);
    int i = go_fs_find(fs, name);
    if (i < 0 || fs->files[i].size != 24) return -1;
    uint64_t h[3]; memcpy(h, fs->files[i].data, 24);
    if (h[0] != 2011101701ULL) return -2;
    *K = h[1]; *count = h[2];
    return 0;
}

int go_graph_header(const go_fs* fs, const char* base, uint64_t* K, uint64_t* flags)
{
    char name[4096]; snprintf(name, sizeof name, "%s.header", base);
    int i = go_fs_find(fs, name);
    if (i < 0 || fs->files[i].size != 24) return -1;
    uint64_t h[3]; memcpy(h, fs->files[i].data, 24);
    if (h[0] != 2011101014ULL) return -2;
    *K = h[1]; *flags = h[2];
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* stand-alone builders / readers in the shape of the reference's unit tests              */
/* ------------------------------------------------------------------------------------ */

/* What testDenseArray.cc:142-167 (and every later case of that file) sets up: a
 * WordyBitVector "v" filled bit by bit with push_backX (WordyBitVector.hh:96-110) over nbits
 * positions, and a DenseSelect::Builder "x" of the given sense fed the positions of the ones
 * (invert = 0) or of the zeros (invert = 1).  `ones` = ascending positions of the set bits. */
int go_write_bits_and_select(go_fs* fs, const char* vname, const char* xname, const uint64_t* ones, size_t n,
                             uint64_t nbits, int invert)
{
    wbv_builder vb; ds_builder b;
    wbv_init(&vb, fs, vname);
    ds_init(&b, fs, xname, invert);
    size_t j = 0;
    for (uint64_t i = 0; i < nbits; ++i)
    {
        int bit = j < n && ones[j] == i;
        if (bit) ++j;
        wbv_push_backx(&vb, bit);
        if (bit != (invert != 0)) ds_push(&b, i);
    }
    wbv_end(&vb);
    ds_end(&b);
    return j == n ? 0 : -1;
}

/* WordyBitVector::Builder::push(pos)* + end() (WordyBitVector.hh:54-134), as testWordyBitVector.cc:44-58 */
int go_write_bits_sparse(go_fs* fs, const char* vname, const uint64_t* ones, size_t n)
{
    wbv_builder vb;
    wbv_init(&vb, fs, vname);
    for (size_t i = 0; i < n; ++i) wbv_push(&vb, ones[i]);
    wbv_end(&vb);
    return 0;
}

static int wbv_open(const go_fs* fs, const char* vname, wbv* v)
{
    int i = go_fs_find(fs, vname);
    if (i < 0) return -1;
    v->w = (const uint64_t*)fs->files[i].data; v->nwords = fs->files[i].size / 8;
    return 0;
}

/* WordyBitVector::get (WordyBitVector.hh:173-179); positions past the last word read 0 */
int go_bits_get(const go_fs* fs, const char* vname, uint64_t pos)
{
    wbv v;
    if (wbv_open(fs, vname, &v)) return -1;
    if (pos / 64 >= v.nwords) return 0;
    return (int)((v.w[pos / 64] >> (pos % 64)) & 1);
}
uint64_t go_bits_words(const go_fs* fs, const char* vname)
{
    wbv v;
    return wbv_open(fs, vname, &v) ? 0 : v.nwords;
}
/* WordyBitVector::select1(from, count) / select0 (WordyBitVector.tcc:17-54): ~0 when out of range */
uint64_t go_bits_select(const go_fs* fs, const char* vname, int invert, uint64_t from, uint64_t count)
{
    wbv v; uint64_t out;
    if (wbv_open(fs, vname, &v) || wbv_select(&v, invert, from, count, &out)) return ~0ULL;
    return out;
}
/* WordyBitVector::popcountRange(begin, end) (WordyBitVector.hh:207-241): set bits in [begin, end) */
uint64_t go_bits_popcount_range(const go_fs* fs, const char* vname, uint64_t begin, uint64_t end)
{
    wbv v; uint64_t c = 0;
    if (wbv_open(fs, vname, &v)) return ~0ULL;
    for (uint64_t i = begin; i < end; ++i)
        if (i / 64 < v.nwords) c += (v.w[i / 64] >> (i % 64)) & 1;
    return c;
}

/* DenseSelect(bits, name, fac, invert).select(i) (DenseArray.cc:36-91,185-258) over the files of
 * go_write_bits_and_select; ~0 on failure */
uint64_t go_dense_select(const go_fs* fs, const char* vname, const char* xname, int invert, uint64_t i)
{
    wbv v; dsel d; char err[256]; uint64_t out;
    if (wbv_open(fs, vname, &v)) return ~0ULL;
    int x = go_fs_find(fs, xname);
    if (x < 0 || dsel_open(&d, fs->files[x].data, fs->files[x].size, &v, invert, err, sizeof err)) return ~0ULL;
    if (dsel_select(&d, i, &out)) return ~0ULL;
    return out;
}

/* VariableByteArray::Builder(name, fac, numItems, frac) + push_back* + end()
 * (VariableByteArray.hh:76-118, VariableByteArray.cc:21-43; frac is ignored there too), as
 * testVariableByteArray.cc:27-47 */
int go_write_vba(go_fs* fs, const char* base, const uint32_t* values, size_t n, uint64_t numItems)
{
    vba_builder b;
    if (vba_init(&b, fs, base, numItems)) return -1;
    for (size_t i = 0; i < n; ++i) vba_push(&b, values[i]);
    vba_end(&b);
    return 0;
}

/* ---- the assertion loops of the reference's unit tests, replayed over a file set ----
 * Each returns the number of failed checks (0 = the reference's test would pass), or ~0 when the
 * structures cannot be opened. */

/* testDenseArray.cc:169-190 (test2), :221-235, :267-281, :313-333, :366-376, :409-431 ...:
 * v.get(i) == bits[i] for every i; a.select(j) == position of the j-th one (zero when inverted);
 * and the pair form select(i, i+j) == (select(i), select(i+j)) for i += 113, j < 197
 * (DenseSelect::select(i, j) is two single selects read side by side, DenseArray.cc:261-423). */
uint64_t go_replay_dense_select(const go_fs* fs, const char* vname, const char* xname, int invert,
                                const uint64_t* ones, size_t n, uint64_t nbits)
{
    wbv v; dsel d; char err[256];
    if (wbv_open(fs, vname, &v)) return ~0ULL;
    int x = go_fs_find(fs, xname);
    if (x < 0 || dsel_open(&d, fs->files[x].data, fs->files[x].size, &v, invert, err, sizeof err)) return ~0ULL;
    uint64_t bad = 0, rank = 0;
    size_t j = 0;
    for (uint64_t i = 0; i < nbits; ++i)
    {
        int bit = j < n && ones[j] == i;
        if (bit) ++j;
        int got = i / 64 < v.nwords ? (int)((v.w[i / 64] >> (i % 64)) & 1) : 0;
        if (got != bit) ++bad;
        if (bit != (invert != 0))
        {
            uint64_t pos;
            if (dsel_select(&d, rank, &pos) || pos != i) ++bad;
            ++rank;
        }
    }
    return bad;
}

/* testSparseArray.cc:66-113 (test1: every position of a small universe) and :166-185, :219-236,
 * :270-289 (test3-5: every stored position): access, rank, accessAndRank, select, iterator.
 * universe = 0: only the stored positions are visited (wide universes). */
uint64_t go_replay_sparse(const go_fs* fs, const char* base, const go_key* pos, size_t n, uint64_t universe)
{
    char err[256];
    go_sparse* s = go_sparse_open(fs, base, err, sizeof err);
    if (!s) return ~0ULL;
    uint64_t bad = 0;
    if (go_sparse_count(s) != n) ++bad;
    for (size_t i = 0; i < n; ++i)
    {
        if (!go_sparse_access(s, pos[i])) ++bad;
        if (go_sparse_rank(s, pos[i]) != i) ++bad;
        go_key g = go_sparse_select(s, i);
        if (g.lo != pos[i].lo || g.hi != pos[i].hi) ++bad;
    }
    size_t j = 0;
    for (uint64_t p = 0; p < universe; ++p)
    {
        go_key k = { p, 0 };
        int bit = j < n && pos[j].hi == 0 && pos[j].lo == p;
        if ((go_sparse_access(s, k) != 0) != bit) ++bad;
        if (go_sparse_rank(s, k) != j) ++bad;
        if (bit) ++j;
    }
    /* testSparseArray.cc:143-151: rank beyond the universe stays at the count */
    if (universe)
        for (int sh = 0; sh < 4; ++sh)
        {
            go_key k = { universe << sh, 0 };
            if (go_sparse_rank(s, k) != n) ++bad;
        }
    go_sparse_close(s);
    return bad;
}

/* The bit vector of a testDenseArray.cc case embedded as the high-bits vector of a SparseArray with
 * split D (element i = (ones[i] - i) << D | i): its -d1 / -d0 files are DenseSelect structures of both
 * senses over exactly that vector, so select through them must give the ones / zeros of the case. */
uint64_t go_replay_sparse_highbits(const go_fs* fs, const char* base, const uint64_t* ones, size_t n, uint64_t nbits)
{
    char err[256];
    go_sparse* s = go_sparse_open(fs, base, err, sizeof err);
    if (!s) return ~0ULL;
    uint64_t bad = 0, zeros = 0;
    size_t j = 0;
    for (uint64_t i = 0; i < nbits; ++i)
    {
        int bit = j < n && ones[j] == i;
        int got = i / 64 < s->hi.nwords ? (int)((s->hi.w[i / 64] >> (i % 64)) & 1) : 0;
        if (got != bit) ++bad;
        if (bit) { if (go_sparse_d1_select(s, j) != i) ++bad; ++j; }
        else { if (go_sparse_d0_select(s, zeros) != i) ++bad; ++zeros; }
    }
    go_sparse_close(s);
    return bad;
}

/* testVariableByteArray.cc:49-64,88-91,117-122,150-156: a[i] == values[i] for every i */
uint64_t go_replay_vba(const go_fs* fs, const char* base, const uint32_t* values, size_t n)
{
    char err[256]; char name[4096]; size_t n0, n1, n2;
    snprintf(name, sizeof name, "%s.ord0", base);
    const uint8_t* o0 = fs_get(fs, name, &n0, err, sizeof err); if (!o0) return ~0ULL;
    snprintf(name, sizeof name, "%s.ord1", base);
    const uint8_t* o1 = fs_get(fs, name, &n1, err, sizeof err); if (!o1) return ~0ULL;
    snprintf(name, sizeof name, "%s.ord2", base);
    const uint8_t* o2 = fs_get(fs, name, &n2, err, sizeof err); if (!o2) return ~0ULL;
    snprintf(name, sizeof name, "%s.ord1p", base);
    go_sparse* p1 = go_sparse_open(fs, name, err, sizeof err); if (!p1) return ~0ULL;
    snprintf(name, sizeof name, "%s.ord2p", base);
    go_sparse* p2 = go_sparse_open(fs, name, err, sizeof err); if (!p2) { go_sparse_close(p1); return ~0ULL; }
    uint64_t bad = n0 != n;
    for (size_t i = 0; i < n && i < n0; ++i)
    {
        /* VariableByteArray::operator[] (VariableByteArray.hh:213-240) */
        uint32_t r = o0[i];
        go_key ki = { i, 0 };
        if (go_sparse_access(p1, ki))
        {
            uint64_t r1 = go_sparse_rank(p1, ki);
            if (r1 >= n1) { ++bad; continue; }
            r |= (uint32_t)o1[r1] << 8;
            go_key kr = { r1, 0 };
            if (go_sparse_access(p2, kr))
            {
                uint64_t r2 = go_sparse_rank(p2, kr);
                if (2 * r2 + 2 > n2) { ++bad; continue; }
                uint16_t w; memcpy(&w, o2 + 2 * r2, 2);
                r |= (uint32_t)w << 16;
            }
        }
        if (r != values[i]) ++bad;
    }
    go_sparse_close(p1); go_sparse_close(p2);
    return bad;
}

/* VByteCodec::encode.  VByteCodec.hh:24-104 (private spill-run format; golden bytes in
 * testVByteCodec.cc:21-62). */
size_t go_vbyte_encode(uint64_t x, uint8_t* out)
{
    size_t n = 0;
    if (x < 0x80) { out[n++] = (uint8_t)x; return n; }
    uint64_t b = 64 - (uint64_t)__builtin_clzll(x);
    uint64_t v = b / 8, l = b % 8;
    if (v + l + 1 <= 8)
    {
        out[n++] = (uint8_t)((x >> (8 * v)) | (uint8_t)~((uint8_t)0xFF >> v));
    }
    else
    {
        if (l != 0) ++v;
        out[n++] = (uint8_t)~((unsigned)0xFF >> v);
    }
    for (uint64_t i = v; i > 0; --i) out[n++] = (uint8_t)(x >> (8 * (i - 1)));
    return n;
}

/* VByteCodec::decode: the count of leading 1 bits of the first byte is the number of
 * payload bytes; remaining bits of the first byte are the most significant payload. */
uint64_t go_vbyte_decode(const uint8_t* in, size_t* used)
{
    uint8_t z = in[0];
    unsigned v = 0;
    while (v < 8 && (z & (0x80u >> v))) ++v;
    uint64_t x = v >= 8 ? 0 : (uint64_t)(z & (0xFFu >> (v + 1)));
    if (v == 7) x = 0;
    for (unsigned i = 0; i < v; ++i) x = (x << 8) | in[1 + i];
    *used = 1 + v;
    return x;
}

/* ------------------------------------------------------------------------------------ */
/* merge-kmer-sets / merge-graphs (GossCmdMerge.tcc:151-326)                             */
/* ------------------------------------------------------------------------------------ */

/* SparseArray::LazyIterator over the whole array: the i-th one of the high-bits bitmap at
 * position p gives ((p - i) << D) + low[i].  SparseArray.hh:185-224,
 * WordyBitVector::LazyIterator1. */
static int sparse_decode_all(const go_fs* fs, const char* base, go_key** out, uint64_t* n, char* err, size_t errcap)
{
    go_sparse* s = go_sparse_open(fs, base, err, errcap);
    if (!s) return -1;
    uint64_t cnt = 0;
    for (uint64_t w = 0; w < s->hi.nwords; ++w) cnt += (uint64_t)__builtin_popcountll(s->hi.w[w]);
    go_key* k = (go_key*)malloc((cnt ? cnt : 1) * sizeof(go_key));
    uint64_t i = 0;
    for (uint64_t w = 0; w < s->hi.nwords; ++w)
    {
        uint64_t x = s->hi.w[w];
        while (x)
        {
            uint64_t b = (uint64_t)__builtin_ctzll(x);
            x &= x - 1;
            u128 pos = (u128)(w * 64 + b - i);
            pos = s->h.D >= 128 ? 0 : (pos << s->h.D);
            pos += sa_low(s, i);
            k[i++] = u2k(pos);
        }
    }
    go_sparse_close(s);
    *out = k; *n = cnt;
    return 0;
}

/* VariableByteArray::GeneralIterator over the whole array.  VariableByteArray.hh:120-195 */
static int vba_decode_all(const go_fs* fs, const char* base, uint64_t n, uint32_t** out, char* err, size_t errcap)
{
    char name[4096]; size_t n0, n1, n2;
    snprintf(name, sizeof name, "%s.ord0", base);
    const uint8_t* o0 = fs_get(fs, name, &n0, err, errcap); if (!o0) return -1;
    snprintf(name, sizeof name, "%s.ord1", base);
    const uint8_t* o1 = fs_get(fs, name, &n1, err, errcap); if (!o1) return -1;
    snprintf(name, sizeof name, "%s.ord2", base);
    const uint8_t* o2 = fs_get(fs, name, &n2, err, errcap); if (!o2) return -1;
    if (n0 < n) { snprintf(err, errcap, "%s.ord0 too short", base); return -1; }
    go_key *p1 = NULL, *p2 = NULL; uint64_t c1 = 0, c2 = 0;
    snprintf(name, sizeof name, "%s.ord1p", base);
    if (sparse_decode_all(fs, name, &p1, &c1, err, errcap)) return -1;
    snprintf(name, sizeof name, "%s.ord2p", base);
    if (sparse_decode_all(fs, name, &p2, &c2, err, errcap)) { free(p1); return -1; }
    uint32_t* c = (uint32_t*)malloc((n ? n : 1) * 4);
    uint64_t i1 = 0, i2 = 0;
    for (uint64_t i = 0; i < n; ++i)
    {
        uint32_t v = o0[i];
        while (i1 < c1 && p1[i1].lo < i) ++i1;
        if (i1 < c1 && p1[i1].lo == i)
        {
            v |= (uint32_t)o1[i1] << 8;
            while (i2 < c2 && p2[i2].lo < i1) ++i2;
            if (i2 < c2 && p2[i2].lo == i1)
            {
                uint16_t w; memcpy(&w, o2 + 2 * i2, 2);
                v |= (uint32_t)w << 16;
            }
        }
        c[i] = v;
    }
    free(p1); free(p2);
    *out = c;
    return 0;
}

typedef struct { go_key* k; uint64_t* c; uint64_t n; } merge_item;

static void merge_item_free(merge_item* m) { free(m->k); free(m->c); m->k = NULL; m->c = NULL; m->n = 0; }

/* Load one object: keys + counts (KmerSet::LazyIterator yields count 1, KmerSet.hh:147-150;
 * Graph::LazyIterator the VariableByteArray value, Graph.hh:288-291). */
static int merge_load(const go_fs* fs, const char* name, int kind, uint64_t* K, uint64_t* cnt, merge_item* out, char* err, size_t errcap)
{
    char base[4096];
    uint64_t hdrK = 0, x = 0;
    if (kind == 0)
    {
        int rc = go_kmer_set_header(fs, name, &hdrK, &x);
        if (rc) { snprintf(err, errcap, rc == -2 ? "%s: version mismatch" : "unable to open graph '%s'", name); return -1; }
        snprintf(base, sizeof base, "%s.kmers", name);
    }
    else
    {
        int rc = go_graph_header(fs, name, &hdrK, &x);
        if (rc) { snprintf(err, errcap, rc == -2 ? "%s: version mismatch" : "unable to open graph '%s'", name); return -1; }
        snprintf(base, sizeof base, "%s-edges", name);
    }
    *K = hdrK;
    uint64_t n = 0;
    if (sparse_decode_all(fs, base, &out->k, &n, err, errcap)) return -1;
    out->n = n;
    out->c = (uint64_t*)malloc((n ? n : 1) * 8);
    if (kind == 0)
    {
        for (uint64_t i = 0; i < n; ++i) out->c[i] = 1;
        *cnt = x;                                       /* header count: KmerSet.hh:180-183 */
    }
    else
    {
        uint32_t* c32 = NULL;
        snprintf(base, sizeof base, "%s-counts", name);
        if (vba_decode_all(fs, base, n, &c32, err, errcap)) return -1;
        for (uint64_t i = 0; i < n; ++i) out->c[i] = c32[i];
        free(c32);
        /* count() = sum of the frequencies in <name>-counts-hist.txt.  Graph.cc:195-216 */
        snprintf(base, sizeof base, "%s-counts-hist.txt", name);
        int fi = go_fs_find(fs, base);
        if (fi < 0) { snprintf(err, errcap, "missing file %s", base); return -1; }
        /* (the file's bytes are not NUL-terminated: the numbers are read within [p, e) -- strtoull on the last
         *  line's '\n' would skip it and go on reading whatever lies behind the buffer) */
        uint64_t tot = 0;
        const char* p = (const char*)fs->files[fi].data; const char* e = p + fs->files[fi].size;
        for (;;)
        {
            uint64_t v[2]; int got = 0;
            for (; got < 2; ++got)
            {
                while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
                if (p >= e || *p < '0' || *p > '9') break;
                uint64_t x2 = 0;
                while (p < e && *p >= '0' && *p <= '9') x2 = x2 * 10 + (uint64_t)(*p++ - '0');
                v[got] = x2;
            }
            if (got < 2) break;
            tot += v[1];
        }
        *cnt = tot;
    }
    return 0;
}

/* One GossCmdMerge::merge (GossCmdMerge.tcc:207-326): k-way merge of the items, equal keys
 * summed (saturating at 2^63), estimate M = tot. */
static void merge_many(merge_item* items, size_t n, merge_item* out)
{
    uint64_t total = 0;
    for (size_t i = 0; i < n; ++i) total += items[i].n;
    go_key* k = (go_key*)malloc((total ? total : 1) * sizeof(go_key));
    uint64_t* c = (uint64_t*)malloc((total ? total : 1) * 8);
    size_t* at = (size_t*)calloc(n ? n : 1, sizeof(size_t));
    uint64_t m = 0;
    for (;;)
    {
        int best = -1;
        for (size_t i = 0; i < n; ++i)
        {
            if (at[i] >= items[i].n) continue;
            if (best < 0 || k2u(items[i].k[at[i]]) < k2u(items[best].k[at[best]])) best = (int)i;
        }
        if (best < 0) break;
        go_key key = items[best].k[at[best]];
        uint64_t cnt = items[best].c[at[best]];
        ++at[best];
        if (m && k[m - 1].lo == key.lo && k[m - 1].hi == key.hi)
        {
            uint64_t s = c[m - 1] + cnt;
            if (s > (1ULL << 63) || s < cnt) s = 1ULL << 63;
            c[m - 1] = s;
        }
        else { k[m] = key; c[m] = cnt; ++m; }
    }
    free(at);
    out->k = k; out->c = c; out->n = m;
}

int go_merge(const go_fs* in, const char* const* names, size_t nin, int kind, uint64_t max_merge,
             go_fs* outfs, const char* out_name, char* err, size_t errcap)
{
    if (nin == 0) { snprintf(err, errcap, "At least one input graph must be supplied either using --graph-in or --graphs-in.\n"); return -1; }
    if (max_merge < 2) max_merge = 2;
    /* todo list: (item, count used for the estimate).  GossCmdMerge.tcc:151-205 */
    merge_item* todo = (merge_item*)calloc(nin * 2 + 2, sizeof(merge_item));
    uint64_t* todo_cnt = (uint64_t*)calloc(nin * 2 + 2, sizeof(uint64_t));
    size_t head = 0, tail = 0;
    uint64_t K0 = 0;
    int rc = 0;
    for (size_t i = 0; i < nin; ++i)
    {
        uint64_t K = 0;
        if (merge_load(in, names[i], kind, &K, &todo_cnt[tail], &todo[tail], err, errcap)) { rc = -1; goto out; }
        if (i == 0) K0 = K;
        else if (K != K0)
        {
            snprintf(err, errcap, "all graphs involved in a merge must have the same kmer-size.\n%s has k=%llu.\n%s has k=%llu.\n",
                     names[0], (unsigned long long)K0, names[i], (unsigned long long)K);
            ++tail; rc = -1; goto out;
        }
        ++tail;
    }
    while (tail - head > max_merge)
    {
        merge_item m;
        merge_many(todo + head, (size_t)max_merge, &m);
        for (size_t i = 0; i < max_merge; ++i) merge_item_free(&todo[head + i]);
        head += (size_t)max_merge;
        if (tail >= nin * 2 + 2) { rc = -1; snprintf(err, errcap, "internal: todo overflow"); merge_item_free(&m); goto out; }
        todo[tail] = m; todo_cnt[tail] = m.n; ++tail;       /* count() of the temporary object */
    }
    {
        uint64_t tot = 0;
        for (size_t i = head; i < tail; ++i) tot += todo_cnt[i];
        merge_item m;
        merge_many(todo + head, tail - head, &m);
        if (kind == 0) rc = go_write_kmer_set(outfs, out_name, (unsigned)K0, m.k, m.n, tot);
        else rc = go_write_graph(outfs, out_name, (unsigned)K0, m.k, m.c, m.n, tot);
        if (rc) snprintf(err, errcap, "write error");
        merge_item_free(&m);
    }
out:
    for (size_t i = head; i < tail; ++i) merge_item_free(&todo[i]);
    free(todo); free(todo_cnt);
    return rc ? -1 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* intersect-kmer-sets / subtract-kmer-set / merge-and-annotate-kmer-sets                */
/* ------------------------------------------------------------------------------------ */

typedef struct { go_key* k; uint64_t n; uint64_t at; } set_iter;

/* The visiting loop of GossCmdIntersectKmerSets.cc:29-79: iterators of empty sets are left
 * out (:38-41); targetKmer starts at zero; a k-mer is visited when every iterator sits on it. */
static uint64_t intersect_visit(set_iter* it, size_t n, go_key* out)
{
    for (size_t i = 0; i < n; ++i) it[i].at = 0;
    uint64_t m = 0;
    int done = 0;
    u128 target = 0;
    while (!done)
    {
        size_t i = 0;
        for (; i < n;)
        {
            set_iter* s = &it[i];
            while (s->at < s->n && k2u(s->k[s->at]) < target) ++s->at;
            if (s->at >= s->n) { done = 1; break; }
            if (k2u(s->k[s->at]) == target) { ++i; continue; }
            target = k2u(s->k[s->at]);
            i = 0;
        }
        if (i >= n)
        {
            if (out) out[m] = u2k(target);
            ++m;
            ++it[0].at;
        }
    }
    return m;
}

int go_intersect_kmer_sets(const go_fs* in, const char* const* names, size_t nin, go_fs* outfs, const char* out_name,
                           char* err, size_t errcap)
{
    if (nin == 0) return 0;                        /* "no input k-mer sets!" (:104-110) */
    set_iter* it = (set_iter*)calloc(nin, sizeof(set_iter));
    size_t n = 0;
    uint64_t K0 = 0;
    int rc = 0;
    for (size_t i = 0; i < nin; ++i)
    {
        uint64_t K = 0, cnt = 0;
        merge_item m = {0};
        if (merge_load(in, names[i], 0, &K, &cnt, &m, err, errcap)) { rc = -1; goto out; }
        if (i == 0) K0 = K;                        /* K of the first input only (:112-116) */
        free(m.c);
        if (m.n == 0) { free(m.k); continue; }     /* invalid iterators are dropped */
        it[n].k = m.k; it[n].n = m.n; ++n;
    }
    if (n == 0) { snprintf(err, errcap, "every input k-mer set is empty (undefined in the reference)"); rc = -1; goto out; }
    {
        uint64_t cnt = intersect_visit(it, n, NULL);          /* Counter pass */
        go_key* keys = (go_key*)malloc((cnt ? cnt : 1) * sizeof(go_key));
        intersect_visit(it, n, keys);
        rc = go_write_kmer_set(outfs, out_name, (unsigned)K0, keys, cnt, cnt);
        if (rc) snprintf(err, errcap, "write error");
        free(keys);
    }
out:
    for (size_t i = 0; i < n; ++i) free(it[i].k);
    free(it);
    return rc ? -1 : 0;
}

/* GossCmdSubtractKmerSet.cc:32-85 */
int go_subtract_kmer_set(const go_fs* in, const char* lhs, const char* rhs, go_fs* outfs, const char* out_name,
                         char* err, size_t errcap)
{
    uint64_t K = 0, K1 = 0, c0 = 0, c1 = 0;
    merge_item a = {0}, b = {0};
    if (merge_load(in, lhs, 0, &K, &c0, &a, err, errcap)) return -1;
    if (merge_load(in, rhs, 0, &K1, &c1, &b, err, errcap)) { merge_item_free(&a); return -1; }
    uint8_t* rem = (uint8_t*)calloc(a.n ? a.n : 1, 1);
    uint64_t remd = 0, j = 0;
    for (uint64_t i = 0; i < a.n; ++i)
    {
        u128 t = k2u(a.k[i]);
        while (j < b.n && k2u(b.k[j]) < t) ++j;
        if (j >= b.n) break;
        if (k2u(b.k[j]) == t) { rem[i] = 1; ++remd; }
    }
    go_key* keys = (go_key*)malloc((a.n ? a.n : 1) * sizeof(go_key));
    uint64_t m = 0;
    for (uint64_t i = 0; i < a.n; ++i) if (!rem[i]) keys[m++] = a.k[i];
    int rc = go_write_kmer_set(outfs, out_name, (unsigned)K, keys, m, c0 - remd);
    if (rc) snprintf(err, errcap, "write error");
    free(keys); free(rem);
    merge_item_free(&a); merge_item_free(&b);
    return rc ? -1 : 0;
}

/* GossCmdGraphToKmerSet.cc:30-59: the edges ((K+1)-mers) of a graph that are their own normal
 * form (edge_type::isNormal, RankSelect.hh:117-124: h(x) < h(rc), or equal hashes and rc >= x),
 * as a KmerSet of k = K + 1 built with the graph's edge count as the size estimate (:45). */
int go_graph_to_kmer_set(const go_fs* in, const char* graph, go_fs* outfs, const char* out_name, char* err, size_t errcap)
{
    uint64_t K = 0, z = 0;
    merge_item a = {0};
    if (merge_load(in, graph, 1, &K, &z, &a, err, errcap)) return -1;
    const unsigned rho = (unsigned)K + 1;
    go_key* keys = (go_key*)malloc((a.n ? a.n : 1) * sizeof(go_key));
    uint64_t m = 0;
    for (uint64_t i = 0; i < a.n; ++i)
    {
        const go_key x = a.k[i], rc = go_revcomp(x, rho);
        const uint64_t h0 = go_hash(x), h1 = go_hash(rc);
        if (h0 < h1 || (h0 == h1 && k2u(rc) >= k2u(x))) keys[m++] = x;
    }
    int rc = go_write_kmer_set(outfs, out_name, rho, keys, m, z);
    if (rc) snprintf(err, errcap, "write error");
    free(keys);
    merge_item_free(&a);
    return rc ? -1 : 0;
}

/* GossCmdMergeAndAnnotateKmerSets.cc:30-206: union of two k-mer sets built with the exact count,
 * plus <out>.lhs-bits / <out>.rhs-bits (WordyBitVector, one bit per k-mer of the union).
 * stats = { lhs count, rhs count, common } (the line printed on stdout, :204). */
int go_merge_and_annotate(const go_fs* in, const char* lhs, const char* rhs, go_fs* outfs, const char* out_name,
                          uint64_t stats[3], char* err, size_t errcap)
{
    uint64_t K = 0, K1 = 0, c0 = 0, c1 = 0;
    merge_item a = {0}, b = {0};
    if (merge_load(in, lhs, 0, &K, &c0, &a, err, errcap)) return -1;
    if (merge_load(in, rhs, 0, &K1, &c1, &b, err, errcap)) { merge_item_free(&a); return -1; }
    if (a.n == 0 || b.n == 0 || K != K1)
    {
        snprintf(err, errcap, "nonsense");
        merge_item_free(&a); merge_item_free(&b);
        return -1;
    }
    go_key* keys = (go_key*)malloc((a.n + b.n) * sizeof(go_key));
    uint8_t* side = (uint8_t*)malloc(a.n + b.n);
    uint64_t l = 0, r = 0, n = 0, c = 0;
    while (l < a.n && r < b.n)
    {
        u128 le = k2u(a.k[l]), re = k2u(b.k[r]);
        if (le < re) { keys[n] = a.k[l]; side[n++] = 1; ++l; continue; }
        if (le > re) { keys[n] = b.k[r]; side[n++] = 2; ++r; continue; }
        keys[n] = a.k[l]; side[n++] = 3; ++c; ++l; ++r;
    }
    while (l < a.n) { keys[n] = a.k[l]; side[n++] = 1; ++l; }
    while (r < b.n) { keys[n] = b.k[r]; side[n++] = 2; ++r; }
    int rc = go_write_kmer_set(outfs, out_name, (unsigned)K, keys, n, n);
    if (!rc)
    {
        char name[4096];
        wbv_builder lb, rb;
        snprintf(name, sizeof name, "%s.lhs-bits", out_name);
        wbv_init(&lb, outfs, name);
        snprintf(name, sizeof name, "%s.rhs-bits", out_name);
        wbv_init(&rb, outfs, name);
        for (uint64_t i = 0; i < n; ++i)
        {
            wbv_push_backx(&lb, side[i] & 1);
            wbv_push_backx(&rb, (side[i] >> 1) & 1);
        }
        wbv_end(&lb); wbv_end(&rb);
    }
    else snprintf(err, errcap, "write error");
    if (stats) { stats[0] = l; stats[1] = r; stats[2] = c; }
    free(keys); free(side);
    merge_item_free(&a); merge_item_free(&b);
    return rc ? -1 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* dump-kmer-set / dump-graph / restore-graph                                            */
/* ------------------------------------------------------------------------------------ */

typedef struct { char* p; size_t n, cap; } text_buf;

static void tb_put(text_buf* t, const char* s, size_t n)
{
    if (t->n + n + 1 > t->cap)
    {
        t->cap = (t->n + n + 1) * 2 + 64;
        t->p = (char*)realloc(t->p, t->cap);
    }
    memcpy(t->p + t->n, s, n);
    t->n += n;
    t->p[t->n] = 0;
}

/* kmerToString (RankSelect.hh:299-308) */
static void tb_kmer(text_buf* t, go_key x, unsigned len)
{
    char s[80];
    u128 v = k2u(x);
    for (unsigned i = 0; i < len; ++i) s[i] = "ACGT"[(unsigned)(v >> (2 * (len - 1 - i))) & 3];
    tb_put(t, s, len);
}

/* GossCmdDumpKmerSet.cc:31-55 (kind 0) / GossCmdDumpGraph.cc:31-61 (kind 1).  The caller frees
 * *text. */
int go_dump(const go_fs* fs, const char* name, int kind, char** text, size_t* len, char* err, size_t errcap)
{
    uint64_t K = 0, cnt = 0;
    merge_item m = {0};
    if (merge_load(fs, name, kind, &K, &cnt, &m, err, errcap)) return -1;
    text_buf t = {0};
    char line[128];
    if (kind == 0)
    {
        int n = snprintf(line, sizeof line, "#%llu\n%llu\t%llu\n", 2011101701ULL, (unsigned long long)K, (unsigned long long)cnt);
        tb_put(&t, line, (size_t)n);
        for (uint64_t i = 0; i < m.n; ++i) { tb_kmer(&t, m.k[i], (unsigned)K); tb_put(&t, "\n", 1); }
    }
    else
    {
        uint64_t flags = 0, hk = 0;
        go_graph_header(fs, name, &hk, &flags);
        int n = snprintf(line, sizeof line, "#%llu\n%llu\t%llu\t%llu\n", 2011101014ULL, (unsigned long long)K,
                         (unsigned long long)m.n, (unsigned long long)(flags & 1));
        tb_put(&t, line, (size_t)n);
        for (uint64_t i = 0; i < m.n; ++i)
        {
            tb_kmer(&t, m.k[i], (unsigned)K + 1);
            n = snprintf(line, sizeof line, "\t%llu\n", (unsigned long long)m.c[i]);
            tb_put(&t, line, (size_t)n);
        }
    }
    merge_item_free(&m);
    *text = t.p; *len = t.n;
    return 0;
}

/* GossCmdRestoreGraph.cc:72-135: the first line is skipped, then "K n flags", then
 * "<edge> <count>" pairs while the stream stays good; Graph::Builder(k, out, fac, n, asymmetric). */
int go_restore_graph(const char* text, size_t len, go_fs* outfs, const char* out_name, char* err, size_t errcap)
{
    size_t p = 0;
    while (p < len && text[p] != '\n') ++p;
    if (p >= len) { snprintf(err, errcap, "unexpected end of file"); return -1; }
    ++p;
    uint64_t hdr[3];
    for (int f = 0; f < 3; ++f)
    {
        while (p < len && isspace((unsigned char)text[p])) ++p;
        size_t b = p; uint64_t v = 0;
        while (p < len && isdigit((unsigned char)text[p])) { v = v * 10 + (uint64_t)(text[p] - '0'); ++p; }
        if (p == b) { snprintf(err, errcap, "unexpected end of file"); return -1; }
        hdr[f] = v;
    }
    if (p >= len) { snprintf(err, errcap, "unexpected end of file"); return -1; }
    const uint64_t k = hdr[0], n = hdr[1], flags = hdr[2];
    size_t cap = 1024, m = 0;
    go_key* keys = (go_key*)malloc(cap * sizeof(go_key));
    uint64_t* counts = (uint64_t*)malloc(cap * 8);
    int rc = 0;
    for (;;)
    {
        while (p < len && isspace((unsigned char)text[p])) ++p;
        size_t b = p;
        while (p < len && !isspace((unsigned char)text[p])) ++p;
        if (p == b) break;
        size_t xl = p - b;
        while (p < len && isspace((unsigned char)text[p])) ++p;
        size_t cb = p; uint64_t c = 0;
        while (p < len && isdigit((unsigned char)text[p]) && c <= 0xFFFFFFFFULL) { c = c * 10 + (uint64_t)(text[p] - '0'); ++p; }
        if (p == cb || c > 0xFFFFFFFFULL || p >= len) break;
        if (xl != k + 1) { snprintf(err, errcap, "sequence %.*s has wrong length", (int)xl, text + b); rc = -1; break; }
        u128 x = 0;
        for (size_t i = 0; i < xl && !rc; ++i)
        {
            switch (text[b + i])
            {
                case 'A': case 'a': x = (x << 2) | 0; break;
                case 'C': case 'c': x = (x << 2) | 1; break;
                case 'G': case 'g': x = (x << 2) | 2; break;
                case 'T': case 't': x = (x << 2) | 3; break;
                default: snprintf(err, errcap, "invalid sequence %.*s", (int)xl, text + b); rc = -1;
            }
        }
        if (rc) break;
        if (m == cap) { cap *= 2; keys = (go_key*)realloc(keys, cap * sizeof(go_key)); counts = (uint64_t*)realloc(counts, cap * 8); }
        keys[m] = u2k(x); counts[m] = c; ++m;
    }
    if (!rc)
    {
        rc = go_write_graph(outfs, out_name, (unsigned)k, keys, counts, m, n);
        if (rc) snprintf(err, errcap, "write error");
        else if (flags & 1)
        {
            char name[4096];
            snprintf(name, sizeof name, "%s.header", out_name);
            int fi = go_fs_find(outfs, name);
            if (fi >= 0 && outfs->files[fi].size >= 24) { uint64_t f = 1; memcpy(outfs->files[fi].data + 16, &f, 8); }
        }
    }
    free(keys); free(counts);
    return rc ? -1 : 0;
}
