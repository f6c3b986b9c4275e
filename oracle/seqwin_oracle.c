/*
 * seqwin_oracle.c -- CPU restatement of Seqwin's cpp/ minimizer-index path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path (seqwin_amd/) never does.
 *
 * It restates, in plain single-threaded C, what the reference computes on the path
 *   seqwin::build  (+ btllib::minimize_sequence / NtHash)  ->  seqwin::get_penalty  ->  seqwin::filter_kmers
 * so that the HIP kernels can be checked bit-for-bit on a box where /root/reference does
 * not exist.  Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 *   - the reference's own golden vector tests/smoke/fixtures/expected/graph.npz, and
 *   - golden vectors produced by the compiled reference itself (oracle/_ref, see
 *     oracle/Makefile and tests/golden/make_golden.py), and, when oracle/_ref is present,
 *   - the compiled reference live on seeded fuzz inputs (tests/test_oracle_vs_ref.py).
 *
 * Deliberate divergences from the reference (both are rejected with an error here and in the
 * product instead of being reproduced):
 *   - k < 3: the reference computes unsigned k-3 (cpp/vendor/btllib/nthash_kmer.hpp:26) and crashes.
 *   - raw bytes 0x01 0x03 0x04 0x05 0x07 inside a sequence: SEED_TAB
 *     (cpp/vendor/btllib/hashing_internals.hpp:136-137) treats them as T,G,A,A,C but CONVERT_TAB
 *     (:354) maps them to 255, so the reference hashes them through an out-of-pattern table index.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include <zlib.h>

#define SO_OK 0
#define SO_ERR_RUNTIME 1   /* std::runtime_error / logic_error in the reference  -> RuntimeError */
#define SO_ERR_VALUE 2     /* std::invalid_argument in the reference             -> ValueError   */

static __thread char g_err[1024];

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *so_last_error(void) { return g_err; }

/* ------------------------------------------------------------------------------------------
 * ntHash (cpp/vendor/btllib/hashing_internals.hpp, nthash_kmer.hpp)
 * ---------------------------------------------------------------------------------------- */

/* hashing_internals.hpp:128-131 */
static const uint64_t SEED[4] = {
    0x3c8bfbb395c60474ULL, /* A */
    0x3193c18562a02b4cULL, /* C */
    0x20323ed082572324ULL, /* G */
    0x295549f54be24456ULL, /* T / U */
};
/* hashing_internals.hpp:76-79 */
#define MULTISHIFT 27
#define MULTISEED 0x90b45d39fb6da1faULL

/* Base code: A0 C1 G2 T3 (CONVERT_TAB hashing_internals.hpp:354-387); -1 = invalid
 * (SEED_TAB == SEED_N, :136-169); -2 = one of the five control bytes we refuse. */
static int base_code(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    case 1: case 3: case 4: case 5: case 7: return -2;
    default: return -1;
    }
}

/* hashing_internals.hpp:29-35: rotate the low 33 bits and the high 31 bits left by one, separately. */
static uint64_t srol1(uint64_t x)
{
    uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}

/* hashing_internals.hpp:69-74 */
static uint64_t sror1(uint64_t x)
{
    uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
    return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}

/* srol applied d times (the reference's srol(x,d) :45-52 and srol_table :347-352 are closed
 * forms of this; the periods are 33 and 31 so d can be reduced mod 1023). */
static uint64_t srol_n(uint64_t x, unsigned d)
{
    d %= 1023u;
    while (d--) x = srol1(x);
    return x;
}

/* nthash_kmer.hpp:22-54 (base_forward_hash): F = XOR_i srol^{k-1-i}(S[s_i]).  The
 * TETRAMER/TRIMER/DIMER tables there are precomputed partial sums of exactly this. */
static uint64_t forward_hash_def(const int8_t *code, unsigned k)
{
    uint64_t h = 0;
    for (unsigned i = 0; i < k; ++i) h = srol1(h) ^ SEED[code[i]];
    return h;
}

/* nthash_kmer.hpp:104-133 (base_reverse_hash): R = XOR_i srol^{i}(S[3 - s_i]). */
static uint64_t reverse_hash_def(const int8_t *code, unsigned k)
{
    uint64_t h = 0;
    for (unsigned i = k; i-- > 0;) h = srol1(h) ^ SEED[3 - code[i]];
    return h;
}

/* hashing_internals.hpp:89-103 (extend_hashes with h = 2): hashes()[0] and hashes()[1]. */
static void extend2(uint64_t fwd, uint64_t rev, unsigned k, uint64_t *h0, uint64_t *h1)
{
    uint64_t b = fwd + rev;                       /* canonical(), :12-17 */
    uint64_t t = b * (1ULL ^ ((uint64_t)k * MULTISEED));
    t ^= t >> MULTISHIFT;
    *h0 = b;
    *h1 = t;
}

typedef struct {
    uint64_t min_hash; /* hashes()[0] */
    uint64_t out_hash; /* hashes()[1] */
    uint64_t pos;
} so_min_t;            /* btllib::Minimizer, minimizer.hpp:10-16 (the `forward` flag is unused downstream) */

typedef struct {
    so_min_t *v;
    size_t n, cap;
} minvec_t;

static int minvec_push(minvec_t *mv, so_min_t m)
{
    if (mv->n == mv->cap) {
        size_t nc = mv->cap ? mv->cap * 2 : 256;
        so_min_t *nv = (so_min_t *)realloc(mv->v, nc * sizeof *nv);
        if (!nv) return fail(SO_ERR_RUNTIME, "out of memory");
        mv->v = nv;
        mv->cap = nc;
    }
    mv->v[mv->n++] = m;
    return SO_OK;
}

/*
 * NtHash iterator state, following nthash_kmer.hpp:315-333 (roll) and :491-511 (init):
 * enumerates, in increasing position, every k-mer whose k characters are all valid.
 */
typedef struct {
    const int8_t *code; /* per-base code, -1 invalid */
    size_t len;
    unsigned k;
    size_t pos;
    int initialized;
    uint64_t fwd, rev;
    uint64_t out_rot[4]; /* srol^k(S[c]) = srol_table(c, k) */
} nthash_t;

static int nthash_init(nthash_t *h)
{
    /* nthash_kmer.hpp:491-505: skip forward until a window of k valid characters is found.
     * (The reference walks the window back-to-front and jumps past the last invalid character;
     * the net effect is "first position >= pos whose k-mer is fully valid".) */
    const size_t len = h->len;
    const unsigned k = h->k;
    size_t pos = h->pos;
    for (;;) {
        if (pos + k > len) return 0;
        int bad = -1;
        for (unsigned i = k; i-- > 0;) {
            if (h->code[pos + i] < 0) { bad = (int)i; break; }
        }
        if (bad < 0) break;
        pos += (size_t)bad + 1;
    }
    h->pos = pos;
    h->fwd = forward_hash_def(h->code + pos, k);
    h->rev = reverse_hash_def(h->code + pos, k);
    h->initialized = 1;
    return 1;
}

static int nthash_roll(nthash_t *h)
{
    if (!h->initialized) return nthash_init(h);                  /* :317-319 */
    if (h->pos >= h->len - h->k) return 0;                        /* :320-322 */
    const int in = h->code[h->pos + h->k];
    if (in < 0) {                                                 /* :323-327 */
        h->pos += h->k;
        return nthash_init(h);
    }
    const int out = h->code[h->pos];
    /* next_forward_hash :65-75 */
    h->fwd = srol1(h->fwd) ^ SEED[in] ^ h->out_rot[out];
    /* next_reverse_hash :145-155 */
    h->rev = sror1(h->rev ^ h->out_rot[3 - in] ^ SEED[3 - out]);
    ++h->pos;
    return 1;
}

/* Translate a raw sequence to codes; returns SO_ERR_VALUE on one of the refused control bytes. */
static int encode_seq(const char *seq, size_t len, int8_t **out)
{
    int8_t *code = (int8_t *)malloc(len + 1);
    if (!code) return fail(SO_ERR_RUNTIME, "out of memory");
    for (size_t i = 0; i < len; ++i) {
        int c = base_code((unsigned char)seq[i]);
        if (c == -2) {
            free(code);
            return fail(SO_ERR_VALUE, "unsupported control byte 0x%02x in sequence at offset %zu",
                        (unsigned char)seq[i], i);
        }
        code[i] = (int8_t)c;
    }
    code[len] = -1; /* std::string's terminating NUL is SEED_N */
    *out = code;
    return SO_OK;
}

static int check_kw(uint64_t k, uint64_t w)
{
    if (k < 3) return fail(SO_ERR_VALUE, "kmerlen must be >= 3 (got %llu)", (unsigned long long)k);
    if (k > 65535) return fail(SO_ERR_VALUE, "kmerlen must be <= 65535 (got %llu)", (unsigned long long)k);
    if (w < 1) return fail(SO_ERR_VALUE, "windowsize must be >= 1 (got %llu)", (unsigned long long)w);
    return SO_OK;
}

/*
 * btllib::minimize_sequence, minimizer.cpp:53-90, with calc_minimizer :14-49 inlined.
 * Kept structurally identical to the reference (ring of w+1 hashed k-mers, rescan when the
 * current minimum leaves the window, `<=` so the rightmost minimum wins, emit when the position
 * advances and the hash is not UINT64_MAX) so that it is an independent check on the
 * reformulation the HIP kernel uses.
 */
static int minimize_codes(const int8_t *code, size_t len, unsigned k, size_t w, minvec_t *out)
{
    if ((size_t)k > len || w > len - k + 1) return SO_OK;        /* :56-58 */

    const size_t ring_n = w + 1;                                   /* :63 */
    so_min_t *ring = (so_min_t *)calloc(ring_n, sizeof *ring);
    if (!ring) return fail(SO_ERR_RUNTIME, "out of memory");

    nthash_t nh;
    memset(&nh, 0, sizeof nh);
    nh.code = code;
    nh.len = len;
    nh.k = k;
    for (int c = 0; c < 4; ++c) nh.out_rot[c] = srol_n(SEED[c], k);

    int64_t min_pos_prev = -1;
    const so_min_t *cur = NULL;
    int rc = SO_OK;

    for (size_t idx = 0; nthash_roll(&nh); ++idx) {               /* :70 */
        so_min_t *hk = &ring[idx % ring_n];
        extend2(nh.fwd, nh.rev, k, &hk->min_hash, &hk->out_hash);
        hk->pos = nh.pos;
        if (idx + 1 < w) continue;                                 /* :77 */

        /* calc_minimizer :23-41 */
        const size_t left = idx + 1 - w, right = idx + 1;
        const so_min_t *min_left = &ring[left % ring_n];
        const so_min_t *min_right = &ring[(right - 1) % ring_n];
        if (cur == NULL || cur->pos < min_left->pos) {
            cur = min_left;
            for (size_t i = left; i < right; ++i) {
                const so_min_t *mi = &ring[i % ring_n];
                if (mi->min_hash <= cur->min_hash) cur = mi;
            }
        } else if (min_right->min_hash <= cur->min_hash) {
            cur = min_right;
        }
        /* :44-48 */
        if ((int64_t)cur->pos > min_pos_prev && cur->min_hash != UINT64_MAX) {
            min_pos_prev = (int64_t)cur->pos;
            if ((rc = minvec_push(out, *cur)) != SO_OK) break;
        }
    }
    free(ring);
    return rc;
}

/* C-ABI: all (min_hash, out_hash, pos) of the valid k-mers of one sequence, in order. */
int so_nthash(const char *seq, size_t len, uint64_t k, uint64_t *min_hash, uint64_t *out_hash,
              uint64_t *pos, size_t cap, size_t *n_out)
{
    int rc = check_kw(k, 1);
    if (rc) return rc;
    int8_t *code;
    if ((rc = encode_seq(seq, len, &code)) != SO_OK) return rc;
    size_t n = 0;
    if (k <= len) {
        nthash_t nh;
        memset(&nh, 0, sizeof nh);
        nh.code = code;
        nh.len = len;
        nh.k = (unsigned)k;
        for (int c = 0; c < 4; ++c) nh.out_rot[c] = srol_n(SEED[c], (unsigned)k);
        while (nthash_roll(&nh)) {
            if (n < cap) {
                extend2(nh.fwd, nh.rev, (unsigned)k, &min_hash[n], &out_hash[n]);
                pos[n] = nh.pos;
            }
            ++n;
        }
    }
    free(code);
    *n_out = n;
    return SO_OK;
}

/* C-ABI: btllib::minimize_sequence on one sequence. Returns the number found in *n_out
 * (which may exceed cap; only the first cap are written). */
int so_minimize(const char *seq, size_t len, uint64_t k, uint64_t w, uint64_t *min_hash,
                uint64_t *out_hash, uint64_t *pos, size_t cap, size_t *n_out)
{
    int rc = check_kw(k, w);
    if (rc) return rc;
    int8_t *code;
    if ((rc = encode_seq(seq, len, &code)) != SO_OK) return rc;
    minvec_t mv = {0};
    rc = minimize_codes(code, len, (unsigned)k, (size_t)w, &mv);
    free(code);
    if (rc == SO_OK) {
        for (size_t i = 0; i < mv.n && i < cap; ++i) {
            min_hash[i] = mv.v[i].min_hash;
            out_hash[i] = mv.v[i].out_hash;
            pos[i] = mv.v[i].pos;
        }
        *n_out = mv.n;
    }
    free(mv.v);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * FASTA reader (cpp/src/utils/fasta_reader.cpp)
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    char *id;
    char *seq;
    size_t len, cap;
} record_t;

typedef struct {
    record_t *v;
    size_t n, cap;
} recvec_t;

static void recvec_free(recvec_t *rv)
{
    for (size_t i = 0; i < rv->n; ++i) {
        free(rv->v[i].id);
        free(rv->v[i].seq);
    }
    free(rv->v);
    rv->v = NULL;
    rv->n = rv->cap = 0;
}

static int is_ws(unsigned char c) /* " \t\n\r\f\v" == isspace() in the C locale, :28,37,84 */
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v';
}

static int ends_with(const char *s, const char *suf)
{
    size_t a = strlen(s), b = strlen(suf);
    return a >= b && memcmp(s + a - b, suf, b) == 0;
}

/* Slurp a whole file (plain via stdio, .gz via zlib gzread: fasta_reader.cpp:97-107,109-160). */
static int slurp(const char *path, char **buf_out, size_t *len_out)
{
    size_t cap = 1 << 16, len = 0;
    char *buf = (char *)malloc(cap);
    if (!buf) return fail(SO_ERR_RUNTIME, "out of memory");
    if (ends_with(path, ".gz")) {                                 /* :209 */
        gzFile gz = gzopen(path, "rb");
        if (!gz) {
            free(buf);
            return fail(SO_ERR_RUNTIME, "Unable to open gzip FASTA: %s", path); /* :112-114 */
        }
        for (;;) {
            if (cap - len < (1 << 16)) {
                cap *= 2;
                char *nb = (char *)realloc(buf, cap);
                if (!nb) { free(buf); gzclose(gz); return fail(SO_ERR_RUNTIME, "out of memory"); }
                buf = nb;
            }
            int got = gzread(gz, buf + len, 1 << 16);
            if (got < 0) {
                int errnum = 0;
                const char *e = gzerror(gz, &errnum);
                fail(SO_ERR_RUNTIME, "gzip read error: %s", e ? e : "unknown"); /* :146-151 */
                free(buf);
                gzclose(gz);
                return SO_ERR_RUNTIME;
            }
            if (got == 0) break;
            len += (size_t)got;
        }
        gzclose(gz);
    } else {
        FILE *f = fopen(path, "rb");
        if (!f) {
            free(buf);
            return fail(SO_ERR_RUNTIME, "Unable to open FASTA: %s", path);      /* :100-102 */
        }
        for (;;) {
            if (cap - len < (1 << 16)) {
                cap *= 2;
                char *nb = (char *)realloc(buf, cap);
                if (!nb) { free(buf); fclose(f); return fail(SO_ERR_RUNTIME, "out of memory"); }
                buf = nb;
            }
            size_t got = fread(buf + len, 1, 1 << 16, f);
            len += got;
            if (got == 0) break;
        }
        fclose(f);
    }
    *buf_out = buf;
    *len_out = len;
    return SO_OK;
}

/* read_fasta_core, fasta_reader.cpp:41-95 */
static int parse_fasta(const char *buf, size_t len, recvec_t *out)
{
    record_t *cur = NULL;
    size_t p = 0;
    while (p < len) {
        /* one std::getline: up to '\n' or end of data; a trailing empty piece is not a line */
        size_t e = p;
        while (e < len && buf[e] != '\n') ++e;
        size_t ls = p, le = e;
        p = (e < len) ? e + 1 : e;
        if (le > ls && buf[le - 1] == '\r') --le;                 /* :51-53 */

        int all_ws = 1;                                             /* :55-57 */
        for (size_t i = ls; i < le; ++i)
            if (!is_ws((unsigned char)buf[i])) { all_ws = 0; break; }
        if (all_ws) continue;

        if (buf[ls] == '>') {                                      /* :58-67 */
            if (out->n == out->cap) {
                size_t nc = out->cap ? out->cap * 2 : 16;
                record_t *nv = (record_t *)realloc(out->v, nc * sizeof *nv);
                if (!nv) return fail(SO_ERR_RUNTIME, "out of memory");
                out->v = nv;
                out->cap = nc;
            }
            cur = &out->v[out->n++];
            memset(cur, 0, sizeof *cur);
            size_t ie = ls + 1;                                     /* extract_id :26-33 */
            while (ie < le && !is_ws((unsigned char)buf[ie])) ++ie;
            cur->id = (char *)malloc(ie - ls);
            if (!cur->id) return fail(SO_ERR_RUNTIME, "out of memory");
            memcpy(cur->id, buf + ls + 1, ie - ls - 1);
            cur->id[ie - ls - 1] = 0;
            continue;
        }
        if (!cur)                                                   /* :69-71 */
            return fail(SO_ERR_RUNTIME, "Invalid FASTA: sequence encountered before header");

        if (cur->len + (le - ls) + 1 > cur->cap) {                  /* :79-88 */
            size_t nc = cur->cap * 2;
            if (nc < cur->len + (le - ls) + 1) nc = cur->len + (le - ls) + 1;
            char *ns = (char *)realloc(cur->seq, nc);
            if (!ns) return fail(SO_ERR_RUNTIME, "out of memory");
            cur->seq = ns;
            cur->cap = nc;
        }
        for (size_t i = ls; i < le; ++i)
            if (!is_ws((unsigned char)buf[i])) cur->seq[cur->len++] = buf[i];
    }
    return SO_OK;
}

static int read_fasta(const char *path, recvec_t *out)
{
    char *buf;
    size_t len;
    int rc = slurp(path, &buf, &len);
    if (rc) return rc;
    rc = parse_fasta(buf, len, out);
    free(buf);
    return rc;
}

/* C-ABI: parse one FASTA file; returns NUL-terminated ids and sequences back to back
 * (id0 \0 seq0 \0 id1 \0 seq1 \0 ...) in a malloc'd blob the caller frees with so_free. */
int so_read_fasta(const char *path, char **blob_out, size_t *blob_len, size_t *n_records)
{
    recvec_t recs = {0};
    int rc = read_fasta(path, &recs);
    if (rc) { recvec_free(&recs); return rc; }
    size_t total = 0;
    for (size_t i = 0; i < recs.n; ++i) total += strlen(recs.v[i].id) + 1 + recs.v[i].len + 1;
    char *blob = (char *)malloc(total ? total : 1);
    if (!blob) { recvec_free(&recs); return fail(SO_ERR_RUNTIME, "out of memory"); }
    size_t o = 0;
    for (size_t i = 0; i < recs.n; ++i) {
        size_t il = strlen(recs.v[i].id);
        memcpy(blob + o, recs.v[i].id, il + 1); o += il + 1;
        if (recs.v[i].len) memcpy(blob + o, recs.v[i].seq, recs.v[i].len);
        o += recs.v[i].len;
        blob[o++] = 0;
    }
    *blob_out = blob;
    *blob_len = total;
    *n_records = recs.n;
    recvec_free(&recs);
    return SO_OK;
}

void so_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------
 * Graph build (cpp/src/seqwin/build.cpp + build_internals.cpp), restated as sort + run-length.
 * ---------------------------------------------------------------------------------------- */

typedef struct { uint32_t pos, record_idx; } so_kmer_t;                       /* graph.hpp:15-20 */
typedef struct { uint64_t hash, start, stop; uint32_t n_tar, n_neg; double penalty; } so_node_t; /* :28-41 */
typedef struct { uint64_t first, second, weight; } so_edge_t;                 /* :46-53 */

typedef struct { uint64_t hash; so_kmer_t km; } occ_t;      /* RawKmer build.cpp:61-64 */
typedef struct { uint64_t first, second; uint32_t assembly; } adj_t;

typedef struct so_graph {
    so_kmer_t *kmers; size_t n_kmers;
    so_node_t *nodes; size_t n_nodes;
    so_edge_t *edges; size_t n_edges;
    uint32_t *record_offsets; size_t n_assemblies;
    char *ids_blob; size_t ids_bytes;      /* NUL-terminated ids, in record order */
    uint64_t total_bp;
} so_graph;

static int cmp_adj(const void *a, const void *b)
{
    const adj_t *x = (const adj_t *)a, *y = (const adj_t *)b;
    if (x->first != y->first) return x->first < y->first ? -1 : 1;
    if (x->second != y->second) return x->second < y->second ? -1 : 1;
    if (x->assembly != y->assembly) return x->assembly < y->assembly ? -1 : 1;
    return 0;
}

/* Stable LSD byte-radix sort of occurrences by hash: the same ordering contract as
 * build_internals.cpp:76-144 (stable, key = hash) applied to insertion-ordered occurrences
 * (build.cpp:153-168, 232-240), which yields (hash, record_idx, pos) order. */
static int sort_occ(occ_t *a, size_t n)
{
    if (n < 2) return SO_OK;
    occ_t *b = (occ_t *)malloc(n * sizeof *b);
    if (!b) return fail(SO_ERR_RUNTIME, "out of memory");
    occ_t *src = a, *dst = b;
    for (int shift = 0; shift < 64; shift += 8) {
        size_t cnt[257] = {0};
        for (size_t i = 0; i < n; ++i) ++cnt[((src[i].hash >> shift) & 0xff) + 1];
        for (int i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; ++i) dst[cnt[(src[i].hash >> shift) & 0xff]++] = src[i];
        occ_t *t = src; src = dst; dst = t;
    }
    /* 8 passes: result is back in `a` */
    free(b);
    return SO_OK;
}

void so_graph_free(so_graph *g)
{
    if (!g) return;
    free(g->kmers);
    free(g->nodes);
    free(g->edges);
    free(g->record_offsets);
    free(g->ids_blob);
    free(g);
}

/* seqwin::build, build.cpp:330-394.  Thread partitioning (:350-367) and the thread merge
 * (build_internals.cpp:295-392) do not change the result (reference test_graph.py:67-127), so the
 * oracle is the n_cpu = 1 semantics; low_memory (:379-390) is result-identical as well
 * (test_graph.py:222-245). */
int so_build(const char *const *paths, size_t n_paths, uint64_t k, uint64_t w, so_graph **out)
{
    int rc = check_kw(k, w);
    if (rc) return rc;
    if (n_paths > UINT32_MAX)
        return fail(SO_ERR_RUNTIME, "Number of input assemblies exceeds uint32 range"); /* :337-339 */

    so_graph *g = (so_graph *)calloc(1, sizeof *g);
    occ_t *occ = NULL; size_t n_occ = 0, cap_occ = 0;
    adj_t *adj = NULL; size_t n_adj = 0, cap_adj = 0;
    size_t ids_cap = 0;
    minvec_t mv = {0};
    if (!g) return fail(SO_ERR_RUNTIME, "out of memory");
    g->record_offsets = (uint32_t *)calloc(n_paths + 1, sizeof(uint32_t));
    g->n_assemblies = n_paths;
    if (!g->record_offsets) { rc = fail(SO_ERR_RUNTIME, "out of memory"); goto done; }

    uint32_t record_idx = 0;
    for (size_t a = 0; a < n_paths; ++a) {
        recvec_t recs = {0};
        if ((rc = read_fasta(paths[a], &recs)) != SO_OK) { recvec_free(&recs); goto done; } /* :131 */
        if (recs.n > (size_t)(UINT32_MAX - record_idx)) {                                   /* :136-140 */
            recvec_free(&recs);
            rc = fail(SO_ERR_RUNTIME, "Total number of FASTA records exceeds uint32 range");
            goto done;
        }
        for (size_t r = 0; r < recs.n; ++r) {
            record_t *rec = &recs.v[r];
            if (rec->len > UINT32_MAX) {                                                    /* :143-147 */
                rc = fail(SO_ERR_RUNTIME, "Sequence length exceeds uint32 range for record %s in assembly %s",
                          rec->id, paths[a]);
                recvec_free(&recs);
                goto done;
            }
            size_t idl = strlen(rec->id) + 1;                                               /* :148 */
            if (g->ids_bytes + idl > ids_cap) {
                ids_cap = (g->ids_bytes + idl) * 2;
                char *nb = (char *)realloc(g->ids_blob, ids_cap);
                if (!nb) { recvec_free(&recs); rc = fail(SO_ERR_RUNTIME, "out of memory"); goto done; }
                g->ids_blob = nb;
            }
            memcpy(g->ids_blob + g->ids_bytes, rec->id, idl);
            g->ids_bytes += idl;
            g->total_bp += rec->len;

            int8_t *code;
            if ((rc = encode_seq(rec->seq ? rec->seq : "", rec->len, &code)) != SO_OK) { recvec_free(&recs); goto done; }
            mv.n = 0;
            rc = minimize_codes(code, rec->len, (unsigned)k, (size_t)w, &mv);              /* :151 */
            free(code);
            if (rc) { recvec_free(&recs); goto done; }

            if (n_occ + mv.n > cap_occ) {
                cap_occ = (n_occ + mv.n) * 2 + 1024;
                occ_t *no = (occ_t *)realloc(occ, cap_occ * sizeof *no);
                if (!no) { recvec_free(&recs); rc = fail(SO_ERR_RUNTIME, "out of memory"); goto done; }
                occ = no;
            }
            if (n_adj + mv.n > cap_adj) {
                cap_adj = (n_adj + mv.n) * 2 + 1024;
                adj_t *na = (adj_t *)realloc(adj, cap_adj * sizeof *na);
                if (!na) { recvec_free(&recs); rc = fail(SO_ERR_RUNTIME, "out of memory"); goto done; }
                adj = na;
            }
            for (size_t i = 0; i < mv.n; ++i) {                                             /* :153-168 */
                occ[n_occ].hash = mv.v[i].out_hash;
                occ[n_occ].km.pos = (uint32_t)mv.v[i].pos;
                occ[n_occ].km.record_idx = record_idx;
                ++n_occ;
            }
            for (size_t i = 0; i + 1 < mv.n; ++i) {                                         /* :177-189 */
                uint64_t u = mv.v[i].out_hash, v = mv.v[i + 1].out_hash;
                if (v < u) { uint64_t t = u; u = v; v = t; }
                adj[n_adj].first = u;
                adj[n_adj].second = v;
                adj[n_adj].assembly = (uint32_t)a;
                ++n_adj;
            }
            ++record_idx;                                                                   /* :169 */
        }
        g->record_offsets[a + 1] = record_idx;                                              /* :191 */
        recvec_free(&recs);
    }

    /* nodes + kmers: build.cpp:196-253 and build_internals.cpp:159-251 */
    if ((rc = sort_occ(occ, n_occ)) != SO_OK) goto done;
    g->n_kmers = n_occ;
    g->kmers = (so_kmer_t *)malloc((n_occ ? n_occ : 1) * sizeof(so_kmer_t));
    size_t n_nodes = 0;
    for (size_t i = 0; i < n_occ; ++i)
        if (i == 0 || occ[i].hash != occ[i - 1].hash) ++n_nodes;
    g->n_nodes = n_nodes;
    g->nodes = (so_node_t *)calloc(n_nodes ? n_nodes : 1, sizeof(so_node_t));
    if (!g->kmers || !g->nodes) { rc = fail(SO_ERR_RUNTIME, "out of memory"); goto done; }
    size_t ni = 0;
    for (size_t i = 0; i < n_occ; ++i) {
        g->kmers[i] = occ[i].km;
        if (i == 0 || occ[i].hash != occ[i - 1].hash) {
            if (ni) g->nodes[ni - 1].stop = i;
            g->nodes[ni].hash = occ[i].hash;
            g->nodes[ni].start = i;
            ++ni;
        }
    }
    if (ni) g->nodes[ni - 1].stop = n_occ;

    /* edges: weight = number of assemblies containing the unordered pair (build.cpp:177-189),
     * sorted by (first, second) (build_internals.cpp:253-291) */
    if (n_adj) qsort(adj, n_adj, sizeof *adj, cmp_adj);   /* (qsort(NULL, 0, ...) is undefined: found by the UBSan build) */
    size_t n_edges = 0;
    for (size_t i = 0; i < n_adj; ++i)
        if (i == 0 || adj[i].first != adj[i - 1].first || adj[i].second != adj[i - 1].second) ++n_edges;
    g->n_edges = n_edges;
    g->edges = (so_edge_t *)calloc(n_edges ? n_edges : 1, sizeof(so_edge_t));
    if (!g->edges) { rc = fail(SO_ERR_RUNTIME, "out of memory"); goto done; }
    size_t ei = 0;
    for (size_t i = 0; i < n_adj; ++i) {
        int new_pair = (i == 0 || adj[i].first != adj[i - 1].first || adj[i].second != adj[i - 1].second);
        if (new_pair) {
            g->edges[ei].first = adj[i].first;
            g->edges[ei].second = adj[i].second;
            g->edges[ei].weight = 1;
            ++ei;
        } else if (adj[i].assembly != adj[i - 1].assembly) {
            ++g->edges[ei - 1].weight;
        }
    }

done:
    free(occ);
    free(adj);
    free(mv.v);
    if (rc != SO_OK) {
        so_graph_free(g);
        return rc;
    }
    *out = g;
    return SO_OK;
}

void so_graph_sizes(const so_graph *g, uint64_t *n_kmers, uint64_t *n_nodes, uint64_t *n_edges,
                    uint64_t *n_assemblies, uint64_t *ids_bytes, uint64_t *total_bp)
{
    *n_kmers = g->n_kmers;
    *n_nodes = g->n_nodes;
    *n_edges = g->n_edges;
    *n_assemblies = g->n_assemblies;
    *ids_bytes = g->ids_bytes;
    *total_bp = g->total_bp;
}

void so_graph_export(const so_graph *g, void *kmers, void *nodes, void *edges, uint32_t *record_offsets,
                     char *ids_blob)
{
    if (g->n_kmers) memcpy(kmers, g->kmers, g->n_kmers * sizeof(so_kmer_t));
    if (g->n_nodes) memcpy(nodes, g->nodes, g->n_nodes * sizeof(so_node_t));
    if (g->n_edges) memcpy(edges, g->edges, g->n_edges * sizeof(so_edge_t));
    memcpy(record_offsets, g->record_offsets, (g->n_assemblies + 1) * sizeof(uint32_t));
    if (g->ids_bytes) memcpy(ids_blob, g->ids_blob, g->ids_bytes);
}

/* ------------------------------------------------------------------------------------------
 * seqwin::get_penalty, cpp/src/seqwin/filter.cpp:15-137
 * ---------------------------------------------------------------------------------------- */
int so_get_penalty(const so_kmer_t *kmers, uint64_t n_kmers, so_node_t *nodes, uint64_t n_nodes,
                   const uint32_t *record_offsets, uint64_t n_record_offsets,
                   const uint8_t *is_targets, uint64_t n_assemblies)
{
    if (n_record_offsets != n_assemblies + 1)                                   /* :33-35 */
        return fail(SO_ERR_VALUE, "len(record_offsets) must equal len(is_targets) + 1");
    if (n_record_offsets == 0 || record_offsets[0] != 0)                        /* :36-38 */
        return fail(SO_ERR_VALUE, "record_offsets must start with 0");
    if (n_assemblies > UINT32_MAX)                                              /* :39-41 */
        return fail(SO_ERR_VALUE, "Number of assemblies exceeds uint32 range");
    uint64_t total_tar = 0, total_neg = 0;
    for (uint64_t i = 0; i < n_assemblies; ++i) {                               /* :45-54 */
        if (record_offsets[i + 1] < record_offsets[i])
            return fail(SO_ERR_VALUE, "record_offsets must be nondecreasing");
        if (is_targets[i]) ++total_tar; else ++total_neg;
    }
    if (total_tar == 0)                                                         /* :55-57 */
        return fail(SO_ERR_VALUE, "is_targets must contain at least one target assembly");
    if (total_neg == 0)                                                         /* :58-60 */
        return fail(SO_ERR_VALUE, "is_targets must contain at least one non-target assembly");

    const uint32_t n_records = record_offsets[n_assemblies];
    uint32_t *last_rec = (uint32_t *)malloc((n_records ? n_records : 1) * sizeof(uint32_t));
    uint8_t *rec_tar = (uint8_t *)malloc(n_records ? n_records : 1);
    if (!last_rec || !rec_tar) { free(last_rec); free(rec_tar); return fail(SO_ERR_RUNTIME, "out of memory"); }
    for (uint64_t a = 0; a < n_assemblies; ++a)                                 /* :68-87 */
        for (uint32_t r = record_offsets[a]; r < record_offsets[a + 1]; ++r) {
            last_rec[r] = record_offsets[a + 1] - 1;
            rec_tar[r] = is_targets[a] ? 1 : 0;
        }
    const double inv_tar = 1.0 / (double)total_tar;                             /* :89-90 */
    const double inv_neg = 1.0 / (double)total_neg;

    int rc = SO_OK;
    for (uint64_t ni = 0; ni < n_nodes && rc == SO_OK; ++ni) {                  /* :92-135 */
        so_node_t *nd = &nodes[ni];
        if (nd->start == nd->stop) {                                            /* :95-100 */
            nd->n_tar = 0; nd->n_neg = 0; nd->penalty = 1.0;
            continue;
        }
        /* The reference is never told len(kmers) and reads past the end on a bad node range
         * (SURVEY 8b); the restatement checks the range instead. */
        if (nd->start > nd->stop || nd->stop > n_kmers) {
            rc = fail(SO_ERR_VALUE, "node range is outside kmers");
            break;
        }
        uint32_t prev = kmers[nd->start].record_idx;
        if (prev >= n_records) { rc = fail(SO_ERR_VALUE, "record_idx is outside record_offsets range"); break; }
        uint32_t last = last_rec[prev];
        uint32_t n_tar = rec_tar[prev], n_neg = 1u - rec_tar[prev];
        for (uint64_t i = nd->start + 1; i < nd->stop; ++i) {
            const uint32_t r = kmers[i].record_idx;
            if (r < prev) { rc = fail(SO_ERR_VALUE, "record_idx must be nondecreasing within each node range"); break; }
            prev = r;
            if (r <= last) continue;
            if (r >= n_records) { rc = fail(SO_ERR_VALUE, "record_idx is outside record_offsets range"); break; }
            last = last_rec[r];
            n_tar += rec_tar[r];
            n_neg += 1u - rec_tar[r];
        }
        if (rc) break;
        nd->n_tar = n_tar;
        nd->n_neg = n_neg;
        /* :132-134 -- separate multiply / add / sqrt in IEEE double; built with -ffp-contract=off */
        const double ft = (double)n_tar * inv_tar;
        const double fn = (double)n_neg * inv_neg;
        const double a = (1.0 - ft) * (1.0 - ft);
        const double b = fn * fn;
        nd->penalty = sqrt(a + b);
    }
    free(last_rec);
    free(rec_tar);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * seqwin::filter_kmers, cpp/src/seqwin/filter.cpp:139-201
 * ---------------------------------------------------------------------------------------- */
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* Two-phase: call with kmers_out == NULL to get the sizes, then again with buffers. */
int so_filter_kmers(const so_kmer_t *kmers, const so_node_t *nodes, uint64_t n_nodes,
                    const uint64_t *used_hashes, uint64_t n_used,
                    so_kmer_t *kmers_out, so_node_t *nodes_out, uint64_t *n_kmers_out, uint64_t *n_nodes_out)
{
    uint64_t *used = (uint64_t *)malloc((n_used ? n_used : 1) * sizeof(uint64_t));
    if (!used) return fail(SO_ERR_RUNTIME, "out of memory");
    if (n_used) memcpy(used, used_hashes, n_used * sizeof(uint64_t));
    if (n_used) qsort(used, n_used, sizeof(uint64_t), cmp_u64);                              /* :145 */
    uint64_t ni = 0, ui = 0, nk = 0, nn = 0;
    while (ni < n_nodes && ui < n_used) {                                        /* :155-173 */
        if (nodes[ni].hash < used[ui]) { ++ni; continue; }
        if (used[ui] < nodes[ni].hash) { ++ui; continue; }
        const uint64_t size = nodes[ni].stop - nodes[ni].start;
        if (kmers_out) {                                                         /* :178-198 */
            nodes_out[nn] = nodes[ni];
            nodes_out[nn].start = nk;
            nodes_out[nn].stop = nk + size;
            memcpy(kmers_out + nk, kmers + nodes[ni].start, size * sizeof(so_kmer_t));
        }
        nk += size;
        ++nn;
        ++ni;
        ++ui;
    }
    free(used);
    *n_kmers_out = nk;
    *n_nodes_out = nn;
    return SO_OK;
}
