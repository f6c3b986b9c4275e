// ingest_dev.hip -- gzip-FASTA ingest on the device: inflate + FASTA parse + 2-bit pack, one file per lane.
//
// Replaces, for inputs made of many .gz files (NCBI's default, src/seqwin/config.py:158), the host side of
// seqwin::internal::read_fasta's gzip branch (reference cpp/src/utils/fasta_reader.cpp:109-203: gzopen / gzread in 64 KiB
// pieces, one worker per file) and the parse of :41-95.  The host only moves the compressed bytes: zlib's inflate runs at
// ~0.3 GB/s of text per core, so 15 000 genomes (75 GB of text) take ~15 s on 16 cores; here every file is one lane of a
// wave -- DEFLATE is serial per stream, but the streams are independent -- with its Huffman tables in LDS (2.3 KB per lane,
// one wave per CU), and the text never leaves HBM.
//
// Three kernels:
//   k_inflate   RFC 1951 (stored / fixed / dynamic blocks) into text[text_off[f] .. + ISIZE); a file qualifies when it is a
//               single gzip member (RFC 1952) whose stream ends exactly at its trailer and inflates to ISIZE bytes.
//   k_parse<0>  the reader's rules (host_ingest.cpp parse_assembly, which restates fasta_reader.cpp:41-95) as a byte-serial
//               state machine: counts records, packed words, valid runs, id bytes; CRC-32 of the text (gzread verifies it).
//   k_parse<1>  the same walk again, writing the packed words and the record / run tables at their final places.
// Anything irregular -- a file that does not qualify, a CRC mismatch, a control byte, sequence before a header -- makes
// device_gz_ingest() return false without raising: the caller then takes the host path, which reports the error the way
// the reference does.  The result is bit-identical to the host path's (tests/test_gpu_parity.py: test_device_gz_ingest*).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "device.hpp"

namespace sw {
namespace {

constexpr int LIT_BITS = 9, DIST_BITS = 7;
constexpr uint32_t ST_OK = 0, ST_BAD_BLOCK = 1, ST_BAD_CODE = 2, ST_TRUNCATED = 3, ST_OVERFLOW = 4, ST_BAD_DIST = 5, ST_BAD_LENS = 6,
                   ST_TRAILING = 7, ST_SHORT = 8;

struct LaneTables {            // per lane, in LDS
    // literal / length table over LIT_BITS bits: [3:0] bits to drop (0: a longer code, or none -- bit by bit), [5:4] n;
    // n = 1..3: that many LITERALS in bytes 1..3 (DNA text is ~2 bits per base: one lookup takes up to three bases);
    // n = 0: one symbol (>= 256: end of block or a length code) in [24:8]
    uint32_t lit[1 << LIT_BITS];
    uint16_t dist[1 << DIST_BITS];  // distance table: symbol << 4 | code length (0: longer than DIST_BITS bits, or no code)
    uint16_t lcount[16], dcount[16], offs[16];
    // the last 128 bytes of output.  Text goes to HBM 64 bytes at a time: a lane's loads wait for ALL its earlier stores (one
    // vmcnt for both on gfx9), so storing every byte at once made every match -- three of four symbols in level-6 DNA --
    // pay a store's round trip before its own (~4 us per match; r03).  Matches that reach into the unwritten tail read it here.
    alignas(8) uint8_t ring[128];
};
static_assert(sizeof(LaneTables) * 64 <= 160 * 1024, "one wave's tables must fit the CU's LDS");
struct LaneScratch {           // per lane, in HBM: what only a block's table set-up and the rare long codes touch
    uint16_t lsym[288];             // symbols in (length, symbol) order: canonical decoding of the long codes
    uint16_t dsym[32];
    uint16_t single[1 << LIT_BITS]; // one-symbol literal / length table the multi-literal one is made from
    uint8_t lens[320];
};

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BitReader {
    // The deflate data is read in aligned 64-bit words, one word ahead of its use (a lane's load latency is ~1 us with one
    // wave per CU: byte loads, each waited for, made the whole decoder run at 3 MB/s per lane).
    const uint64_t *w;      // the compressed arena as words (16 readable bytes behind its last file)
    uint64_t pos, end;      // next byte to take, end of the deflate data (the gzip trailer starts there)
    uint64_t cur, nxt;      // the words holding byte pos and the one behind it
    uint64_t buf;
    uint32_t cnt;           // valid bits in buf
    __device__ void init(const uint8_t *base, uint64_t start, uint64_t stop)
    {
        w = reinterpret_cast<const uint64_t *>(base);
        pos = start;
        end = stop;
        cur = w[pos >> 3];
        nxt = w[(pos >> 3) + 1];
        buf = 0;
        cnt = 0;
    }
    __device__ void refill()    // tops buf up to more than 32 bits (or to the end of the data)
    {
        if (cnt > 32 || pos >= end) return;
        const uint32_t off = (uint32_t)(pos & 7u);
        uint64_t v = cur >> (8u * off);
        if (off > 4) v |= nxt << (64u - 8u * off);            // the four bytes straddle the word boundary
        uint32_t take = 4;
        if (end - pos < 4) {
            take = (uint32_t)(end - pos);
            v &= (1ull << (8u * take)) - 1ull;
        }
        buf |= (v & 0xFFFFFFFFull) << cnt;
        cnt += 8u * take;
        pos += take;
        if (off + take >= 8) {                                 // into the next word: fetch the one behind it
            cur = nxt;
            nxt = w[(pos >> 3) + 1];
        }
    }
    // n <= 32; false: the data ends before n bits
    __device__ bool need(uint32_t n)
    {
        if (cnt < n) refill();
        return cnt >= n;
    }
    __device__ uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    __device__ void drop(uint32_t n)
    {
        buf >>= n;
        cnt -= n;
    }
};

// canonical Huffman tables from code lengths (RFC 1951 3.2.2).  Returns 0: complete code, > 0: incomplete, < 0: over-subscribed.
__device__ int build_tables(const uint8_t *lens, uint32_t n, uint16_t *count, uint16_t *symbols, uint16_t *primary, int pbits, uint16_t *offs)
{
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t s = 0; s < n; ++s) ++count[lens[s]];
    for (uint32_t i = 0; i < (1u << pbits); ++i) primary[i] = 0;
    if (count[0] == n) return 0;               // no codes at all: complete, decoding any symbol fails
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= (int)count[l];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (uint32_t s = 0; s < n; ++s)
        if (lens[s]) symbols[offs[lens[s]]++] = (uint16_t)s;
    uint32_t code = 0, idx = 0;
    for (int l = 1; l <= pbits; ++l) {
        for (uint32_t j = 0; j < count[l]; ++j) {
            const uint32_t sym = symbols[idx++];
            const uint32_t r = __brev(code) >> (32 - l);   // the stream carries a code's bits most significant first
            for (uint32_t k = r; k < (1u << pbits); k += 1u << l) primary[k] = (uint16_t)((sym << 4) | (uint32_t)l);
            ++code;
        }
        code <<= 1;
    }
    return left;
}

// a code bit by bit through the (length, symbol)-ordered list (puff-style): the codes longer than the table's window
__device__ int decode_long(BitReader &br, const uint16_t *count, const uint16_t *symbols)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        if (br.cnt < 1) return -1;
        code |= (int)(br.buf & 1u);
        br.drop(1);
        const int c = (int)count[l];
        if (code - c < first) return (int)symbols[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// one symbol: the primary table, or bit by bit through the (length, symbol)-ordered list (puff-style) for the long codes
__device__ int decode_symbol(BitReader &br, const uint16_t *primary, int pbits, const uint16_t *count, const uint16_t *symbols)
{
    br.refill();
    const uint32_t e = primary[br.peek((uint32_t)pbits)];
    if (e & 15u) {
        if (br.cnt < (e & 15u)) return -1;
        br.drop(e & 15u);
        return (int)(e >> 4);
    }
    return decode_long(br, count, symbols);
}

struct InflateArgs {
    const uint8_t *comp;
    const uint64_t *data_start, *data_end;   // deflate data of file f: comp[data_start[f], data_end[f])
    uint8_t *text;
    const uint64_t *text_off;                // [n_files + 1]; file f may hold text_off[f + 1] - text_off[f] >= its ISIZE bytes
    const uint32_t *isize;
    uint32_t n_files;
    uint32_t *status;
    LaneScratch *scratch;                    // [n_files]
    unsigned long long *prof;                // (SEQWIN_AMD_DEBUG_TIMING) file 0: clocks in table set-up / decoding, blocks, lookups, matches
};

__global__ __launch_bounds__(64) void k_inflate(const InflateArgs A)
{
    __shared__ LaneTables T[64];
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= A.n_files) return;
    LaneTables &t = T[threadIdx.x];
    LaneScratch &g = A.scratch[f];
    BitReader br;
    br.init(A.comp, A.data_start[f], A.data_end[f]);
    uint8_t *out = A.text + A.text_off[f];
    const uint64_t cap = A.isize[f];
    uint64_t n = 0, fl = 0;                   // bytes produced / bytes in HBM (a multiple of 64 until the end)
    uint32_t st = ST_OK;
    bool last = false;
    uint64_t *const outw = reinterpret_cast<uint64_t *>(out);
    auto flush64 = [&]() {
        const uint64_t *r = reinterpret_cast<const uint64_t *>(&t.ring[fl & 127u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) outw[(fl >> 3) + j] = r[j];
        fl += 64;
    };
    auto emit1 = [&](uint32_t b) {
        t.ring[n & 127u] = (uint8_t)b;
        ++n;
        if (n - fl >= 64) flush64();
    };
    const bool prof = A.prof && f == 0;
    unsigned long long p_build = 0, p_dec = 0, p_blocks = 0, p_look = 0, p_match = 0, p_t0 = 0;
    while (!last && st == ST_OK) {
        if (prof) { p_t0 = clock64(); ++p_blocks; }
        if (!br.need(3)) { st = ST_TRUNCATED; break; }
        last = br.peek(1) != 0;
        const uint32_t type = (br.peek(3) >> 1);
        br.drop(3);
        if (type == 0) {                        // stored: to a byte boundary, LEN, NLEN, bytes
            br.drop(br.cnt & 7u);
            if (!br.need(32)) { st = ST_TRUNCATED; break; }
            const uint32_t v = br.peek(32);
            br.drop(32);
            const uint32_t len = v & 0xFFFFu;
            if ((len ^ (v >> 16)) != 0xFFFFu) { st = ST_BAD_BLOCK; break; }
            if (n + len > cap) { st = ST_OVERFLOW; break; }
            for (uint32_t i = 0; i < len; ++i) {
                if (!br.need(8)) { st = ST_TRUNCATED; break; }
                emit1(br.peek(8));
                br.drop(8);
            }
            continue;
        }
        if (type == 3) { st = ST_BAD_BLOCK; break; }
        uint32_t nlen, ndist;
        if (type == 1) {                        // fixed code (3.2.6)
            for (int s = 0; s < 144; ++s) g.lens[s] = 8;
            for (int s = 144; s < 256; ++s) g.lens[s] = 9;
            for (int s = 256; s < 280; ++s) g.lens[s] = 7;
            for (int s = 280; s < 288; ++s) g.lens[s] = 8;
            for (int s = 288; s < 318; ++s) g.lens[s] = 5;
            nlen = 288;
            ndist = 30;
        } else {                                // dynamic code (3.2.7)
            if (!br.need(14)) { st = ST_TRUNCATED; break; }
            nlen = br.peek(5) + 257;
            br.drop(5);
            ndist = br.peek(5) + 1;
            br.drop(5);
            const uint32_t ncode = br.peek(4) + 4;
            br.drop(4);
            if (nlen > 286 || ndist > 30) { st = ST_BAD_LENS; break; }
            for (int i = 0; i < 19; ++i) g.lens[i] = 0;
            for (uint32_t i = 0; i < ncode; ++i) {
                if (!br.need(3)) { st = ST_TRUNCATED; break; }
                g.lens[kClOrder[i]] = (uint8_t)br.peek(3);
                br.drop(3);
            }
            if (st != ST_OK) break;
            // the code-length code: its tables live in the distance tables' space until the lengths are read
            if (build_tables(g.lens, 19, t.dcount, g.dsym, t.dist, DIST_BITS, t.offs) != 0) { st = ST_BAD_LENS; break; }
            uint32_t idx = 0;
            uint8_t *ll = g.lens;                // (the 19 lengths above have been consumed)
            while (idx < nlen + ndist) {
                const int sym = decode_symbol(br, t.dist, DIST_BITS, t.dcount, g.dsym);
                if (sym < 0) { st = ST_BAD_CODE; break; }
                if (sym < 16) {
                    ll[idx++] = (uint8_t)sym;
                } else {
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) { st = ST_BAD_LENS; break; }
                        val = ll[idx - 1];
                        if (!br.need(2)) { st = ST_TRUNCATED; break; }
                        rep = 3 + br.peek(2);
                        br.drop(2);
                    } else if (sym == 17) {
                        if (!br.need(3)) { st = ST_TRUNCATED; break; }
                        rep = 3 + br.peek(3);
                        br.drop(3);
                    } else {
                        if (!br.need(7)) { st = ST_TRUNCATED; break; }
                        rep = 11 + br.peek(7);
                        br.drop(7);
                    }
                    if (idx + rep > nlen + ndist) { st = ST_BAD_LENS; break; }
                    while (rep--) ll[idx++] = (uint8_t)val;
                }
            }
            if (st != ST_OK) break;
            if (ll[256] == 0) { st = ST_BAD_LENS; break; }     // no end-of-block code
        }
        {
            const int e1 = build_tables(g.lens, nlen, t.lcount, g.lsym, g.single, LIT_BITS, t.offs);
            // (the fixed distance code is incomplete by definition: 30 of 32 five-bit codes)
            if (type == 2 && e1 && (e1 < 0 || nlen != (uint32_t)t.lcount[0] + t.lcount[1])) { st = ST_BAD_LENS; break; }   // incomplete: one code only
            const int e2 = build_tables(g.lens + nlen, ndist, t.dcount, g.dsym, t.dist, DIST_BITS, t.offs);
            if (type == 2 && e2 && (e2 < 0 || ndist != (uint32_t)t.dcount[0] + t.dcount[1])) { st = ST_BAD_LENS; break; }
        }
        for (uint32_t i = 0; i < (1u << LIT_BITS); ++i) {    // up to three literals per entry
            const uint32_t a = g.single[i], la = a & 15u;
            uint32_t e = 0;
            if (la) {
                if ((a >> 4) >= 256u) {
                    e = ((a >> 4) << 8) | la;
                } else {
                    uint32_t total = la, cnt = 1, lits = a >> 4;
                    for (int more = 0; more < 2; ++more) {
                        // the bits behind the codes taken so far; a code found there counts if all its bits lie inside the window
                        const uint32_t b = g.single[i >> total], lb = b & 15u;
                        if (!lb || total + lb > (uint32_t)LIT_BITS || (b >> 4) >= 256u) break;
                        lits |= (b >> 4) << (8 * cnt);
                        total += lb;
                        ++cnt;
                    }
                    e = (lits << 8) | (cnt << 4) | total;
                }
            }
            t.lit[i] = e;
        }
        if (prof) { const unsigned long long c = clock64(); p_build += c - p_t0; p_t0 = c; }
        for (;;) {                              // every iteration writes a byte, ends the block or fails: <= cap + 1 iterations
            int sym;
            if (prof) ++p_look;
            {
                br.refill();
                const uint32_t e = t.lit[br.peek(LIT_BITS)];
                const uint32_t tl = e & 15u;
                if (tl) {
                    if (br.cnt < tl) { st = ST_TRUNCATED; break; }
                    const uint32_t cnt = (e >> 4) & 3u;
                    if (cnt) {
                        if (n + cnt > cap) { st = ST_OVERFLOW; break; }
                        t.ring[n & 127u] = (uint8_t)(e >> 8);
                        if (cnt > 1) t.ring[(n + 1) & 127u] = (uint8_t)(e >> 16);
                        if (cnt > 2) t.ring[(n + 2) & 127u] = (uint8_t)(e >> 24);
                        n += cnt;
                        if (n - fl >= 64) flush64();
                        br.drop(tl);
                        continue;
                    }
                    br.drop(tl);
                    sym = (int)(e >> 8);
                } else {
                    sym = decode_long(br, t.lcount, g.lsym);
                    if (sym < 0) { st = br.cnt == 0 && br.pos >= br.end ? ST_TRUNCATED : ST_BAD_CODE; break; }
                    if (sym < 256) {
                        if (n >= cap) { st = ST_OVERFLOW; break; }
                        emit1((uint32_t)sym);
                        continue;
                    }
                }
            }
            if (sym == 256) break;
            if (prof) ++p_match;
            if (sym > 285) { st = ST_BAD_CODE; break; }
            const uint32_t li = (uint32_t)sym - 257u;
            if (!br.need(kLenExtra[li])) { st = ST_TRUNCATED; break; }
            const uint32_t len = kLenBase[li] + br.peek(kLenExtra[li]);
            br.drop(kLenExtra[li]);
            const int ds = decode_symbol(br, t.dist, DIST_BITS, t.dcount, g.dsym);
            if (ds < 0 || ds > 29) { st = ST_BAD_CODE; break; }
            if (!br.need(kDistExtra[ds])) { st = ST_TRUNCATED; break; }
            const uint64_t dist = (uint64_t)kDistBase[ds] + br.peek(kDistExtra[ds]);
            br.drop(kDistExtra[ds]);
            if (dist > n) { st = ST_BAD_DIST; break; }
            if (n + len > cap) { st = ST_OVERFLOW; break; }
            if (dist >= 8) {
                // Eight source bytes per step: from HBM (two aligned words, requested together) when they have been written
                // there, else from the ring.  dist >= 8: the bytes a step reads were all produced before it.
                const uint64_t *ow = reinterpret_cast<const uint64_t *>(A.text);
                for (uint32_t i = 0; i < len; i += 8) {
                    const uint64_t sp = n - dist;                    // position of the step's first source byte
                    const uint32_t m = min(8u, len - i);
                    uint64_t v;
                    if (sp + 8 <= fl) {
                        const uint64_t a = A.text_off[f] + sp;
                        const uint32_t off = (uint32_t)(a & 7u);
                        const uint64_t w0 = ow[a >> 3], w1 = ow[(a >> 3) + 1];
                        v = off ? (w0 >> (8u * off)) | (w1 << (64u - 8u * off)) : w0;
                    } else {                                         // (then sp >= n - 128: fl >= n - 71)
                        v = 0;
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j)
                            if (j < m) v |= (uint64_t)t.ring[(sp + j) & 127u] << (8u * j);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if (j < m) t.ring[(n + j) & 127u] = (uint8_t)(v >> (8u * j));
                    n += m;
                    if (n - fl >= 64) flush64();
                }
            } else {
                for (uint32_t i = 0; i < len; ++i) emit1(t.ring[(n - dist) & 127u]);   // (overlapping: byte by byte, forwards)
            }
        }
        if (prof) p_dec += clock64() - p_t0;
    }
    for (; fl < n; ++fl) out[fl] = t.ring[fl & 127u];   // the tail
    if (prof) {
        A.prof[0] = p_build; A.prof[1] = p_dec; A.prof[2] = p_blocks; A.prof[3] = p_look; A.prof[4] = p_match; A.prof[5] = n;
    }
    if (st == ST_OK) {
        const uint64_t used = br.pos - (br.cnt >> 3);     // whole bytes still in the bit buffer were not consumed
        if (used != br.end) st = ST_TRAILING;              // another member, or garbage before the trailer
        else if (n != cap) st = ST_SHORT;
    }
    A.status[f] = st;
}

// ---- FASTA text -> records, 2-bit words, valid runs (host_ingest.cpp: parse_assembly / Packer, byte by byte) -------------
constexpr uint32_t PE_SEQ_BEFORE_HEADER = 1, PE_CONTROL_BYTE = 2;

struct ParseCounts {
    uint32_t n_rec, n_runs, err, crc;
    uint64_t n_words, n_id, total_bp;
};

struct ParseArgs {
    const uint8_t *text;
    const uint64_t *text_off;       // [n_files + 1]
    const uint32_t *isize;
    uint32_t n_files;
    ParseCounts *counts;            // <0>: written; <1>: read (for the checks only)
    // <1>: where file f's output starts
    const uint64_t *word_base, *id_base;
    const uint32_t *rec_base_idx, *run_base;
    uint64_t *words;                // the batch's packed stream, 64-bit words
    uint32_t *rec_len, *rec_run_off, *run_pos, *run_len;
    uint64_t *rec_base;
    char *ids;
};

template <bool WRITE> __global__ __launch_bounds__(64) void k_parse(const ParseArgs A)
{
    __shared__ uint8_t cls[256];
    __shared__ uint32_t crc_tab[4][256];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) {
        uint8_t c = 4;                                                  // CharTable of host_ingest.cpp
        if (i == 'A' || i == 'a') c = 0;
        else if (i == 'C' || i == 'c') c = 1;
        else if (i == 'G' || i == 'g') c = 2;
        else if (i == 'T' || i == 't' || i == 'U' || i == 'u') c = 3;
        else if (i == ' ' || i == '\t' || i == '\n' || i == '\r' || i == '\f' || i == '\v') c = 5;
        else if (i == 1 || i == 3 || i == 4 || i == 5 || i == 7) c = 6;
        cls[i] = c;
        if (!WRITE) {
            uint32_t r = i;
            for (int b = 0; b < 8; ++b) r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1u)));
            crc_tab[0][i] = r;
        }
    }
    __syncthreads();
    if (!WRITE) {
        for (uint32_t i = threadIdx.x; i < 256; i += 64)
            for (int s = 1; s < 4; ++s) crc_tab[s][i] = (crc_tab[s - 1][i] >> 8) ^ crc_tab[0][crc_tab[s - 1][i] & 0xFFu];
        __syncthreads();
    }
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= A.n_files) return;
    const uint8_t *t = A.text + A.text_off[f];     // (16-byte aligned)
    const uint64_t n = A.isize[f];

    // The text is read 16 bytes at a time, one load ahead of its use (16-byte aligned, and readable up to the next multiple
    // of 16: text_off is a multiple of 16 and the arena ends 16 bytes behind the last file).
    const uint4 *tv = reinterpret_cast<const uint4 *>(t);
    const uint64_t n_chunks = (n + 15) / 16;
    uint32_t crc = 0xFFFFFFFFu;
    if (!WRITE) {                                  // CRC-32 (RFC 1952 8.), four bytes per step
        uint4 nx = n_chunks ? tv[0] : uint4{0, 0, 0, 0};
        for (uint64_t ch = 0; ch < n_chunks; ++ch) {
            const uint4 cu = nx;
            if (ch + 1 < n_chunks) nx = tv[ch + 1];
            const uint32_t wv[4] = {cu.x, cu.y, cu.z, cu.w};
            const uint64_t left = n - ch * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (left >= (uint64_t)(4 * q + 4)) {
                    const uint32_t v = crc ^ wv[q];
                    crc = crc_tab[3][v & 0xFFu] ^ crc_tab[2][(v >> 8) & 0xFFu] ^ crc_tab[1][(v >> 16) & 0xFFu] ^ crc_tab[0][v >> 24];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (left > (uint64_t)(4 * q + j)) crc = crc_tab[0][(crc ^ (wv[q] >> (8 * j))) & 0xFFu] ^ (crc >> 8);
                }
            }
        }
        crc = ~crc;
    }

    uint32_t n_rec = 0, n_runs = 0, err = 0;
    uint64_t n_words = 0, n_id = 0, total_bp = 0;
    uint64_t wb = 0, ib = 0;
    uint32_t rb = 0, ub = 0;
    if (WRITE) {
        wb = A.word_base[f];
        ib = A.id_base[f];
        rb = A.rec_base_idx[f];
        ub = A.run_base[f];
    }
    bool have = false;
    int mode = 0;                                  // 0 first byte of a line, 1 record id, 2 rest of a header, 3 sequence line
    uint64_t acc = 0, len = 0;
    uint32_t nacc = 0;
    int64_t run_start = -1;
    auto close_record = [&]() {
        if (run_start >= 0) {
            if (WRITE) {
                A.run_pos[ub + n_runs] = (uint32_t)run_start;
                A.run_len[ub + n_runs] = (uint32_t)(len - (uint64_t)run_start);
            }
            ++n_runs;
        }
        if (nacc) {
            if (WRITE) A.words[wb + n_words] = acc;
            ++n_words;
        }
        if (WRITE) A.rec_len[rb + n_rec - 1] = (uint32_t)len;     // (len <= ISIZE < 2^32)
        total_bp += len;
    };
    auto step = [&](const uint32_t c) {
        if (c == '\n') {
            if (mode == 1) {                       // the id ran to the end of its line
                if (WRITE) A.ids[ib + n_id] = 0;
                ++n_id;
            }
            mode = 0;
            return;
        }
        if (mode == 0) {
            if (c == '>') {                        // fasta_reader.cpp:58-67
                if (have) close_record();
                if (WRITE) {
                    A.rec_base[rb + n_rec] = (wb + n_words) * 32;
                    A.rec_run_off[rb + n_rec] = ub + n_runs;
                }
                ++n_rec;
                acc = 0;
                nacc = 0;
                len = 0;
                run_start = -1;
                have = true;
                mode = 1;
                return;
            }
            mode = 3;
        }
        const uint32_t k = cls[c];
        if (mode == 1) {                           // extract_id, :26-33: up to the first whitespace
            if (k == 5) {
                if (WRITE) A.ids[ib + n_id] = 0;
                ++n_id;
                mode = 2;
            } else {
                if (WRITE) A.ids[ib + n_id] = (char)c;
                ++n_id;
            }
            return;
        }
        if (mode == 2 || k == 5) return;
        if (k == 6) { err |= PE_CONTROL_BYTE; return; }
        if (!have) { err |= PE_SEQ_BEFORE_HEADER; return; }   // :69-71
        if (k < 4) {                               // Packer::push
            if (run_start < 0) run_start = (int64_t)len;
            acc |= (uint64_t)k << (2 * nacc);
        } else if (run_start >= 0) {
            if (WRITE) {
                A.run_pos[ub + n_runs] = (uint32_t)run_start;
                A.run_len[ub + n_runs] = (uint32_t)(len - (uint64_t)run_start);
            }
            ++n_runs;
            run_start = -1;
        }
        ++len;
        if (++nacc == 32) {
            if (WRITE) A.words[wb + n_words] = acc;
            ++n_words;
            acc = 0;
            nacc = 0;
        }
    };
    {
        uint4 nx = n_chunks ? tv[0] : uint4{0, 0, 0, 0};
        for (uint64_t ch = 0; ch < n_chunks; ++ch) {
            const uint4 cu = nx;
            if (ch + 1 < n_chunks) nx = tv[ch + 1];
            const uint32_t wv[4] = {cu.x, cu.y, cu.z, cu.w};
            const uint64_t left = n - ch * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t v = wv[q];
                if (mode == 3 && have && left >= (uint64_t)(4 * q + 4)) {
                    // four bytes inside a sequence line: if all are bases (the rule on 80-column lines: 18 of 20 words) they go
                    // into the accumulator together -- Packer::push_block for nb = 4, all valid
                    const uint32_t k0 = cls[v & 0xFFu], k1 = cls[(v >> 8) & 0xFFu], k2 = cls[(v >> 16) & 0xFFu], k3 = cls[v >> 24];
                    if ((k0 | k1 | k2 | k3) < 4u) {
                        const uint64_t codes = k0 | (k1 << 2) | (k2 << 4) | (k3 << 6);
                        if (run_start < 0) run_start = (int64_t)len;
                        acc |= codes << (2 * nacc);
                        len += 4;
                        nacc += 4;
                        if (nacc >= 32) {
                            if (WRITE) A.words[wb + n_words] = acc;
                            ++n_words;
                            nacc -= 32;
                            acc = nacc ? codes >> (8 - 2 * nacc) : 0;     // the bases that did not fit the word
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (left > (uint64_t)(4 * q + j)) step((v >> (8 * j)) & 0xFFu);
            }
        }
    }
    if (mode == 1) {                               // the file ends inside an id
        if (WRITE) A.ids[ib + n_id] = 0;
        ++n_id;
    }
    if (have) close_record();
    if (!WRITE) {
        ParseCounts pc;
        pc.n_rec = n_rec;
        pc.n_runs = n_runs;
        pc.err = err;
        pc.crc = crc;
        pc.n_words = n_words;
        pc.n_id = n_id;
        pc.total_bp = total_bp;
        A.counts[f] = pc;
    }
}

std::atomic<uint64_t> g_device_gz_batches{0};

bool ends_with_gz(const char *s)
{
    const size_t n = strlen(s);
    return n >= 3 && !strcmp(s + n - 3, ".gz");
}

// RFC 1952 2.3: the offset of the deflate data in a member that starts at h[0], or 0 if the header is not a plain one
uint64_t gzip_header_len(const uint8_t *h, uint64_t n)
{
    if (n < 18 || h[0] != 0x1F || h[1] != 0x8B || h[2] != 8 || (h[3] & 0xE0)) return 0;
    const uint8_t flg = h[3];
    uint64_t p = 10;
    if (flg & 4) {                                  // FEXTRA
        if (p + 2 > n) return 0;
        p += 2 + ((uint64_t)h[p] | ((uint64_t)h[p + 1] << 8));
    }
    for (int bit : {8, 16})                         // FNAME, FCOMMENT: NUL-terminated
        if (flg & bit) {
            while (p < n && h[p]) ++p;
            ++p;
        }
    if (flg & 2) p += 2;                            // FHCRC
    return p + 8 <= n ? p : 0;
}

struct Pinned {   // one pinned staging buffer per worker, kept for the life of the process (allocating costs ~0.25 ms per MiB)
    static constexpr size_t BYTES = 4u << 20;
    char *p = nullptr;
    hipStream_t st = nullptr;
};
Pinned &pinned_slot(size_t i)
{
    static std::mutex mu;
    static std::vector<Pinned *> *slots = new std::vector<Pinned *>;   // leaked on purpose
    std::lock_guard<std::mutex> lock(mu);
    while (slots->size() <= i) slots->push_back(nullptr);
    if (!(*slots)[i]) {
        Pinned *s = new Pinned;
        SW_HIP(hipHostMalloc((void **)&s->p, Pinned::BYTES, hipHostMallocDefault));
        SW_HIP(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
        (*slots)[i] = s;
    }
    return *(*slots)[i];
}

template <class T> void to_host(std::vector<T> &dst, const DevArray<T> &src, size_t n)
{
    dst.resize(n);
    if (n) SW_HIP(hipMemcpy(dst.data(), src.p, n * sizeof(T), hipMemcpyDeviceToHost));
}

}  // namespace

// All paths end in ".gz" and there are enough of them (SEQWIN_AMD_DEVICE_INFLATE=1: any number; =0: never): inflate, parse
// and pack on the device.  true: b.host / b.d_packed / b.packed_words are filled as ingest_to_device's host route fills
// them (the caller uploads the record tables); false: nothing was changed -- take the host route.
bool device_gz_ingest(const char *const *paths, size_t n_paths, uint64_t n_cpu, sw_batch &b)
{
    const char *mode = getenv("SEQWIN_AMD_DEVICE_INFLATE");
    if (mode && !strcmp(mode, "0")) return false;
    const bool forced = mode && !strcmp(mode, "1");
    // A lane inflates ~1.6 MB of text per second whatever the number of files (every match is a memory round trip of the
    // lane's own: 3.3 us per symbol, r03), a host thread ~370 MB/s: the device wins from ~320 files per host thread on
    // (measured: 8 192 files of 1 Mbp, 16 threads: 843 ms against 1 387 ms; 1 024 files: 802 against 181 ms).
    const uint64_t host_threads = std::min<uint64_t>(std::max<uint64_t>(1, n_cpu), 64);
    if (n_paths == 0 || n_paths >= 0xFFFFFFFFull || (!forced && n_paths < 320 * host_threads)) return false;
    for (size_t i = 0; i < n_paths; ++i)
        if (!ends_with_gz(paths[i])) return false;
    const bool timing = getenv("SEQWIN_AMD_DEBUG_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto decline = [&](const char *why, size_t file, unsigned long long v) {
        if (timing) fprintf(stderr, "[seqwin_amd] device gz ingest declined: %s (file %zu: %s, %llu)\n", why, file, paths[file], v);
        return false;
    };

    // -- sizes, then the compressed bytes to the device (every file at a 16-byte boundary) ---------------------------
    std::vector<uint64_t> fsize(n_paths), coff(n_paths + 1, 0);
    for (size_t i = 0; i < n_paths; ++i) {
        struct stat st;
        if (stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) return decline("not a regular file of 18 bytes or more", i, 0);
        fsize[i] = (uint64_t)st.st_size;
        coff[i + 1] = coff[i] + ((fsize[i] + 15) & ~15ull);
    }
    size_t free_b = 0, total_b = 0;
    SW_HIP(hipMemGetInfo(&free_b, &total_b));
    if (coff[n_paths] * 8 > free_b) return decline("not enough free HBM", 0, free_b);          // (text ~4x the compressed bytes, + packed words + tables: the host route streams)
    DevArray<uint8_t> d_comp(coff[n_paths] + 16);
    std::vector<uint64_t> dstart(n_paths), dend(n_paths);
    std::vector<uint32_t> isize(n_paths), crc_want(n_paths);
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    std::atomic<size_t> bad_file{0};
    const size_t n_workers = std::max<size_t>(1, std::min<size_t>({(size_t)std::max<uint64_t>(1, n_cpu), n_paths, 32}));
    auto reader = [&](size_t w) {
        try {
            Pinned &pin = pinned_slot(w);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_paths || !ok.load()) break;
                const int fd = open(paths[i], O_RDONLY);
                if (fd < 0) { bad_file.store(i); ok.store(false); break; }
                uint64_t done = 0;
                bool first = true, bad = false;
                uint8_t tail[8] = {0};
                while (done < fsize[i]) {
                    SW_HIP(hipStreamSynchronize(pin.st));              // the slot's previous copy has left the host
                    const size_t want = (size_t)std::min<uint64_t>(Pinned::BYTES, fsize[i] - done);
                    size_t got = 0;
                    while (got < want) {
                        const ssize_t r = read(fd, pin.p + got, want - got);
                        if (r <= 0) break;
                        got += (size_t)r;
                    }
                    if (got != want) { bad = true; break; }
                    if (first) {
                        dstart[i] = gzip_header_len((const uint8_t *)pin.p, std::min<uint64_t>(got, fsize[i]));
                        if (dstart[i] == 0) { bad = true; break; }
                        first = false;
                    }
                    for (uint64_t k = 0; k < 8; ++k) {                 // the trailer: the file's last eight bytes
                        const uint64_t at = fsize[i] - 8 + k;
                        if (at >= done && at < done + got) tail[k] = (uint8_t)pin.p[at - done];
                    }
                    SW_HIP(hipMemcpyAsync(d_comp.p + coff[i] + done, pin.p, got, hipMemcpyHostToDevice, pin.st));
                    done += got;
                }
                close(fd);
                if (bad || dstart[i] + 8 > fsize[i]) { bad_file.store(i); ok.store(false); break; }
                crc_want[i] = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
                isize[i] = (uint32_t)tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
                dend[i] = coff[i] + fsize[i] - 8;
                dstart[i] += coff[i];
            }
            SW_HIP(hipStreamSynchronize(pin.st));
        } catch (...) {
            ok.store(false);
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t w = 0; w < n_workers; ++w) th.emplace_back(reader, w);
        for (auto &t : th) t.join();
    }
    if (!ok.load()) return decline("unreadable, or not a plain gzip header", bad_file.load(), 0);
    const auto t1 = std::chrono::steady_clock::now();

    // -- inflate -------------------------------------------------------------------------------------------------------
    std::vector<uint64_t> toff(n_paths + 1, 0);
    for (size_t i = 0; i < n_paths; ++i) toff[i + 1] = toff[i] + (((uint64_t)isize[i] + 15) & ~15ull);
    SW_HIP(hipMemGetInfo(&free_b, &total_b));
    if (toff[n_paths] + toff[n_paths] / 2 > free_b) return decline("not enough free HBM for the text", 0, toff[n_paths]);
    const uint32_t nf = (uint32_t)n_paths;
    DevArray<uint8_t> d_text(toff[n_paths] + 16);
    DevArray<uint64_t> d_dstart(nf), d_dend(nf), d_toff(nf + 1);
    DevArray<uint32_t> d_isize(nf), d_status(nf);
    SW_HIP(hipMemcpy(d_dstart.p, dstart.data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_dend.p, dend.data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_toff.p, toff.data(), (nf + 1) * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_isize.p, isize.data(), nf * 4ull, hipMemcpyHostToDevice));
    const unsigned blocks = (nf + 63) / 64;
    DevArray<LaneScratch> d_scratch(nf);
    DevArray<unsigned long long> d_prof(8);
    SW_HIP(hipMemset(d_prof.p, 0, 64));
    InflateArgs ia{d_comp.p, d_dstart.p, d_dend.p, d_text.p, d_toff.p, d_isize.p, nf, d_status.p, d_scratch.p, timing ? d_prof.p : nullptr};
    hipLaunchKernelGGL(k_inflate, dim3(blocks), dim3(64), 0, nullptr, ia);
    SW_HIP(hipGetLastError());
    if (timing) {
        unsigned long long hp[8];
        SW_HIP(hipMemcpy(hp, d_prof.p, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[seqwin_amd] inflate, file 0: %llu bytes, %llu blocks, %llu lookups, %llu matches; clocks: table set-up %llu, decoding %llu\n",
                hp[5], hp[2], hp[3], hp[4], hp[0], hp[1]);
    }
    std::vector<uint32_t> status;
    to_host(status, d_status, nf);
    for (uint32_t i = 0; i < nf; ++i)
        if (status[i] != ST_OK) return decline("inflate status", i, status[i]);   // corrupt, or more than one member: the host route decides
    d_comp.release();
    const auto t2 = std::chrono::steady_clock::now();

    // -- parse: count, place, write ------------------------------------------------------------------------------------
    DevArray<ParseCounts> d_counts(nf);
    ParseArgs pa{};
    pa.text = d_text.p;
    pa.text_off = d_toff.p;
    pa.isize = d_isize.p;
    pa.n_files = nf;
    pa.counts = d_counts.p;
    hipLaunchKernelGGL(k_parse<false>, dim3(blocks), dim3(64), 0, nullptr, pa);
    SW_HIP(hipGetLastError());
    std::vector<ParseCounts> counts;
    to_host(counts, d_counts, nf);
    std::vector<uint64_t> word_base(nf + 1, 0), id_base(nf + 1, 0);
    std::vector<uint32_t> rec_idx(nf + 1, 0), run_base(nf + 1, 0);
    uint64_t total_bp = 0;
    for (uint32_t i = 0; i < nf; ++i) {
        if (counts[i].err) return decline("parse error flags", i, counts[i].err);
        if (counts[i].crc != crc_want[i]) return decline("CRC-32 mismatch", i, counts[i].crc);
        word_base[i + 1] = word_base[i] + counts[i].n_words;
        id_base[i + 1] = id_base[i] + counts[i].n_id;
        const uint64_t r = (uint64_t)rec_idx[i] + counts[i].n_rec, u = (uint64_t)run_base[i] + counts[i].n_runs;
        if (r > UINT32_MAX || u > UINT32_MAX) return decline("more than 2^32-1 records or runs", i, r);    // (the host route raises: build.cpp:136-140)
        rec_idx[i + 1] = (uint32_t)r;
        run_base[i + 1] = (uint32_t)u;
        total_bp += counts[i].total_bp;
    }
    const uint64_t n_words = word_base[nf], n_rec = rec_idx[nf], n_runs = run_base[nf], n_id = id_base[nf];
    DevArray<uint32_t> d_packed(n_words * 2 + 8);
    DevArray<uint64_t> d_wb(nf), d_ib(nf), d_rec_base(n_rec);
    DevArray<uint32_t> d_ri(nf), d_ub(nf), d_rec_len(n_rec), d_rec_run_off(n_rec), d_run_pos(n_runs), d_run_len(n_runs);
    DevArray<char> d_ids(n_id);
    SW_HIP(hipMemcpy(d_wb.p, word_base.data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_ib.p, id_base.data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_ri.p, rec_idx.data(), nf * 4ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_ub.p, run_base.data(), nf * 4ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemsetAsync(d_packed.p + 2 * n_words, 0, 8 * 4, nullptr));   // the read slack
    pa.word_base = d_wb.p;
    pa.id_base = d_ib.p;
    pa.rec_base_idx = d_ri.p;
    pa.run_base = d_ub.p;
    pa.words = reinterpret_cast<uint64_t *>(d_packed.p);
    pa.rec_len = d_rec_len.p;
    pa.rec_run_off = d_rec_run_off.p;
    pa.run_pos = d_run_pos.p;
    pa.run_len = d_run_len.p;
    pa.rec_base = d_rec_base.p;
    pa.ids = d_ids.p;
    hipLaunchKernelGGL(k_parse<true>, dim3(blocks), dim3(64), 0, nullptr, pa);
    SW_HIP(hipGetLastError());

    // -- the tables the planner reads, back on the host ----------------------------------------------------------------
    HostBatch h;
    h.n_assemblies = n_paths;
    h.total_bp = total_bp;
    h.record_offsets.assign(rec_idx.begin(), rec_idx.end());
    to_host(h.rec_len, d_rec_len, n_rec);
    to_host(h.rec_base, d_rec_base, n_rec);
    to_host(h.rec_run_off, d_rec_run_off, n_rec);
    h.rec_run_off.push_back((uint32_t)n_runs);
    to_host(h.run_pos, d_run_pos, n_runs);
    to_host(h.run_len, d_run_len, n_runs);
    std::vector<char> ids;
    to_host(ids, d_ids, n_id);
    h.ids_blob.assign(ids.data(), ids.size());
    h.chunks.resize(n_paths);
    h.chunk_word0.assign(word_base.begin(), word_base.end());
    SW_HIP(hipDeviceSynchronize());
    b.host = std::move(h);
    b.d_packed = std::move(d_packed);
    b.packed_words = n_words * 2 + 8;
    g_device_gz_batches.fetch_add(1);
    if (timing)
        fprintf(stderr, "[seqwin_amd] device gz ingest: %zu files, read + upload %.1f ms, inflate %.1f ms, parse + tables %.1f ms\n", n_paths,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    return true;
}

}  // namespace sw

extern "C" uint64_t sw_device_gz_batches(void) { return sw::g_device_gz_batches.load(); }
