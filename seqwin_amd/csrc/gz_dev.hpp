// gz_dev.hpp -- DEFLATE decoder and FASTA parser / 2-bit packer of the device gzip ingest (ingest_dev.hip), one stream per
// caller.  Plain C++ over pointers: the kernels of ingest_dev.hip run it one file per lane (all decoder state in LDS);
// tests/tools/gz_dev_san_driver.cpp compiles THE SAME code for the host with AddressSanitizer + UBSan and feeds it
// well-formed and broken streams (tests/test_abi_cpu.py) -- GPU sanitizers do not exist on the target.
#pragma once
#include <cstdint>

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define SW_GZ_CLOCK() clock64()
#else
#define SW_GZ_CLOCK() 0ull
#endif
#if defined(__HIPCC__)
#define SW_GZ_FN __device__ __host__ inline
#define SW_GZ_CONST __device__ const
#else
#define SW_GZ_FN inline
#define SW_GZ_CONST static const
#endif

namespace sw {
namespace gz {

SW_GZ_FN uint32_t brev32(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}

constexpr int LIT_BITS = 8, DIST_BITS = 7;
constexpr uint32_t ST_OK = 0, ST_BAD_BLOCK = 1, ST_BAD_CODE = 2, ST_TRUNCATED = 3, ST_OVERFLOW = 4, ST_BAD_DIST = 5, ST_BAD_LENS = 6,
                   ST_TRAILING = 7, ST_SHORT = 8;

struct LaneTables {            // per lane, in LDS: everything a stream's decoder touches except its input and output
    // literal / length table over LIT_BITS bits: [3:0] bits to drop (0: a longer code, or none -- bit by bit), [5:4] n;
    // n = 1..3: that many LITERALS in bytes 1..3 (DNA text is ~2 bits per base: one lookup takes up to three bases);
    // n = 0: one symbol (>= 256: end of block or a length code) in [24:8]
    uint32_t lit[1 << LIT_BITS];
    uint16_t dist[1 << DIST_BITS];  // distance table: symbol << 4 | code length (0: longer than DIST_BITS bits, or no code)
    uint16_t lsym[288];             // symbols in (length, symbol) order: canonical decoding of the long codes
    uint16_t dsym[32];
    uint16_t lcount[16], dcount[16], offs[16];
    uint8_t lens[320];              // code lengths of a block's two alphabets (table set-up)
    // the last 128 bytes of output.  Text goes to HBM 64 bytes at a time: a lane's loads wait for ALL its earlier stores (one
    // vmcnt for both on gfx9), so storing every byte at once made every match -- three of four symbols in level-6 DNA --
    // pay a store's round trip before its own (r03).  Matches that reach into the unwritten tail read it here.
    alignas(8) uint8_t ring[128];
};
static_assert(sizeof(LaneTables) * 64 <= 160 * 1024, "one wave's tables must fit the CU's LDS");

// RFC 1951 3.2.5: base value and extra bits of length code 257 + li and of distance code ds, as arithmetic (a table in
// memory would be a per-lane global load -- a memory round trip -- four times per match)
SW_GZ_FN void length_code(uint32_t li, uint32_t &base, uint32_t &extra)
{
    extra = li < 4u ? 0u : (li - 4u) >> 2;
    base = li < 8u ? 3u + li : ((4u + (li & 3u)) << extra) + 3u;
    if (li == 28u) { base = 258u; extra = 0u; }
}
SW_GZ_FN void distance_code(uint32_t ds, uint32_t &base, uint32_t &extra)
{
    extra = ds < 2u ? 0u : (ds >> 1) - 1u;
    base = ds < 2u ? 1u + ds : ((2u + (ds & 1u)) << extra) + 1u;
}
SW_GZ_CONST uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BitReader {
    // The deflate data is read in aligned 64-bit words, one word ahead of its use (a lane's load latency is ~1 us with one
    // wave per CU: byte loads, each waited for, made the whole decoder run at 3 MB/s per lane).
    const uint64_t *w;      // the compressed arena as words (16 readable bytes behind its last file)
    uint64_t pos, end;      // next byte to take, end of the deflate data (the gzip trailer starts there)
    uint64_t cur, nxt;      // the words holding byte pos and the one behind it
    uint64_t buf;
    uint32_t cnt;           // valid bits in buf
    SW_GZ_FN void init(const uint8_t *base, uint64_t start, uint64_t stop)
    {
        w = reinterpret_cast<const uint64_t *>(base);
        pos = start;
        end = stop;
        cur = w[pos >> 3];
        nxt = w[(pos >> 3) + 1];
        buf = 0;
        cnt = 0;
    }
    SW_GZ_FN void refill()    // tops buf up to more than 32 bits (or to the end of the data)
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
    SW_GZ_FN bool need(uint32_t n)
    {
        if (cnt < n) refill();
        return cnt >= n;
    }
    SW_GZ_FN uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    SW_GZ_FN void drop(uint32_t n)
    {
        buf >>= n;
        cnt -= n;
    }
};

// canonical Huffman tables from code lengths (RFC 1951 3.2.2).  Returns 0: complete code, > 0: incomplete, < 0: over-subscribed.
template <class P>
SW_GZ_FN int build_tables(const uint8_t *lens, uint32_t n, uint16_t *count, uint16_t *symbols, P *primary, int pbits, uint16_t *offs)
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
            const uint32_t r = brev32(code) >> (32 - l);   // the stream carries a code's bits most significant first
            for (uint32_t k = r; k < (1u << pbits); k += 1u << l) primary[k] = (P)((sym << 4) | (uint32_t)l);
            ++code;
        }
        code <<= 1;
    }
    return left;
}

// a code bit by bit through the (length, symbol)-ordered list (puff-style): the codes longer than the table's window
SW_GZ_FN int decode_long(BitReader &br, const uint16_t *count, const uint16_t *symbols)
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
SW_GZ_FN int decode_symbol(BitReader &br, const uint16_t *primary, int pbits, const uint16_t *count, const uint16_t *symbols)
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


// One deflate stream comp[start, end) -> arena[text_off, text_off + cap) (arena: 8-byte aligned, text_off a multiple of 16,
// 16 readable bytes behind every region).  Returns ST_OK only if the stream ends exactly at `end` after exactly cap bytes.
// prof_out (may be null): clocks in table set-up / decoding, blocks, lookups, matches, bytes.
SW_GZ_FN uint32_t inflate_one(LaneTables &t, const uint8_t *comp, uint64_t start, uint64_t end, uint8_t *arena,
                              uint64_t text_off, uint64_t cap, unsigned long long *prof_out)
{
    BitReader br;
    br.init(comp, start, end);
    uint8_t *out = arena + text_off;
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
    const bool prof = prof_out != nullptr;
    unsigned long long p_build = 0, p_dec = 0, p_blocks = 0, p_look = 0, p_match = 0, p_t0 = 0, p_load = 0, p_nload = 0;
    while (!last && st == ST_OK) {
        if (prof) { p_t0 = SW_GZ_CLOCK(); ++p_blocks; }
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
            for (int s = 0; s < 144; ++s) t.lens[s] = 8;
            for (int s = 144; s < 256; ++s) t.lens[s] = 9;
            for (int s = 256; s < 280; ++s) t.lens[s] = 7;
            for (int s = 280; s < 288; ++s) t.lens[s] = 8;
            for (int s = 288; s < 318; ++s) t.lens[s] = 5;
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
            for (int i = 0; i < 19; ++i) t.lens[i] = 0;
            for (uint32_t i = 0; i < ncode; ++i) {
                if (!br.need(3)) { st = ST_TRUNCATED; break; }
                t.lens[kClOrder[i]] = (uint8_t)br.peek(3);
                br.drop(3);
            }
            if (st != ST_OK) break;
            // the code-length code: its tables live in the distance tables' space until the lengths are read
            if (build_tables(t.lens, 19, t.dcount, t.dsym, t.dist, DIST_BITS, t.offs) != 0) { st = ST_BAD_LENS; break; }
            uint32_t idx = 0;
            uint8_t *ll = t.lens;                // (the 19 lengths above have been consumed)
            while (idx < nlen + ndist) {
                const int sym = decode_symbol(br, t.dist, DIST_BITS, t.dcount, t.dsym);
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
            const int e1 = build_tables(t.lens, nlen, t.lcount, t.lsym, t.lit, LIT_BITS, t.offs);
            // (the fixed distance code is incomplete by definition: 30 of 32 five-bit codes)
            if (type == 2 && e1 && (e1 < 0 || nlen != (uint32_t)t.lcount[0] + t.lcount[1])) { st = ST_BAD_LENS; break; }   // incomplete: one code only
            const int e2 = build_tables(t.lens + nlen, ndist, t.dcount, t.dsym, t.dist, DIST_BITS, t.offs);
            if (type == 2 && e2 && (e2 < 0 || ndist != (uint32_t)t.dcount[0] + t.dcount[1])) { st = ST_BAD_LENS; break; }
        }
        // t.lit holds one symbol per entry (symbol << 4 | length); up to three literals per entry are packed IN PLACE, from the
        // last entry down: entry i looks at entries i >> (bits taken) < i only, which are still in their one-symbol form
        for (uint32_t i = (1u << LIT_BITS); i-- > 0;) {
            const uint32_t a = t.lit[i], la = a & 15u;
            uint32_t e = 0;
            if (la) {
                if ((a >> 4) >= 256u) {
                    e = ((a >> 4) << 8) | la;
                } else {
                    uint32_t total = la, cnt = 1, lits = a >> 4;
                    for (int more = 0; more < 2; ++more) {
                        // the bits behind the codes taken so far; a code found there counts if all its bits lie inside the window
                        const uint32_t b = t.lit[i >> total], lb = b & 15u;     // (i == 0: its own entry, not yet rewritten)
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
        if (prof) { const unsigned long long c = SW_GZ_CLOCK(); p_build += c - p_t0; p_t0 = c; }
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
                    sym = decode_long(br, t.lcount, t.lsym);
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
            uint32_t lbase, lextra;
            length_code(li, lbase, lextra);
            if (!br.need(lextra)) { st = ST_TRUNCATED; break; }
            const uint32_t len = lbase + br.peek(lextra);
            br.drop(lextra);
            const int ds = decode_symbol(br, t.dist, DIST_BITS, t.dcount, t.dsym);
            if (ds < 0 || ds > 29) { st = ST_BAD_CODE; break; }
            uint32_t dbase, dextra;
            distance_code((uint32_t)ds, dbase, dextra);
            if (!br.need(dextra)) { st = ST_TRUNCATED; break; }
            const uint64_t dist = (uint64_t)dbase + br.peek(dextra);
            br.drop(dextra);
            if (dist > n) { st = ST_BAD_DIST; break; }
            if (n + len > cap) { st = ST_OVERFLOW; break; }
            if (dist >= 8) {
                // Eight source bytes per step: from HBM (two aligned words, requested together) when they have been written
                // there, else from the ring.  dist >= 8: the bytes a step reads were all produced before it.
                const uint64_t *ow = reinterpret_cast<const uint64_t *>(arena);
                for (uint32_t i = 0; i < len; i += 8) {
                    const uint64_t sp = n - dist;                    // position of the step's first source byte
                    const uint32_t m = (len - i < 8u ? len - i : 8u);
                    uint64_t v;
                    if (sp + 8 <= fl) {
                        const uint64_t a = text_off + sp;
                        const uint32_t off = (uint32_t)(a & 7u);
                        unsigned long long c0 = 0;
                        if (prof) c0 = SW_GZ_CLOCK();
                        const uint64_t w0 = ow[a >> 3], w1 = ow[(a >> 3) + 1];
                        v = off ? (w0 >> (8u * off)) | (w1 << (64u - 8u * off)) : w0;
                        if (prof) {                                  // (the stamp waits for the words: v feeds it)
                            p_load += SW_GZ_CLOCK() - c0 + (v & 0ull);
                            ++p_nload;
                        }
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
        if (prof) p_dec += SW_GZ_CLOCK() - p_t0;
    }
    for (; fl < n; ++fl) out[fl] = t.ring[fl & 127u];   // the tail
    if (prof) {
        prof_out[0] = p_build; prof_out[1] = p_dec; prof_out[2] = p_blocks; prof_out[3] = p_look; prof_out[4] = p_match; prof_out[5] = n; prof_out[6] = p_load; prof_out[7] = p_nload;
    }
    if (st == ST_OK) {
        const uint64_t used = br.pos - (br.cnt >> 3);     // whole bytes still in the bit buffer were not consumed
        if (used != br.end) st = ST_TRAILING;              // another member, or garbage before the trailer
        else if (n != cap) st = ST_SHORT;
    }
    return st;
}

// ---- FASTA text -> records, 2-bit words, valid runs (host_ingest.cpp: parse_assembly / Packer, byte by byte) -------------
constexpr uint32_t PE_SEQ_BEFORE_HEADER = 1, PE_CONTROL_BYTE = 2;

struct ParseCounts {
    uint32_t n_rec, n_runs, err, crc;
    uint64_t n_words, n_id, total_bp;
};
struct ParseDst {                   // where a file's output starts (WRITE pass)
    uint64_t word_base, id_base;
    uint32_t rec_base_idx, run_base;
    uint64_t *words;                // the batch's packed stream, 64-bit words
    uint32_t *rec_len, *rec_run_off, *run_pos, *run_len;
    uint64_t *rec_base;
    char *ids;
};
struct alignas(16) Quad { uint32_t x, y, z, w; };

SW_GZ_FN uint8_t char_class(uint32_t i)     // CharTable of host_ingest.cpp
{
    if (i == 'A' || i == 'a') return 0;
    if (i == 'C' || i == 'c') return 1;
    if (i == 'G' || i == 'g') return 2;
    if (i == 'T' || i == 't' || i == 'U' || i == 'u') return 3;
    if (i == ' ' || i == '\t' || i == '\n' || i == '\r' || i == '\f' || i == '\v') return 5;
    if (i == 1 || i == 3 || i == 4 || i == 5 || i == 7) return 6;
    return 4;
}
SW_GZ_FN uint32_t crc_entry(uint32_t i)      // CRC-32 (RFC 1952 8.) of one byte
{
    uint32_t r = i;
    for (int b = 0; b < 8; ++b) r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1u)));
    return r;
}

// cls[256] = char_class; crc_tab[s][i] (WRITE == false only) = the slicing-by-4 tables made from crc_entry.
// t: 16-byte aligned, readable up to the next multiple of 16 behind n.
template <bool WRITE>
SW_GZ_FN void parse_one(const uint8_t *cls, const uint32_t (*crc_tab)[256], const uint8_t *t, uint64_t n, const ParseDst &D,
                        ParseCounts &pc)
{
    // The text is read 16 bytes at a time, one load ahead of its use (16-byte aligned, and readable up to the next multiple
    // of 16: text_off is a multiple of 16 and the arena ends 16 bytes behind the last file).
    const Quad *tv = reinterpret_cast<const Quad *>(t);
    const uint64_t n_chunks = (n + 15) / 16;
    uint32_t crc = 0xFFFFFFFFu;
    if (!WRITE) {                                  // CRC-32 (RFC 1952 8.), four bytes per step
        Quad nx = n_chunks ? tv[0] : Quad{0, 0, 0, 0};
        for (uint64_t ch = 0; ch < n_chunks; ++ch) {
            const Quad cu = nx;
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
        wb = D.word_base;
        ib = D.id_base;
        rb = D.rec_base_idx;
        ub = D.run_base;
    }
    bool have = false;
    int mode = 0;                                  // 0 first byte of a line, 1 record id, 2 rest of a header, 3 sequence line
    uint64_t acc = 0, len = 0;
    uint32_t nacc = 0;
    int64_t run_start = -1;
    auto close_record = [&]() {
        if (run_start >= 0) {
            if (WRITE) {
                D.run_pos[ub + n_runs] = (uint32_t)run_start;
                D.run_len[ub + n_runs] = (uint32_t)(len - (uint64_t)run_start);
            }
            ++n_runs;
        }
        if (nacc) {
            if (WRITE) D.words[wb + n_words] = acc;
            ++n_words;
        }
        if (WRITE) D.rec_len[rb + n_rec - 1] = (uint32_t)len;     // (len <= ISIZE < 2^32)
        total_bp += len;
    };
    auto step = [&](const uint32_t c) {
        if (c == '\n') {
            if (mode == 1) {                       // the id ran to the end of its line
                if (WRITE) D.ids[ib + n_id] = 0;
                ++n_id;
            }
            mode = 0;
            return;
        }
        if (mode == 0) {
            if (c == '>') {                        // fasta_reader.cpp:58-67
                if (have) close_record();
                if (WRITE) {
                    D.rec_base[rb + n_rec] = (wb + n_words) * 32;
                    D.rec_run_off[rb + n_rec] = ub + n_runs;
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
                if (WRITE) D.ids[ib + n_id] = 0;
                ++n_id;
                mode = 2;
            } else {
                if (WRITE) D.ids[ib + n_id] = (char)c;
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
                D.run_pos[ub + n_runs] = (uint32_t)run_start;
                D.run_len[ub + n_runs] = (uint32_t)(len - (uint64_t)run_start);
            }
            ++n_runs;
            run_start = -1;
        }
        ++len;
        if (++nacc == 32) {
            if (WRITE) D.words[wb + n_words] = acc;
            ++n_words;
            acc = 0;
            nacc = 0;
        }
    };
    {
        Quad nx = n_chunks ? tv[0] : Quad{0, 0, 0, 0};
        for (uint64_t ch = 0; ch < n_chunks; ++ch) {
            const Quad cu = nx;
            if (ch + 1 < n_chunks) nx = tv[ch + 1];
            const uint32_t wv[4] = {cu.x, cu.y, cu.z, cu.w};
            const uint64_t left = n - ch * 16;
            if (mode == 3 && have && left >= 16) {
                // a whole chunk inside a sequence line (four of the five chunks of an 80-column line): sixteen look-ups in
                // flight, and if all are bases they go into the accumulator together (Packer::push_block, all valid)
                uint32_t any = 0;
                uint64_t codes = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t k = cls[(wv[j >> 2] >> (8 * (j & 3))) & 0xFFu];
                    any |= k;
                    codes |= (uint64_t)k << (2 * j);
                }
                if (any < 4u) {
                    if (run_start < 0) run_start = (int64_t)len;
                    acc |= codes << (2 * nacc);
                    len += 16;
                    nacc += 16;
                    if (nacc >= 32) {
                        if (WRITE) D.words[wb + n_words] = acc;
                        ++n_words;
                        nacc -= 32;
                        acc = nacc ? codes >> (32 - 2 * nacc) : 0;    // the bases that did not fit the word
                    }
                    continue;
                }
            }
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
                            if (WRITE) D.words[wb + n_words] = acc;
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
        if (WRITE) D.ids[ib + n_id] = 0;
        ++n_id;
    }
    if (have) close_record();
    pc.n_rec = n_rec;
    pc.n_runs = n_runs;
    pc.err = err;
    pc.crc = crc;
    pc.n_words = n_words;
    pc.n_id = n_id;
    pc.total_bp = total_bp;
}

// RFC 1952 2.3: the offset of the deflate data in a member that starts at h[0], or 0 if the header is not a plain one
inline uint64_t gzip_header_len(const uint8_t *h, uint64_t n)
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

}  // namespace gz
}  // namespace sw
