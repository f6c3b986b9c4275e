// fast_inflate.hpp -- whole-buffer gzip (RFC 1952) / DEFLATE (RFC 1951) decoder for the host ingest (r05).
//
// The reference reads .fna.gz -- its DEFAULT input (src/seqwin/config.py:158) -- through zlib's gzread with a 64 KiB buffer
// (cpp/src/utils/fasta_reader.cpp:109-203).  zlib's inflate decodes one symbol per table lookup from a bit buffer it refills
// byte by byte: ~0.4 GB/s of text per core on the MI355X host, which made a 16-core ingest of gzip FASTA 4 x slower than
// that of plain FASTA (bench.py e2e.gz).  This decoder is written for the case the ingest has: the whole compressed file in
// memory, the whole output buffer allocated (ISIZE says how much), so the hot loop has no "need more input / output" states:
//   * a 64-bit bit buffer refilled with ONE unaligned 8-byte load (>= 56 valid bits after every refill: a length/distance pair
//     with all its extra bits -- at most 48 bits -- or three literals never need a second one);
//   * an 11-bit first-level table for literals / lengths and an 8-bit one for distances whose entries carry the code length,
//     the number of extra bits and the base value in one 32-bit word; longer codes go through second-level tables;
//   * matches are copied eight bytes at a time (distance >= 8), by a byte fill (distance 1: runs) or byte by byte.
//
// CONTRACT.  gunzip_fast() either returns true with exactly the bytes zlib's gzread loop would have delivered for a file that
// consists of one or more well-formed gzip members, CRC-32 and ISIZE of every member checked -- or it returns false, having
// promised nothing, and the caller takes the zlib route, which IS the reference's behaviour for everything unusual
// (truncated streams, trailing garbage, corrupt data, reserved flags, incomplete or over-subscribed code sets ...).
// Nothing here trusts the input: every table index is masked, every distance is checked against the bytes produced so
// far, the output never passes its capacity, consumed input is checked against the real length (the caller pads the input
// with 16 zero bytes so that refills never read outside the allocation).  tests/tools/ingest_san_driver.cpp runs it against
// zlib on well-formed and mutated streams under AddressSanitizer + UBSan.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <initializer_list>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace sw {
namespace finf {

constexpr unsigned LL_BITS = 11, D_BITS = 8;          // first-level table widths
constexpr unsigned MAX_LL_SYMS = 288, MAX_D_SYMS = 32, MAX_CODE_LEN = 15;
constexpr unsigned LL_TABLE = (1u << LL_BITS) + 1024; // + room for second-level tables (<= 2^(15-11) entries per long prefix; bounded below)
constexpr unsigned D_TABLE = (1u << D_BITS) + 512;
constexpr size_t IN_PAD = 16;                          // zero bytes the caller appends to the compressed data

// entry: [7:0] bits of the Huffman code (first level: of the whole code, or -- SUB -- the first-level width; second level: the
// remaining bits), [11:8] extra bits, [15:12] kind, [31:16] base value / literal / index of the second-level table
enum : uint32_t { K_LIT = 0x8000u, K_EOB = 0x4000u, K_SUB = 0x2000u, K_BAD = 0x1000u };

// r06b: the SUPER table -- what the next LL_BITS bits of the stream decode to when they hold MORE than one symbol: up to two literals
// and / or a whole match (length code + its extra bits + distance code; the distance's extra bits are read from the stream behind).
// Level-6 DNA text is three matches in four symbols, a match is a length code of 2-3 bits + a distance code of 2-5 bits + ~10 extra
// bits, a literal 4 bits: "literal, match" and "match" fit the 11 index bits in ~9 of 10 cases, and one table load decodes what took
// two or three dependent ones (refill -> ll -> d: ~19 cycles per match; the loop is that chain).  Layout of an entry:
//   [5:0] ALL the bits the entry consumes (literal codes + length code + length extra + distance code + distance extra bits: the
//   shift count of the bit buffer, the one field on the loop's dependency chain -- a 64-bit shift takes its low six bits as they are),
//   [9:6] the code bits among them (= where the distance's extra bits start), [11:10] literals (0-2), [12] a match follows them,
//   [21:13] match length, [36:22] distance base, [44:37] [52:45] the literals.
// 0 = nothing to gain here (end of block, long codes, a length whose distance code is not inside the index): the generic path.
constexpr uint64_t S_ANY = 0x1C00u, S_MATCH = 0x1000u;
struct Tables {
    uint32_t ll[LL_TABLE];
    uint32_t d[D_TABLE];
    uint64_t sup[1u << LL_BITS];
    bool d_usable;   // false: the block declared no distance code at all (a block of literals only): any match is an error
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bit_reverse(uint32_t code, unsigned len)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman code of `lens[0..n)` -> two-level decoding table.  `first_bits` = width of the first level, `cap` = entries of
// `table`.  sym_entry(s) gives the entry of symbol s without its code length.  Returns false for an over-subscribed or incomplete
// set (zlib rejects these, with one exception handled by the caller) and for anything that would not fit.
template <class SymEntry>
inline bool build_table(const uint8_t *lens, unsigned n, unsigned first_bits, uint32_t *table, unsigned cap, SymEntry sym_entry,
                        bool allow_single_code)
{
    unsigned count[MAX_CODE_LEN + 1] = {0};
    for (unsigned s = 0; s < n; ++s) {
        if (lens[s] > MAX_CODE_LEN) return false;
        ++count[lens[s]];
    }
    const unsigned used = n - count[0];
    if (used == 0) return false;
    // Kraft sum: must be exactly 1 (complete), except a single code of length 1 where allowed (zlib: incomplete distance set with one code)
    uint32_t left = 1;
    for (unsigned l = 1; l <= MAX_CODE_LEN; ++l) {
        left <<= 1;
        if (count[l] > left) return false;   // over-subscribed
        left -= count[l];
    }
    const bool single = used == 1 && count[1] == 1;
    if (left != 0 && !(single && allow_single_code)) return false;
    uint32_t next_code[MAX_CODE_LEN + 2];
    {
        uint32_t code = 0;
        for (unsigned l = 1; l <= MAX_CODE_LEN; ++l) {
            next_code[l] = code;
            code = (code + count[l]) << 1;
        }
    }
    const unsigned first_size = 1u << first_bits;
    for (unsigned i = 0; i < first_size; ++i) table[i] = K_BAD | first_bits;
    // second-level tables: for every first-level prefix that longer codes share, the longest such code decides the table's width
    uint8_t sub_bits[1u << LL_BITS];   // (first_bits <= LL_BITS)
    memset(sub_bits, 0, first_size);
    uint32_t codes[MAX_LL_SYMS];
    {
        uint32_t nc[MAX_CODE_LEN + 2];
        memcpy(nc, next_code, sizeof nc);
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            const uint32_t rev = bit_reverse(nc[l]++, l);
            codes[s] = rev;
            if (l > first_bits) {
                const unsigned prefix = rev & (first_size - 1u);
                if (l - first_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - first_bits);
            }
        }
    }
    unsigned next_free = first_size;
    for (unsigned p = 0; p < first_size; ++p)
        if (sub_bits[p]) {
            const unsigned sz = 1u << sub_bits[p];
            if (next_free + sz > cap) return false;
            table[p] = K_SUB | ((uint32_t)next_free << 16) | ((uint32_t)sub_bits[p] << 8) | first_bits;   // [11:8]: width of the second level
            for (unsigned i = 0; i < sz; ++i) table[next_free + i] = K_BAD | sub_bits[p];
            next_free += sz;
        }
    for (unsigned s = 0; s < n; ++s) {
        const unsigned l = lens[s];
        if (!l) continue;
        const uint32_t rev = codes[s], e = sym_entry(s);
        if (l <= first_bits) {
            for (uint32_t i = rev; i < first_size; i += 1u << l) table[i] = e | l;
        } else {
            const unsigned prefix = rev & (first_size - 1u), sb = sub_bits[prefix], base = table[prefix] >> 16, rl = l - first_bits;
            for (uint32_t i = rev >> first_bits; i < (1u << sb); i += 1u << rl) table[base + i] = e | rl;
        }
    }
    return true;
}

inline uint32_t ll_entry(unsigned s)
{
    if (s < 256) return K_LIT | ((uint32_t)s << 16);
    if (s == 256) return K_EOB;
    if (s > 285) return K_BAD;   // 286, 287: in the fixed code, never valid in data
    return ((uint32_t)kLenBase[s - 257] << 16) | ((uint32_t)kLenExtra[s - 257] << 8);
}
inline uint32_t d_entry(unsigned s)
{
    if (s > 29) return K_BAD;
    return ((uint32_t)kDistBase[s] << 16) | ((uint32_t)kDistExtra[s] << 8);
}

// One entry of the super table, decoded straight from the block's two tables.  Index bits are consumed from the low end; behind
// `used` consumed bits only LL_BITS - used bits are known, so a symbol counts only if its first-level entry is a plain one (no
// second level, not invalid) whose code is no longer than that.  (The definition: build_super below must agree with it entry for
// entry -- tests/tools/finf_san_driver.cpp compares them on every case's last dynamic block.)
inline uint64_t super_entry(const Tables &t, unsigned i)
{
    unsigned used = 0, nlit = 0;
    uint64_t lits = 0, s = 0;
    uint32_t e = t.ll[i];
    while (nlit < 2 && (e & K_LIT) && (e & 0xFFu) <= LL_BITS - used) {
        lits |= (uint64_t)((e >> 16) & 0xFFu) << (8 * nlit);
        ++nlit;
        used += e & 0xFFu;
        e = t.ll[i >> used];
    }
    if (!(e & (K_LIT | K_EOB | K_SUB | K_BAD))) {   // a length code
        const unsigned lc = e & 0xFFu, lx = (e >> 8) & 0xFu;
        if (lc + lx <= LL_BITS - used) {
            const unsigned len = (e >> 16) + ((i >> (used + lc)) & ((1u << lx) - 1u)), u2 = used + lc + lx;
            const uint32_t de = t.d[(i >> u2) & ((1u << D_BITS) - 1u)];
            if (!(de & (K_SUB | K_BAD)) && (de & 0xFFu) <= LL_BITS - u2)
                s = (uint64_t)(u2 + (de & 0xFFu) + ((de >> 8) & 0xFu)) | (uint64_t)(u2 + (de & 0xFFu)) << 6 | S_MATCH | (uint64_t)len << 13 |
                    (uint64_t)(de >> 16) << 22;
        }
    }
    if (!s) s = used | (uint64_t)used << 6;   // (literals only -- or nothing: 0, since no literal means used = 0)
    return s | (uint64_t)nlit << 10 | lits << 37;
}

// The super table of a block (built per dynamic block, 2 048 entries), in ascending index order with two short cuts:
//   * an entry that ends in a match depends on its `code` low index bits only (decoding stopped there), so it is the entry of
//     every index with those low bits: computed once, at the smallest such index, and copied to the others;
//   * an entry that starts with a literal of l bits is that literal in front of what the remaining bits decode to -- entry
//     [i >> l], already there (i >> l < i), valid here if it holds at most one literal and depends on no more index bits than are
//     left: one table load instead of a chain of up to four.
inline void build_super(Tables &t)
{
    if (!t.d_usable) return;   // (a block of literals only never enters the fast loop)
    constexpr unsigned N = 1u << LL_BITS;
    memset(t.sup, 0, sizeof t.sup);
    for (unsigned i = 0; i < N; ++i) {
        if (t.sup[i]) continue;   // (a copy of a smaller index's match entry)
        const uint32_t e = t.ll[i];
        uint64_t s = 0;
        if (i == 0) {
            s = super_entry(t, 0);   // (refers to itself in the recurrence)
        } else if (e & K_LIT) {
            const unsigned l0 = e & 0xFFu, left = LL_BITS - l0;
            const uint64_t lit0 = (e >> 16) & 0xFFu, sj = t.sup[i >> l0];
            const unsigned cj = (unsigned)(sj >> 6) & 0xFu, nj = (unsigned)(sj >> 10) & 3u;
            if ((sj & S_ANY) && nj <= 1 && cj <= left) {
                s = (sj & (S_MATCH | 0x1FFull << 13 | 0x7FFFull << 22)) | (uint64_t)((sj & 63u) + l0) | (uint64_t)(cj + l0) << 6 |
                    (uint64_t)(nj + 1) << 10 | (lit0 | ((sj >> 37) & 0xFFu) << 8) << 37;
            } else {
                const uint32_t e1 = t.ll[i >> l0];   // a second literal?
                if ((e1 & K_LIT) && (e1 & 0xFFu) <= left) {
                    const unsigned used = l0 + (e1 & 0xFFu);
                    s = used | (uint64_t)used << 6 | 2u << 10 | (lit0 | (uint64_t)((e1 >> 16) & 0xFFu) << 8) << 37;
                } else {
                    s = l0 | (uint64_t)l0 << 6 | 1u << 10 | lit0 << 37;
                }
            }
        } else if (!(e & (K_EOB | K_SUB | K_BAD))) {   // a length code: the match, if its distance code lies inside the index
            const unsigned lc = e & 0xFFu, lx = (e >> 8) & 0xFu;
            if (lc + lx <= LL_BITS) {
                const unsigned len = (e >> 16) + ((i >> lc) & ((1u << lx) - 1u)), u2 = lc + lx;
                const uint32_t de = t.d[(i >> u2) & ((1u << D_BITS) - 1u)];
                if (!(de & (K_SUB | K_BAD)) && (de & 0xFFu) <= LL_BITS - u2)
                    s = (uint64_t)(u2 + (de & 0xFFu) + ((de >> 8) & 0xFu)) | (uint64_t)(u2 + (de & 0xFFu)) << 6 | S_MATCH | (uint64_t)len << 13 |
                        (uint64_t)(de >> 16) << 22;
            }
        }
        t.sup[i] = s;
        if (s & S_MATCH) {
            const unsigned step = 1u << ((unsigned)(s >> 6) & 0xFu);
            for (unsigned j = i + step; j < N; j += step) t.sup[j] = s;
        }
    }
}

struct BitReader {
    const uint8_t *in, *in_end;   // in_end: end of the REAL data; IN_PAD zero bytes follow it
    uint64_t buf = 0;
    unsigned cnt = 0;
    inline void refill()
    {
        uint64_t w;
        memcpy(&w, in, 8);           // (little-endian host: x86-64)
        buf |= w << cnt;
        in += (63u - cnt) >> 3;
        cnt |= 56u;
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void drop(unsigned n) { buf >>= n; cnt -= n; }
    // bytes of real input consumed so far (whole bytes still in the buffer are not consumed)
    inline const uint8_t *position() const { return in - (cnt >> 3); }
    inline bool overrun() const { return position() > in_end; }
};

enum Result { OK = 0, BAD = 1, NEED_OUT = 2 };

// the fixed Huffman code of RFC 1951 section 3.2.6 (built once; a function-local static: thread-safe)
inline const Tables &fixed_tables()
{
    static const Tables fixed = [] {
        Tables t;
        uint8_t lens[MAX_LL_SYMS];
        for (unsigned s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        uint8_t dl[MAX_D_SYMS];
        for (unsigned s = 0; s < 32; ++s) dl[s] = 5;
        (void)build_table(lens, 288, LL_BITS, t.ll, LL_TABLE, ll_entry, false);
        (void)build_table(dl, 32, D_BITS, t.d, D_TABLE, d_entry, false);
        t.d_usable = true;
        build_super(t);
        return t;
    }();
    return fixed;
}

// One DEFLATE stream: the bit reader at its current bit, out_begin..out_cap the output buffer, `out` the write position (history =
// [out_begin, out)), the block that is open.  inflate_stream() below runs one stream from its first bit to its end.  (r06 also had
// TWO streams advancing in one loop, 1.2 x on the loop as it was then; the super table made the one-stream loop twice as fast as
// that pair loop, which is gone: NOTES.md.)
struct Stream {
    BitReader br;
    uint8_t *out_begin = nullptr, *out = nullptr, *out_cap = nullptr;
    Tables *T = nullptr;              // storage for the tables of a dynamic block
    const Tables *tb = nullptr;       // the open block's tables
    bool in_block = false, final_block = false, ended = false;
};

// Block headers from the reader's position on: stored blocks are copied out here; returns with a Huffman-coded block open
// (s.in_block) or with the stream's last block behind it (s.ended).
inline Result open_block(Stream &s)
{
    BitReader &br = s.br;
    uint8_t *&out = s.out;
    uint8_t *const out_cap = s.out_cap;
    Tables &T = *s.T;
    // Every refill outside the fast loop is followed by an overrun() test before the next one: a refill moves `in` by at most 7
    // bytes and leaves >= 56 bits, so "not overrun" means in <= in_end + 7 and the NEXT refill's 8-byte load ends at or before
    // in_end + 15 < in_end + IN_PAD (the block header and the code-length-code loop refilled without that test, and a
    // stream truncated inside a dynamic header read up to 6 bytes behind the pad).
    for (;;) {
        br.refill();
        if (br.overrun()) return BAD;
        const unsigned final_block = br.peek(1), type = (br.peek(3) >> 1);
        br.drop(3);
        const Tables *tp = &T;
        if (type == 0) {   // stored
            br.drop(br.cnt & 7u);
            const uint8_t *p = br.position();
            if (p + 4 > br.in_end) return BAD;
            const unsigned len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return BAD;
            p += 4;
            if ((size_t)(br.in_end - p) < len) return BAD;
            if ((size_t)(out_cap - out) < len) return NEED_OUT;
            memcpy(out, p, len);
            out += len;
            br.in = p + len;
            br.buf = 0;
            br.cnt = 0;
            if (final_block) {
                s.ended = true;
                return OK;
            }
            continue;
        } else if (type == 1) {
            tp = &fixed_tables();
        } else if (type == 2) {
            const unsigned hlit = br.peek(5) + 257, hdist = (br.peek(10) >> 5) + 1, hclen = (br.peek(14) >> 10) + 4;
            br.drop(14);
            if (hlit > 286 || hdist > 30) return BAD;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            br.refill();
            if (br.overrun()) return BAD;
            for (unsigned i = 0; i < hclen; ++i) {   // (19 x 3 = 57 bits: at most one more refill)
                if (br.cnt < 3) {
                    br.refill();
                    if (br.overrun()) return BAD;
                }
                cl[order[i]] = (uint8_t)br.peek(3);
                br.drop(3);
            }
            if (br.overrun()) return BAD;   // (the header's last bits were padding: truncated)
            uint32_t clt[128 + 8];
            if (!build_table(cl, 19, 7, clt, 128, [](unsigned s) { return (uint32_t)s << 16; }, false)) return BAD;
            uint8_t lens[MAX_LL_SYMS + MAX_D_SYMS];
            unsigned i = 0;
            const unsigned total = hlit + hdist;
            while (i < total) {
                br.refill();
                const uint32_t e = clt[br.peek(7)];
                if (e & K_BAD) return BAD;
                br.drop(e & 0xFFu);
                const unsigned s = e >> 16;
                if (s < 16) {
                    lens[i++] = (uint8_t)s;
                } else {
                    unsigned rep, v = 0;
                    if (s == 16) {
                        if (i == 0) return BAD;
                        v = lens[i - 1];
                        rep = 3 + br.peek(2);
                        br.drop(2);
                    } else if (s == 17) {
                        rep = 3 + br.peek(3);
                        br.drop(3);
                    } else {
                        rep = 11 + br.peek(7);
                        br.drop(7);
                    }
                    if (i + rep > total) return BAD;
                    memset(lens + i, (int)v, rep);
                    i += rep;
                }
                if (br.overrun()) return BAD;
            }
            if (lens[256] == 0) return BAD;   // no end-of-block code (zlib: "invalid code -- missing end-of-block")
            if (!build_table(lens, hlit, LL_BITS, T.ll, LL_TABLE, ll_entry, false)) return BAD;
            // distance codes: all lengths zero = a block of literals only (allowed); one code of length 1 = allowed (incomplete)
            bool any = false;
            for (unsigned s = 0; s < hdist; ++s) any = any || lens[hlit + s];
            T.d_usable = any;
            if (any && !build_table(lens + hlit, hdist, D_BITS, T.d, D_TABLE, d_entry, true)) return BAD;
            build_super(T);
        } else {
            return BAD;
        }
        s.tb = tp;
        s.final_block = final_block != 0;
        s.in_block = true;
        return OK;
    }
}

// The open block's symbols while a refill cannot reach the end of the real input and 320 bytes of output are free.
// The stream's state lives in locals (a byte store may alias anything behind a reference), and the loop is COUNTED: an iteration
// consumes < 9 bytes of input (at most two literals + a length / distance pair with all extra bits: 70 bits; `in` runs at most 8 bytes
// ahead of what is consumed) and writes <= 260 bytes (+ 16 of copy overshoot), so K iterations are safe when K is derived from what
// is left of both buffers -- one decrement per symbol instead of two pointer comparisons, and K is recomputed when it runs out.
inline Result fast_loop(Stream &s, bool &block_done)
{
    const Tables &tb = *s.tb;
    if (!tb.d_usable) return OK;   // (a block of literals only stays out of the fast loop: its one test per match would be paid by all)
    uint64_t buf = s.br.buf;
    unsigned cnt = s.br.cnt;
    const uint8_t *in = s.br.in, *const in_end = s.br.in_end;
    uint8_t *out = s.out, *const out_begin = s.out_begin, *const out_cap = s.out_cap;
    Result res = OK;
    constexpr uint32_t IDX = (1u << LL_BITS) - 1u;
#define SW_FINF_REFILL()                 \
    do {                                 \
        uint64_t w_;                     \
        memcpy(&w_, in, 8);              \
        buf |= w_ << cnt;                \
        in += (63u - cnt) >> 3;          \
        cnt |= 56u;                      \
    } while (0)
    for (;;) {
        const ptrdiff_t in_left = in_end - in;   // (signed: `in` may stand up to 8 bytes behind the end of the real input, inside the pad)
        const size_t out_left = (size_t)(out_cap - out);
        if (in_left < 64 || out_left < 320) break;
        size_t k = (size_t)in_left / 16;   // 9 (k + 1) + 16 <= 16 k from k = 4 on: every refill of k iterations and of the look-ahead behind them reads real input
        {
            const size_t k_out = (out_left - 320) / 260 + 1;
            if (k_out < k) k = k_out;
        }
        const bool far = (size_t)(out - out_begin) >= 32768;   // no distance reaches the start of the output any more
        SW_FINF_REFILL();
        uint64_t se = tb.sup[buf & IDX];
        for (; k; --k) {
            unsigned len;
            size_t dist;
            uint64_t se_next;
            if (__builtin_expect((se & S_ANY) != 0, 1)) {
                // the super entry: up to two literals (both bytes are stored whatever their number: what is not a literal is
                // overwritten by the next symbol) and / or a whole match, from ONE table load
                const uint16_t two = (uint16_t)(se >> 37);
                memcpy(out, &two, 2);
                out += (se >> 10) & 3u;
                const unsigned all = (unsigned)se & 63u, code = (unsigned)(se >> 6) & 0xFu;   // (all <= 11 + 13 of the >= 56 bits)
                const uint32_t extra = (uint32_t)(buf >> code) & ((1u << (all - code)) - 1u);
                buf >>= all;
                cnt -= all;
                // The next iteration's entry: looked up BEFORE the refill (>= 32 valid bits are left, 11 are needed), so that the loop's
                // dependency chain is table load -> shift -> mask -> table load, without the refill's shift and or; and before the
                // match is copied: the entry does not depend on the bytes a match copies, its load overlaps the copy.
                se_next = tb.sup[buf & IDX];
                SW_FINF_REFILL();
                if (!(se & S_MATCH)) {
                    se = se_next;
                    continue;
                }
                len = (unsigned)(se >> 13) & 0x1FFu;
                dist = ((size_t)(se >> 22) & 0x7FFFu) + extra;
            } else {
                uint32_t e = tb.ll[buf & IDX];
                if (e & K_LIT) {
                    // up to three literals straight from the first-level table (3 x 11 bits of the >= 56 in the buffer: every entry is
                    // found through valid bits, and a non-literal entry behind them is still good after the refill below)
                    buf >>= (e & 0xFFu);
                    cnt -= (e & 0xFFu);
                    *out++ = (uint8_t)(e >> 16);
                    e = tb.ll[buf & IDX];
                    if (e & K_LIT) {
                        buf >>= (e & 0xFFu);
                        cnt -= (e & 0xFFu);
                        *out++ = (uint8_t)(e >> 16);
                        e = tb.ll[buf & IDX];
                        if (e & K_LIT) {
                            buf >>= (e & 0xFFu);
                            cnt -= (e & 0xFFu);
                            *out++ = (uint8_t)(e >> 16);
                            goto next_entry;
                        }
                    }
                    SW_FINF_REFILL();   // a length / distance pair may need 48 bits (the bits `e` was found through stay where they are)
                }
                if (__builtin_expect(e & (K_SUB | K_EOB | K_BAD), 0)) {
                    if (e & K_SUB) {
                        buf >>= LL_BITS;
                        cnt -= LL_BITS;
                        e = tb.ll[(e >> 16) + (uint32_t)(buf & ((1u << ((e >> 8) & 0xFu)) - 1u))];
                        if (e & K_LIT) {
                            buf >>= (e & 0xFFu);
                            cnt -= (e & 0xFFu);
                            *out++ = (uint8_t)(e >> 16);
                            goto next_entry;
                        }
                    }
                    if (e & (K_EOB | K_BAD)) {
                        buf >>= (e & 0xFFu);
                        cnt -= (e & 0xFFu);
                        if (e & K_BAD) res = BAD;
                        else block_done = true;
                        goto done;
                    }
                }
                {
                    // length: code + extra bits in one step
                    const unsigned lc = e & 0xFFu, lx = (e >> 8) & 0xFu;
                    len = (e >> 16) + ((uint32_t)(buf >> lc) & ((1u << lx) - 1u));
                    buf >>= (lc + lx);
                    cnt -= (lc + lx);
                    uint32_t de = tb.d[buf & ((1u << D_BITS) - 1u)];
                    if (__builtin_expect(de & (K_SUB | K_BAD), 0)) {
                        if (de & K_SUB) {
                            buf >>= D_BITS;
                            cnt -= D_BITS;
                            de = tb.d[(de >> 16) + (uint32_t)(buf & ((1u << ((de >> 8) & 0xFu)) - 1u))];
                        }
                        if (de & K_BAD) {
                            res = BAD;
                            goto done;
                        }
                    }
                    const unsigned dc = de & 0xFFu, dx = (de >> 8) & 0xFu;
                    dist = (de >> 16) + ((uint32_t)(buf >> dc) & ((1u << dx) - 1u));
                    buf >>= (dc + dx);
                    cnt -= (dc + dx);
                }
                SW_FINF_REFILL();
                se_next = tb.sup[buf & IDX];
            }
            if (__builtin_expect(!far && dist > (size_t)(out - out_begin), 0)) {   // ("invalid distance too far back")
                res = BAD;
                goto done;
            }
            {
                const uint8_t *src = out - dist;
                uint8_t *dst = out;
                out += len;
                if (__builtin_expect(dist >= 16, 1)) {
                    struct W16 { uint64_t a, b; } w;   // (most matches of DNA text are 3 ... 16 bytes: one 16-byte move; spare bytes behind `out`)
                    memcpy(&w, src, 16);
                    memcpy(dst, &w, 16);
                    if (__builtin_expect(len > 16, 0)) {
                        do {
                            src += 16;
                            dst += 16;
                            memcpy(&w, src, 16);
                            memcpy(dst, &w, 16);
                        } while (dst + 16 < out);
                    }
                } else if (dist >= 8) {
                    uint64_t w;
                    do {
                        memcpy(&w, src, 8);
                        memcpy(dst, &w, 8);
                        src += 8;
                        dst += 8;
                    } while (dst < out);
                } else if (dist == 1) {
                    memset(dst, *src, len);
                } else {
                    do *dst++ = *src++; while (dst < out);
                }
            }
            se = se_next;
            continue;
        next_entry:
            SW_FINF_REFILL();
            se = tb.sup[buf & IDX];
        }
    }
done:
#undef SW_FINF_REFILL
    s.br.buf = buf;
    s.br.cnt = cnt;
    s.br.in = in;
    s.out = out;
    return res;
}

// ... and the careful loop for the ends of the buffers: until the block's end
inline Result careful_loop(Stream &s)
{
    BitReader &br = s.br;
    uint8_t *&out = s.out;
    uint8_t *const out_begin = s.out_begin, *const out_cap = s.out_cap;
    const Tables &tb = *s.tb;
    const bool block_done = false;
    {
        // ---- ... and the careful loop for the ends of the buffers ----
        while (!block_done) {
            br.refill();
            if (br.overrun()) return BAD;                       // (decoding zero padding: the stream is truncated)
            uint32_t e = tb.ll[br.peek(LL_BITS)];
            if (e & K_SUB) {
                br.drop(LL_BITS);
                e = tb.ll[(e >> 16) + br.peek((e >> 8) & 0xFu)];
            }
            if (e & K_LIT) {   // up to three literals per refill (3 x 15 bits <= 56)
                if ((size_t)(out_cap - out) < 3) {   // the last bytes of the buffer: one literal at a time
                    if (out == out_cap) return NEED_OUT;
                    br.drop(e & 0xFFu);
                    *out++ = (uint8_t)(e >> 16);
                    continue;
                }
                br.drop(e & 0xFFu);
                *out++ = (uint8_t)(e >> 16);
                e = tb.ll[br.peek(LL_BITS)];
                if (!(e & K_LIT)) continue;   // (anything but a first-level literal: next iteration, after a refill)
                br.drop(e & 0xFFu);
                *out++ = (uint8_t)(e >> 16);
                e = tb.ll[br.peek(LL_BITS)];
                if (!(e & K_LIT)) continue;
                br.drop(e & 0xFFu);
                *out++ = (uint8_t)(e >> 16);
                continue;
            }
            br.drop(e & 0xFFu);
            if (e & (K_EOB | K_BAD)) {
                if (e & K_BAD) return BAD;
                break;
            }
            const unsigned lx = (e >> 8) & 0xFu;
            const unsigned len = (e >> 16) + br.peek(lx);
            br.drop(lx);
            if (!tb.d_usable) return BAD;
            uint32_t de = tb.d[br.peek(D_BITS)];
            if (de & K_SUB) {
                br.drop(D_BITS);
                de = tb.d[(de >> 16) + br.peek((de >> 8) & 0xFu)];
            }
            if (de & K_BAD) return BAD;
            br.drop(de & 0xFFu);
            const unsigned dx = (de >> 8) & 0xFu;
            const size_t dist = (de >> 16) + br.peek(dx);
            br.drop(dx);
            if (dist > (size_t)(out - out_begin)) return BAD;   // before the start of the output ("invalid distance too far back")
            if ((size_t)(out_cap - out) < (size_t)len + 8) {    // near the end of the buffer: exact copy, or more room
                if ((size_t)(out_cap - out) < len) return NEED_OUT;
                for (unsigned i = 0; i < len; ++i) out[i] = out[i - (ptrdiff_t)dist];
                out += len;
                continue;
            }
            const uint8_t *src = out - dist;
            uint8_t *dst = out;
            out += len;
            if (dist >= 8) {
                do {
                    uint64_t w;
                    memcpy(&w, src, 8);
                    memcpy(dst, &w, 8);
                    src += 8;
                    dst += 8;
                } while (dst < out);
            } else if (dist == 1) {
                memset(dst, *src, len);
            } else {
                do *dst++ = *src++; while (dst < out);
            }
        }
    }
    return OK;
}

inline Result close_block(Stream &s)
{
    if (s.br.overrun()) return BAD;
    s.in_block = false;
    if (s.final_block) s.ended = true;
    return OK;
}

// from wherever the stream stands to its end
inline Result run_stream(Stream &s)
{
    for (;;) {
        if (!s.in_block) {
            if (s.ended) return OK;
            const Result r = open_block(s);
            if (r != OK) return r;
            if (s.ended) return OK;
        }
        bool block_done = false;
        Result r = fast_loop(s, block_done);
        if (r != OK) return r;
        if (!block_done) {
            r = careful_loop(s);
            if (r != OK) return r;
        }
        r = close_block(s);
        if (r != OK) return r;
    }
}

// One DEFLATE stream: br at its first bit; out_begin..out_cap the output buffer, `out` the write position (history = [out_begin, out)).
inline Result inflate_stream(BitReader &br, uint8_t *out_begin, uint8_t *&out, uint8_t *out_cap, Tables &T)
{
    Stream s;
    s.br = br;
    s.out_begin = out_begin;
    s.out = out;
    s.out_cap = out_cap;
    s.T = &T;
    const Result r = run_stream(s);
    br = s.br;
    out = s.out;
    return r;
}

// ---- CRC-32 (IEEE, reflected: the gzip trailer's) ---------------------------------------------------------------------------------
// Slicing-by-8 tables built at first use; with PCLMULQDQ the bulk is folded 64 bytes at a time (Gopal et al., "Fast CRC
// Computation for Generic Polynomials Using PCLMULQDQ Instruction"; constants for the reflected polynomial 0xEDB88320).  Both are
// checked against each other -- and by the tests against zlib's crc32 -- before the folded one is trusted.
struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
    }
};
inline const CrcTables &crc_tables()
{
    static const CrcTables T;
    return T;
}
inline uint32_t crc32_slice8(uint32_t crc, const uint8_t *p, size_t n)   // crc: running value, pre/post-inverted by the caller
{
    const CrcTables &T = crc_tables();
    while (n && ((uintptr_t)p & 7u)) {
        crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xFFu];
        --n;
    }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= crc;
        crc = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^ T.t[3][(w >> 32) & 0xFF] ^
              T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][(w >> 56) & 0xFF];
        p += 8;
        n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xFFu];
    return crc;
}

#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_pclmul(uint32_t crc, const uint8_t *p, size_t n)   // n >= 64, n % 16 == 0
{
    // fold constants x^(512+32) mod P ... (reflected domain), as in the Linux kernel's crc32-pclmul and zlib-ng
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000000000ll, 0x0000000163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x00000001f7011641ll, 0x00000001db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16)),
            x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64;
    n -= 64;
    while (n >= 64) {   // fold by 4 x 128 bits
        __m128i t1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), t2 = _mm_clmulepi64_si128(x2, k1k2, 0x00),
                t3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), t4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, t1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, t2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, t3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, t4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        n -= 64;
    }
    // fold the four into one (a lambda would not inherit the target attribute)
#define SW_FINF_FOLD1(a, b) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128((a), k3k4, 0x11), _mm_clmulepi64_si128((a), k3k4, 0x00)), (b))
    x1 = SW_FINF_FOLD1(x1, x2);
    x1 = SW_FINF_FOLD1(x1, x3);
    x1 = SW_FINF_FOLD1(x1, x4);
    while (n >= 16) {
        const __m128i nx = _mm_loadu_si128((const __m128i *)p);
        x1 = SW_FINF_FOLD1(x1, nx);
        p += 16;
        n -= 16;
    }
#undef SW_FINF_FOLD1
    // 128 -> 64 bits
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5k0, 0x00), t);
    // Barrett reduction 64 -> 32 bits
    t = _mm_and_si128(x1, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// 1: the folded form agrees with the tables on this CPU, 2: unavailable / disagrees (examined once: a function-local static)
inline int crc_fold_state()
{
#if defined(__x86_64__)
    static const int state = [] {
        if (!(__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"))) return 2;
        uint8_t probe[64 * 5 + 16];
        uint32_t x = 0x2545F491u;
        for (size_t i = 0; i < sizeof probe; ++i) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            probe[i] = (uint8_t)x;
        }
        bool ok = true;
        for (size_t len : {(size_t)80, (size_t)96, (size_t)128, (size_t)336})
            ok = ok && crc32_pclmul(0xFFFFFFFFu, probe, len) == crc32_slice8(0xFFFFFFFFu, probe, len) &&
                 crc32_pclmul(0x12345678u, probe + 3, len - 16) == crc32_slice8(0x12345678u, probe + 3, len - 16);
        return ok ? 1 : 2;
    }();
    return state;
#else
    return 2;
#endif
}
inline uint32_t crc32(const uint8_t *p, size_t n)   // the gzip trailer's CRC-32 of p[0, n)
{
    uint32_t crc = 0xFFFFFFFFu;
#if defined(__x86_64__)
    if (n >= 64 && crc_fold_state() == 1) {
        const size_t bulk = n & ~(size_t)15;
        crc = crc32_pclmul(crc, p, bulk);
        p += bulk;
        n -= bulk;
    }
#endif
    return ~crc32_slice8(crc, p, n);
}

// A gzip member's header at p (RFC 1952): the position of its DEFLATE stream, or nullptr for what is left to zlib
inline const uint8_t *gzip_header(const uint8_t *p, const uint8_t *end)
{
    if (end - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return nullptr;
    const unsigned flg = p[3];
    if (flg & 0xE0u) return nullptr;      // reserved bits
    p += 10;
    if (flg & 4u) {                       // FEXTRA
        if (end - p < 2) return nullptr;
        const unsigned xl = p[0] | (p[1] << 8);
        p += 2;
        if ((size_t)(end - p) < xl) return nullptr;
        p += xl;
    }
    for (unsigned bit : {8u, 16u})        // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
            if (!z) return nullptr;
            p = z + 1;
        }
    if (flg & 2u) return nullptr;         // FHCRC: zlib checks the header CRC -- rare enough to leave to it
    if (end - p < 8) return nullptr;
    return p;
}
// ... and its trailer at p behind the stream: CRC-32 and ISIZE of member_begin[0, n_out)
inline bool gzip_trailer_ok(const uint8_t *p, const uint8_t *end, const uint8_t *member_begin, size_t n_out)
{
    if (end - p < 8) return false;
    const uint32_t want_crc = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    const uint32_t isize = p[4] | (p[5] << 8) | (p[6] << 16) | ((uint32_t)p[7] << 24);
    return (uint32_t)n_out == isize && crc32(member_begin, n_out) == want_crc;
}

// One or more gzip members in in[0, n) (followed by IN_PAD readable zero bytes) -> out[0, *out_len), at most out_cap bytes.
// OK: every member well-formed, its CRC-32 and ISIZE right, nothing behind the last one.  NEED_OUT: out_cap is too small.
inline Result gunzip_members(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len, Tables &T)
{
    const uint8_t *p = in, *end = in + n;
    uint8_t *o = out;
    if (n < 18) return BAD;
    while (p < end) {
        p = gzip_header(p, end);
        if (!p) return BAD;
        BitReader br;
        br.in = p;
        br.in_end = end;
        uint8_t *member_begin = o;
        const Result r = inflate_stream(br, member_begin, o, out + out_cap, T);
        if (r != OK) return r;
        p = br.position();
        if (!gzip_trailer_ok(p, end, member_begin, (size_t)(o - member_begin))) return BAD;
        p += 8;
    }
    *out_len = (size_t)(o - out);
    return OK;
}

}  // namespace finf
}  // namespace sw
