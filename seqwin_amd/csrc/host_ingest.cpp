// host_ingest.cpp -- FASTA / gzip-FASTA reader and 2-bit packer (host side of the hot path).
//
// Replaces seqwin::internal::read_fasta (reference cpp/src/utils/fasta_reader.cpp:207-213; core
// :41-95, gzip :109-203, id extraction :26-33) and feeds the GPU: instead of std::string records it
// produces the 2-bit packed stream + valid-run table described in common.hpp.  Record boundaries,
// ids and base coordinates are identical to the reference's: a line is what std::getline returns,
// one trailing '\r' is dropped, empty / whitespace-only lines are skipped, a line whose first byte is
// '>' opens a record whose id is the header up to the first whitespace, every other line
// contributes its non-whitespace bytes (case kept by the reference; here folded into the 2-bit code,
// which is all the hash ever looks at: SEED_TAB, cpp/vendor/btllib/hashing_internals.hpp:136-169).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sched.h>
#include <zlib.h>

#include "fast_inflate.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "common.hpp"

namespace sw {

void check_kw(uint64_t k, uint64_t w)
{
    // k < 3 crashes the reference (unsigned k-3, nthash_kmer.hpp:26); k is truncated to uint16
    // there (hashing_internals.hpp:10).  Both are refused here.
    if (k < 3) raise(SW_ERR_VALUE, "kmerlen must be >= 3 (got %llu)", (unsigned long long)k);
    if (k > 65535) raise(SW_ERR_VALUE, "kmerlen must be <= 65535 (got %llu)", (unsigned long long)k);
    if (w < 1) raise(SW_ERR_VALUE, "windowsize must be >= 1 (got %llu)", (unsigned long long)w);
    // (any w: windows above SW_MAX_WINDOW take the two-step route of index.hip's select_large_windows)
}

namespace {

// 0..3 = A C G T/U (either case); 4 = invalid base (SEED_N); 5 = whitespace; 6 = refused control byte
// (0x01 0x03 0x04 0x05 0x07: SEED_TAB says valid, CONVERT_TAB says 255 -- see DESIGN.md).
struct CharTable {
    uint8_t t[256];
    CharTable()
    {
        for (int i = 0; i < 256; ++i) t[i] = 4;
        t['A'] = t['a'] = 0;
        t['C'] = t['c'] = 1;
        t['G'] = t['g'] = 2;
        t['T'] = t['t'] = t['U'] = t['u'] = 3;
        for (unsigned char c : {' ', '\t', '\n', '\r', '\f', '\v'}) t[c] = 5;
        for (unsigned char c : {1, 3, 4, 5, 7}) t[c] = 6;
    }
};
const CharTable kChar;

bool ends_with(const std::string &s, const char *suf)
{
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// A worker's reusable file buffer: grown with realloc, never value-initialised (a std::vector<char> zero-fills what it
// grows by -- 8 MB of stores per 5 MB file in front of the read that overwrites them).
struct RawBuf {
    char *p = nullptr;
    size_t cap = 0, len = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        const size_t ncap = std::max(n, cap + cap / 2);
        char *q = (char *)realloc(p, ncap);
        if (!q) raise(SW_ERR_RUNTIME, "out of memory reading a FASTA file (%zu bytes)", ncap);
        p = q;
        cap = ncap;
    }
};

// The bytes of a file, as they are, + finf::IN_PAD zero bytes behind them (buf.len excludes the padding).
bool read_raw(const std::string &path, RawBuf &buf)
{
    buf.len = 0;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    const size_t expect = (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) ? (size_t)st.st_size : 0;
    bool ok = true;
    for (;;) {
        try {
            buf.reserve(std::max<size_t>(buf.len + (1u << 20), expect + 1) + finf::IN_PAD);
        } catch (...) {
            close(fd);
            throw;
        }
        const ssize_t got = read(fd, buf.p + buf.len, buf.cap - finf::IN_PAD - buf.len);
        if (got < 0) {
            if (errno == EINTR) continue;
            ok = false;
            break;
        }
        if (got == 0) break;
        buf.len += (size_t)got;
    }
    close(fd);
    if (ok) memset(buf.p + buf.len, 0, finf::IN_PAD);
    return ok;
}

// r05: a .gz file that is one or more well-formed gzip members is inflated by fast_inflate.hpp from the whole file in memory
// (a second per-worker buffer) -- the bytes zlib's gzread loop would deliver, CRC-32 and ISIZE checked; anything else
// (and SEQWIN_AMD_ZLIB_INFLATE=1) takes the gzread loop below, which is the reference's (fasta_reader.cpp:109-203).
bool slurp_gz_fast(const std::string &path, RawBuf &buf)
{
    static thread_local RawBuf raw;   // the compressed bytes: one buffer per worker thread, reused from file to file
    static const bool off = getenv("SEQWIN_AMD_ZLIB_INFLATE") != nullptr;
    if (off || !read_raw(path, raw) || raw.len < 18) return false;
    const uint8_t *in = (const uint8_t *)raw.p;
    const uint32_t isize = in[raw.len - 4] | (in[raw.len - 3] << 8) | (in[raw.len - 2] << 16) | ((uint32_t)in[raw.len - 1] << 24);
    static thread_local finf::Tables tables;
    // ISIZE of the last member is exact for the one-member files NCBI ships -- but it is four bytes anyone can write: the first
    // reservation trusts it only up to 8 x the compressed size (FASTA deflates 3-5 x); a stream that really expands more finds
    // its room by doubling (8 attempts: 2 048 x, above DEFLATE's 1 032 x)
    size_t cap = std::min<size_t>(std::max<size_t>((size_t)isize, raw.len), raw.len * 8 + 4096) + 64;
    // BGZF (bgzip: many members, each with its block size in a 'BC' extra subfield; the last one is empty, so its ISIZE says 0):
    // the members' ISIZEs are summed by hopping from block to block -- the exact size, one attempt (sized from the
    // last member alone such a file was inflated three or four times over)
    {
        size_t off = 0, total = 0;
        bool bgzf = raw.len >= 28;
        while (bgzf && off < raw.len) {
            if (raw.len - off < 18 || in[off] != 0x1f || in[off + 1] != 0x8b || in[off + 2] != 8 || !(in[off + 3] & 4)) { bgzf = false; break; }
            const size_t xlen = in[off + 10] | (in[off + 11] << 8);
            if (raw.len - off < 12 + xlen + 8) { bgzf = false; break; }
            size_t bsize = 0;
            for (size_t x = off + 12, xe = off + 12 + xlen; x + 4 <= xe;) {
                const size_t slen = in[x + 2] | (in[x + 3] << 8);
                if (in[x] == 'B' && in[x + 1] == 'C' && slen == 2 && x + 6 <= xe) bsize = (size_t)(in[x + 4] | (in[x + 5] << 8)) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || bsize > raw.len - off) { bgzf = false; break; }
            const uint8_t *t = in + off + bsize - 4;
            total += (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
            off += bsize;
        }
        if (bgzf && total <= raw.len * 1040 + 4096) cap = total + 64;   // (DEFLATE expands at most 1 032 x: a larger claim is a lie)
    }
    for (int attempt = 0; attempt < 8; ++attempt) {
        buf.len = 0;
        try {
            buf.reserve(cap);
        } catch (...) {
            return false;   // (a stream that claims more room than the host has: the gzread loop grows with the REAL output and raises what it raises)
        }
        size_t got = 0;
        const finf::Result r = finf::gunzip_members(in, raw.len, (uint8_t *)buf.p, cap, &got, tables);
        if (r == finf::OK) {
            buf.len = got;
            return true;
        }
        if (r != finf::NEED_OUT) return false;
        cap *= 2;   // (several members, or more than 4 GiB of text: ISIZE does not say)
    }
    return false;
}

void slurp(const std::string &path, RawBuf &buf)
{
    buf.len = 0;
    if (ends_with(path, ".gz")) {  // fasta_reader.cpp:209
        if (slurp_gz_fast(path, buf)) return;
        buf.len = 0;
        gzFile gz = gzopen(path.c_str(), "rb");
        if (!gz) raise(SW_ERR_RUNTIME, "Unable to open gzip FASTA: %s", path.c_str());
        gzbuffer(gz, 1u << 20);
        for (;;) {
            try {
                buf.reserve(std::max<size_t>(buf.len + (1u << 20), 1u << 22));
            } catch (...) {
                gzclose(gz);
                throw;
            }
            int got = gzread(gz, buf.p + buf.len, 1u << 20);
            if (got < 0) {
                int errnum = 0;
                const char *e = gzerror(gz, &errnum);
                std::string msg = std::string("gzip read error: ") + (e ? e : "unknown");
                gzclose(gz);
                raise(SW_ERR_RUNTIME, "%s", msg.c_str());
            }
            if (got == 0) break;
            buf.len += (size_t)got;
        }
        gzclose(gz);
    } else {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) raise(SW_ERR_RUNTIME, "Unable to open FASTA: %s", path.c_str());   // fasta_reader.cpp:100-102
        struct stat st;
        size_t expect = (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) ? (size_t)st.st_size : 0;
        for (;;) {   // (the size is a hint: the file is read until read() returns 0)
            try {
                buf.reserve(std::max<size_t>(buf.len + (1u << 20), expect + 1));
            } catch (...) {
                close(fd);
                throw;
            }
            const ssize_t got = read(fd, buf.p + buf.len, buf.cap - buf.len);
            if (got < 0) {
                if (errno == EINTR) continue;
                close(fd);
                raise(SW_ERR_RUNTIME, "Unable to read FASTA: %s", path.c_str());
            }
            if (got == 0) break;
            buf.len += (size_t)got;
        }
        close(fd);
    }
}

// One assembly, parsed and packed by one worker.
struct alignas(128) Assembly {   // neighbours are filled by different workers: no shared cache lines
    std::vector<uint32_t> rec_len;
    std::vector<uint64_t> rec_base;  // local (within this assembly's packed stream)
    std::vector<uint32_t> rec_run_off;
    std::vector<uint32_t> run_pos, run_len;
    WordBuf packed;                  // 32 bases per word; private to the worker until the assembly is published
    uint64_t n_words = 0;            // packed.size() at publication (packed itself may have gone to the sink)
    std::string ids;
    uint64_t total_bp = 0;
};

struct Packer {
    Assembly &a;
    WordBuf &words;                 // a local of the parsing function while packing (moved into a.packed at the end)
    uint64_t acc = 0;
    unsigned nacc = 0;     // bases in acc
    uint64_t len = 0;      // bases in the current record
    int64_t run_start = -1;
    Packer(Assembly &as, WordBuf &w) : a(as), words(w) {}

    void open_record()
    {
        a.rec_base.push_back((uint64_t)words.size() * 32);
        a.rec_run_off.push_back((uint32_t)a.run_pos.size());
        acc = 0;
        nacc = 0;
        len = 0;
        run_start = -1;
    }
    void close_record(const std::string &path, const std::string &id)
    {
        if (run_start >= 0) {
            a.run_pos.push_back((uint32_t)run_start);
            a.run_len.push_back((uint32_t)(len - (uint64_t)run_start));
        }
        if (nacc) words.push_back(acc);
        if (len > UINT32_MAX)  // build.cpp:143-147
            raise(SW_ERR_RUNTIME, "Sequence length exceeds uint32 range for record %s in assembly %s",
                  id.c_str(), path.c_str());
        a.rec_len.push_back((uint32_t)len);
        a.total_bp += len;
    }
    // nb <= 32 bases at once: codes = 2 bits per base (zero above 2 nb bits and at invalid bases), valid = 1 bit per base
    inline void push_block(uint64_t codes, unsigned nb, uint32_t valid)
    {
        const uint32_t full = nb == 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
        valid &= full;
        if (valid == full) {
            if (run_start < 0) run_start = (int64_t)len;
        } else {
            unsigned pos = 0;
            while (pos < nb) {
                if (run_start < 0) {
                    const uint32_t t = valid >> pos;
                    if (!t) break;
                    pos += (unsigned)__builtin_ctz(t);
                    run_start = (int64_t)(len + pos);
                } else {
                    const uint32_t t = (~valid & full) >> pos;
                    if (!t) break;
                    pos += (unsigned)__builtin_ctz(t);
                    a.run_pos.push_back((uint32_t)run_start);
                    a.run_len.push_back((uint32_t)(len + pos - (uint64_t)run_start));
                    run_start = -1;
                }
            }
        }
        len += nb;
        const unsigned total = nacc + nb;
        if (nacc == 0) {
            acc = codes;
        } else {
            acc |= codes << (2 * nacc);
        }
        if (total >= 32) {
            words.push_back(acc);
            acc = nacc ? (codes >> (2 * (32 - nacc))) : 0;
            nacc = total - 32;
        } else {
            nacc = total;
        }
    }
    // nb <= 64 bases at once (r06; the 64-byte chunks of pack_chunks_avx512): lo / hi = 2 bits per base of bases 0..31 / 32..63, zero
    // above 2 nb bits.  A chunk of valid bases only -- the rule -- is appended to the accumulator in one step and leaves as up to two
    // words; anything else takes the 32-base blocks above.
    inline void push_chunk(uint64_t lo, uint64_t hi, unsigned nb, uint64_t valid)
    {
        const uint64_t full = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
        if (nb == 0 || (valid & full) != full) {
            if (nb >= 32) {
                push_block(lo, 32, (uint32_t)valid);
                if (nb > 32) push_block(hi, nb - 32, (uint32_t)(valid >> 32));
            } else if (nb) {
                push_block(lo, nb, (uint32_t)valid);
            }
            return;
        }
        if (run_start < 0) run_start = (int64_t)len;
        len += nb;
        const unsigned s = 2 * nacc;   // 0 .. 62
        uint64_t w0, w1, w2;
        if (s == 0) {
            w0 = lo;
            w1 = hi;
            w2 = 0;
        } else {
            w0 = acc | (lo << s);
            w1 = (lo >> (64 - s)) | (hi << s);
            w2 = hi >> (64 - s);
        }
        const unsigned total = nacc + nb;   // < 96
        if (total >= 64) {
            words.push_back2(w0, w1);
            acc = w2;
            nacc = total - 64;
        } else if (total >= 32) {
            words.push_back(w0);
            acc = w1;
            nacc = total - 32;
        } else {
            acc = w0;
            nacc = total;
        }
    }
    inline void push(unsigned code)
    {
        if (code < 4) {
            if (run_start < 0) run_start = (int64_t)len;
            acc |= (uint64_t)code << (2 * nacc);
        } else if (run_start >= 0) {
            a.run_pos.push_back((uint32_t)run_start);
            a.run_len.push_back((uint32_t)(len - (uint64_t)run_start));
            run_start = -1;
        }
        ++len;
        if (++nacc == 32) {
            words.push_back(acc);
            acc = 0;
            nacc = 0;
        }
    }
};

#if defined(__x86_64__)
#define SW_HAVE_AVX2_PACKER 1
// 32 sequence bytes -> 2-bit codes, validity mask, and a mask of bytes <= 0x20 (whitespace / control: the caller
// falls back to the byte loop for such a chunk).  A/C/G/T/U in either case are told apart by their low nibble.
__attribute__((target("avx2,bmi2"))) inline void classify32(const char *q, uint64_t &codes, uint32_t &valid, uint32_t &special)
{
    const __m256i v = _mm256_loadu_si256((const __m256i *)q);
    const __m256i lo = _mm256_and_si256(v, _mm256_set1_epi8(0x0F));
    // low nibble -> expected upper-case letter (0xFF: none) and 2-bit code
    const __m256i exp_lut = _mm256_setr_epi8((char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', (char)0xFF, (char)0xFF,
                                             (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF,
                                             (char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', (char)0xFF, (char)0xFF,
                                             (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF);
    const __m256i code_lut = _mm256_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i up = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
    const __m256i ok = _mm256_cmpeq_epi8(up, _mm256_shuffle_epi8(exp_lut, lo));
    const __m256i code = _mm256_and_si256(_mm256_shuffle_epi8(code_lut, lo), ok);
    valid = (uint32_t)_mm256_movemask_epi8(ok);
    const __m256i c20 = _mm256_set1_epi8(0x20);
    special = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(v, c20), c20));
    const uint64_t m = 0x0303030303030303ull;
    codes = _pext_u64((uint64_t)_mm256_extract_epi64(code, 0), m) | (_pext_u64((uint64_t)_mm256_extract_epi64(code, 1), m) << 16) |
            (_pext_u64((uint64_t)_mm256_extract_epi64(code, 2), m) << 32) | (_pext_u64((uint64_t)_mm256_extract_epi64(code, 3), m) << 48);
}

// One sequence line [ls, le).  Returns false when a chunk holds whitespace / control bytes at `*stop` (byte loop takes over).
__attribute__((target("avx2,bmi2"))) const char *pack_line_avx2(Packer &pk, const char *ls, const char *le)
{
    const char *q = ls;
    uint64_t codes;
    uint32_t valid, special;
    while (le - q >= 32) {
        classify32(q, codes, valid, special);
        if (special) return q;
        pk.push_block(codes, 32, valid);
        q += 32;
    }
    const unsigned tail = (unsigned)(le - q);
    if (tail) {
        alignas(32) char tmp[32];
        memset(tmp, 'A', 32);
        memcpy(tmp, q, tail);
        classify32(tmp, codes, valid, special);
        if (special) return q;
        pk.push_block(tail == 32 ? codes : (codes & ((1ull << (2 * tail)) - 1ull)), tail, valid);
        q = le;
    }
    return q;
}

// The same with 64 bytes per step (AVX-512 BW): validity and blanks come out as mask registers, four 2-bit codes are folded
// into a byte by two multiply-adds (c0 + 4 c1, then + 16 (c2 + 4 c3)) and the dwords narrowed to bytes -- no extracts, no pext;
// a line's tail is a masked load (a 60- or 80-column line is one or two steps).
__attribute__((target("avx512f,avx512bw"))) inline void classify64(const __m512i v, uint64_t codes[2], uint64_t &valid, uint64_t &special)
{
    const __m512i lo = _mm512_and_si512(v, _mm512_set1_epi8(0x0F));
    const __m512i exp_lut = _mm512_broadcast_i32x4(_mm_setr_epi8((char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', (char)0xFF,
                                                                 (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF,
                                                                 (char)0xFF));
    const __m512i code_lut = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0));
    const __m512i up = _mm512_and_si512(v, _mm512_set1_epi8((char)0xDF));
    const __mmask64 ok = _mm512_cmpeq_epi8_mask(up, _mm512_shuffle_epi8(exp_lut, lo));
    const __m512i code = _mm512_maskz_shuffle_epi8(ok, code_lut, lo);
    valid = (uint64_t)ok;
    special = (uint64_t)_mm512_cmple_epu8_mask(v, _mm512_set1_epi8(0x20));
    const __m512i t16 = _mm512_maddubs_epi16(code, _mm512_set1_epi16(0x0401));
    const __m512i t32 = _mm512_madd_epi16(t16, _mm512_set1_epi32(0x00100001));
    const __m128i packed = _mm512_cvtepi32_epi8(t32);   // byte j = the codes of bases 4 j .. 4 j + 3
    codes[0] = (uint64_t)_mm_extract_epi64(packed, 0);
    codes[1] = (uint64_t)_mm_extract_epi64(packed, 1);
}

__attribute__((target("avx512f,avx512bw"))) const char *pack_line_avx512(Packer &pk, const char *ls, const char *le)
{
    const char *q = ls;
    uint64_t codes[2], valid, special;
    while (q < le) {
        const size_t left = (size_t)(le - q);
        const unsigned nb = left >= 64 ? 64u : (unsigned)left;
        const __mmask64 m = nb == 64 ? ~(__mmask64)0 : (((__mmask64)1 << nb) - 1);
        classify64(nb == 64 ? _mm512_loadu_si512((const void *)q) : _mm512_maskz_loadu_epi8(m, (const void *)q), codes, valid, special);
        if (special & (uint64_t)m) return q;   // whitespace / control bytes in the chunk: the byte loop takes over here
        pk.push_block(codes[0], nb >= 32 ? 32u : nb, (uint32_t)valid);
        if (nb > 32) pk.push_block(codes[1], nb - 32, (uint32_t)(valid >> 32));
        q += nb;
    }
    return q;
}

// r05: sequence text 64 bytes at a time ACROSS line ends (AVX-512 VBMI2): a chunk that holds nothing but sequence bytes and '\n' has
// its line ends squeezed out in the register (vpcompressb on the codes, pext on the validity mask) and goes to the packer as one
// or two blocks -- no memchr, no per-line tail, no third block per 80-column line.  Stops in front of the first chunk that holds
// anything else (a blank or control byte other than '\n', or a '>', which may start a header: the line code below takes that
// line, from where this stopped) or when fewer than 64 bytes are left.
__attribute__((target("avx512f,avx512bw,avx512vbmi,avx512vbmi2,bmi2,popcnt"))) const char *pack_chunks_avx512(Packer &pk, const char *p, const char *end)
{
    const __m512i exp_lut = _mm512_broadcast_i32x4(_mm_setr_epi8((char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', (char)0xFF,
                                                                 (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF,
                                                                 (char)0xFF));
    const __m512i code_lut = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0));
    while (end - p >= 64) {
        const __m512i v = _mm512_loadu_si512((const void *)p);
        const __mmask64 nl = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('\n'));
        const __mmask64 stop = (_mm512_cmple_epu8_mask(v, _mm512_set1_epi8(0x20)) & ~nl) | _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('>'));
        if (stop) break;
        const __m512i lo = _mm512_and_si512(v, _mm512_set1_epi8(0x0F));
        const __m512i up = _mm512_and_si512(v, _mm512_set1_epi8((char)0xDF));
        const __mmask64 ok = _mm512_cmpeq_epi8_mask(up, _mm512_shuffle_epi8(exp_lut, lo));
        const __mmask64 keep = ~nl;
        const uint64_t nlm = (uint64_t)nl;
        const __m512i raw = _mm512_maskz_shuffle_epi8(ok, code_lut, lo);
        __m512i code;
        // r06: a chunk of a file with lines of 64 columns or more holds at most ONE line end, and taking one byte out of the vector is a
        // byte permutation -- idx[i] = i + (i >= j), one vpermb -- where vpcompressb is microcoded on the Zen cores of the target host
        if (nlm == 0) {
            code = raw;
        } else if ((nlm & (nlm - 1)) == 0) {
            const __m512i iota = _mm512_set_epi8(63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47, 46, 45, 44, 43, 42, 41, 40, 39, 38, 37, 36, 35,
                                                 34, 33, 32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6,
                                                 5, 4, 3, 2, 1, 0);
            const __m512i idx = _mm512_mask_add_epi8(iota, (__mmask64)~(nlm - 1), iota, _mm512_set1_epi8(1));   // + 1 from the line end on
            code = _mm512_maskz_permutexvar_epi8((__mmask64)(~0ull >> 1), idx, raw);                             // (zero behind the last base)
        } else {
            code = _mm512_maskz_compress_epi8(keep, raw);   // (zero behind the last base)
        }
        const unsigned nb = (unsigned)__builtin_popcountll((uint64_t)keep);
        const uint64_t valid = _pext_u64((uint64_t)ok, (uint64_t)keep);
        const __m512i t16 = _mm512_maddubs_epi16(code, _mm512_set1_epi16(0x0401));
        const __m512i t32 = _mm512_madd_epi16(t16, _mm512_set1_epi32(0x00100001));
        const __m128i packed = _mm512_cvtepi32_epi8(t32);   // byte j = the codes of bases 4 j .. 4 j + 3
        pk.push_chunk((uint64_t)_mm_extract_epi64(packed, 0), (uint64_t)_mm_extract_epi64(packed, 1), nb, valid);
        p += 64;
    }
    return p;
}

bool have_chunk_packer()
{
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi") &&
                           __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt") && !SW_TEST_GETENV("SEQWIN_AMD_SCALAR_INGEST") &&
                           !SW_TEST_GETENV("SEQWIN_AMD_NO_AVX512") && !SW_TEST_GETENV("SEQWIN_AMD_LINE_PACKER");
    return ok;
}

bool have_avx512_packer()
{
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                           !SW_TEST_GETENV("SEQWIN_AMD_SCALAR_INGEST") && !SW_TEST_GETENV("SEQWIN_AMD_NO_AVX512");
    return ok;
}

bool have_avx2_packer()
{
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !SW_TEST_GETENV("SEQWIN_AMD_SCALAR_INGEST");
    return ok;
}
#endif

// Bytes of one input file: read into the worker's reusable buffer, or (single worker, plain regular file) mapped.
struct FileBytes {
    const char *p = nullptr;
    size_t n = 0;
    void *map = nullptr;
    // use_mmap: see ingest_fasta (r05: also with several workers)
    FileBytes(const std::string &path, RawBuf &buf, int use_mmap)   // 0: read(), 1: mmap, 2: mmap + MAP_POPULATE
    {
        if (use_mmap && !ends_with(path, ".gz")) {
            const int fd = open(path.c_str(), O_RDONLY);
            if (fd < 0) raise(SW_ERR_RUNTIME, "Unable to open FASTA: %s", path.c_str());   // fasta_reader.cpp:100-102
            struct stat st;
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | (use_mmap > 1 ? MAP_POPULATE : 0), fd, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    map = m;
                    p = (const char *)m;
                    n = (size_t)st.st_size;
                }
            }
            close(fd);
            if (map) return;
        }
        slurp(path, buf);
        p = buf.p;
        n = buf.len;
    }
    ~FileBytes()
    {
        if (map) munmap(map, n);
    }
    FileBytes(const FileBytes &) = delete;
    FileBytes &operator=(const FileBytes &) = delete;
};

// The reader's line loop (fasta_reader.cpp:41-95) + the packer, fed with the text of one assembly in one piece or in several:
// every piece ends behind a '\n', except the last one of the file.
struct TextParser {
    const std::string &path;
    Assembly &a;
    WordBuf words;
    Packer pk;
    bool have = false;
    std::string cur_id;
#ifdef SW_HAVE_AVX2_PACKER
    const bool simd512 = have_avx512_packer(), simd = !simd512 && have_avx2_packer(), chunks = have_chunk_packer();
#endif
    TextParser(const std::string &path_, Assembly &a_, WordBuf &&storage) : path(path_), a(a_), words(std::move(storage)), pk(a_, words)
    {
        words.clear();
    }
    void expect_bytes(size_t n) { words.reserve(n / 32 + 64); }

    void feed(const char *p, const char *end)
    {
        bool midline = false;   // p is inside a sequence line whose head pack_chunks_avx512 has taken
        while (p < end) {
#ifdef SW_HAVE_AVX2_PACKER
            if (chunks && have && end - p >= 64) {
                const char *q = pack_chunks_avx512(pk, p, end);
                if (q != p) {
                    midline = q[-1] != '\n';
                    p = q;
                    if (p >= end) break;
                }
            }
#endif
            const bool rest_of_line = midline;   // (then a '>' at its front is a sequence byte, not a header's)
            midline = false;
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *ls = p, *le = nl ? nl : end;
            p = nl ? nl + 1 : end;
            if (le > ls && le[-1] == '\r') --le;  // fasta_reader.cpp:51-53
            // empty or whitespace-only (:55-57): cheap test on the first byte, full scan only if it is ws
            if (le == ls) continue;
            if (kChar.t[(unsigned char)*ls] == 5) {
                const char *q = ls;
                while (q < le && kChar.t[(unsigned char)*q] == 5) ++q;
                if (q == le) continue;
            }
            if (*ls == '>' && !rest_of_line) {  // :58-67
                if (have) pk.close_record(path, cur_id);
                const char *ie = ls + 1;  // extract_id :26-33
                while (ie < le && kChar.t[(unsigned char)*ie] != 5) ++ie;
                cur_id.assign(ls + 1, ie);
                a.ids.append(cur_id);
                a.ids.push_back('\0');
                pk.open_record();
                have = true;
                continue;
            }
            if (!have) raise(SW_ERR_RUNTIME, "Invalid FASTA: sequence encountered before header");  // :69-71
            const char *q = ls;
#ifdef SW_HAVE_AVX2_PACKER
            if (simd512) q = pack_line_avx512(pk, ls, le);   // (both stop early at a chunk with whitespace / control bytes)
            else if (simd) q = pack_line_avx2(pk, ls, le);
#endif
            for (; q < le; ++q) {  // :73-88
                const unsigned code = kChar.t[(unsigned char)*q];
                if (code == 5) continue;
                if (code == 6)
                    raise(SW_ERR_VALUE, "unsupported control byte 0x%02x in sequence of record %s in assembly %s",
                          (unsigned)(unsigned char)*q, cur_id.c_str(), path.c_str());
                pk.push(code);
            }
        }
    }
    void finish()
    {
        if (have) pk.close_record(path, cur_id);
        a.rec_run_off.push_back((uint32_t)a.run_pos.size());
        a.packed = std::move(words);
        a.n_words = a.packed.size();
    }
};

// r05: a plain file read by read() goes through the parser block by block (256 KiB: the block the kernel has just copied is
// parsed out of the core's L2, and the buffer it lands in stays there from block to block, instead of 5 MB written to memory and
// read back); the rest of a block behind its last '\n' moves to the front of the next.  A line longer than a block makes the
// buffer grow until its end is in.  SEQWIN_AMD_READ_BLOCK_KB=0: the whole file first (r01-r04).
size_t read_block_bytes()
{
    static const size_t n = [] {
        const char *e = SW_TEST_GETENV("SEQWIN_AMD_READ_BLOCK_KB");
        return (size_t)(e ? std::max(0, atoi(e)) : 256) << 10;
    }();
    return n;
}

void stream_plain_file(const std::string &path, RawBuf &buf, size_t block, TextParser &tp)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) raise(SW_ERR_RUNTIME, "Unable to open FASTA: %s", path.c_str());   // fasta_reader.cpp:100-102
    struct Close {
        int fd;
        ~Close() { close(fd); }
    } closer{fd};
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) tp.expect_bytes((size_t)st.st_size);
    size_t fill = 0;   // bytes of a line whose end has not been read yet, at the front of the buffer
    for (;;) {
        buf.reserve(fill + block);
        const ssize_t got = read(fd, buf.p + fill, block);
        if (got < 0) {
            if (errno == EINTR) continue;
            raise(SW_ERR_RUNTIME, "Unable to read FASTA: %s", path.c_str());
        }
        if (got == 0) break;
        const char *nl = (const char *)memrchr(buf.p + fill, '\n', (size_t)got);
        fill += (size_t)got;
        if (!nl) continue;
        tp.feed(buf.p, nl + 1);
        const size_t rest = fill - (size_t)(nl + 1 - buf.p);
        memmove(buf.p, nl + 1, rest);
        fill = rest;
    }
    tp.feed(buf.p, buf.p + fill);   // the last line, if the file does not end with '\n'
    buf.len = 0;
}

void parse_assembly(const std::string &path, RawBuf &buf, int use_mmap, WordBuf &&storage, Assembly &a)
{
    TextParser tp(path, a, std::move(storage));
    const size_t block = read_block_bytes();
    if (!use_mmap && block && !ends_with(path, ".gz")) {
        stream_plain_file(path, buf, block, tp);
    } else {
        const FileBytes file(path, buf, use_mmap);
        tp.expect_bytes(file.n);
        tp.feed(file.p, file.p + file.n);
    }
    tp.finish();
}

// Recycles the packed-word buffers of assemblies that have been handed to the sink, so that a streaming ingest
// keeps only ~n_workers buffers alive (and does not unmap / fault in fresh memory for every file).
struct BufferPool {
    std::mutex mu;
    std::vector<WordBuf> free_list;
    WordArena *arena = nullptr;   // fresh buffers ask it first (the sink's page-locked blocks)
    WordBuf get()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (free_list.empty()) return WordBuf(arena);
        WordBuf v = std::move(free_list.back());
        free_list.pop_back();
        return v;
    }
    void put(WordBuf &&v)
    {
        std::lock_guard<std::mutex> lock(mu);
        free_list.push_back(std::move(v));
    }
};

}  // namespace

// CPUs this process may run on at once: hardware threads, narrowed by the affinity mask and by the cgroup CPU quota (v2 cpu.max,
// v1 cfs quota / period), at least 1
size_t usable_cpus()
{
    size_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<size_t>(n, (size_t)std::max(1, CPU_COUNT(&set)));
    auto read_two = [](const char *path, long long &a, long long &b) {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char x[64] = {0}, y[64] = {0};
        const int got = fscanf(f, "%63s %63s", x, y);
        fclose(f);
        if (got < 1 || !strcmp(x, "max")) return false;
        a = atoll(x);
        b = got > 1 ? atoll(y) : 0;
        return true;
    };
    long long q = 0, per = 0;
    if (read_two("/sys/fs/cgroup/cpu.max", q, per) && q > 0 && per > 0) n = std::min<size_t>(n, (size_t)((q + per - 1) / per));
    else if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, per) && q > 0) {
        long long p2 = 0, dummy = 0;
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p2, dummy) && p2 > 0) n = std::min<size_t>(n, (size_t)((q + p2 - 1) / p2));
    }
    return std::max<size_t>(n, 1);
}

void ingest_fasta(const char *const *paths, size_t n_paths, uint64_t n_cpu, HostBatch &out, ChunkSink *sink)
{
    if (n_paths > UINT32_MAX)  // build.cpp:337-339
        raise(SW_ERR_RUNTIME, "Number of input assemblies exceeds uint32 range");
    std::vector<Assembly> asms(n_paths);
    std::vector<std::unique_ptr<Error>> errors(n_paths);
    size_t n_workers = std::max<uint64_t>(1, n_cpu);  // build.cpp:342-347
    if (n_paths > 0) n_workers = std::min(n_workers, n_paths);
    // r05: the cap follows the CPUs the process may really use.  Round 4 capped at 64 workers "because they contend for the
    // address space" (19 Gbp/s end to end at 32-64 workers, 5.6 at 256, on 2 x EPYC 9575F) -- the box in fact grants the container
    // 16 CPUs (cgroup cpu.max), and 256 runnable threads burn a 100 ms period's quota in its first quarter and sit out the rest.
    // Four workers per usable CPU is where the measured optimum lay then (64 on 16); with the faster packer, the block-wise read and
    // the page-locked word buffers of the end of r05 it is two (FASTA -> numpy at n_cpu 16 / 32 / 64 / 128: 38 / 44 / 29 / 27 Gbp/s
    // with four, gpurun_out/r5aa); on an unrestricted host that is no cap at all.
    // SEQWIN_AMD_INGEST_WORKERS_MAX overrides it (scaling tables).  The result does not depend on the worker count.
    {
        size_t cap = std::min<size_t>(2 * usable_cpus(), 128);   // (128: an absolute ceiling until a host without a quota has been measured -- ADVICE r5)
        if (const char *e = getenv("SEQWIN_AMD_INGEST_WORKERS_MAX")) cap = (size_t)std::max(1, atoi(e));
        n_workers = std::min(n_workers, std::max<size_t>(cap, 1));
    }

    const bool timing = getenv("SEQWIN_AMD_DEBUG_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    out = HostBatch();
    const bool threaded = (n_workers > 1 && n_paths > 1) || (sink && n_paths > 0);
    // Plain files are mapped only by a single worker; several workers read() into their buffers.  r05 measured mapping from many
    // workers again (round 4's "serialises on the address-space lock" had been measured while 256 threads fought over a 16-CPU
    // quota): 2 048 genomes of 5 Mbp in /dev/shm, FASTA -> numpy at 32 workers (tests/tools/e2e_ingest_ab.py, Gbp/s) -- one box
    // read() 25.7 / mmap 26.9 / mmap + MAP_POPULATE 30.3, the next box 23.6-26.7 / 22.4-27.1 / 16.7-17.5, and inside bench.py on a
    // third 15 (all worker counts alike: serialised).  Not robust across boxes: read() stays.  SEQWIN_AMD_MMAP=0 / 1 / 2 forces
    // read() / mmap / mmap + populate for such measurements, SEQWIN_AMD_NO_MMAP=1 is read() also for a single worker.
    int mmap_mode = threaded ? 0 : 1;
    if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_MMAP")) mmap_mode = atoi(e);
    if (SW_TEST_GETENV("SEQWIN_AMD_NO_MMAP")) mmap_mode = 0;
    const int use_mmap = mmap_mode;
    BufferPool pool;
    if (sink) pool.arena = sink->arena();
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::mutex done_mu;
    std::condition_variable done_cv;
    std::vector<char> done(n_paths, 0);
    // r05: with a sink, the parsers stay within a window of assemblies ahead of the one the sink is at.  Unbounded (r01-r05a) they ran
    // as far ahead as the CPUs let them -- 2 048 genomes parsed while the sink thread was at the 500th: 2.6 GB of word buffers
    // faulted in fresh instead of ~n_workers buffers going round (page faults and zeroing on the parsers' CPU time, 95-220 ms of
    // munmap when ingest_fasta returned: SEQWIN_AMD_DEBUG_TIMING, gpurun_out/r5z).
    size_t sunk = 0;   // assemblies the sink thread is done with (under done_mu)
    std::condition_variable sunk_cv;
    size_t window = n_workers + n_workers / 2 + 4;
    if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_INGEST_WINDOW")) window = (size_t)std::max(1, atoi(e));
    auto worker = [&]() {
        RawBuf buf;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_paths) break;
            if (sink) {
                std::unique_lock<std::mutex> lock(done_mu);
                sunk_cv.wait(lock, [&] { return i < sunk + window; });
            }
            try {
                parse_assembly(paths[i], buf, use_mmap, pool.get(), asms[i]);
            } catch (const Error &e) {
                errors[i].reset(new Error(e));
            } catch (const std::exception &e) {
                errors[i].reset(new Error(SW_ERR_RUNTIME, e.what()));
            }
            if (errors[i]) failed.store(true);
            if (threaded) {
                {
                    std::lock_guard<std::mutex> lock(done_mu);
                    done[i] = 1;
                }
                done_cv.notify_all();
            }
        }
    };
    out.chunks.resize(n_paths);
    out.chunk_word0.assign(n_paths + 1, 0);
    if (!threaded) {
        worker();
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < n_workers; ++t) th.emplace_back(worker);
        if (sink) {
            // this thread hands the finished assemblies to the sink in order while the workers go on parsing
            uint64_t est_bytes = 0;
            for (size_t i = 0; i < n_paths; ++i) {
                struct stat st;
                if (stat(paths[i], &st) == 0 && S_ISREG(st.st_mode))
                    est_bytes += (uint64_t)st.st_size * (ends_with(paths[i], ".gz") ? 5 : 1);
            }
            bool sink_ok = true;
            std::unique_ptr<Error> sink_error;
            try {
                sink->begin(est_bytes / 32 + est_bytes / 2048 + 1024 * n_paths);
            } catch (const Error &e) {
                sink_ok = false;
                sink_error.reset(new Error(e));
            }
            uint64_t word_off = 0;
            double wait_ms = 0;
            for (size_t i = 0; i < n_paths; ++i) {
                {
                    const auto w0 = std::chrono::steady_clock::now();
                    std::unique_lock<std::mutex> lock(done_mu);
                    sunk = i;   // (assembly i itself is inside every window)
                    sunk_cv.notify_all();
                    done_cv.wait(lock, [&] { return done[i] != 0; });
                    wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                }
                if (failed.load() || !sink_ok) continue;   // keep draining the flags; the error is raised below
                out.chunk_word0[i] = word_off;
                const size_t n_words = asms[i].packed.size();
                try {
                    if (n_words) sink->chunk(asms[i].packed, word_off);
                } catch (const Error &e) {
                    sink_ok = false;
                    sink_error.reset(new Error(e));
                    continue;
                }
                word_off += n_words;
                // (the sink has copied it: the next file's buffer -- unless the sink offers page-locked ones, which a recycled
                // malloc'd buffer would keep its parser from ever getting)
                if (asms[i].packed.has_storage() && !pool.arena) pool.put(std::move(asms[i].packed));
                asms[i].packed = WordBuf();
            }
            {
                std::lock_guard<std::mutex> lock(done_mu);
                sunk = n_paths;
            }
            sunk_cv.notify_all();
            out.chunk_word0[n_paths] = word_off;
            if (timing) fprintf(stderr, "[seqwin_amd] ingest: sink thread waited %.1f ms for the parsers\n", wait_ms);
            for (auto &t : th) t.join();
            if (sink_error) throw *sink_error;
        } else {
            for (auto &t : th) t.join();
        }
    }
    for (size_t i = 0; i < n_paths; ++i)
        if (errors[i]) throw *errors[i];
    const auto t_parsed = std::chrono::steady_clock::now();

    // tables in assembly order (= global record order, build.cpp:135,169,191); the packed words stay in their per-assembly chunks
    out.n_assemblies = n_paths;
    out.record_offsets.assign(n_paths + 1, 0);
    uint64_t n_rec = 0, n_runs = 0;
    for (size_t i = 0; i < n_paths; ++i) {
        n_rec += asms[i].rec_len.size();
        if (n_rec > UINT32_MAX)  // build.cpp:136-140
            raise(SW_ERR_RUNTIME, "Total number of FASTA records exceeds uint32 range");
        out.record_offsets[i + 1] = (uint32_t)n_rec;
        n_runs += asms[i].run_pos.size();
    }
    if (n_runs > UINT32_MAX) raise(SW_ERR_RUNTIME, "Total number of valid-base runs exceeds uint32 range");
    out.rec_len.reserve(n_rec);
    out.rec_base.reserve(n_rec);
    out.rec_run_off.reserve(n_rec + 1);
    out.run_pos.reserve(n_runs);
    out.run_len.reserve(n_runs);
    uint64_t word_off = 0;
    for (size_t i = 0; i < n_paths; ++i) {
        Assembly &a = asms[i];
        const uint32_t run_base = (uint32_t)out.run_pos.size();
        for (size_t r = 0; r < a.rec_len.size(); ++r) {
            out.rec_len.push_back(a.rec_len[r]);
            out.rec_base.push_back(a.rec_base[r] + word_off * 32);
            out.rec_run_off.push_back(a.rec_run_off[r] + run_base);
        }
        out.run_pos.insert(out.run_pos.end(), a.run_pos.begin(), a.run_pos.end());
        out.run_len.insert(out.run_len.end(), a.run_len.begin(), a.run_len.end());
        out.chunk_word0[i] = word_off;
        word_off += a.n_words;
        out.chunks[i] = std::move(a.packed);   // empty when it has been streamed to the sink
        out.ids_blob.append(a.ids);
        out.total_bp += a.total_bp;
    }
    out.chunk_word0[n_paths] = word_off;
    if (timing)
        fprintf(stderr, "[seqwin_amd] ingest: %zu files, %zu workers: parse %.1f ms, tables %.1f ms\n", n_paths, n_workers,
                std::chrono::duration<double, std::milli>(t_parsed - t_begin).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_parsed).count());
    out.rec_run_off.push_back((uint32_t)out.run_pos.size());
}

}  // namespace sw
