// Differential fuzzer for seqwin_amd/csrc/fast_inflate.hpp against zlib (CPU only; `make -C seqwin_amd/csrc finfsan` builds it with
// AddressSanitizer + UBSan, tests/test_abi_cpu.py runs it).
//   finf_san <seed> <cases>      exit code 0 = the contract held on every case
// Contract (fast_inflate.hpp): gunzip_members() == OK  =>  its output is byte for byte what zlib delivers for the same bytes (and
// zlib accepts them as complete gzip members); well-formed members made by zlib's deflate at any level / strategy MUST be OK; on
// mutated, truncated or extended streams it may say BAD (the caller then takes zlib), but it never reads or writes outside its
// buffers and never says OK with other bytes than zlib's.  Also: finf::crc32 == zlib's crc32 on random spans.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../seqwin_amd/csrc/fast_inflate.hpp"

static uint64_t rng_state;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

static std::string make_text(size_t n, int kind)
{
    std::string s(n, 0);
    switch (kind) {
    case 0: for (auto &c : s) c = "ACGT"[rnd() & 3]; break;                                    // DNA
    case 1: for (size_t i = 0; i < n; ++i) s[i] = (i % 81 == 80) ? '\n' : "ACGTN"[rnd() % 5]; break;   // FASTA-like lines
    case 2: for (auto &c : s) c = (char)rnd(); break;                                           // noise
    case 3: for (size_t i = 0; i < n; ++i) s[i] = "ACGGTCA"[i % 7]; break;                      // tandem repeat: long matches
    case 4: for (auto &c : s) c = 'A'; break;                                                   // run: distance 1
    case 6: {                                                                                   // words: many literal codes of many lengths, second-level tables
        static const char *words[] = {"contig", "scaffold", "Escherichia", "coli", "plasmid", "whole", "genome", "shotgun", "sequence", "NZ_", "strain", "K-12", "\n>", "length=", " "};
        size_t i = 0;
        while (i < n) {
            const char *w = words[rnd() % 15];
            for (; *w && i < n; ++w) s[i++] = *w;
            if (rnd() % 4 == 0 && i < n) s[i++] = (char)('0' + rnd() % 10);
        }
        break;
    }
    case 7: {                                                                                   // soft-masked FASTA: headers, 60-column lines, lower-case stretches, N runs
        size_t i = 0, col = 0;
        bool lower = false;
        while (i < n) {
            if (rnd() % 5000 == 0) {
                const char *h = "\n>seq description text\n";
                for (; *h && i < n; ++h) s[i++] = *h;
                col = 0;
                continue;
            }
            if (rnd() % 300 == 0) lower = !lower;
            if (col == 60) { s[i++] = '\n'; col = 0; continue; }
            const char c = rnd() % 400 == 0 ? 'N' : "ACGT"[rnd() & 3];
            s[i++] = lower ? (char)(c | 0x20) : c;
            ++col;
        }
        break;
    }
    default: {                                                                                  // mixture with far matches
        for (auto &c : s) c = "ACGT"[rnd() & 3];
        for (int r = 0; r < 20 && n > 100; ++r) {
            const size_t a = rnd() % (n - 50), b = rnd() % (n - 50), l = 3 + rnd() % 47;
            for (size_t i = 0; i < l; ++i) s[b + i] = s[a + i];
        }
    }
    }
    return s;
}

static std::vector<uint8_t> gzip_member(const std::string &text, int level, int strategy)
{
    z_stream z{};
    if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&z, text.size()) + 64);
    z.next_in = (Bytef *)text.data();
    z.avail_in = (uInt)text.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// zlib's verdict: all members inflate completely and nothing is left -> true + the bytes
static bool zlib_gunzip(const std::vector<uint8_t> &in, std::string &out)
{
    out.clear();
    size_t at = 0;
    while (at < in.size()) {
        z_stream z{};
        if (inflateInit2(&z, 15 + 16) != Z_OK) abort();
        z.next_in = (Bytef *)in.data() + at;
        z.avail_in = (uInt)(in.size() - at);
        int rc;
        do {
            char buf[65536];
            z.next_out = (Bytef *)buf;
            z.avail_out = sizeof buf;
            rc = inflate(&z, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&z);
                return false;
            }
            out.append(buf, sizeof buf - z.avail_out);
            if (rc == Z_OK && z.avail_in == 0 && z.avail_out != 0) {   // input exhausted mid-stream
                inflateEnd(&z);
                return false;
            }
        } while (rc != Z_STREAM_END);
        at = in.size() - z.avail_in;
        inflateEnd(&z);
    }
    return true;
}

#include <chrono>
static bool read_all(const char *path, std::vector<uint8_t> &gz)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint8_t b[65536];
    size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) gz.insert(gz.end(), b, b + n);
    fclose(f);
    return true;
}

// finf_san time <a.gz>: MB/s of text, this decoder (CRC-32 included) against zlib's inflate, one thread, best of five
static int time_file(const char *path)
{
    std::vector<uint8_t> gz;
    if (!read_all(path, gz)) return 2;
    std::string ref;
    if (!zlib_gunzip(gz, ref)) return 3;
    std::vector<uint8_t> in(gz.size() + sw::finf::IN_PAD, 0), out(ref.size() + 64);
    memcpy(in.data(), gz.data(), gz.size());
    static sw::finf::Tables T;
    double best[2] = {1e9, 1e9};
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        size_t got = 0;
        if (sw::finf::gunzip_members(in.data(), gz.size(), out.data(), out.size(), &got, T) != sw::finf::OK || got != ref.size() ||
            memcmp(out.data(), ref.data(), got) != 0)
            return 4;
        auto t1 = std::chrono::steady_clock::now();
        std::string r2;
        zlib_gunzip(gz, r2);
        auto t2 = std::chrono::steady_clock::now();
        best[0] = std::min(best[0], std::chrono::duration<double>(t1 - t0).count());
        best[1] = std::min(best[1], std::chrono::duration<double>(t2 - t1).count());
    }
    printf("%s: %zu -> %zu bytes; fast_inflate (with CRC-32) %.0f MB/s, zlib inflate %.0f MB/s: %.2f x\n", path, gz.size(), ref.size(),
           ref.size() / best[0] / 1e6, ref.size() / best[1] / 1e6, best[1] / best[0]);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 2 && !strcmp(argv[1], "time")) return time_file(argv[2]);
    rng_state = argc > 1 ? strtoull(argv[1], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1 : 88172645463325252ull;
    const long cases = argc > 2 ? atol(argv[2]) : 2000;
    static sw::finf::Tables T;
    long n_ok = 0, n_bad = 0, n_mut_ok = 0, n_sup = 0;
    // CRC-32 against zlib on random spans and alignments
    {
        std::vector<uint8_t> buf(70000);
        for (auto &b : buf) b = (uint8_t)rnd();
        for (int i = 0; i < 400; ++i) {
            const size_t off = rnd() % 64, len = (i < 200) ? rnd() % 400 : rnd() % (buf.size() - 64);
            const uint32_t a = sw::finf::crc32(buf.data() + off, len), b = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data() + off, (uInt)len);
            if (a != b) {
                fprintf(stderr, "crc32 mismatch: off %zu len %zu: %08x != %08x\n", off, len, a, b);
                return 1;
            }
        }
        printf("crc32: 400 spans equal to zlib's (folded form %s)\n", sw::finf::crc_fold_state() == 1 ? "in use" : "not in use");
    }
    // Streams that END inside a dynamic block's header (the header loops refilled without an overrun test and read
    // behind IN_PAD): 0-8 empty fixed blocks in front (every bit alignment), then BTYPE=2 with HCLEN=19 and j of the 19 code-length
    // codes present, cut there -- and every such stream cut again at each earlier byte.  Exact-size input allocations.
    {
        long n_hdr = 0;
        for (int nf = 0; nf < 9 * 4; ++nf)
            for (int j = 0; j <= 19; ++j)
                for (int fill = 0; fill < 2; ++fill) {
                    const int nfixed = nf % 9, nlit = nf / 9;   // nlit fixed blocks of one 9-bit literal (19 bits each: odd alignments too)
                    std::vector<uint8_t> gz = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
                    uint64_t acc = 0;
                    unsigned nb = 0;
                    auto put = [&](uint32_t v, unsigned n) {
                        acc |= (uint64_t)v << nb;
                        nb += n;
                        while (nb >= 8) { gz.push_back((uint8_t)acc); acc >>= 8; nb -= 8; }
                    };
                    for (int b = 0; b < nfixed; ++b) { put(0, 1); put(1, 2); put(0, 7); }   // BFINAL=0, fixed, end-of-block
                    for (int b = 0; b < nlit; ++b) { put(0, 1); put(1, 2); put(0x13, 9); put(0, 7); }   // ... with literal 144 (code 110010000, MSB first)
                    put(0, 1); put(2, 2); put(fill ? 29 : 0, 5); put(fill ? 29 : 0, 5); put(15, 4);
                    for (int c = 0; c < j; ++c) put(fill ? 7 : 1, 3);
                    if (nb) gz.push_back((uint8_t)(acc | (fill ? 0xFFu << nb : 0)));
                    for (size_t len = gz.size(); len >= 10; --len) {
                        std::vector<uint8_t> in(len + sw::finf::IN_PAD, 0);
                        memcpy(in.data(), gz.data(), len);
                        std::vector<uint8_t> out(64);
                        size_t got = 0;
                        if (sw::finf::gunzip_members(in.data(), len, out.data(), out.size(), &got, T) == sw::finf::OK) {
                            fprintf(stderr, "a stream cut inside a dynamic header was accepted (nfixed %d, j %d, len %zu)\n", nfixed, j, len);
                            return 1;
                        }
                        ++n_hdr;
                    }
                }
        printf("%ld streams ending inside a dynamic block header: all declined, no access outside the buffers\n", n_hdr);
    }
    for (long c = 0; c < cases; ++c) {
        const int kind = (int)(rnd() % 8);
        static const size_t sizes[] = {0, 1, 2, 17, 300, 5000, 70000, 300000};
        const size_t n = sizes[rnd() % 8] + (rnd() % 7);
        std::vector<uint8_t> gz;
        std::string text;
        const int members = 1 + (rnd() % 8 == 0 ? (int)(rnd() % 3) : 0);
        for (int m = 0; m < members; ++m) {
            const std::string t = make_text(m ? n / 3 : n, kind);
            static const int levels[] = {0, 1, 6, 9}, strategies[] = {Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
            const std::vector<uint8_t> one = gzip_member(t, levels[rnd() % 4], strategies[rnd() % 6]);
            gz.insert(gz.end(), one.begin(), one.end());
            text += t;
        }
        const bool mutate = c % 3 == 2;
        if (mutate && !gz.empty()) {
            const int how = (int)(rnd() % 4);
            if (how == 0) for (int i = 0, k = 1 + (int)(rnd() % 3); i < k; ++i) gz[rnd() % gz.size()] ^= (uint8_t)(1u << (rnd() % 8));
            else if (how == 1) gz.resize(rnd() % gz.size());
            else if (how == 2) for (int i = 0, k = 1 + (int)(rnd() % 9); i < k; ++i) gz.push_back((uint8_t)rnd());
            else gz[rnd() % gz.size()] = (uint8_t)rnd();
        }
        // exact-size allocations, so that AddressSanitizer sees any access beyond IN_PAD / the output capacity
        std::vector<uint8_t> in(gz.size() + sw::finf::IN_PAD, 0);
        memcpy(in.data(), gz.data(), gz.size());
        size_t cap = text.size() + (rnd() % 3 == 0 ? 0 : rnd() % 100);
        if (rnd() % 10 == 0 && cap > 4) cap /= 2;   // too small on purpose
        std::vector<uint8_t> out(cap ? cap : 1);
        size_t got = 0;
        const sw::finf::Result r = sw::finf::gunzip_members(in.data(), gz.size(), out.data(), cap, &got, T);
        // the super table (r06: literal(s) + a whole match from one lookup): what build_super left in T for the stream's last dynamic
        // block -- built by the recurrence over shorter indices -- must be, entry for entry, what super_entry decodes from the two
        // tables directly (whatever the stream was: T always holds consistent tables of SOME block, or zeros)
        if (T.d_usable) {
            sw::finf::Tables &chk = T;
            static std::vector<uint64_t> built(1u << sw::finf::LL_BITS);
            memcpy(built.data(), chk.sup, sizeof chk.sup);
            sw::finf::build_super(chk);   // (of the tables as they stand: a header that failed half-way may have replaced ll or d since)
            for (unsigned i = 0; i < (1u << sw::finf::LL_BITS); ++i)
                if (chk.sup[i] != sw::finf::super_entry(chk, i)) {
                    fprintf(stderr, "case %ld: super table entry %u is %016llx, the tables decode to %016llx\n", c, i, (unsigned long long)chk.sup[i],
                            (unsigned long long)sw::finf::super_entry(chk, i));
                    return 1;
                }
            ++n_sup;
        }
        std::string ref;
        const bool zok = zlib_gunzip(gz, ref);
        if (r == sw::finf::OK) {
            ++n_ok;
            n_mut_ok += mutate;
            if (!zok || got != ref.size() || memcmp(out.data(), ref.data(), got) != 0) {
                fprintf(stderr, "case %ld: OK, but zlib %s (%zu vs %zu bytes)\n", c, zok ? "gives other bytes" : "rejects the stream", got, ref.size());
                return 1;
            }
        } else {
            ++n_bad;
            if (!mutate && !(r == sw::finf::NEED_OUT && cap < text.size())) {
                fprintf(stderr, "case %ld: a well-formed stream (%d members, %zu bytes of text, kind %d) was refused (%d), capacity %zu\n", c, members,
                        text.size(), kind, (int)r, cap);
                return 1;
            }
        }
    }
    printf("%ld cases: %ld accepted (%ld of them mutated streams zlib accepts as well), %ld left to zlib; %ld super tables equal to their definition\n", cases,
           n_ok, n_mut_ok, n_bad, n_sup);
    return 0;
}
