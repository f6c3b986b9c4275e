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
#include <cstdlib>
#include <memory>
#include <thread>

#include <zlib.h>

#include "common.hpp"

namespace sw {

void check_kw(uint64_t k, uint64_t w)
{
    // k < 3 crashes the reference (unsigned k-3, nthash_kmer.hpp:26); k is truncated to uint16
    // there (hashing_internals.hpp:10).  Both are refused here.
    if (k < 3) raise(SW_ERR_VALUE, "kmerlen must be >= 3 (got %llu)", (unsigned long long)k);
    if (k > 65535) raise(SW_ERR_VALUE, "kmerlen must be <= 65535 (got %llu)", (unsigned long long)k);
    if (w < 1) raise(SW_ERR_VALUE, "windowsize must be >= 1 (got %llu)", (unsigned long long)w);
    if (w > SW_MAX_WINDOW)
        raise(SW_ERR_VALUE, "windowsize must be <= %u on the GPU path (got %llu)", SW_MAX_WINDOW,
              (unsigned long long)w);
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

void slurp(const std::string &path, std::vector<char> &buf)
{
    buf.clear();
    if (ends_with(path, ".gz")) {  // fasta_reader.cpp:209
        gzFile gz = gzopen(path.c_str(), "rb");
        if (!gz) raise(SW_ERR_RUNTIME, "Unable to open gzip FASTA: %s", path.c_str());
        gzbuffer(gz, 1u << 20);
        size_t len = 0;
        for (;;) {
            if (buf.size() - len < (1u << 20)) buf.resize(std::max<size_t>(buf.size() * 2, 1u << 22));
            int got = gzread(gz, buf.data() + len, 1u << 20);
            if (got < 0) {
                int errnum = 0;
                const char *e = gzerror(gz, &errnum);
                std::string msg = std::string("gzip read error: ") + (e ? e : "unknown");
                gzclose(gz);
                raise(SW_ERR_RUNTIME, "%s", msg.c_str());
            }
            if (got == 0) break;
            len += (size_t)got;
        }
        gzclose(gz);
        buf.resize(len);
    } else {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) raise(SW_ERR_RUNTIME, "Unable to open FASTA: %s", path.c_str());
        size_t len = 0;
        for (;;) {
            if (buf.size() - len < (1u << 20)) buf.resize(std::max<size_t>(buf.size() * 2, 1u << 22));
            size_t got = fread(buf.data() + len, 1, 1u << 20, f);
            len += got;
            if (got == 0) break;
        }
        fclose(f);
        buf.resize(len);
    }
}

// One assembly, parsed and packed by one worker.
struct Assembly {
    std::vector<uint32_t> rec_len;
    std::vector<uint64_t> rec_base;  // local (within this assembly's packed stream)
    std::vector<uint32_t> rec_run_off;
    std::vector<uint32_t> run_pos, run_len;
    std::vector<uint64_t> packed;  // 32 bases per word
    std::string ids;
    uint64_t total_bp = 0;
};

struct Packer {
    Assembly &a;
    uint64_t acc = 0;
    unsigned nacc = 0;     // bases in acc
    uint64_t len = 0;      // bases in the current record
    int64_t run_start = -1;
    explicit Packer(Assembly &as) : a(as) {}

    void open_record()
    {
        a.rec_base.push_back((uint64_t)a.packed.size() * 32);
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
        if (nacc) a.packed.push_back(acc);
        if (len > UINT32_MAX)  // build.cpp:143-147
            raise(SW_ERR_RUNTIME, "Sequence length exceeds uint32 range for record %s in assembly %s",
                  id.c_str(), path.c_str());
        a.rec_len.push_back((uint32_t)len);
        a.total_bp += len;
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
            a.packed.push_back(acc);
            acc = 0;
            nacc = 0;
        }
    }
};

void parse_assembly(const std::string &path, std::vector<char> &buf, Assembly &a)
{
    slurp(path, buf);
    const char *p = buf.data();
    const char *end = p + buf.size();
    Packer pk(a);
    bool have = false;
    std::string cur_id;
    a.packed.reserve(buf.size() / 32 + 16);

    while (p < end) {
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
        if (*ls == '>') {  // :58-67
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
        for (const char *q = ls; q < le; ++q) {  // :73-88
            const unsigned code = kChar.t[(unsigned char)*q];
            if (code == 5) continue;
            if (code == 6)
                raise(SW_ERR_VALUE, "unsupported control byte 0x%02x in sequence of record %s in assembly %s",
                      (unsigned)(unsigned char)*q, cur_id.c_str(), path.c_str());
            pk.push(code);
        }
    }
    if (have) pk.close_record(path, cur_id);
    a.rec_run_off.push_back((uint32_t)a.run_pos.size());
}

}  // namespace

void ingest_fasta(const char *const *paths, size_t n_paths, uint64_t n_cpu, HostBatch &out)
{
    if (n_paths > UINT32_MAX)  // build.cpp:337-339
        raise(SW_ERR_RUNTIME, "Number of input assemblies exceeds uint32 range");
    std::vector<Assembly> asms(n_paths);
    std::vector<std::unique_ptr<Error>> errors(n_paths);
    size_t n_workers = std::max<uint64_t>(1, n_cpu);  // build.cpp:342-347
    if (n_paths > 0) n_workers = std::min(n_workers, n_paths);

    std::atomic<size_t> next{0};
    auto worker = [&]() {
        std::vector<char> buf;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n_paths) break;
            try {
                parse_assembly(paths[i], buf, asms[i]);
            } catch (const Error &e) {
                errors[i].reset(new Error(e));
            } catch (const std::exception &e) {
                errors[i].reset(new Error(SW_ERR_RUNTIME, e.what()));
            }
        }
    };
    if (n_workers <= 1 || n_paths <= 1) {
        worker();
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < n_workers; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < n_paths; ++i)
        if (errors[i]) throw *errors[i];

    // concatenate in assembly order (= global record order, build.cpp:135,169,191)
    out = HostBatch();
    out.n_assemblies = n_paths;
    out.record_offsets.assign(n_paths + 1, 0);
    uint64_t n_rec = 0, n_runs = 0, n_words = 0;
    for (size_t i = 0; i < n_paths; ++i) {
        n_rec += asms[i].rec_len.size();
        if (n_rec > UINT32_MAX)  // build.cpp:136-140
            raise(SW_ERR_RUNTIME, "Total number of FASTA records exceeds uint32 range");
        out.record_offsets[i + 1] = (uint32_t)n_rec;
        n_runs += asms[i].run_pos.size();
        n_words += asms[i].packed.size();
    }
    if (n_runs > UINT32_MAX) raise(SW_ERR_RUNTIME, "Total number of valid-base runs exceeds uint32 range");
    out.rec_len.reserve(n_rec);
    out.rec_base.reserve(n_rec);
    out.rec_run_off.reserve(n_rec + 1);
    out.run_pos.reserve(n_runs);
    out.run_len.reserve(n_runs);
    out.packed.resize(n_words * 2 + 8, 0);  // + slack so the kernel's word reads never leave the buffer
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
        if (!a.packed.empty()) memcpy(out.packed.data() + word_off * 2, a.packed.data(), a.packed.size() * 8);
        word_off += a.packed.size();
        out.ids_blob.append(a.ids);
        out.total_bp += a.total_bp;
        Assembly().packed.swap(a.packed);
    }
    out.rec_run_off.push_back((uint32_t)out.run_pos.size());
}

}  // namespace sw
