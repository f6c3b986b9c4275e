// Sanitizer driver for the device gzip ingest (seqwin_amd/csrc/gz_dev.hpp: DEFLATE decoder + FASTA parser / packer, the code
// the kernels of ingest_dev.hip run one file per lane): built for the HOST by `make -C seqwin_amd/csrc gzsan` with
// -fsanitize=address,undefined and run by tests/test_abi_cpu.py on well-formed and broken .gz files -- the GPU has no
// sanitizers.  Buffers are laid out and sized exactly as device_gz_ingest() lays them out in HBM.
//   gz_dev_san <dump file> <path.gz>...
// exit code 0: every file qualified; the dump holds record_offsets | ids blob | record lengths | decoded bases of every record
//              ('N' = invalid base) -- the format of ingest_san, so the test compares it with the host reader's;
// exit code 4: the device route would decline (first reason on stderr) -- the library then takes the host route.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../seqwin_amd/csrc/gz_dev.hpp"

using namespace sw::gz;

static int decline(const char *why, const char *path, unsigned long long v)
{
    fprintf(stderr, "declined: %s (%s, %llu)\n", why, path, v);
    return 4;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const size_t nf = (size_t)(argc - 2);
    std::vector<std::vector<uint8_t>> raw(nf);
    std::vector<uint64_t> coff(nf + 1, 0), dstart(nf), dend(nf), toff(nf + 1, 0);
    std::vector<uint32_t> isize(nf), crc_want(nf);
    for (size_t i = 0; i < nf; ++i) {
        FILE *f = fopen(argv[i + 2], "rb");
        if (!f) return decline("unreadable", argv[i + 2], 0);
        int c;
        while ((c = fgetc(f)) != EOF) raw[i].push_back((uint8_t)c);
        fclose(f);
        if (raw[i].size() < 18) return decline("not a regular file of 18 bytes or more", argv[i + 2], raw[i].size());
        const uint64_t h = gzip_header_len(raw[i].data(), raw[i].size());
        if (h == 0 || h + 8 > raw[i].size()) return decline("not a plain gzip header", argv[i + 2], 0);
        coff[i + 1] = coff[i] + ((raw[i].size() + 15) & ~15ull);
        const uint8_t *t = raw[i].data() + raw[i].size() - 8;
        crc_want[i] = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        isize[i] = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        dstart[i] = coff[i] + h;
        dend[i] = coff[i] + raw[i].size() - 8;
        toff[i + 1] = toff[i] + (((uint64_t)isize[i] + 15) & ~15ull);
    }
    uint8_t *comp = (uint8_t *)aligned_alloc(16, coff[nf] + 16);
    memset(comp, 0xA5, coff[nf] + 16);
    for (size_t i = 0; i < nf; ++i) memcpy(comp + coff[i], raw[i].data(), raw[i].size());
    uint8_t *text = (uint8_t *)aligned_alloc(16, toff[nf] + 16);
    memset(text, 0x5A, toff[nf] + 16);

    static LaneTables tables;
    for (size_t i = 0; i < nf; ++i) {
        const uint32_t st = inflate_one(tables, comp, dstart[i], dend[i], text, toff[i], isize[i], nullptr);
        if (st != ST_OK) return decline("inflate status", argv[i + 2], st);
    }
    static uint8_t cls[256];
    static uint32_t crc_tab[4][256];
    for (uint32_t i = 0; i < 256; ++i) {
        cls[i] = char_class(i);
        crc_tab[0][i] = crc_entry(i);
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int s = 1; s < 4; ++s) crc_tab[s][i] = (crc_tab[s - 1][i] >> 8) ^ crc_tab[0][crc_tab[s - 1][i] & 0xFFu];
    std::vector<ParseCounts> counts(nf);
    std::vector<uint64_t> word_base(nf + 1, 0), id_base(nf + 1, 0);
    std::vector<uint32_t> rec_idx(nf + 1, 0), run_base(nf + 1, 0);
    for (size_t i = 0; i < nf; ++i) {
        ParseDst none{};
        parse_one<false>(cls, crc_tab, text + toff[i], isize[i], none, counts[i]);
        if (counts[i].err) return decline("parse error flags", argv[i + 2], counts[i].err);
        if (counts[i].crc != crc_want[i]) return decline("CRC-32 mismatch", argv[i + 2], counts[i].crc);
        word_base[i + 1] = word_base[i] + counts[i].n_words;
        id_base[i + 1] = id_base[i] + counts[i].n_id;
        rec_idx[i + 1] = rec_idx[i] + counts[i].n_rec;
        run_base[i + 1] = run_base[i] + counts[i].n_runs;
    }
    // exact sizes: an overrun of the second walk is an AddressSanitizer report
    std::vector<uint64_t> words(word_base[nf]), rec_base(rec_idx[nf]);
    std::vector<uint32_t> rec_len(rec_idx[nf]), rec_run_off(rec_idx[nf]), run_pos(run_base[nf]), run_len(run_base[nf]);
    std::vector<char> ids(id_base[nf]);
    for (size_t i = 0; i < nf; ++i) {
        ParseDst D{word_base[i], id_base[i], rec_idx[i], run_base[i], words.data(), rec_len.data(), rec_run_off.data(), run_pos.data(),
                   run_len.data(), rec_base.data(), ids.data()};
        ParseCounts again;
        parse_one<true>(cls, crc_tab, text + toff[i], isize[i], D, again);
        if (again.n_words != counts[i].n_words || again.n_rec != counts[i].n_rec || again.n_runs != counts[i].n_runs ||
            again.n_id != counts[i].n_id)
            return decline("the two walks disagree", argv[i + 2], again.n_words);
    }
    FILE *dump = fopen(argv[1], "wb");
    if (!dump) return 2;
    fwrite(rec_idx.data(), 4, nf + 1, dump);
    fwrite(ids.data(), 1, ids.size(), dump);
    fwrite(rec_len.data(), 4, rec_len.size(), dump);
    uint64_t bp = 0;
    for (size_t r = 0; r < rec_len.size(); ++r) {
        std::string s(rec_len[r], 'N');
        const uint32_t u0 = rec_run_off[r], u1 = r + 1 < rec_len.size() ? rec_run_off[r + 1] : run_base[nf];
        for (uint32_t u = u0; u < u1; ++u)
            for (uint32_t q = 0; q < run_len[u]; ++q) {
                const uint64_t b = rec_base[r] + run_pos[u] + q;
                s[run_pos[u] + q] = "ACGT"[(words[b >> 5] >> (2 * (b & 31))) & 3];
            }
        fwrite(s.data(), 1, s.size(), dump);
        bp += rec_len[r];
    }
    fclose(dump);
    printf("%zu assemblies %zu records %llu bp\n", nf, rec_len.size(), (unsigned long long)bp);
    free(comp);
    free(text);
    return 0;
}
