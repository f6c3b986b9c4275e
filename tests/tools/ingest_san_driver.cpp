// Sanitizer driver for the host FASTA reader / 2-bit packer (seqwin_amd/csrc/host_ingest.cpp): built by
// `make -C seqwin_amd/csrc asan` with -fsanitize=address,undefined, run by tests/test_abi_cpu.py on hostile bytes.
// CPU only (sanitizers never run on the GPU box).  Prints one line per file set: a digest of everything the reader
// produced (record table, ids, decoded bases), which the test compares with the regular library's.
//   ingest_san <n_cpu> <dump file> <path>...   exit code 0 = parsed, 3 = the reader refused the input (message on stderr)
// The dump holds record_offsets | ids blob | record lengths | decoded bases of every record ('N' = invalid base).
#include <cinttypes>
#include <cstdio>
#include <cstdlib>

#include "../../seqwin_amd/csrc/common.hpp"

namespace sw {
void set_last_error(const char *) {}
}  // namespace sw

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ULL;
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const uint64_t n_cpu = strtoull(argv[1], nullptr, 10);
    FILE *dump = fopen(argv[2], "wb");
    if (!dump) return 2;
    sw::HostBatch h;
    try {
        sw::ingest_fasta(argv + 3, (size_t)(argc - 3), n_cpu, h);
    } catch (const sw::Error &e) {
        fprintf(stderr, "refused (%d): %s\n", e.code, e.what());
        return 3;
    }
    uint64_t d = 0xcbf29ce484222325ULL;
    d = fnv(d, h.record_offsets.data(), h.record_offsets.size() * 4);
    d = fnv(d, h.ids_blob.data(), h.ids_blob.size());
    d = fnv(d, h.rec_len.data(), h.rec_len.size() * 4);
    fwrite(h.record_offsets.data(), 4, h.record_offsets.size(), dump);
    fwrite(h.ids_blob.data(), 1, h.ids_blob.size(), dump);
    fwrite(h.rec_len.data(), 4, h.rec_len.size(), dump);
    for (size_t r = 0; r < h.rec_len.size(); ++r) {
        std::string seq(h.rec_len[r], 'N');
        for (uint32_t q = h.rec_run_off[r]; q < h.rec_run_off[r + 1]; ++q)
            for (uint64_t p = h.run_pos[q]; p < (uint64_t)h.run_pos[q] + h.run_len[q]; ++p) {
                const uint64_t b = h.rec_base[r] + p;
                seq[p] = "ACGT"[(h.word32(b / 16) >> (2 * (b % 16))) & 3u];
            }
        d = fnv(d, seq.data(), seq.size());
        fwrite(seq.data(), 1, seq.size(), dump);
    }
    fclose(dump);
    printf("%" PRIu64 " assemblies %zu records %" PRIu64 " bp digest %016" PRIx64 "\n", h.n_assemblies, h.rec_len.size(), h.total_bp, d);
    return 0;
}
