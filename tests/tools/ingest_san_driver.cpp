// Sanitizer driver for the host FASTA reader / 2-bit packer (seqwin_amd/csrc/host_ingest.cpp): built by
// `make -C seqwin_amd/csrc asan` with -fsanitize=address,undefined, run by tests/test_abi_cpu.py on hostile bytes.
// CPU only (sanitizers never run on the GPU box).  Prints one line per file set: a digest of everything the reader
// produced (record table, ids, decoded bases), which the test compares with the regular library's.
//   ingest_san <n_cpu> <dump file> <path>...   exit code 0 = parsed, 3 = the reader refused the input (message on stderr)
// INGEST_SAN_SINK=1: the streaming form sw_build uses -- the parsers' word buffers come from an arena (here: malloc'd blocks that are
// poisoned when they come back), the chunks go to a sink in assembly order (here: into one host vector, every buffer held for a
// few chunks as if its DMA were still in flight), the parsers stay within their window of the sink (`make tsan`: the same under
// ThreadSanitizer).
// The dump holds record_offsets | ids blob | record lengths | decoded bases of every record ('N' = invalid base).
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>

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

struct TestArena : sw::WordArena {
    std::mutex mu;
    size_t live = 0, handed = 0;
    uint64_t *get(size_t min_words, size_t *cap_words) override
    {
        std::lock_guard<std::mutex> lock(mu);
        if (++handed % 5 == 0) return nullptr;   // (declines now and then: the parser mallocs, the sink copies)
        ++live;
        *cap_words = min_words + handed % 3;
        return (uint64_t *)malloc(*cap_words * 8);
    }
    void put(uint64_t *p, size_t cap_words) override
    {
        memset(p, 0xA5, cap_words * 8);
        free(p);
        std::lock_guard<std::mutex> lock(mu);
        --live;
    }
};
struct TestSink : sw::ChunkSink {
    TestArena arena_;
    std::vector<uint64_t> stream;
    std::deque<sw::WordBuf> in_flight;
    uint64_t next_off = 0;
    void begin(uint64_t expected_words64) override { stream.reserve(expected_words64); }
    void chunk(sw::WordBuf &words, uint64_t word_off) override
    {
        if (word_off != next_off) abort();   // chunks arrive in stream order
        stream.insert(stream.end(), words.data(), words.data() + words.size());
        next_off += words.size();
        if (words.in_arena()) {
            in_flight.push_back(std::move(words));
            if (in_flight.size() > 3) in_flight.pop_front();
        }
    }
    sw::WordArena *arena() override { return &arena_; }
};

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const uint64_t n_cpu = strtoull(argv[1], nullptr, 10);
    FILE *dump = fopen(argv[2], "wb");
    if (!dump) return 2;
    sw::HostBatch h;
    TestSink sink;
    const bool streaming = getenv("INGEST_SAN_SINK") != nullptr;
    try {
        sw::ingest_fasta(argv + 3, (size_t)(argc - 3), n_cpu, h, streaming ? &sink : nullptr);
    } catch (const sw::Error &e) {
        fprintf(stderr, "refused (%d): %s\n", e.code, e.what());
        return 3;
    }
    sink.in_flight.clear();
    if (streaming && (sink.arena_.live != 0 || sink.next_off != (h.chunk_word0.empty() ? 0 : h.chunk_word0.back()))) {
        fprintf(stderr, "ERROR: %zu arena blocks not returned, %llu of %llu words streamed\n", sink.arena_.live,
                (unsigned long long)sink.next_off, (unsigned long long)(h.chunk_word0.empty() ? 0 : h.chunk_word0.back()));
        return 4;
    }
    auto word32 = [&](uint64_t i) -> uint32_t {
        if (!streaming) return h.word32(i);
        const uint64_t w = i >> 1;
        return w < sink.stream.size() ? (uint32_t)(sink.stream[w] >> (32 * (i & 1))) : 0u;
    };
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
                seq[p] = "ACGT"[(word32(b / 16) >> (2 * (b % 16))) & 3u];
            }
        d = fnv(d, seq.data(), seq.size());
        fwrite(seq.data(), 1, seq.size(), dump);
    }
    fclose(dump);
    printf("%" PRIu64 " assemblies %zu records %" PRIu64 " bp digest %016" PRIx64 "\n", h.n_assemblies, h.rec_len.size(), h.total_bp, d);
    return 0;
}
