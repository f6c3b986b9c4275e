// common.hpp -- shared declarations of libseqwin_hip.so (host side).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <map>
#include <mutex>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/seqwin_hip.h"

// A/B switches.  Alternatives that lost their measurement and are the default for no input (the one-tile-per-workgroup radix
// pass, the extra tile shapes, the look-back form of the unsort, rocPRIM's run-length pass, the descent sweeps over the payloads,
// the "tails" tile plan, ...) are compiled OUT of the release library: their environment variables read as
// unset and their kernels are not instantiated.  `make -C seqwin_amd/csrc ab` builds libseqwin_hip_ab.so with -DSW_AB, where
// they are live (tests/tools/*_time.py, NOTES.md).  Switches that force a size-dependent DEFAULT path on small inputs (the
// suite's way of reaching the 15 000-genome branches), debug output and the fault injection stay in the release library.
#ifdef SW_AB
#define SW_AB_GETENV(name) getenv(name)
#else
#define SW_AB_GETENV(name) ((const char *)nullptr)
#endif

// Test hooks (r06): switches that force a path which is the DEFAULT only at sizes no unit test has (how the
// suite and the fuzzer reach the 15 000-genome branches), the safe alternatives the guards fall back to, the host packer's narrower
// forms, the fault injection and the lowered index bound.  They are live in the TEST library (`make -C seqwin_amd/csrc test` ->
// seqwin_amd/libseqwin_hip_test.so, -DSW_TEST_HOOKS: what tests/conftest.py and the fuzzer load) and in the A/B build; in the
// release library they read as unset -- same kernels, same code, the switches' strings are not even in the binary
// (tests/test_abi_cpu.py checks that) -- which leaves a deployment the "use" switches of DESIGN.md section 8a only.
#if defined(SW_TEST_HOOKS) || defined(SW_AB)
#define SW_TEST_GETENV(name) getenv(name)
#else
#define SW_TEST_GETENV(name) ((const char *)nullptr)
#endif

namespace sw {

// ---- error plumbing: C++ exceptions inside, int codes at the C ABI -------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void raise(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

// More items than 32-bit indices address on one device (2^32 - 2 minimizer occurrences of a shard, or rows of a slice: the
// permutation words of the sorts are 32-bit).  sw_build catches it and splits the job into more (logical) shards instead of failing
// where the reference succeeds (its indices are size_t, cpp/include/seqwin/graph.hpp:28-41); every other caller sees a RuntimeError.
// SEQWIN_AMD_OCC_CAP lowers the bound so that a unit-sized input takes that route (tests).
struct OccCapError : Error {
    uint64_t n;
    OccCapError(uint64_t n_, const std::string &m) : Error(SW_ERR_RUNTIME, m), n(n_) {}
};
uint64_t occ_cap();                                             // api.hip
[[noreturn]] void raise_occ_cap(uint64_t n, const char *what);  // api.hip
uint64_t last_occ_cap_n();   // api.hip: n of the OccCapError the calling thread's last failed C-ABI call ended with (0: none)

void set_last_error(const char *msg);
void note_occ_cap(uint64_t n);   // api.hip
// api.hip: an always-on order guard of the index build tripped (which: 0 node sort, 1 edge-key sort); counted (sw_order_guard_trips),
// logged as a WARNING through the log callback and on stderr.  The caller then re-sorts without the LDS-atomic ranking.
void order_guard_tripped(int which, uint32_t places);
void log_info(const char *fmt, ...);   // api.hip: an "info" line through the log callback (sw_set_log_callback; dropped if none)

template <class F> int guarded(F &&f)
{
    try {
        f();
        return SW_OK;
    } catch (const OccCapError &e) {
        set_last_error(e.what());
        note_occ_cap(e.n);
        return e.code;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return SW_ERR_RUNTIME;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return SW_ERR_RUNTIME;
    }
}

// ---- packed words of one assembly ------------------------------------------------------------
// Where the parsers' word buffers come from when the ingest streams to the device (r05): page-locked blocks that the DMA engine
// reads where the packer wrote them (api.hip: PinnedArena).  get() may decline (nullptr: the limit of pinned memory is reached).
struct WordArena {
    virtual uint64_t *get(size_t min_words, size_t *cap_words) = 0;
    virtual void put(uint64_t *p, size_t cap_words) = 0;
    virtual ~WordArena() {}
};

// A growing array of 64-bit words (what std::vector<uint64_t> was until r05) whose storage is malloc'd or an arena's.
class WordBuf {
    uint64_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    WordArena *home_ = nullptr;    // owner of p_ (nullptr: malloc)
    WordArena *arena_ = nullptr;   // where to ask first when growing
    void drop()
    {
        if (p_) {
            if (home_) home_->put(p_, cap_);
            else free(p_);
        }
        p_ = nullptr;
        n_ = cap_ = 0;
        home_ = nullptr;
    }
    void grow(size_t want)
    {
        size_t ncap = std::max<size_t>(std::max(want, cap_ + cap_ / 2), 1024), got = 0;
        uint64_t *q = arena_ ? arena_->get(ncap, &got) : nullptr;
        WordArena *home = q ? arena_ : nullptr;
        if (!q) {
            q = (uint64_t *)malloc(ncap * 8);
            got = ncap;
            if (!q) throw std::bad_alloc();
        }
        if (n_) memcpy(q, p_, n_ * 8);
        const size_t n = n_;
        drop();
        p_ = q;
        n_ = n;
        cap_ = got;
        home_ = home;
    }

public:
    WordBuf() = default;
    explicit WordBuf(WordArena *arena) : arena_(arena) {}
    WordBuf(WordBuf &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), home_(o.home_), arena_(o.arena_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; o.home_ = nullptr; }
    WordBuf &operator=(WordBuf &&o) noexcept
    {
        if (this != &o) {
            drop();
            p_ = o.p_; n_ = o.n_; cap_ = o.cap_; home_ = o.home_; arena_ = o.arena_;
            o.p_ = nullptr; o.n_ = o.cap_ = 0; o.home_ = nullptr;
        }
        return *this;
    }
    WordBuf(const WordBuf &) = delete;
    WordBuf &operator=(const WordBuf &) = delete;
    ~WordBuf() { drop(); }
    void push_back(uint64_t x)
    {
        if (n_ == cap_) grow(n_ + 1);
        p_[n_++] = x;
    }
    void push_back2(uint64_t x, uint64_t y)
    {
        if (n_ + 2 > cap_) grow(n_ + 2);
        p_[n_] = x;
        p_[n_ + 1] = y;
        n_ += 2;
    }
    void reserve(size_t c) { if (c > cap_) grow(c); }
    void clear() { n_ = 0; }
    void set_arena(WordArena *a) { arena_ = a; }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    bool empty() const { return n_ == 0; }
    bool has_storage() const { return p_ != nullptr; }
    bool in_arena() const { return home_ != nullptr; }
    const uint64_t *data() const { return p_; }
    uint64_t operator[](size_t i) const { return p_[i]; }
};

// ---- host-side result of FASTA ingest (k-independent) ---------------------------------------
// Bases are packed 2 bits each (A0 C1 G2 T/U3; invalid bases are stored as 0 and described by
// the run table), 16 per uint32 word, base i of the stream in bits [2*(i%16), 2*(i%16)+2) of word
// i/16.  Every record starts on a 32-base boundary.
struct HostBatch {
    uint64_t n_assemblies = 0;
    uint64_t total_bp = 0;
    std::vector<uint32_t> record_offsets;   // [n_assemblies + 1]
    std::string ids_blob;                   // NUL-terminated ids in record order
    std::vector<uint32_t> rec_len;          // [R]
    std::vector<uint64_t> rec_base;         // [R] offset of the record's first base in the packed stream
    std::vector<uint32_t> rec_run_off;      // [R + 1] index of the record's first valid run
    std::vector<uint32_t> run_pos, run_len; // maximal runs of valid bases, in (record, pos) order
    // 2-bit stream: one chunk per assembly, never concatenated on the host.  When the ingest streams to a ChunkSink
    // the chunks are gone by the time it returns (chunks[i] empty); chunk_word0 always describes the layout.
    std::vector<WordBuf> chunks;                 // [n_assemblies]
    std::vector<uint64_t> chunk_word0;           // [n_assemblies + 1] index of the chunk's first 64-bit word in the stream
    uint64_t packed_words32() const { return (chunk_word0.empty() ? 0 : chunk_word0.back()) * 2 + 8; }   // + read slack
    uint32_t word32(uint64_t i) const            // test / debug accessor (non-streamed batches)
    {
        const uint64_t w = i >> 1;
        const size_t c = std::upper_bound(chunk_word0.begin(), chunk_word0.end(), w) - chunk_word0.begin() - 1;
        if (c >= chunks.size() || w - chunk_word0[c] >= chunks[c].size()) return 0;
        return (uint32_t)(chunks[c][w - chunk_word0[c]] >> (32 * (i & 1)));
    }
};

// Receives the packed chunks in assembly order while later files are still being parsed (pipelined upload).
struct ChunkSink {
    virtual void begin(uint64_t expected_words64) = 0;                                   // estimate, before the first chunk
    // word_off: the chunk's place in the stream.  The sink may keep the buffer (an arena's, until its copy has left the host)
    // or leave it to the caller, who recycles it.
    virtual void chunk(WordBuf &words, uint64_t word_off) = 0;
    virtual WordArena *arena() { return nullptr; }                                       // where the parsers' buffers should come from
    virtual ~ChunkSink() {}
};



// host_ingest.cpp
void ingest_fasta(const char *const *paths, size_t n_paths, uint64_t n_cpu, HostBatch &out, ChunkSink *sink = nullptr);
void check_kw(uint64_t k, uint64_t w);
size_t usable_cpus();   // host_ingest.cpp: hardware threads narrowed by the affinity mask and the cgroup CPU quota

}  // namespace sw
