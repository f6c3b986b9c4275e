// multi_choreography.cpp -- the host choreography of csrc/multi.hip (worker threads, rendezvous, peer pulls, buffer release
// points) and the caching pool of csrc/api.hip (block hand-over between threads and streams, spare events per device, the staged
// route) under ThreadSanitizer, with NON-ZERO sizes, on the mock HIP runtime (hip_mock.cpp).
//
// multi.hip is compiled a second time with its engine calls (sw_batch_from_fasta ... sw_index_edge_hash_attach) renamed
// (multi_mock_engine.h) and the fake engine below behind them.  A fake "kernel" is a function enqueued on the worker's stream with
// hip_mock_enqueue(): on the stream's own thread it READS its input buffers -- every 32-bit word must carry the stamp of
// (this job, the kind of buffer it is supposed to be) -- and WRITES its outputs with their stamp.  So
//   * a buffer pulled before its producer's stream finished, or released -- and handed to another worker by the pool -- while a
//     peer's pull or an own kernel is still queued, is a data race ThreadSanitizer reports, and
//   * bytes that are stale, unwritten or of another kind fail the stamp check (g_bad).
// Everything else is the real code: build_multi_device's workers and rendezvous, pull() (peer copies / the staged route), the
// DevArray allocations under StreamScope, dev_alloc / dev_free with their events.  Counts are random per (source, owner) pair, zero
// included; MOCK_FAIL (per mille of engine calls) injects errors: the broken rendezvous and the order of the clean-up.
// usage: tsan_multi [jobs] [seed]          env: HIP_MOCK_DEVICES (default 4), HIP_MOCK_NO_PEER, SEQWIN_MULTI_NO_P2P, MOCK_FAIL
#include "multi_mock_engine.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../seqwin_amd/csrc/device.hpp"
#include "hip_mock.h"

static std::atomic<uint64_t> g_bad{0}, g_kernels{0}, g_injected{0};
static std::atomic<uint64_t> g_rng{0x9E3779B97F4A7C15ull};
static int g_fail_permille = 0;
static std::atomic<uint64_t> g_job{0};

static uint64_t mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
static uint64_t rnd() { return mix(g_rng.fetch_add(0x9E3779B97F4A7C15ull)); }

enum Kind { K_PACKED = 1, K_TMP, K_ROWS, K_RANKS, K_KMERS, K_NODES, K_HASHES, K_KEYS, K_CAND, K_EDGES, K_REQ, K_ANS };
static uint32_t stamp(unsigned kind) { return (uint32_t)mix(g_job.load() * 64 + kind) | 1u; }

// ---- fake kernels ---------------------------------------------------------------------------------------------------
struct Range {
    const void *p;
    size_t words32;
    unsigned kind;
};
struct Kernel {
    std::vector<Range> in, out;
    const char *what;
    uint32_t stamps[16];
};
static void run_kernel(void *arg)
{
    Kernel *kn = (Kernel *)arg;
    if ((rnd() & 7) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rnd() % 300));
    for (const Range &r : kn->in) {
        const uint32_t *p = (const uint32_t *)r.p, want = kn->stamps[r.kind];
        for (size_t i = 0; i < r.words32; ++i)
            if (p[i] != want) {
                if (g_bad.fetch_add(1) < 8)
                    fprintf(stderr, "BAD DATA read by %s: word %zu of %zu (kind %u) is %08x, not %08x\n", kn->what, i, r.words32, r.kind, p[i], want);
                break;
            }
    }
    for (const Range &r : kn->out) {
        uint32_t *p = (uint32_t *)r.p;
        const uint32_t v = kn->stamps[r.kind];
        for (size_t i = 0; i < r.words32; ++i) p[i] = v;
    }
    ++g_kernels;
    delete kn;
}
static void launch(void *st, const char *what, std::vector<Range> in, std::vector<Range> out)
{
    Kernel *kn = new Kernel{std::move(in), std::move(out), what, {}};
    for (unsigned k = 0; k < 16; ++k) kn->stamps[k] = stamp(k);   // (of the job that launches: a late kernel of a failed job keeps its own)
    hip_mock_enqueue((hipStream_t)st, run_kernel, kn);
}

// ---- the fake engine (prototypes: include/seqwin_hip.h through the macros of multi_mock_engine.h) ---------------------------
struct FakeBatch {
    uint64_t n_asm;
    std::vector<uint32_t> offs;
    std::string ids;
    sw::DevArray<uint32_t> packed;
};
struct FakeOcc {
    FakeBatch *b;
    uint64_t n_occ, n_keys = 0, n_cand = 0;
};
struct FakeSlice {
    std::vector<uint64_t> req_cnt;
    uint64_t n_req = 0;
};
static std::mutex g_slice_mu;
static std::map<const sw_index *, FakeSlice> g_slices;

static bool inject(const char *where)
{
    if (g_fail_permille && (int)(rnd() % 1000) < g_fail_permille) {
        sw::set_last_error((std::string("injected failure in ") + where).c_str());
        ++g_injected;
        return true;
    }
    return false;
}
#define MAYBE_FAIL(where) do { if (inject(where)) return SW_ERR_RUNTIME; } while (0)

extern "C" {

int sw_batch_from_fasta(const char *const *, size_t n, uint64_t, sw_batch **out)
{
    MAYBE_FAIL("sw_batch_from_fasta");
    FakeBatch *b = new FakeBatch;
    b->n_asm = n;
    b->offs.assign(1, 0);
    for (size_t a = 0; a < n; ++a) {
        b->offs.push_back(b->offs.back() + 1 + (uint32_t)(rnd() % 3));
        b->ids += "r" + std::to_string(a) + '\0';
    }
    // the streaming ingest: a fresh pool block written by ANOTHER stream than the worker's (the upload ring's DMA), then fenced
    b->packed.alloc(1 + rnd() % 8192);
    static thread_local hipStream_t up = nullptr;
    static thread_local int up_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!up || up_dev != dev) {
        if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) return SW_ERR_DEVICE;
        up_dev = dev;
    }
    launch(up, "upload", {}, {{b->packed.p, b->packed.n, K_PACKED}});
    (void)hipStreamSynchronize(up);
    *out = (sw_batch *)b;
    return SW_OK;
}
int sw_batch_info(const sw_batch *b_, uint64_t *na, uint64_t *nr, uint64_t *bp, uint64_t *bytes)
{
    const FakeBatch *b = (const FakeBatch *)b_;
    *na = b->n_asm;
    *nr = b->offs.back();
    *bp = 1000 * (uint64_t)b->offs.back();
    *bytes = b->packed.bytes();
    return SW_OK;
}
int sw_batch_records(const sw_batch *b_, uint32_t *offs, char *blob, uint64_t cap, uint64_t *need)
{
    const FakeBatch *b = (const FakeBatch *)b_;
    memcpy(offs, b->offs.data(), b->offs.size() * 4);
    *need = b->ids.size();
    if (blob && cap >= b->ids.size()) memcpy(blob, b->ids.data(), b->ids.size());
    return SW_OK;
}
void sw_batch_free(sw_batch *b) { delete (FakeBatch *)b; }

int sw_occ_sketch(const sw_batch *b_, uint64_t, uint64_t, void *st, sw_occ **out)
{
    MAYBE_FAIL("sw_occ_sketch");
    FakeBatch *b = (FakeBatch *)b_;
    FakeOcc *o = new FakeOcc{b, 0};
    sw::StreamScope scope((hipStream_t)st);
    sw::DevArray<uint32_t> stage(1 + rnd() % 16384);   // a temporary: released while the "kernel" that uses it is still queued
    launch(st, "sketch", {{b->packed.p, b->packed.n, K_PACKED}}, {{stage.p, stage.n, K_TMP}});
    o->n_occ = rnd() % 6 == 0 ? 0 : rnd() % 6000;
    *out = (sw_occ *)o;
    return SW_OK;
}
int sw_occ_sketch_paths(const char *const *paths, size_t n, uint64_t k, uint64_t w, uint64_t n_cpu, uint64_t, void *st, sw_batch **b, sw_occ **o)
{
    const int rc = sw_batch_from_fasta(paths, n, n_cpu, b);
    return rc != SW_OK ? rc : sw_occ_sketch(*b, k, w, st, o);
}
int sw_occ_size(const sw_occ *o, uint64_t *n, double *ms)
{
    *n = ((const FakeOcc *)o)->n_occ;
    *ms = 0.1;
    return SW_OK;
}
void sw_occ_free(sw_occ *o) { delete (FakeOcc *)o; }

int sw_occ_partition(const sw_occ *o_, const uint64_t *, uint64_t n_bounds, uint64_t, void *rows, void *, uint64_t *counts, void *st)
{
    MAYBE_FAIL("sw_occ_partition");
    const FakeOcc *o = (const FakeOcc *)o_;
    const uint32_t P = (uint32_t)n_bounds + 1;
    uint64_t left = o->n_occ;
    for (uint32_t q = 0; q < P; ++q) {
        counts[q] = q + 1 == P ? left : (rnd() % 4 == 0 ? 0 : rnd() % (left + 1));
        left -= counts[q];
    }
    launch(st, "partition", {}, {{rows, (size_t)o->n_occ * 4, K_ROWS}});
    return SW_OK;
}

int sw_slice_build(const void *rows, uint64_t n, uint64_t, const uint32_t *, const uint8_t *, uint64_t, void *rank_out, void *st, sw_index **out)
{
    MAYBE_FAIL("sw_slice_build");
    sw_index *ix = new sw_index;
    (void)hipGetDevice(&ix->device);
    sw::StreamScope scope((hipStream_t)st);
    ix->n_kmers = n;
    ix->n_nodes = n ? 1 + rnd() % n : 0;
    ix->kmers.alloc(n);
    ix->nodes.alloc(ix->n_nodes);
    ix->ranks_marked = true;
    ix->last_stream = (hipStream_t)st;
    sw::DevArray<uint32_t> sort_buf(4 * n + 1);   // the sort's buffers: pool blocks released under this stream, still in use
    launch(st, "slice_build", {{rows, (size_t)n * 4, K_ROWS}},
           {{sort_buf.p, sort_buf.n, K_TMP}, {ix->kmers.p, (size_t)n * 2, K_KMERS}, {ix->nodes.p, (size_t)ix->n_nodes * 10, K_NODES}, {rank_out, (size_t)n, K_RANKS}});
    {
        std::lock_guard<std::mutex> lock(g_slice_mu);
        g_slices[ix] = FakeSlice{};
    }
    *out = ix;
    return SW_OK;
}
int sw_index_node_hashes(const sw_index *ix, void *dst, void *st)
{
    MAYBE_FAIL("sw_index_node_hashes");
    launch(st, "node_hashes", {{ix->nodes.p, (size_t)ix->n_nodes * 10, K_NODES}}, {{dst, (size_t)ix->n_nodes * 2, K_HASHES}});
    return SW_OK;
}
int sw_occ_adjacency_pairs(const sw_occ *o_, const void *rank_by_row, const uint64_t *, uint64_t n_owners, uint64_t, const uint64_t *, uint64_t, void *keys,
                           uint64_t *counts, uint64_t *cand_counts, uint64_t *key_bits, void *st)
{
    MAYBE_FAIL("sw_occ_adjacency_pairs");
    FakeOcc *o = (FakeOcc *)o_;
    o->n_keys = o->n_occ ? rnd() % (o->n_occ + 1) : 0;
    o->n_cand = rnd() % 3 == 0 ? rnd() % 200 : 0;
    uint64_t left = o->n_keys, cleft = o->n_cand;
    for (uint32_t q = 0; q < n_owners; ++q) {
        counts[q] = q + 1 == n_owners ? left : rnd() % (left + 1);
        left -= counts[q];
        cand_counts[q] = q + 1 == n_owners ? cleft : rnd() % (cleft + 1);
        cleft -= cand_counts[q];
    }
    key_bits[0] = 20;
    key_bits[1] = 24;
    launch(st, "adjacency", {{rank_by_row, (size_t)o->n_occ, K_RANKS}}, {{keys, (size_t)o->n_keys * 2, K_KEYS}});
    return SW_OK;
}
int sw_occ_candidates(const sw_occ *o_, void *rows, void *st)
{
    MAYBE_FAIL("sw_occ_candidates");
    const FakeOcc *o = (const FakeOcc *)o_;
    launch(st, "candidates", {}, {{rows, (size_t)o->n_cand * 4, K_CAND}});
    return SW_OK;
}
int sw_slice_edges_pairs(sw_index *ix, void *keys, uint64_t m, const void *cand, uint64_t c, uint64_t, uint64_t, uint64_t, uint64_t, const void *table,
                         const uint64_t *node_base, uint64_t n_owners, uint64_t pad, void *st)
{
    MAYBE_FAIL("sw_slice_edges_pairs");
    sw::StreamScope scope((hipStream_t)st);
    ix->n_edges = m ? 1 + rnd() % m : 0;
    ix->edges.alloc(ix->n_edges);
    ix->edges_hold_ranks = table == nullptr;
    std::vector<Range> in = {{keys, (size_t)m * 2, K_KEYS}, {cand, (size_t)c * 4, K_CAND}};
    if (table)
        for (uint64_t o = 0; o < n_owners; ++o) in.push_back({(const char *)table + o * pad * 8, (size_t)(node_base[o + 1] - node_base[o]) * 2, K_HASHES});
    launch(st, "slice_edges", in, {{ix->edges.p, (size_t)ix->n_edges * 6, K_EDGES}});
    return SW_OK;
}
int sw_index_edge_hash_requests(sw_index *ix, const uint64_t *node_base, uint64_t n_owners, uint64_t *counts, uint64_t *n_requests, void *)
{
    MAYBE_FAIL("sw_index_edge_hash_requests");
    std::lock_guard<std::mutex> lock(g_slice_mu);
    FakeSlice &fs = g_slices[ix];
    fs.req_cnt.assign(n_owners, 0);
    fs.n_req = 0;
    for (uint64_t o = 0; o < n_owners; ++o) {
        const uint64_t nn = node_base[o + 1] - node_base[o];
        counts[o] = fs.req_cnt[o] = (nn && ix->n_edges) ? rnd() % (std::min<uint64_t>(nn, 2 * ix->n_edges) + 1) : 0;
        fs.n_req += counts[o];
    }
    *n_requests = fs.n_req;
    return SW_OK;
}
int sw_index_edge_hash_request_rows(const sw_index *ix, void *dst, void *st)
{
    MAYBE_FAIL("sw_index_edge_hash_request_rows");
    uint64_t n;
    {
        std::lock_guard<std::mutex> lock(g_slice_mu);
        n = g_slices[ix].n_req;
    }
    launch(st, "request_rows", {{ix->edges.p, (size_t)ix->n_edges * 6, K_EDGES}}, {{dst, (size_t)n, K_REQ}});
    return SW_OK;
}
int sw_index_node_hash_lookup(const sw_index *ix, const void *ranks, uint64_t n, void *hashes, void *st)
{
    MAYBE_FAIL("sw_index_node_hash_lookup");
    launch(st, "hash_lookup", {{ranks, (size_t)n, K_REQ}, {ix->nodes.p, (size_t)ix->n_nodes * 10, K_NODES}}, {{hashes, (size_t)n * 2, K_ANS}});
    return SW_OK;
}
int sw_index_edge_hash_attach(sw_index *ix, const void *replies, uint64_t n, void *st)
{
    MAYBE_FAIL("sw_index_edge_hash_attach");
    launch(st, "hash_attach", {{replies, (size_t)n * 2, K_ANS}, {ix->edges.p, (size_t)ix->n_edges * 6, K_EDGES}}, {{ix->edges.p, (size_t)ix->n_edges * 6, K_EDGES}});
    ix->edges_hold_ranks = false;
    return SW_OK;
}

}  // extern "C"

int main(int argc, char **argv)
{
    const long jobs = argc > 1 ? atol(argv[1]) : 200;
    if (argc > 2) g_rng = strtoull(argv[2], nullptr, 10) * 0x9E3779B97F4A7C15ull + 12345;
    if (const char *e = getenv("MOCK_FAIL")) g_fail_permille = atoi(e);
    int n_dev = 0;
    (void)hipGetDeviceCount(&n_dev);
    long ok = 0, failed = 0, wrong = 0;
    std::vector<std::string> names;
    for (int a = 0; a < 64; ++a) names.push_back("a" + std::to_string(a) + ".fa");
    for (long j = 0; j < jobs; ++j) {
        g_job = (uint64_t)j + 1;
        const uint32_t P = 2 + (uint32_t)(rnd() % 7);
        std::vector<int> devs;
        const int mode = (int)(rnd() % 3);   // 0: distinct devices round robin, 1: random (repeats), 2: all on one card
        for (uint32_t p = 0; p < P; ++p) devs.push_back(mode == 0 ? (int)(p % n_dev) : mode == 1 ? (int)(rnd() % n_dev) : 0);
        const size_t n_paths = rnd() % 5 == 0 ? 1 + rnd() % P : P + rnd() % 40;
        std::vector<const char *> paths;
        for (size_t a = 0; a < n_paths; ++a) paths.push_back(names[a].c_str());
        if (rnd() % 2) setenv("SEQWIN_DIST_HASH_ROUTE", "requests", 1);
        else setenv("SEQWIN_DIST_HASH_ROUTE", "table", 1);
        sw::MultiGraph mg;
        try {
            sw::build_multi_device(paths.data(), n_paths, 21, 200, 8, devs, mg, rnd() % 4 == 0 ? 1000 : 0);
            ++ok;
            // the result: every slice's arrays carry this job's stamps (read on the NULL stream of the slice's device)
            if (mg.slices.size() != std::min<size_t>(P, n_paths) || mg.record_offsets.size() != n_paths + 1 || mg.n_assemblies != n_paths) ++wrong;
            for (auto &s : mg.slices) {
                if (!s) { ++wrong; continue; }
                (void)hipSetDevice(s->device);
                launch(nullptr, "result", {{s->kmers.p, (size_t)s->n_kmers * 2, K_KMERS}, {s->nodes.p, (size_t)s->n_nodes * 10, K_NODES},
                                           {s->edges.p, (size_t)s->n_edges * 6, K_EDGES}}, {});
                (void)hipStreamSynchronize(nullptr);
                if (s->edges_hold_ranks) ++wrong;
            }
        } catch (const sw::Error &e) {
            ++failed;
            if (!g_fail_permille || !strstr(e.what(), "injected")) {
                fprintf(stderr, "job %ld: unexpected error: %s\n", j, e.what());
                ++wrong;
            }
        }
        {
            std::lock_guard<std::mutex> lock(g_slice_mu);
            g_slices.clear();
        }
        mg.slices.clear();
        (void)hipSetDevice(0);
        if (j % 50 == 49) sw::dev_pool_trim();   // hipFree of every cached block: nothing may still be in use
    }
    uint64_t st[4];
    hip_mock_stats(st);
    printf("%ld jobs: %ld built, %ld failed (%llu injected errors), %ld wrong results, %llu stamp mismatches; %llu fake kernels, %llu copies, %llu peer copies, "
           "%llu hipFree; pool %.1f MiB\n", jobs, ok, failed, (unsigned long long)g_injected.load(), wrong, (unsigned long long)g_bad.load(),
           (unsigned long long)g_kernels.load(), (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3], sw::dev_pool_bytes() / 1048576.0);
    return (wrong || g_bad.load()) ? 1 : 0;
}
