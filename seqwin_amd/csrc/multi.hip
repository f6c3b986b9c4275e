// multi.hip -- ONE sw_build over several GPUs of the node, inside the one process the unmodified `seqwin` CLI runs in.
//
// The reference parallelises inside the one call: seqwin::build splits the assemblies over n_cpu worker threads
// (cpp/src/seqwin/build.cpp:342-367) and merges the per-thread graphs (merge_thread_graphs,
// cpp/src/seqwin/build_internals.cpp:295-392).  Here a "worker" is a GPU: SEQWIN_DEVICES=0,1,... (or "all") makes sw_build
// start one host thread + one HIP stream per listed device, give device g the assemblies of "thread g" (the same partition
// formula), and run the tuple-exchange form of the sharded build -- the choreography of seqwin_amd/dist.py, with the same
// slice kernels behind the same C-ABI entry points, but with the exchanges done by peer-to-peer copies (hipMemcpyPeerAsync
// over xGMI: the OWNER pulls its piece from every source; a direct copy per pair uses all links of a GPU at once) instead of
// RCCL collectives between processes.  The result is the concatenation of the slices in owner order = the single-device
// arrays, bit for bit (shard-count invariance, reference tests/smoke/test_graph.py:67-127).
//
// A device may be listed more than once (SEQWIN_DEVICES=0,0,0): the shards are then logical and share one card -- that is how
// the path is tested on a one-GPU box (tests/test_gpu_parity.py); on real multi-GPU hardware it is UNMEASURED so far.
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "device.hpp"

namespace sw {
namespace {

typedef unsigned __int128 u128;

uint64_t isqrt128(u128 v)   // floor(sqrt(v)) for v < 2^126, bit by bit (math.isqrt of dist.py)
{
    u128 r = 0;
    for (int b = 62; b >= 0; --b) {
        const u128 t = r | ((u128)1 << b);
        if (t * t <= v) r = t;
    }
    return (uint64_t)r;
}

// one stream per (device, worker slot), kept for the life of the process: blocks of the caching pool remember the stream they
// were released under, so a worker's stream must outlive the build
hipStream_t worker_stream(int dev, uint32_t slot)
{
    static std::mutex &mu = *new std::mutex;
    static std::map<std::pair<int, uint32_t>, hipStream_t> &streams = *new std::map<std::pair<int, uint32_t>, hipStream_t>;
    std::lock_guard<std::mutex> lock(mu);
    hipStream_t &st = streams[std::make_pair(dev, slot)];
    if (!st) SW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

// seqwin_amd/dist.py: hash_bounds (node splitters only) and rank_bounds, in exact integer arithmetic
std::vector<uint64_t> hash_splitters(uint32_t P)
{
    std::vector<uint64_t> nb;
    for (uint32_t j = 1; j < P; ++j) nb.push_back((uint64_t)((((u128)j << 64) + P - 1) / P));   // ceil(j * 2^64 / P)
    return nb;
}
std::vector<uint64_t> rank_splitters(uint32_t P, uint64_t total_nodes)
{
    std::vector<uint64_t> rb;
    for (uint32_t j = 1; j < P; ++j)
        rb.push_back(total_nodes - isqrt128(((u128)(P - j) * total_nodes * total_nodes) / P));   // the j/P quantile of min(u, v)
    return rb;
}

// a barrier that can be broken: a worker that fails wakes everybody, and every later arrival fails as well
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    unsigned n, waiting = 0;
    uint64_t generation = 0;
    bool broken = false;
    explicit Rendezvous(unsigned n_) : n(n_) {}
    void arrive()
    {
        std::unique_lock<std::mutex> lock(mu);
        if (broken) raise(SW_ERR_RUNTIME, "multi-device build: another device failed");
        const uint64_t gen = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return;
        }
        cv.wait(lock, [&] { return generation != gen || broken; });
        if (generation == gen) raise(SW_ERR_RUNTIME, "multi-device build: another device failed");
    }
    void fail()
    {
        std::lock_guard<std::mutex> lock(mu);
        broken = true;
        cv.notify_all();
    }
};

void ck(int rc)   // a C-ABI call made from a worker: its message lives in this thread's sw_last_error
{
    if (rc == SW_OK) return;
    if (const uint64_t n = last_occ_cap_n()) throw OccCapError(n, sw_last_error());   // (sw_build splits the job further)
    raise(rc, "%s", sw_last_error());
}

struct DevBuf {   // a plain device buffer of one worker (the pool is per device; released under the worker's stream scope)
    DevArray<unsigned char> a;
    void *p() const { return a.p; }
};

// what a worker publishes for the others (filled before a rendezvous, read after it)
struct Shared {
    // step 1
    std::vector<uint32_t> offs;            // record offsets of the shard's assemblies (local)
    std::string ids;
    uint64_t total_bp = 0;
    std::vector<uint64_t> cnt;             // tuples by owner
    const unsigned char *rows = nullptr;   // partitioned rows (16 B each), on this worker's device
    uint64_t n_occ = 0;
    // step 2 (as owner)
    uint64_t n_nodes = 0;
    int marked = 0;
    const unsigned char *ranks = nullptr;  // u32 slice-local rank of every received row
    const unsigned char *hashes = nullptr; // u64 node hashes of the slice
    // step 3 (as source)
    std::vector<uint64_t> acnt, ccnt;      // adjacency keys / candidate rows by edge owner
    const unsigned char *keys = nullptr, *cand = nullptr;
    uint64_t key_bits[2] = {0, 0};
    // step 5 (as edge owner) / 6 (as node owner)
    std::vector<uint64_t> req_cnt;
    const unsigned char *req = nullptr;
    std::vector<const unsigned char *> ans;   // per edge owner
};

// How bytes travel between two workers' devices (r05).  Before the workers start, peer access is queried and enabled for every
// ordered pair of distinct devices (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess): a pair that has it copies directly over
// xGMI (hipMemcpyPeerAsync on the owner's stream), a pair that has not -- or every pair under SEQWIN_MULTI_NO_P2P=1, which is how
// the fallback is exercised on one card -- goes through a pinned host buffer of the pulling worker, 32 MiB at a time.  The routes
// are logged once per build.
struct Routes {
    uint32_t P = 0;
    std::vector<uint8_t> staged;   // [dst worker * P + src worker]: 1 = through the host
    bool is_staged(uint32_t dst, uint32_t src) const { return staged[(size_t)dst * P + src] != 0; }
};
struct HostStage {   // one per worker, allocated on first use
    static constexpr size_t BYTES = 32u << 20;
    void *p = nullptr;
    ~HostStage() { if (p) (void)hipHostFree(p); }
};

void pull(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st, bool staged, HostStage &hs)
{
    if (!bytes) return;
    if (!staged) {
        if (dst_dev == src_dev) SW_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        else SW_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st));
        return;
    }
    // no peer access: device -> pinned host (on the source's device, blocking) -> device (this worker's stream).  The source's
    // bytes are complete (every pull follows a rendezvous behind the producer's stream synchronisation).
    // (portable: the source device's copy engine writes it, the pulling device's reads it -- two different devices on a real node)
    if (!hs.p) SW_HIP(hipHostMalloc(&hs.p, HostStage::BYTES, hipHostMallocPortable));
    for (size_t o = 0; o < bytes; o += HostStage::BYTES) {
        const size_t n = std::min(HostStage::BYTES, bytes - o);
        if (src_dev != dst_dev) SW_HIP(hipSetDevice(src_dev));
        const hipError_t e = hipMemcpy(hs.p, (const char *)src + o, n, hipMemcpyDeviceToHost);
        if (src_dev != dst_dev) SW_HIP(hipSetDevice(dst_dev));
        SW_HIP(e);
        SW_HIP(hipMemcpyAsync((char *)dst + o, hs.p, n, hipMemcpyHostToDevice, st));
        SW_HIP(hipStreamSynchronize(st));   // (the one buffer is reused by the next piece)
    }
}

Routes make_routes(const std::vector<int> &devs, std::string &summary)
{
    Routes R;
    R.P = (uint32_t)devs.size();
    R.staged.assign((size_t)R.P * R.P, 0);
    const bool forced = getenv("SEQWIN_MULTI_NO_P2P") && atoi(getenv("SEQWIN_MULTI_NO_P2P")) != 0;
    int home = 0;
    SW_HIP(hipGetDevice(&home));
    std::map<std::pair<int, int>, int> access;   // (device, peer) -> 1 direct, 0 staged
    size_t n_pairs = 0, n_direct = 0;
    for (uint32_t a = 0; a < R.P; ++a)
        for (uint32_t b = 0; b < R.P; ++b) {
            const int da = devs[a], db = devs[b];
            if (da == db) {
                R.staged[(size_t)a * R.P + b] = forced ? 1 : 0;
                continue;
            }
            auto it = access.find(std::make_pair(da, db));
            if (it == access.end()) {
                int can = 0;
                if (!forced) {
                    if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess) { (void)hipGetLastError(); can = 0; }
                    if (can) {
                        SW_HIP(hipSetDevice(da));
                        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
                        (void)hipGetLastError();
                    }
                }
                it = access.emplace(std::make_pair(da, db), can).first;
                ++n_pairs;
                n_direct += can ? 1 : 0;
            }
            R.staged[(size_t)a * R.P + b] = it->second ? 0 : 1;
        }
    SW_HIP(hipSetDevice(home));
    char buf[256];
    if (forced)
        snprintf(buf, sizeof buf, "every exchange staged through pinned host memory (SEQWIN_MULTI_NO_P2P=1)");
    else if (n_pairs == 0)
        snprintf(buf, sizeof buf, "logical shards of one device (device-to-device copies)");
    else
        snprintf(buf, sizeof buf, "peer access on %zu of %zu ordered device pairs%s", n_direct, n_pairs,
                 n_direct == n_pairs ? " (all exchanges direct, hipMemcpyPeerAsync)" : "; the others are staged through pinned host memory");
    summary = buf;
    return R;
}

}  // namespace

std::vector<int> devices_from_env()
{
    std::vector<int> devs;
    const char *e = getenv("SEQWIN_DEVICES");
    if (!e || !*e) return devs;
    int count = 0;
    SW_HIP(hipGetDeviceCount(&count));
    if (!strcmp(e, "all")) {
        for (int d = 0; d < count; ++d) devs.push_back(d);
    } else {
        const char *s = e;
        while (*s) {
            char *end = nullptr;
            const long v = strtol(s, &end, 10);
            if (end == s || v < 0 || v >= count) raise(SW_ERR_VALUE, "SEQWIN_DEVICES=%s: expected \"all\" or a comma-separated list of device indices below %d", e, count);
            devs.push_back((int)v);
            s = end;
            if (*s == ',') ++s;
            else if (*s) raise(SW_ERR_VALUE, "SEQWIN_DEVICES=%s: expected \"all\" or a comma-separated list of device indices below %d", e, count);
        }
    }
    if (devs.size() < 2) devs.clear();   // one device: the ordinary build on the current device
    return devs;
}

void build_multi_device(const char *const *paths, size_t n_paths, uint64_t k, uint64_t w, uint64_t n_cpu, std::vector<int> devs,
                        MultiGraph &out, uint64_t chunk_bp)
{
    if (devs.size() > n_paths) devs.resize(std::max<size_t>(1, n_paths));   // (build.cpp:344-346: no more workers than assemblies)
    const uint32_t P = (uint32_t)devs.size();
    int home = 0;
    SW_HIP(hipGetDevice(&home));
    // contiguous ranges, the first `rem` workers get one more (build.cpp:350-356)
    std::vector<size_t> first(P + 1, 0);
    {
        const size_t base = n_paths / P, rem = n_paths % P;
        for (uint32_t t = 0; t < P; ++t) first[t + 1] = first[t] + base + (t < rem ? 1 : 0);
    }
    const std::vector<uint64_t> nb = hash_splitters(P);
    std::vector<Shared> sh(P);
    std::vector<std::unique_ptr<sw_index>> slices(P);
    Rendezvous meet(P);
    std::mutex err_mu;
    std::exception_ptr err;
    // job-wide values a rendezvous makes known (written by worker 0 between two rendezvous, read by all after the second)
    std::vector<uint32_t> record_offsets;
    std::vector<uint64_t> rec_base(P + 1, 0), node_base, rb;
    bool by_request = false;
    uint64_t pad = 1;

    // What a worker holds on its device.  Owned HERE, not by the worker thread: a worker that fails only synchronises
    // its own stream, while its peers may still have pulls from its buffers in flight until they meet the broken rendezvous -- so
    // nothing a peer can read is released before every thread has been joined (the cleanup loop below).  On the success path the
    // workers release their buffers themselves, each behind the rendezvous that makes it safe, to keep the footprint down.
    struct WorkerState {
        hipStream_t st = nullptr;
        sw_batch *batch = nullptr;
        sw_occ *occ = nullptr;
        sw_index *ix = nullptr;
        DevBuf rows, ranks_by_row, keys, cand, r_rows, r_ranks, hashes, table, r_keys, r_cand, req, got, answers, replies;
        HostStage stage;
    };
    std::vector<WorkerState> wstate(P);
    std::string route_summary;
    const Routes routes = make_routes(devs, route_summary);
    {
        std::string list;
        for (uint32_t q = 0; q < P; ++q) list += (q ? "," : "") + std::to_string(devs[q]);
        log_info("multi-device build: %u workers on devices [%s]; %s", P, list.c_str(), route_summary.c_str());
    }

    auto worker = [&](uint32_t p) {
        WorkerState &ws = wstate[p];
        hipStream_t &st = ws.st;
        sw_batch *&batch = ws.batch;
        sw_occ *&occ = ws.occ;
        sw_index *&ix = ws.ix;
        DevBuf &rows = ws.rows, &ranks_by_row = ws.ranks_by_row, &keys = ws.keys, &cand = ws.cand, &r_rows = ws.r_rows, &r_ranks = ws.r_ranks,
               &hashes = ws.hashes, &table = ws.table, &r_keys = ws.r_keys, &r_cand = ws.r_cand, &req = ws.req, &got = ws.got,
               &answers = ws.answers, &replies = ws.replies;
        try {
            const int dev = devs[p];
            SW_HIP(hipSetDevice(dev));
            st = worker_stream(dev, p);
            Shared &me = sh[p];
            // ---- 1: ingest, sketch, partition by hash range ------------------------------------------------------------
            const uint64_t cpu_share = std::max<uint64_t>(1, n_cpu / P);
            if (chunk_bp) ck(sw_occ_sketch_paths(paths + first[p], first[p + 1] - first[p], k, w, cpu_share, chunk_bp, st, &batch, &occ));
            else ck(sw_batch_from_fasta(paths + first[p], first[p + 1] - first[p], cpu_share, &batch));
            {
                uint64_t na = 0, nr = 0, bp = 0, bytes = 0, idb = 0;
                ck(sw_batch_info(batch, &na, &nr, &bp, &bytes));
                me.total_bp = bp;
                me.offs.resize(na + 1);
                ck(sw_batch_records(batch, me.offs.data(), nullptr, 0, &idb));
                me.ids.resize(idb);
                if (idb) ck(sw_batch_records(batch, me.offs.data(), &me.ids[0], idb, &idb));
            }
            if (!chunk_bp) ck(sw_occ_sketch(batch, k, w, st, &occ));
            double sk_ms = 0;
            ck(sw_occ_size(occ, &me.n_occ, &sk_ms));
            meet.arrive();                                   // (A) every shard's record count is known
            if (p == 0) {
                record_offsets.assign(1, 0);
                uint64_t total = 0;
                for (uint32_t q = 0; q < P; ++q) {
                    rec_base[q] = total;
                    for (size_t a = 1; a < sh[q].offs.size(); ++a) {
                        const uint64_t v = total + sh[q].offs[a];
                        if (v > 0xFFFFFFFFull) raise(SW_ERR_RUNTIME, "Total number of FASTA records exceeds uint32 range");
                        record_offsets.push_back((uint32_t)v);
                    }
                    total += sh[q].offs.empty() ? 0 : sh[q].offs.back();
                }
                rec_base[P] = total;
            }
            meet.arrive();                                   // (B) record_offsets / rec_base are set
            {
                StreamScope scope(st);
                rows.a.alloc((size_t)me.n_occ * 16);
            }
            me.cnt.assign(P, 0);
            ck(sw_occ_partition(occ, nb.data(), nb.size(), rec_base[p], rows.p(), nullptr, me.cnt.data(), st));
            me.rows = rows.a.p;
#ifndef SW_MULTI_TEST_SKIP_SYNC   // (tests/tools/hip_mock builds one variant without it: the harness must notice)
            SW_HIP(hipStreamSynchronize(st));
#endif
            meet.arrive();                                   // (C) every source's rows and counts are ready
            // ---- 2: this device's hash range: pull the rows, build the slice ------------------------------------------------
            uint64_t n_mine = 0, kmer_base = 0;
            for (uint32_t q = 0; q < P; ++q) n_mine += sh[q].cnt[p];
            for (uint32_t o = 0; o < p; ++o)
                for (uint32_t q = 0; q < P; ++q) kmer_base += sh[q].cnt[o];
            {
                StreamScope scope(st);
                r_rows.a.alloc((size_t)n_mine * 16);
                r_ranks.a.alloc((size_t)n_mine * 4);
            }
            {
                size_t at = 0;
                for (uint32_t q = 0; q < P; ++q) {
                    size_t off = 0;
                    for (uint32_t o = 0; o < p; ++o) off += sh[q].cnt[o];
                    pull(r_rows.a.p + at * 16, dev, sh[q].rows + off * 16, devs[q], (size_t)sh[q].cnt[p] * 16, st, routes.is_staged(p, q), ws.stage);
                    at += sh[q].cnt[p];
                }
            }
            ck(sw_slice_build(r_rows.p(), n_mine, kmer_base, record_offsets.data(), nullptr, n_paths, r_ranks.p(), st, &ix));
            {
                uint64_t nk = 0, nn = 0, ne = 0;
                ck(sw_index_sizes(ix, &nk, &nn, &ne));
                me.n_nodes = nn;
                ck(sw_index_ranks_marked(ix, &me.marked));
                StreamScope scope(st);
                r_rows.a.release();
                hashes.a.alloc((size_t)std::max<uint64_t>(nn, 1) * 8);
            }
            ck(sw_index_node_hashes(ix, hashes.p(), st));
            me.ranks = r_ranks.a.p;
            me.hashes = hashes.a.p;
            SW_HIP(hipStreamSynchronize(st));
            meet.arrive();                                   // (D) every slice's nodes, ranks and hashes are ready
            if (p == 0) {
                node_base.assign(1, 0);
                uint64_t mx = 1;
                for (uint32_t o = 0; o < P; ++o) {
                    uint64_t rows_o = 0;
                    for (uint32_t q = 0; q < P; ++q) rows_o += sh[q].cnt[o];
                    // (a slice that received no tuple returns no rank: whatever it reports must not stop the job -- a job without any
                    //  record reports "unmarked" for every slice; found by the SEQWIN_DEVICES fuzz campaign, r04)
                    if (!sh[o].marked && rows_o)
                        raise(SW_ERR_RUNTIME, "multi-device build: a slice holds 2^31 nodes or more (use more devices)");
                    node_base.push_back(node_base.back() + sh[o].n_nodes);
                    mx = std::max(mx, sh[o].n_nodes);
                }
                pad = mx;
                rb = rank_splitters(P, node_base.back());
                // rank -> hash: the whole table on every device while it is small, else by request (dist.py: hash_route)
                uint64_t limit_mb = 4096;
                if (const char *e = getenv("SEQWIN_DIST_TABLE_LIMIT_MB")) limit_mb = (uint64_t)std::max(0ll, atoll(e));
                by_request = (u128)P * pad * 8 > ((u128)limit_mb << 20);
                if (const char *e = getenv("SEQWIN_DIST_HASH_ROUTE")) {
                    if (!strcmp(e, "requests")) by_request = true;
                    else if (!strcmp(e, "table")) by_request = false;
                }
            }
            meet.arrive();                                   // (E) node_base / rb / route are set
            {
                StreamScope scope(st);
                rows.a.release();                            // (every owner has pulled its piece: (D))
                ranks_by_row.a.alloc((size_t)std::max<uint64_t>(me.n_occ, 1) * 4);
            }
            // ---- 3: ranks back to the sources, the rank -> hash table, adjacency keys ----------------------------------------
            {
                size_t at = 0;
                for (uint32_t o = 0; o < P; ++o) {
                    size_t off = 0;
                    for (uint32_t q = 0; q < p; ++q) off += sh[q].cnt[o];
                    pull(ranks_by_row.a.p + at * 4, dev, sh[o].ranks + off * 4, devs[o], (size_t)me.cnt[o] * 4, st, routes.is_staged(p, o), ws.stage);
                    at += me.cnt[o];
                }
            }
            if (!by_request) {
                StreamScope scope(st);
                table.a.alloc((size_t)P * pad * 8);
                for (uint32_t o = 0; o < P; ++o)
                    pull(table.a.p + (size_t)o * pad * 8, dev, sh[o].hashes, devs[o], (size_t)sh[o].n_nodes * 8, st, routes.is_staged(p, o), ws.stage);
            }
            {
                StreamScope scope(st);
                keys.a.alloc((size_t)std::max<uint64_t>(me.n_occ, 1) * 8);
            }
            me.acnt.assign(P, 0);
            me.ccnt.assign(P, 0);
            ck(sw_occ_adjacency_pairs(occ, ranks_by_row.p(), node_base.data(), P, first[p], rb.data(), rb.size(), keys.p(), me.acnt.data(),
                                      me.ccnt.data(), me.key_bits, st));
            {
                uint64_t nc = 0;
                for (uint64_t c : me.ccnt) nc += c;
                StreamScope scope(st);
                cand.a.alloc((size_t)std::max<uint64_t>(nc, 1) * 16);
                ranks_by_row.a.release();
            }
            ck(sw_occ_candidates(occ, cand.p(), st));
            me.keys = keys.a.p;
            me.cand = cand.a.p;
            SW_HIP(hipStreamSynchronize(st));
            meet.arrive();                                   // (F) every source's keys and candidate rows are ready
            // ---- 4: this device's rank range: pull keys and candidates, build the edges ----------------------------------------
            uint64_t m_mine = 0, c_mine = 0;
            for (uint32_t q = 0; q < P; ++q) m_mine += sh[q].acnt[p], c_mine += sh[q].ccnt[p];
            {
                StreamScope scope(st);
                r_ranks.a.release();                         // (every source has pulled its ranks: (F))
                r_keys.a.alloc((size_t)std::max<uint64_t>(m_mine, 1) * 8);
                r_cand.a.alloc((size_t)std::max<uint64_t>(c_mine, 1) * 16);
            }
            {
                size_t at = 0, cat = 0;
                for (uint32_t q = 0; q < P; ++q) {
                    if (sh[q].key_bits[0] != sh[0].key_bits[0] || sh[q].key_bits[1] != sh[0].key_bits[1])
                        raise(SW_ERR_RUNTIME, "internal error: the sources derived different adjacency key layouts");
                    size_t off = 0, coff = 0;
                    for (uint32_t o = 0; o < p; ++o) off += sh[q].acnt[o], coff += sh[q].ccnt[o];
                    pull(r_keys.a.p + at * 8, dev, sh[q].keys + off * 8, devs[q], (size_t)sh[q].acnt[p] * 8, st, routes.is_staged(p, q), ws.stage);
                    pull(r_cand.a.p + cat * 16, dev, sh[q].cand + coff * 16, devs[q], (size_t)sh[q].ccnt[p] * 16, st, routes.is_staged(p, q), ws.stage);
                    at += sh[q].acnt[p];
                    cat += sh[q].ccnt[p];
                }
            }
            unsigned asm_bits = 1;
            while (((uint64_t)1 << asm_bits) <= n_paths && asm_bits < 63) ++asm_bits;   // max(1, bit_length(n_assemblies))
            ck(sw_slice_edges_pairs(ix, r_keys.p(), m_mine, r_cand.p(), c_mine, sh[0].key_bits[0], sh[0].key_bits[1], p ? rb[p - 1] : 0,
                                    asm_bits, by_request ? nullptr : table.p(), node_base.data(), P, pad, st));
            SW_HIP(hipStreamSynchronize(st));
            meet.arrive();                                   // (G) every owner has pulled: the sources' keys may go
            {
                StreamScope scope(st);
                keys.a.release();
                cand.a.release();
                r_keys.a.release();
                r_cand.a.release();
                table.a.release();
            }
            if (by_request) {
                // ---- 5: the edges hold global ranks: ask the node owners for the hashes of the distinct endpoints -------------
                me.req_cnt.assign(P, 0);
                uint64_t n_req = 0;
                ck(sw_index_edge_hash_requests(ix, node_base.data(), P, me.req_cnt.data(), &n_req, st));
                {
                    StreamScope scope(st);
                    req.a.alloc((size_t)std::max<uint64_t>(n_req, 1) * 4);
                }
                ck(sw_index_edge_hash_request_rows(ix, req.p(), st));
                me.req = req.a.p;
                SW_HIP(hipStreamSynchronize(st));
                meet.arrive();                               // (H) every edge owner's requests are ready
                uint64_t n_got = 0;
                for (uint32_t q = 0; q < P; ++q) n_got += sh[q].req_cnt[p];
                {
                    StreamScope scope(st);
                    got.a.alloc((size_t)std::max<uint64_t>(n_got, 1) * 4);
                    answers.a.alloc((size_t)std::max<uint64_t>(n_got, 1) * 8);
                }
                me.ans.assign(P, nullptr);
                {
                    size_t at = 0;
                    for (uint32_t q = 0; q < P; ++q) {
                        size_t off = 0;
                        for (uint32_t o = 0; o < p; ++o) off += sh[q].req_cnt[o];
                        pull(got.a.p + at * 4, dev, sh[q].req + off * 4, devs[q], (size_t)sh[q].req_cnt[p] * 4, st, routes.is_staged(p, q), ws.stage);
                        me.ans[q] = answers.a.p + at * 8;
                        at += sh[q].req_cnt[p];
                    }
                }
                ck(sw_index_node_hash_lookup(ix, got.p(), n_got, answers.p(), st));
                SW_HIP(hipStreamSynchronize(st));
                meet.arrive();                               // (I) every node owner's answers are ready
                {
                    StreamScope scope(st);
                    replies.a.alloc((size_t)std::max<uint64_t>(n_req, 1) * 8);
                }
                {
                    size_t at = 0;
                    for (uint32_t o = 0; o < P; ++o) {
                        pull(replies.a.p + at * 8, dev, sh[o].ans[p], devs[o], (size_t)me.req_cnt[o] * 8, st, routes.is_staged(p, o), ws.stage);
                        at += me.req_cnt[o];
                    }
                }
                ck(sw_index_edge_hash_attach(ix, replies.p(), n_req, st));
                SW_HIP(hipStreamSynchronize(st));
                meet.arrive();                               // (J) every edge owner has its answers: the node owners' buffers may go
                StreamScope scope(st);
                req.a.release();
                got.a.release();
                answers.a.release();
                replies.a.release();
            }
            {
                StreamScope scope(st);
                hashes.a.release();
            }
            SW_HIP(hipStreamSynchronize(st));
            ix->last_stream = nullptr;                       // (all its work is complete; it is exported from the caller's thread)
            slices[p].reset(ix);
            ix = nullptr;
        } catch (...) {
            {
                std::lock_guard<std::mutex> g(err_mu);
                if (!err) err = std::current_exception();
            }
            meet.fail();
        }
        if (st) (void)hipStreamSynchronize(st);   // (what is left is released by the caller's cleanup loop, after every thread has joined)
    };

    std::vector<std::thread> th;
    try {
        for (uint32_t p = 0; p < P; ++p) th.emplace_back(worker, p);
    } catch (...) {              // a thread could not be started: the ones that run must not wait for it for ever
        {
            std::lock_guard<std::mutex> g(err_mu);
            if (!err) err = std::current_exception();
        }
        meet.fail();
    }
    for (auto &t : th) t.join();
    // every worker has stopped and synchronised its stream: no copy of any peer is in flight any more
    for (uint32_t p = 0; p < P; ++p) {
        WorkerState &ws = wstate[p];
        if (hipSetDevice(devs[p]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (ws.st) (void)hipStreamSynchronize(ws.st);
        StreamScope scope(ws.st);
        if (ws.ix) sw_index_free(ws.ix);
        if (ws.occ) sw_occ_free(ws.occ);
        if (ws.batch) sw_batch_free(ws.batch);
        for (DevBuf *b : {&ws.rows, &ws.ranks_by_row, &ws.keys, &ws.cand, &ws.r_rows, &ws.r_ranks, &ws.hashes, &ws.table, &ws.r_keys,
                          &ws.r_cand, &ws.req, &ws.got, &ws.answers, &ws.replies})
            b->a.release();
        if (ws.stage.p) { (void)hipHostFree(ws.stage.p); ws.stage.p = nullptr; }
    }
    SW_HIP(hipSetDevice(home));
    if (err) std::rethrow_exception(err);

    out.slices = std::move(slices);
    out.record_offsets = std::move(record_offsets);
    out.n_assemblies = n_paths;
    out.total_bp = 0;
    out.ids_blob.clear();
    for (uint32_t p = 0; p < P; ++p) {
        out.total_bp += sh[p].total_bp;
        out.ids_blob += sh[p].ids;
    }
    out.hash_route = by_request ? "requests" : "table";
    out.copy_route = route_summary;
}

}  // namespace sw
