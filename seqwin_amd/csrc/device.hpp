// device.hpp -- device-side objects shared by sketch.hip / index.hip / api.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <functional>
#include <memory>
#include <string>

#include "common.hpp"

namespace sw {

#define SW_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            ::sw::raise(SW_ERR_DEVICE, "HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), \
                        __FILE__, __LINE__, #expr);                                                    \
    } while (0)

// ---- position-dependent checksums of the output arrays (r06: EVERY field of an element enters together with the element's index) ----
// Element i of array X contributes  sum over its fields f of  mix64(((base + i) * G ^ K_f) + value_f)  (mod 2^64), so two
// elements that swap ANY field -- a node's start / stop / counts / penalty, an edge's second / weight -- change the sum (until r05
// only kmers, nodes.hash and edges.first carried the index).  `base` = the element's index in the whole array
// when the array at hand is a slice of it: the shares of the slices of a sharded index add up modulo 2^64.  One definition for
// the device kernels (index.hip: k_checksum, k_identity), the host threads of sw_get_penalty's identity test (api.hip:
// host_identity) and -- restated in numpy -- seqwin_amd/device.py: host_checksums.  Whole node rows compared: tests/smoke/test_graph.py:281-291.
__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
constexpr uint64_t CK_G = 0x9E3779B97F4A7C15ULL;
constexpr uint64_t CK_K1 = 0xA0761D6478BD642FULL, CK_K2 = 0xE7037ED1A0B428DBULL, CK_K3 = 0x8EBC6AF09C88C6E3ULL,
                   CK_K4 = 0x589965CC75374CC3ULL;
__host__ __device__ inline uint64_t ck_kmer(uint64_t i, const sw_kmer &k)
{
    return mix64(i * CK_G + ((uint64_t)k.pos | ((uint64_t)k.record_idx << 32)));
}
// the immutable part of a node (what sw_get_penalty's identity test compares): hash, start, stop
__host__ __device__ inline uint64_t ck_node_identity(uint64_t i, const sw_node &n)
{
    const uint64_t x = i * CK_G;
    return mix64(x + n.hash) + mix64((x ^ CK_K1) + n.start) + mix64((x ^ CK_K2) + n.stop);
}
__host__ __device__ inline uint64_t ck_node(uint64_t i, const sw_node &n)
{
    const uint64_t x = i * CK_G;
    uint64_t pbits;
    memcpy(&pbits, &n.penalty, 8);   // the f64 by bit pattern (-0.0 and 0.0 differ, a NaN's payload counts)
    return ck_node_identity(i, n) + mix64((x ^ CK_K3) + ((uint64_t)n.n_tar << 32 | n.n_neg)) + mix64((x ^ CK_K4) + pbits);
}
__host__ __device__ inline uint64_t ck_edge(uint64_t i, const sw_edge &e)
{
    const uint64_t x = i * CK_G;
    return mix64(x + e.first) + mix64((x ^ CK_K1) + e.second) + mix64((x ^ CK_K2) + e.weight);
}

// ---- caching device allocator (steady-state index builds do no hipMalloc), stream-aware: api.hip ----------
void *dev_alloc(size_t bytes);
void dev_free(void *p);
void dev_pool_trim();
uint64_t dev_pool_bytes();
// The stream a C-ABI call works on, for the calling thread: blocks released inside the scope are tagged with it and
// blocks taken inside it are fenced against the stream context they were released under.
struct StreamScope {
    explicit StreamScope(hipStream_t s);
    ~StreamScope();
    StreamScope(const StreamScope &) = delete;
    StreamScope &operator=(const StreamScope &) = delete;
private:
    hipStream_t prev_main;
    std::vector<hipStream_t> prev_side;
};
// Between fork and join of a side stream, blocks released by this thread also carry an event of that stream.
void alloc_fork(hipStream_t side);
void alloc_join(hipStream_t side);

// hipEvent_t with a destructor (no leak when an exception unwinds through a build)
struct Event {
    hipEvent_t e = nullptr;
    explicit Event(bool timing = true)
    {
        SW_HIP(timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    ~Event() { if (e) (void)hipEventDestroy(e); }
    Event(const Event &) = delete;
    Event &operator=(const Event &) = delete;
    operator hipEvent_t() const { return e; }
};

template <class T> struct DevArray {
    T *p = nullptr;
    size_t n = 0;
    DevArray() = default;
    explicit DevArray(size_t count) { alloc(count); }
    DevArray(const DevArray &) = delete;
    DevArray &operator=(const DevArray &) = delete;
    DevArray(DevArray &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevArray &operator=(DevArray &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevArray() { release(); }
    void alloc(size_t count)
    {
        release();
        n = count;
        p = (T *)dev_alloc((count ? count : 1) * sizeof(T));
    }
    void release()
    {
        if (p) dev_free(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

// One tile of a fast class, as its workgroup reads it at start-up (32 B, one scalar load).  Written on the device by
// k_plan_tiles (sketch.hip) from the per-record tables: the host never loops over tiles.
struct alignas(32) TileDesc {
    uint64_t bfirst;   // base index (batch-wide, in bases) of the tile's first element: rec_base + pos0 + E0
    uint32_t ne;       // elements (k-mers) the tile holds: window ends + halo
    uint32_t i0;       // idx of the tile's first window end (w - 1 for the first tile of a record)
    uint32_t kpos;     // position in the record of the tile's first element (pos0 + E0)
    uint32_t rec;      // record
    uint32_t gid;      // global tile id (record-major, window order: the order pass follows it)
    uint32_t flags;    // bit 0: the tile's reach crosses an invalid-base gap -> done by the generic kernel (gap_list)
};

// ---- per-(k, w) launch plan of a batch ---------------------------------------------------------
// Valid k-mers of a record are numbered 0..n_valid-1 in increasing position ("idx space",
// minimizer.cpp:69-70).  A segment is a maximal run of valid bases of length >= k; it contributes
// run_len-k+1 consecutive idx values.  A tile is TW consecutive window ends of one record.
struct Plan {
    uint32_t k = 0, w = 0;
    uint32_t w_full = 0;   // the caller's window; > w when it exceeds SW_MAX_WINDOW: the tile kernels then sketch with the
                           // smaller w (a superset of the answer) and order_tuples selects the minimizers of w_full from it
    uint32_t L = 0;        // k-mers hashed per thread (odd, <= w)
    uint32_t NE = 0;       // elements per tile = 256 * L
    uint32_t TW = 0;       // window ends per tile = NE - w
    uint32_t Lf = 0, Lg_list = 0, halo_f = 0;  // fast classes: run length 32 / 16 / 0 = unavailable; list-mode run length; halo
    uint32_t n_tiles = 0, n_tiles_gen = 0;
    // Fast tile classes: [0] 256-thread workgroups (256 * L elements per tile), [1] 64-thread workgroups (64 * L) for
    // short records: a tile occupies its workgroup's LDS whatever its fill, so a 1 kbp contig in a 256-thread tile keeps
    // 3 of 4 waves idle and the CU at a quarter of its occupancy.  Every tile carries its own (record, first window end,
    // global id), so the classes can mix inside a record (get_plan: only behind SEQWIN_AMD_SKETCH=tails, measured slower).
    struct FastClass {
        uint32_t B = 0;                // threads per workgroup
        uint32_t TW = 0;               // window ends per tile (0 = class unavailable)
        uint32_t n_tiles = 0, n_gap = 0;
        DevArray<uint32_t> rec_off;    // [R + 1] first tile of every record in this class
        DevArray<TileDesc> desc;       // [n_tiles] one descriptor per tile: everything a workgroup needs (no search, ONE load)
        DevArray<uint32_t> gap_list;   // [n_gap] class ids of the tiles whose reach crosses an invalid-base gap (-> generic kernel)
    } fc[2];
    uint64_t n_windows = 0;
    uint64_t n_valid = 0;
    size_t lds_bytes = 0;
    DevArray<uint32_t> rec_seg_off;   // [R + 1]
    DevArray<uint32_t> rec_nvalid;    // [R]
    DevArray<uint32_t> rec_tile_off;  // [R + 1] global tile numbering
    DevArray<uint32_t> gen_tile_off;  // [R + 1] tiles of the generic class
    DevArray<uint32_t> gen_tile_rec;  // [n_tiles_gen]
    uint32_t slot_cap = 0;            // stage entries reserved per tile (1.5 x the expected number of minimizers + 16)
    DevArray<uint32_t> seg_pos;       // [S]
    DevArray<uint32_t> seg_idx;       // [S]
    DevArray<uint64_t> lut;           // [40] roll tables, see sketch.hip
    DevArray<uint64_t> t4;            // [256][2] 4-base warm-up table
    uint64_t mult = 0;                // 1 ^ (k * MULTISEED)
    double build_ms = 0;              // wall time get_plan took to build this plan (host loops + table uploads)
};

}  // namespace sw

struct sw_batch {
    int device = 0;
    sw::HostBatch host;               // host.packed is released after upload
    uint64_t n_records = 0;
    uint64_t packed_words = 0;
    sw::DevArray<uint32_t> d_packed;
    sw::DevArray<uint64_t> d_rec_base;
    sw::DevArray<uint32_t> d_rec_asm;  // assembly index of each record
    std::mutex plan_mu;
    std::map<std::pair<uint32_t, uint32_t>, sw::Plan> plans;
};

namespace sw { struct OrderedOcc; }
namespace sw {
// ingest_dev.hip: many .gz files -> batch, inflated / parsed / packed on the device (false: take the host route)
bool device_gz_ingest(const char *const *paths, size_t n_paths, uint64_t n_cpu, sw_batch &b);
}

struct sw_occ {   // ordered tuple stream of one shard (tuple-exchange form of the multi-GPU build)
    const sw_batch *batch = nullptr;
    sw::OrderedOcc *occ = nullptr;
    float sketch_ms = 0.f;
    hipStream_t last_stream = nullptr;   // the stream its arrays were last used on: sw_occ_free releases them under it
    ~sw_occ();
};

namespace sw { struct EdgeHashJob; }
struct sw_index {
    int device = 0;
    uint64_t n_kmers = 0, n_nodes = 0, n_edges = 0;
    sw::DevArray<sw_kmer> kmers;
    sw::DevArray<sw_node> nodes;
    sw::DevArray<sw_edge> edges;
    sw_timings timings{};
    bool ranks_marked = false;   // slice build: the returned ranks carry "node recurs in its assembly" in bit 31
    hipStream_t last_stream = nullptr;   // the stream its arrays were last used on: sw_index_free releases them under it
    // multi-GPU slices built without the job-wide rank -> hash table: edges hold global RANKS until the hashes have been
    // asked from the node owners (index.hip: edge_hash_requests / edge_hash_attach)
    bool edges_hold_ranks = false;
    sw::EdgeHashJob *hash_job = nullptr;
    sw_index() = default;
    sw_index(const sw_index &) = delete;
    sw_index &operator=(const sw_index &) = delete;
    ~sw_index();
};

namespace sw {

// multi.hip: one build over several devices inside one process (SEQWIN_DEVICES).  slices[o] = owner o's hash range of the
// graph, resident on its device; their concatenation in owner order is the single-device result.
struct MultiGraph {
    std::vector<std::unique_ptr<sw_index>> slices;
    std::vector<uint32_t> record_offsets;
    std::string ids_blob;
    uint64_t n_assemblies = 0, total_bp = 0;
    const char *hash_route = "";
    std::string copy_route;   // how the exchanges travelled: peer access per device pair, or staged through the host (multi.hip: Routes)
};
std::vector<int> devices_from_env();   // SEQWIN_DEVICES: "all" or a list of device indices (repeats allowed); empty: one device
// chunk_bp > 0: every worker streams its shard through HBM in chunks of about that many bases (low_memory / the HBM budget per shard)
void build_multi_device(const char *const *paths, size_t n_paths, uint64_t k, uint64_t w, uint64_t n_cpu, std::vector<int> devs,
                        MultiGraph &out, uint64_t chunk_bp = 0);

Plan &get_plan(sw_batch &b, uint64_t k, uint64_t w, bool *cached = nullptr);

// sketch.hip: runs the fused ntHash + window-minimum kernel over every tile of the plan.
// Output: tuples per tile in (stage_hash, stage_kmer) -- a tile's own slot at tile * slot_cap, or a range of the
// overflow area behind the slots -- plus per-tile (offset, count).
struct SketchOut {
    DevArray<uint64_t> stage_hash;
    DevArray<uint64_t> stage_kmer;   // pos | record_idx << 32
    DevArray<uint32_t> tile_count;
    DevArray<uint64_t> tile_offset;
    uint64_t n_occ = 0;
    uint64_t launches = 0;
    uint64_t n_ovf_tiles = 0;   // fast-class tiles done by the generic kernel's list pass (gap tiles + more than RC suffix records in a run)
};
void run_sketch(const sw_batch &b, const Plan &plan, hipStream_t stream, SketchOut &out, float *sketch_ms);

// radix.hip: stable LSD radix sort of 64-bit keys by bits [begin_bit, end_bit); (keys, alt) is a double buffer, on return
// `keys` points at the sorted data; *d_fail (device word, zeroed by the caller) becomes non-zero if a pass gave up
// perm_hi32: the keys' upper halves are a permutation of 0 .. n-1 (the unsort's words): the digit histograms of a sort on
// bits >= 32 then follow from n alone and the sweep that counts them is skipped
// d_hist_given: digit counts the producer of the keys took while writing them (layout: radix_layout), scanned in place
void radix_sort_keys64(uint64_t *&keys, uint64_t *&alt, uint64_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream,
                       uint32_t *d_fail, bool perm_hi32 = false, unsigned long long *d_hist_given = nullptr, unsigned layout_bits = 0);
void radix_layout(unsigned bits, unsigned *digit_bits, unsigned *n_passes);
// keys[i] >> 32 a permutation of 0 .. n-1: grouped by the index bits [low_bits, nbit) in two unstable passes (false: not taken)
bool radix_unsort_perm(uint64_t *&keys, uint64_t *&alt, uint64_t n, unsigned low_bits, unsigned nbit, hipStream_t stream, uint32_t *d_fail);

// index.hip: radix.hip or rocPRIM; d_fail: zeroed device word, to be read back and handed to check_sort_failed
void sort_keys64(uint64_t *&keys, uint64_t *&keys_alt, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream,
                 uint32_t *d_fail, bool perm_hi32 = false, unsigned long long *d_hist_given = nullptr, unsigned layout_bits = 0);
bool sort_keys64_is_own(size_t n);   // radix.hip (digit counts may be handed in) or rocPRIM
void check_sort_failed(uint32_t fail_word);

// index.hip
// What the node sort moves with every occurrence (16 B): the low half of its hash (the high half is the sort key), its
// (pos, record_idx) -- so that `kmers` comes out of the sort in order, without a gather -- and its place in the
// (record_idx, pos) stream, which brings its node's rank back to that stream for the adjacency (index.hip: unsort).
struct alignas(16) OccPay {
    uint32_t low, pos, rec, idx;
};
// radix.hip: stable sort of (key32, OccPay) pairs by key bits [0, end_bit), end_bit in {8, 16, 24, 32}; double buffers
int radix_rank_mode();          // how radix.hip ranks keys inside a wave on the current device: 1 LDS atomics, 0 ballots (runs the self-check once)
bool radix_pairs_available();   // false on a device that does not pass the LDS-atomic ranking self-check (rocPRIM sorts the pairs then)
uint64_t radix_trim_state();    // frees the look-back state buffers of idle (device, stream) pairs; called by dev_pool_trim
void radix_demote_rank();       // an order guard (k_nodes, k_rle_keys, k_check_ascending) tripped: ballots / rocPRIM on this device from now on
// The sketch stage as the input of the sort's first pass (no ordered copy in between: the pass does what k_order does on
// the way in).  Dense index g = place of a tuple in (record_idx, pos) order; tile T holds [dst_off[T], dst_off[T] + tile_count[T]).
constexpr uint32_t STAGE_WIN = 32;                 // tiles from the first of a wave's 448 tuples on that the pass resolves without a search
constexpr uint32_t STAGE_ROW = STAGE_WIN + 9;      // 8-byte words of a directory row (radix.hip, k_rs_stage_prepare)
constexpr uint32_t STAGE_PAD = 1;                  // dst_off has n_tiles + STAGE_PAD entries; the pad entry holds n
struct StageSource {
    const uint64_t *stage_hash, *stage_kmer;       // canonical hash (extend_hashes is applied by the reader), pos | record_idx << 32
    const uint32_t *tile_count;
    const uint64_t *tile_offset;                   // where a tile's tuples lie in the stage
    const uint64_t *dst_off;                       // exclusive sum of tile_count (padded, see above)
    uint32_t n_tiles;
    uint64_t mult;                                 // extend_hashes multiplier of the plan
    uint32_t *rec_out;                             // record_idx of every tuple, dense order (the adjacency reads it)
    const unsigned long long *chunk_dir;           // (filled in by radix_sort_pairs32) one row per 448 dense indices
    const uint64_t *rows;                          // non-null: the source is n rows of (out_hash, pos | record_idx << 32) instead
                                                   // (a slice's received tuples: only this field is read)
};
// src: pass 0 reads the stage and writes (keys_alt, vals_alt); after_first() runs once that pass is enqueued and must leave
// (keys, vals) pointing at n-element buffers (the stage may be released there: the pool is stream-ordered)
void radix_sort_pairs32(uint32_t *&keys, uint32_t *&keys_alt, OccPay *&vals, OccPay *&vals_alt, uint64_t n, unsigned end_bit,
                        hipStream_t stream, uint32_t *d_fail, const StageSource *src = nullptr,
                        const std::function<void()> &after_first = std::function<void()>(), uint32_t *low_out = nullptr);
// (low_out: the last pass also writes OccPay::low of every element, in sorted order, into this array)
// index.hip: radix.hip's pair passes or rocPRIM's (small inputs, SEQWIN_AMD_SORT / SEQWIN_AMD_PAIR_SORT); d_fail as for sort_keys64
// returns true if low_out (may be null) was written (radix.hip's passes only)
bool sort_pairs32(uint32_t *&keys, uint32_t *&keys_alt, OccPay *&vals, OccPay *&vals_alt, uint64_t n, unsigned end_bit,
                  hipStream_t stream, uint32_t *d_fail, uint32_t *low_out = nullptr);
struct PartState;                       // index.hip: offsets of the last tuple partition (the way back walks them again)
void part_state_delete(PartState *p);
uint32_t occ_partition_owners(const struct OrderedOcc &occ);   // owners of the last tuple partition (0: none)
struct OrderedOcc {
    // exchange form (multi-GPU tuple exchange, sw_sketch): the tuples themselves
    DevArray<uint64_t> hash;   // out_hash in (record_idx, pos) order
    DevArray<uint64_t> kmer;   // pos | record_idx << 32
    // index form (single-GPU build): the node sort's input, consumed by it, and the records for the adjacency
    DevArray<uint32_t> key32;  // out_hash >> 32: first-phase sort key
    DevArray<OccPay> pay;
    DevArray<uint32_t> rec;    // record_idx in (record_idx, pos) order
    DevArray<uint64_t> cand_rows;   // pairs form of the adjacency exchange: {pair key, assembly} of the candidate records, by owner
    // staged index form: key32 / pay are not made; the node sort's first pass reads the sketch stage (StageSource) and writes rec
    bool staged = false;
    SketchOut stage;
    DevArray<uint64_t> dst_off;
    uint32_t stage_tiles = 0;
    uint64_t stage_mult = 0;
    uint64_t n = 0;
    PartState *part = nullptr;
    OrderedOcc() = default;
    OrderedOcc(OrderedOcc &&o) noexcept { *this = std::move(o); }
    OrderedOcc &operator=(OrderedOcc &&o) noexcept
    {
        if (this != &o) {
            hash = std::move(o.hash); kmer = std::move(o.kmer); key32 = std::move(o.key32); pay = std::move(o.pay);
            rec = std::move(o.rec); cand_rows = std::move(o.cand_rows); n = o.n; o.n = 0;
            staged = o.staged; o.staged = false; stage = std::move(o.stage); dst_off = std::move(o.dst_off);
            stage_tiles = o.stage_tiles; stage_mult = o.stage_mult;
            part_state_delete(part); part = o.part; o.part = nullptr;
        }
        return *this;
    }
    ~OrderedOcc() { part_state_delete(part); }
};
// take_stage (index form only): if the node sort can read the stage itself, `sk` is moved into `out` (out.staged) instead of
// being copied into key32 / pay
void order_tuples(SketchOut &sk, const Plan &plan, hipStream_t stream, OrderedOcc &out, bool index_form = false, bool take_stage = false);
// index-form streams of consecutive assembly chunks -> one stream (chunk c's records follow rec_base[c] earlier ones);
// the chunks are emptied
void concat_occ(std::vector<OrderedOcc> &chunks, const std::vector<uint64_t> &rec_base, hipStream_t stream, OrderedOcc &out);
// occ: index form (key32 / pay / rec); d_rec_asm[n_records] = assembly of every record of the stream
void build_index(const uint32_t *d_rec_asm, uint64_t n_records, uint64_t n_assemblies, OrderedOcc &occ,
                 const uint8_t *d_is_target, uint64_t n_targets, uint64_t n_non_targets, hipStream_t stream, sw_index &ix);
void pack_export(const sw_node *nodes, uint64_t n_nodes, uint64_t n_kmers, const sw_edge *edges, uint64_t n_edges, uint64_t per, uint32_t *pn,
                 uint64_t *bases, uint32_t *pe, uint32_t *flag, hipStream_t stream);
void device_get_penalty(const sw_kmer *d_kmers, uint64_t n_kmers, sw_node *d_nodes, uint64_t n_nodes,
                        const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target,
                        uint64_t n_targets, uint64_t n_non_targets, hipStream_t stream, uint64_t *err_flags_host);
void device_filter_kmers(const sw_kmer *d_kmers, uint64_t n_kmers, const sw_node *d_nodes, uint64_t n_nodes,
                         const uint64_t *d_used_sorted, uint64_t n_used, hipStream_t stream,
                         DevArray<sw_kmer> &kmers_out, DevArray<sw_node> &nodes_out, uint64_t *n_kmers_out,
                         uint64_t *n_nodes_out);
void device_identity(const sw_index &ix, hipStream_t stream, uint64_t *sums2, uint64_t kbase = 0, uint64_t nbase = 0);
void slice_get_penalty(sw_index &ix, uint64_t kmer_base, const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target,
                       uint64_t n_targets, uint64_t n_non_targets, hipStream_t stream, uint64_t *err_flags_host);
void device_checksums(const sw_index &ix, hipStream_t stream, uint64_t *sums3, uint64_t kbase = 0, uint64_t nbase = 0,
                      uint64_t ebase = 0);
void index_threshold_sums(const sw_index &ix, hipStream_t stream, uint64_t *sums3);
void index_verify(const sw_index &ix, uint64_t n_assemblies, bool scored, hipStream_t stream, uint64_t *out10);
void index_filter_graph(const sw_index &ix, uint64_t weight_th, hipStream_t stream, sw_index &out);
void index_occ_rows(const sw_index &ix, uint64_t rec_offset, uint64_t *d_rows, hipStream_t stream);
void index_splits(const sw_index &ix, const uint64_t *node_bounds, const uint64_t *edge_bounds, uint32_t n_bounds,
                  uint64_t *occ_split, uint64_t *edge_split, hipStream_t stream);
void merge_build(const uint64_t *d_occ_rows, uint64_t n, const uint64_t *d_edge_rows, uint64_t m, uint64_t kmer_base,
                 const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target, uint64_t n_targets,
                 uint64_t n_non_targets, hipStream_t stream, sw_index &ix, uint32_t *d_rank_out = nullptr);
void occ_partition(const OrderedOcc &occ, const uint64_t *bounds, uint32_t n_bounds, uint64_t rec_offset, uint64_t *d_rows,
                   uint32_t *d_perm, uint64_t *counts_host, hipStream_t stream);
void occ_adjacency(const OrderedOcc &occ, const uint32_t *d_rec_asm, const uint32_t *d_perm, const uint32_t *d_rank_by_row,
                   unsigned nb, unsigned ab, uint64_t asm_base, const uint64_t *rank_bounds, uint32_t n_bounds,
                   uint64_t *d_rows_out, uint64_t *counts_host, hipStream_t stream);
void slice_edges(sw_index &ix, const uint64_t *d_adj_rows, uint64_t m, unsigned nb, unsigned ab, const uint64_t *d_rank_hash,
                 hipStream_t stream);
void occ_adjacency_pairs(OrderedOcc &occ, const uint32_t *d_rec_asm, const uint32_t *d_rank_by_row, const uint64_t *node_base,
                         uint64_t asm_base, const uint64_t *rank_bounds, uint32_t n_bounds, uint64_t *d_keys_out,
                         uint64_t *counts_host, uint64_t *cand_counts_host, uint64_t *key_bits_host, hipStream_t stream);
void slice_edges_pairs(sw_index &ix, uint64_t *d_keys, uint64_t m, const uint64_t *d_cand_rows, uint64_t c, unsigned lo_bits,
                       unsigned hi_bits, uint64_t lo_base, unsigned ab, const uint64_t *d_rank_hash, const uint64_t *node_base,
                       uint32_t n_owners, uint64_t pad, hipStream_t stream);
void index_node_hashes(const sw_index &ix, uint64_t *d_out, hipStream_t stream);
// rank -> hash by request instead of the job-wide table (slices whose edges hold ranks): the distinct endpoint ranks of the
// slice's edges as owner-local ranks grouped by node owner (counts_host[n_owners]); lookups at the node owner; the replies,
// in the order of the requests, put into the edges
uint64_t edge_hash_requests(sw_index &ix, const uint64_t *node_base, uint32_t n_owners, uint64_t *counts_host, hipStream_t stream);
void edge_hash_request_rows(const sw_index &ix, uint32_t *d_out, hipStream_t stream);
void node_hash_lookup(const sw_index &ix, const uint32_t *d_local_ranks, uint64_t n, uint64_t *d_out, hipStream_t stream);
void edge_hash_attach(sw_index &ix, const uint64_t *d_replies, uint64_t n, hipStream_t stream);

}  // namespace sw
