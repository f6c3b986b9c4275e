// index.hip -- device-side index build: tuple ordering, nodes / kmers, per-node assembly counts,
// adjacency edges, filter.  Replaces (reference paths relative to /root/reference):
//   node map + grouped scatter     cpp/src/seqwin/build.cpp:153-168, 196-253
//   edge map with per-assembly +1  cpp/src/seqwin/build.cpp:177-189
//   radix sort + run-length merge  cpp/src/seqwin/build_internals.cpp:76-144, 159-291
//   get_penalty                    cpp/src/seqwin/filter.cpp:15-137
//   filter_kmers                   cpp/src/seqwin/filter.cpp:139-201
// The reference builds per-thread ankerl hash maps and erases their iteration order with a stable
// LSD radix sort; here the tuple stream is already in (record_idx, pos) order, so ONE stable device
// radix sort by out_hash yields `kmers` in the reference's (hash, record_idx, pos) order and a
// run-length pass yields `nodes`.  Edges are sorted as (rank_lo, rank_hi) pairs of dense node ranks
// (rank order == hash order), stably, so equal pairs stay in assembly order and the number of
// assemblies containing a pair is a count of assembly changes inside its run.
// The large sorts are radix.hip's (this library's onesweep passes: pairs for the nodes, keys for the edges and the unsort); rocPRIM
// supplies the small ones, the prefix sums over block counts, and -- behind knobs -- the library forms of the large primitives.
#include <cstdlib>
#include <cstring>  // rocprim's texture iterator needs ::memset declared first
#include <map>
#include <memory>

#include <rocprim/rocprim.hpp>

#include "device.hpp"

namespace sw {

namespace {

constexpr int TPB = 256;
inline unsigned blocks_for(uint64_t n, int per = TPB) { return (unsigned)((n + per - 1) / per); }
inline bool ranks_by_table()
{
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_RANKS");
    return e && !strcmp(e, "table");
}

// ---- rocPRIM wrappers (temp storage from the caching allocator) ------------------------------
template <class K, class V>
void sort_pairs(K *&keys, K *&keys_alt, V *&vals, V *&vals_alt, size_t n, unsigned begin_bit, unsigned end_bit,
                hipStream_t stream)
{
    rocprim::double_buffer<K> dk(keys, keys_alt);
    rocprim::double_buffer<V> dv(vals, vals_alt);
    size_t tmp_bytes = 0;
    SW_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, n, begin_bit, end_bit, stream));
    DevArray<unsigned char> tmp(tmp_bytes);
    SW_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, dk, dv, n, begin_bit, end_bit, stream));
    keys = dk.current();
    keys_alt = dk.alternate();
    vals = dv.current();
    vals_alt = dv.alternate();
}

}  // namespace

// keys-only stable sort of 64-bit keys on bits [begin_bit, end_bit): the hand-written onesweep of radix.hip for large
// inputs, rocPRIM's for small ones; SEQWIN_AMD_SORT=own|rocprim forces one (A/B, tests).  On return `keys` is the sorted buffer.  d_fail: a zeroed device word the caller
// reads back at its next host synchronisation -- non-zero means a pass of radix.hip gave up waiting for a lower-numbered
// workgroup (it relies on in-order dispatch, like rocPRIM's onesweep; never seen) and the result is not sorted:
// check_sort_failed() then raises instead of the device hanging.
bool sort_keys64_is_own(size_t n)
{
    // measured per build, edges stage (r04d, gpurun_out/r4v; keys = occurrences): 0.38 M keys 0.30 against 0.23 ms with rocPRIM,
    // 3 M 0.41 / 0.37, 6 M 0.53 / 0.55, 12 M 0.76 / 0.81, 24 M 1.22 / 1.47 (the launches and the look-back of few tiles weigh
    // more on small inputs) -> the own passes from 2^23 keys on (r03, with ballot ranking and state resets: from 2^26)
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_SORT");
    const bool own = e ? !strcmp(e, "own") : n >= (1ull << 23);
    return own && !(e && !strcmp(e, "rocprim"));
}

void sort_keys64(uint64_t *&keys, uint64_t *&keys_alt, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream,
                 uint32_t *d_fail, bool perm_hi32, unsigned long long *d_hist_given, unsigned layout_bits)
{
    if (sort_keys64_is_own(n)) {
        radix_sort_keys64(keys, keys_alt, n, begin_bit, end_bit, stream, d_fail, perm_hi32, d_hist_given, layout_bits);
        return;
    }
    rocprim::double_buffer<uint64_t> dk(keys, keys_alt);
    size_t tmp_bytes = 0;
    SW_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, dk, n, begin_bit, end_bit, stream));
    DevArray<unsigned char> tmp(tmp_bytes);
    SW_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, dk, n, begin_bit, end_bit, stream));
    keys = dk.current();
    keys_alt = dk.alternate();
}

// The node sort's phase 1 -- lsd_radix_sort (build_internals.cpp:76-110) over (key32, OccPay) -- by radix.hip's pair passes for
// large inputs, like the keys-only sorts (745 M pairs: 26.7 ms against 28.0 for rocPRIM's onesweep, r04: 7168-element tiles, one
// workgroup per CU, keys and payloads staged through LDS together); rocPRIM's for small ones, for key widths that are no multiple
// of 8 (a test knob), and on a device that fails the LDS-atomic ranking self-check.  SEQWIN_AMD_SORT=own|rocprim forces one for
// all sorts, SEQWIN_AMD_PAIR_SORT=own|rocprim for this one only (A/B).
// From 2^20 pairs on (r04d: nodes stage 0.27 against 0.38 ms at 0.76 M occurrences, 0.42 / 0.47 at 3 M, 0.99 / 1.05 at 12 M --
// with the first pass reading the sketch stage, which comes with these passes; at 0.38 M the whole build is the same either way).
bool sort_pairs_is_own(size_t n, unsigned bits)
{
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_PAIR_SORT"), *all = SW_TEST_GETENV("SEQWIN_AMD_SORT");
    if (e && !strcmp(e, "rocprim")) return false;
    if (bits % 8 != 0 || bits > 32 || n >= 0xFFFFFFFFull) return false;
    bool own = n >= (1ull << 20);
    if (all) own = !strcmp(all, "own");
    if (e && !strcmp(e, "own")) own = true;
    return own && radix_pairs_available();
}

bool sort_pairs32(uint32_t *&keys, uint32_t *&keys_alt, OccPay *&vals, OccPay *&vals_alt, uint64_t n, unsigned end_bit,
                  hipStream_t stream, uint32_t *d_fail, uint32_t *low_out)
{
    if (sort_pairs_is_own(n, end_bit)) {
        radix_sort_pairs32(keys, keys_alt, vals, vals_alt, n, end_bit, stream, d_fail, nullptr, std::function<void()>(), low_out);
        return low_out != nullptr && n != 0 && end_bit != 0;
    }
    sort_pairs(keys, keys_alt, vals, vals_alt, n, 0, end_bit, stream);
    return false;
}

void check_sort_failed(uint32_t fail_word)
{
    if (fail_word)
        raise(SW_ERR_RUNTIME, "internal error: a radix pass gave up waiting for a lower-numbered workgroup (set SEQWIN_AMD_SORT=rocprim)");
}

namespace {

// NOTE: the temp storage goes back to the caching allocator when the wrapper returns, while the scan / sort may
// still be running.  That is safe only because every later user of the block is ordered after it on the SAME
// stream; work on a second stream must keep its temp storage alive itself (inclusive_sum_keep).
template <class InIt, class OutIt, class T>
void inclusive_sum(InIt in, OutIt out, size_t n, T, hipStream_t stream)
{
    size_t tmp_bytes = 0;
    SW_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, in, out, n, rocprim::plus<T>(), stream));
    DevArray<unsigned char> tmp(tmp_bytes);
    SW_HIP(rocprim::inclusive_scan(tmp.p, tmp_bytes, in, out, n, rocprim::plus<T>(), stream));
}

template <class InIt, class OutIt, class T>
void inclusive_sum_keep(InIt in, OutIt out, size_t n, T, hipStream_t stream, DevArray<unsigned char> &tmp)
{
    size_t tmp_bytes = 0;
    SW_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, in, out, n, rocprim::plus<T>(), stream));
    tmp.alloc(tmp_bytes);
    SW_HIP(rocprim::inclusive_scan(tmp.p, tmp_bytes, in, out, n, rocprim::plus<T>(), stream));
}

template <class InIt, class OutIt, class T>
void exclusive_sum(InIt in, OutIt out, size_t n, T init, hipStream_t stream)
{
    size_t tmp_bytes = 0;
    SW_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, init, n, rocprim::plus<T>(), stream));
    DevArray<unsigned char> tmp(tmp_bytes);
    SW_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, init, n, rocprim::plus<T>(), stream));
}

struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return v; }
};

// head-of-run flag of a sorted key array, evaluated on the fly
struct HeadFlag {
    const uint64_t *keys;
    uint64_t sentinel;  // record boundaries: sort last, never a head
    __host__ __device__ uint32_t operator()(uint64_t s) const
    {
        const uint64_t k = keys[s];
        return (k != sentinel && (s == 0 || k != keys[s - 1])) ? 1u : 0u;
    }
};

// ---- tuple ordering -----------------------------------------------------------------------------
// One wave per tile: the tile's tuples move from its stage slot to their place in (record_idx, pos) order; the staged
// canonical hash becomes out_hash = extend_hashes (hashing_internals.hpp:89-103).  INDEX form writes the node sort's
// input (key32, OccPay) and the record of every occurrence; exchange form writes the tuples (hash, pos | record << 32).
template <bool INDEX, bool RAW = false>
__global__ void k_order(const uint64_t *__restrict__ stage_hash, const uint64_t *__restrict__ stage_kmer,
                        const uint32_t *__restrict__ tile_count, const uint64_t *__restrict__ tile_offset,
                        const uint64_t *__restrict__ dst_off, uint32_t n_tiles, uint64_t mult, uint64_t *__restrict__ hash,
                        uint64_t *__restrict__ kmer, uint32_t *__restrict__ key32, OccPay *__restrict__ pay,
                        uint32_t *__restrict__ rec)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= n_tiles) return;
    const uint32_t c = tile_count[wave];
    const uint64_t src = tile_offset[wave], dst = dst_off[wave];
    for (uint32_t i = lane; i < c; i += 64) {
        uint64_t h = stage_hash[src + i];
        if (!RAW) {   // extend_hashes, hashing_internals.hpp:89-103 (RAW: the canonical hash, what the windows compare)
            h *= mult;
            h ^= h >> 27;
        }
        const uint64_t km = stage_kmer[src + i];
        if (INDEX) {
            key32[dst + i] = (uint32_t)(h >> 32);
            OccPay p;
            p.low = (uint32_t)h;
            p.pos = (uint32_t)km;
            p.rec = (uint32_t)(km >> 32);
            p.idx = (uint32_t)(dst + i);
            pay[dst + i] = p;
            rec[dst + i] = p.rec;
            if (hash) hash[dst + i] = h;   // (only for the hash-table rank lookup, SEQWIN_AMD_RANKS=table)
        } else {
            hash[dst + i] = h;
            kmer[dst + i] = km;
        }
    }
}

// rows[n][2] = {hash, pos | record << 32} (exchanged tuples) -> the node sort's input
__global__ void k_rows_to_pay(const uint64_t *__restrict__ rows, uint64_t n, uint32_t *__restrict__ key32, OccPay *__restrict__ pay)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = rows[2 * i], km = rows[2 * i + 1];
    key32[i] = (uint32_t)(h >> 32);
    OccPay p;
    p.low = (uint32_t)h;
    p.pos = (uint32_t)km;
    p.rec = (uint32_t)(km >> 32);
    p.idx = (uint32_t)i;
    pay[i] = p;
}

__global__ void k_iota(uint32_t *v, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// ---- nodes / kmers / ranks ------------------------------------------------------------------------
// The sorted occurrences (key32[s] = top half of the hash, pay[s] = low half, pos, record, original index) stream through
// once: kmers[s] = (pos, record); a head of a run of equal hashes starts node cum[s] - 1; (original index, node) goes to
// the unsort, which brings every occurrence's node rank back to (record_idx, pos) order; and -- when the counts are wanted --
// one bit per occurrence says "first occurrence of a target (T) / non-target (N) assembly in its node" (the occurrences of
// a node are in record order, records are assembly-major, filter.cpp:62-136), so a node's counts are two popcounts.
// A workgroup takes NODES_TILE consecutive occurrences, NODES_THREADS at a time (lane s of a wave = occurrence s of a
// 64-aligned group: the bits of a group are one ballot).
#ifndef SW_NODES_ROWS
#define SW_NODES_ROWS 4
#endif
constexpr int NODES_ROWS = SW_NODES_ROWS;            // rows per tile; a row = NODES_THREADS lanes x 2 consecutive occurrences (-DSW_NODES_ROWS: A/B)
constexpr int NODES_THREADS = 1024;                  // 16 waves: one workgroup per CU
constexpr int NODES_WAVES = NODES_THREADS / 64;
constexpr uint32_t NODES_ROW = NODES_THREADS * 2;
constexpr uint32_t NODES_TILE = NODES_ROW * NODES_ROWS;   // 8192 occurrences: one ticket, one look-back
constexpr uint32_t UNSORT_BITS = 14;                 // the unsort's last step handles 2^14 consecutive indices in LDS
constexpr uint32_t UNSORT_RANGE = 1u << UNSORT_BITS;
constexpr uint64_t UNSORT_DIRECT_MAX = 1ull << 25;   // up to here (128 MiB of ranks) the array stays in the 256 MiB Infinity Cache and a
                                                     // direct scatter is as fast (measured: 24 M occurrences 7.74 against 7.84 ms per build;
                                                     // 745 M: 239.0 against 232.3 ms)

// The node of an occurrence is the number of run heads before it: a prefix sum over the whole array, done in the same pass
// by chained tiles (decoupled look-back: a tile publishes the number of its heads at once, then adds up the published
// numbers of its predecessors until it meets one that already knows its inclusive total).  Tiles take their number from a
// ticket counter, so every predecessor of a running tile is itself running or finished: the look-back cannot wait for a
// workgroup that has not started.  state = status << 62 | value; status 0 = nothing yet, 1 = own heads, 2 = all heads up to
// and including the tile.  A walk passes every tile that is in flight (none of them knows its total yet), 64 per step and
// a round trip to the memory side per step: with 2048-occurrence tiles and four workgroups per CU that was 1024 tiles = 16
// steps per tile and the kernel's bound (13.4 ms at 745 M occurrences); 8192-occurrence tiles of 1024 threads, one workgroup
// per CU, walk 4 steps (nodes stage 58.3 -> 54.0 ms; 1024 x 2 rows 56.4, 1024 x 6 rows 53.7, 1024 x 8 rows spills).
constexpr unsigned long long TS_AGG = 1ull << 62, TS_INC = 2ull << 62;

// bits of a (lanes 0..31's even positions) and b (odd positions) interleaved: result bit 2 l + e = e ? b[l] : a[l]
__device__ __forceinline__ unsigned long long interleave32(uint32_t a, uint32_t b)
{
    unsigned long long x = a, y = b;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; y = (y | (y << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;  y = (y | (y << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;  y = (y | (y << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;  y = (y | (y << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;  y = (y | (y << 1)) & 0x5555555555555555ull;
    return x | (y << 1);
}

// Every lane takes TWO consecutive occurrences per row (16-B stores of kmers and of the unsort words, one 8-B load of
// the keys, 32 B of payload), a wave 128, so the bits of a wave's row are two 64-bit words: ballots of the even and of the
// odd occurrences, interleaved (wave-uniform values: scalar ALU).
// REP: bit 31 of the rank word marks an occurrence whose node occurs more than once in its assembly (its neighbour in this
// order has the same node and assembly) -- only adjacencies touching such an occurrence can repeat a pair inside one
// assembly (k_adj_pairs).  Needs rec_flag and fewer than 2^31 nodes.
constexpr uint32_t RANK_REP = 0x80000000u;
constexpr uint8_t OWNER_DROP = 0xFF;   // owner byte of a key that is no row (record boundary): lands behind the last owner
template <bool BITS, bool REP>
__global__ __launch_bounds__(NODES_THREADS) void k_nodes(const uint32_t *__restrict__ key32, const OccPay *__restrict__ pay, uint64_t n,
                                               uint64_t base, const uint32_t *__restrict__ rec_flag, sw_kmer *__restrict__ kmers,
                                               uint64_t *__restrict__ node_hash, uint32_t *__restrict__ node_start,
                                               uint32_t *__restrict__ rank_direct,
                                               uint64_t *__restrict__ uval,
                                               unsigned long long *__restrict__ tbits, unsigned long long *__restrict__ nbits,
                                               unsigned long long *__restrict__ tile_state, uint32_t *__restrict__ ticket,
                                               uint32_t *__restrict__ n_nodes_out, uint32_t *__restrict__ order_bad)
{
    __shared__ uint32_t s_tile, s_excl;
    __shared__ uint32_t s_row[NODES_ROWS * NODES_WAVES];   // heads of (row, wave): counts, then exclusive offsets
    static_assert(NODES_ROWS * NODES_WAVES <= 128, "the (row, wave) group counts are scanned by one wave, two per lane");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t s0 = (uint64_t)tile * NODES_TILE;
    uint32_t k[NODES_ROWS][2], within[NODES_ROWS], headm = 0, prec[NODES_ROWS];
    OccPay p[NODES_ROWS][2];
    unsigned long long misorder = 0;   // (wave-uniform: lanes that saw (hash, stream index) not ascending)
#pragma unroll
    for (int r = 0; r < NODES_ROWS; ++r) {
        const uint64_t s = s0 + (uint64_t)r * NODES_ROW + 2 * threadIdx.x;   // occurrences s, s + 1
        k[r][0] = k[r][1] = 0;
        p[r][0] = p[r][1] = OccPay{0, 0, 0, 0};
        if (s + 1 < n) {
            const uint2 kk = *reinterpret_cast<const uint2 *>(key32 + s);
            k[r][0] = kk.x;
            k[r][1] = kk.y;
            p[r][0] = pay[s];
            p[r][1] = pay[s + 1];
        } else if (s < n) {
            k[r][0] = key32[s];
            p[r][0] = pay[s];
        }
        // the occurrence before s: the odd one of the lane below, or (lane 0) a load
        uint32_t pk = __shfl_up(k[r][1], 1, 64), plow = __shfl_up(p[r][1].low, 1, 64), pidx = __shfl_up(p[r][1].idx, 1, 64);
        prec[r] = __shfl_up(p[r][1].rec, 1, 64);
        if (lane == 0 && s < n && s) {
            pk = key32[s - 1];
            const OccPay q = pay[s - 1];
            plow = q.low;
            prec[r] = q.rec;
            pidx = q.idx;
        }
        const bool h0 = s < n && (s == 0 || k[r][0] != pk || p[r][0].low != plow);
        const bool h1 = s + 1 < n && (k[r][1] != k[r][0] || p[r][1].low != p[r][0].low);
        // Always-on order guard (r05): what arrives here must be THE stable sort by hash of the (record_idx, pos) stream
        // (build_internals.cpp:76-144, 173, 203-218) -- hashes ascending, and inside a run of equal hashes the stream indices
        // ascending.  The radix passes rank by an LDS atomic whose lane order is checked at start-up but is no architectural
        // promise (radix.hip); a violation is counted here and group_occurrences re-sorts without that ranking.
#ifndef SW_NO_ORDER_GUARD   // (A/B timing of the guard only: tests/tools/build_variant.sh noguard -DSW_NO_ORDER_GUARD)
        {   // (hash, idx) as (key32, low) lexicographic then idx: one 64-bit compare per pair + the equal-hash index test
            const unsigned long long hp = ((unsigned long long)pk << 32) | plow, h0v = ((unsigned long long)k[r][0] << 32) | p[r][0].low,
                                     h1v = ((unsigned long long)k[r][1] << 32) | p[r][1].low;
            const bool bad0 = s < n && s && (h0v < hp || (!h0 && p[r][0].idx <= pidx));
            const bool bad1 = s + 1 < n && (h1v < h0v || (!h1 && p[r][1].idx <= p[r][0].idx));
            misorder |= __ballot(bad0 || bad1);
        }
#else
        (void)pidx;
#endif
        const unsigned long long b0 = __ballot(h0), b1 = __ballot(h1), lt = (1ull << lane) - 1ull;
        within[r] = (uint32_t)__popcll(b0 & lt) + (uint32_t)__popcll(b1 & lt);   // heads of the wave's row before occurrence s
        if (h0) headm |= 1u << (2 * r);
        if (h1) headm |= 2u << (2 * r);
        if (lane == 0) s_row[r * NODES_WAVES + wave] = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
    }
    if (misorder && lane == 0) atomicAdd(order_bad, (uint32_t)__popcll(misorder));   // (never, on a device whose LDS unit serves the lanes of an atomic in lane order)
    __syncthreads();
    if (wave == 0) {
        // exclusive offsets of the (row, wave) groups (two per lane), the tile's total, then the look-back
        constexpr uint32_t GROUPS = NODES_ROWS * NODES_WAVES;
        const uint32_t c0 = (2 * lane < GROUPS) ? s_row[2 * lane] : 0u, c1 = (2 * lane + 1 < GROUPS) ? s_row[2 * lane + 1] : 0u;
        uint32_t incl = c0 + c1;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        if (2 * lane < GROUPS) s_row[2 * lane] = incl - c0 - c1;
        if (2 * lane + 1 < GROUPS) s_row[2 * lane + 1] = incl - c1;
        if (lane == 0)
            __hip_atomic_store(&tile_state[tile], (tile == 0 ? TS_INC : TS_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile) {
            int64_t look = (int64_t)tile - 1;
            for (;;) {
                const int64_t idx = look - lane;   // lane 0 reads the nearest predecessor
                const unsigned long long st = idx >= 0 ? __hip_atomic_load(&tile_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : TS_INC;   // before the first tile: nothing
                const unsigned long long inc = __ballot((st >> 62) == 2), none = __ballot((st >> 62) == 0);
                const uint32_t first_inc = inc ? (uint32_t)__builtin_ctzll(inc) : 64u;
                const unsigned long long needed = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
                if (none & needed) {   // a predecessor in reach has not published yet (it is running: see above)
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t v = (lane <= first_inc) ? (uint32_t)st : 0u;
                for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
                excl += v;
                if (first_inc < 64) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&tile_state[tile], TS_INC | (unsigned long long)(excl + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            if (s0 + NODES_TILE >= n) *n_nodes_out = excl + total;   // the last tile
        }
    }
    __syncthreads();
    const uint32_t excl = s_excl;
#pragma unroll
    for (int r = 0; r < NODES_ROWS; ++r) {
        const uint64_t s = s0 + (uint64_t)r * NODES_ROW + 2 * threadIdx.x;
        const bool live0 = s < n, live1 = s + 1 < n;
        const bool h0 = (headm >> (2 * r)) & 1u, h1 = (headm >> (2 * r + 1)) & 1u;
        // node = heads up to and including the occurrence, - 1
        const uint32_t nid0 = excl + s_row[r * NODES_WAVES + wave] + within[r] + (h0 ? 1u : 0u) - 1u;
        const uint32_t nid1 = nid0 + (h1 ? 1u : 0u);
        uint32_t w0 = nid0, w1 = nid1;   // the rank words handed to the adjacency
        uint32_t f0 = 0, f1 = 0, pf = 0;
        if (BITS || REP) {
            // rec_flag[r] = assembly << 1 | is_target
            f0 = live0 ? rec_flag[p[r][0].rec] : 0u;
            f1 = live1 ? rec_flag[p[r][1].rec] : 0u;
            pf = __shfl_up(f1, 1, 64);
            if (lane == 0 && live0 && s) pf = rec_flag[prec[r]];
        }
        if (REP) {
            // the occurrence after s + 1: the even one of the lane above, or (lane 63) loads
            uint32_t nk = __shfl_down(k[r][0], 1, 64), nlow = __shfl_down(p[r][0].low, 1, 64), nf = __shfl_down(f0, 1, 64);
            bool nlive = __shfl_down((int)live0, 1, 64) != 0;
            if (lane == 63) {
                nlive = s + 2 < n;
                if (nlive) {
                    nk = key32[s + 2];
                    const OccPay q = pay[s + 2];
                    nlow = q.low;
                    nf = rec_flag[q.rec];
                }
            }
            const bool same01 = live1 && !h1 && (f1 >> 1) == (f0 >> 1);
            const bool same_p0 = live0 && !h0 && (f0 >> 1) == (pf >> 1);
            const bool same_1n = live1 && nlive && nk == k[r][1] && nlow == p[r][1].low && (nf >> 1) == (f1 >> 1);
            if (same_p0 || same01) w0 |= RANK_REP;
            if (same01 || same_1n) w1 |= RANK_REP;
        }
        if (live1) {
            *reinterpret_cast<uint4 *>(kmers + s) = make_uint4(p[r][0].pos, p[r][0].rec, p[r][1].pos, p[r][1].rec);
            if (uval)
                *reinterpret_cast<ulonglong2 *>(uval + s) = make_ulonglong2(((uint64_t)p[r][0].idx << 32) | w0,
                                                                          ((uint64_t)p[r][1].idx << 32) | w1);
        } else if (live0) {
            sw_kmer km;
            km.pos = p[r][0].pos;
            km.record_idx = p[r][0].rec;
            kmers[s] = km;
            if (uval) uval[s] = ((uint64_t)p[r][0].idx << 32) | w0;   // the unsort's element: index above, node below
        }
        if (rank_direct) {
            if (live0) rank_direct[p[r][0].idx] = w0;
            if (live1) rank_direct[p[r][1].idx] = w1;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool head = e ? h1 : h0;
            if (head) {   // (heads are live)
                const uint32_t nid = e ? nid1 : nid0;
                // r05: the head leaves its hash and its position in two DENSE arrays (12 B per node, consecutive nodes of a tile
                // side by side) instead of 16 B inside the 40-byte node (three partly written lines per 128 B); k_finish_nodes
                // streams over them and writes every node whole, coalesced -- and the edges read the dense hashes directly.
                node_hash[nid] = ((uint64_t)k[r][e] << 32) | p[r][e].low;
                node_start[nid] = (uint32_t)(s + e);
            }
        }
        if (BITS) {
            const bool first0 = live0 && (h0 || (f0 >> 1) != (pf >> 1)), first1 = live1 && (h1 || (f1 >> 1) != (f0 >> 1));
            const unsigned long long t0 = __ballot(first0 && (f0 & 1u)), t1 = __ballot(first1 && (f1 & 1u));
            const unsigned long long g0 = __ballot(first0 && !(f0 & 1u)), g1 = __ballot(first1 && !(f1 & 1u));
            // the wave's 128 occurrences start at a multiple of 128: words (s >> 6) and (s >> 6) + 1 of lane 0's s
            if (lane == 0 && live0) {
                const uint64_t wd = s >> 6;
                tbits[wd] = interleave32((uint32_t)t0, (uint32_t)t1);
                nbits[wd] = interleave32((uint32_t)g0, (uint32_t)g1);
                if (s + 64 < n) {
                    tbits[wd + 1] = interleave32((uint32_t)(t0 >> 32), (uint32_t)(t1 >> 32));
                    nbits[wd + 1] = interleave32((uint32_t)(g0 >> 32), (uint32_t)(g1 >> 32));
                }
            }
        }
    }
}

// bits [a, b) of a bitmap of 64-bit words
__device__ __forceinline__ uint32_t popc_range(const unsigned long long *__restrict__ w, uint64_t a, uint64_t b)
{
    if (a >= b) return 0;
    const uint64_t wa = a >> 6, wb = (b - 1) >> 6;
    const unsigned long long ma = ~0ull << (a & 63u), mb = ~0ull >> (63u - ((b - 1) & 63u));
    if (wa == wb) return (uint32_t)__popcll(w[wa] & ma & mb);
    uint32_t c = (uint32_t)__popcll(w[wa] & ma) + (uint32_t)__popcll(w[wb] & mb);
    for (uint64_t i = wa + 1; i < wb; ++i) c += (uint32_t)__popcll(w[i]);
    return c;
}

// Completes the nodes (r05; k_pen_bits / k_node_stops before): node i = {hash, start, stop, n_tar, n_neg, penalty} from the dense
// arrays k_nodes left (hash, position of the node's first occurrence; stop = the next node's start, `end` for the last one) and
// -- BITS -- the per-node distinct target / non-target assemblies from the two first-of-assembly bitmaps + the penalty
// (filter.cpp:89-90, 125-134); without BITS the counts are zero (python_bindings.cpp:50-90 hands out n_tar = n_neg = penalty = 0).
// A workgroup stages its 256 nodes (10 KiB) in LDS and writes them with 16-byte stores, consecutive lanes to consecutive
// addresses: every line of the node array is written once, whole (one GPU's share of 100 000 iid genomes has 5.4e8 nodes:
// 21.7 GB of nodes; k_pen_bits wrote 24 of every 40 bytes at a 40-byte stride, 11.3 ms).
constexpr int FIN_THREADS = 256;
template <bool BITS>
__global__ __launch_bounds__(FIN_THREADS) void k_finish_nodes(sw_node *__restrict__ nodes, uint64_t n_nodes, const uint64_t *__restrict__ node_hash,
                                                              const uint32_t *__restrict__ node_start, uint64_t base, uint64_t end,
                                                              const unsigned long long *__restrict__ tbits,
                                                              const unsigned long long *__restrict__ nbits, double inv_tar, double inv_neg)
{
    static_assert(sizeof(sw_node) == 40, "five 8-byte words per node");
    __shared__ __align__(16) unsigned long long stage[FIN_THREADS * 5];
    const uint64_t i0 = (uint64_t)blockIdx.x * FIN_THREADS, i = i0 + threadIdx.x;
    if (i < n_nodes) {
        const uint32_t a = node_start[i];
        const uint64_t b = i + 1 < n_nodes ? (uint64_t)node_start[i + 1] : end - base;
        uint32_t n_tar = 0, n_neg = 0;
        double pen = 0.0;
        if (BITS) {
            n_tar = popc_range(tbits, a, b);
            n_neg = popc_range(nbits, a, b);
            {
// filter.cpp:132-134 evaluated with separate IEEE multiply / add / sqrt (no FMA contraction)
#pragma clang fp contract(off)
                const double ft = (double)n_tar * inv_tar;
                const double fn = (double)n_neg * inv_neg;
                const double omf = 1.0 - ft;
                const double aa = omf * omf;
                const double bb = fn * fn;
                const double sum = aa + bb;
                pen = __dsqrt_rn(sum);
            }
        }
        unsigned long long *w = stage + threadIdx.x * 5;
        w[0] = node_hash[i];
        w[1] = base + a;
        w[2] = base + b;
        w[3] = ((unsigned long long)n_neg << 32) | n_tar;
        w[4] = (unsigned long long)__double_as_longlong(pen);
    }
    __syncthreads();
    const uint64_t cnt = min((uint64_t)FIN_THREADS, n_nodes - i0);          // nodes of this workgroup
    const uint32_t n16 = (uint32_t)(cnt * 40 / 16), tail8 = (uint32_t)((cnt * 40) % 16) / 8;   // 16-byte pieces (+ one 8-byte word)
    uint4 *dst = reinterpret_cast<uint4 *>(nodes + i0);                      // (i0 * 40 is a multiple of 16)
    const uint4 *src = reinterpret_cast<const uint4 *>(stage);
    for (uint32_t t = threadIdx.x; t < n16; t += FIN_THREADS) dst[t] = src[t];
    if (tail8 && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(nodes + i0)[n16 * 2] = stage[n16 * 2];
}

__global__ void k_rec_flag(const uint32_t *__restrict__ rec_asm, const uint8_t *__restrict__ is_target, uint64_t n_records,
                           uint32_t *__restrict__ rec_flag)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_records) {
        const uint32_t a = rec_asm[r];
        rec_flag[r] = (a << 1) | ((is_target && is_target[a]) ? 1u : 0u);
    }
}

// ---- open-addressing table hash -> node rank (the ankerl::unordered_dense role, build.cpp:66-73, 153-168) ----------
// Linear probing over 16-B slots {hash, rank}, claimed with a 64-bit atomicCAS on the hash word, load <= 0.5.  Built
// from the distinct node hashes once the sort has produced them, then probed once per occurrence to bring the node
// rank to (record_idx, pos) order -- the alternative to the unsort below, kept behind SEQWIN_AMD_RANKS=table because
// it measured slower (one random 128-B line per occurrence against two streaming passes).
constexpr unsigned long long TABLE_EMPTY = ~0ull;
__device__ __forceinline__ uint64_t table_mix(uint64_t x)   // the hashes are already uniform: a cheap fold is enough
{
    return x ^ (x >> 29);
}

__global__ void k_table_build(const sw_node *__restrict__ nodes, uint64_t n_nodes, ulonglong2 *__restrict__ slots, uint64_t mask,
                              uint32_t *__restrict__ rank_of_empty_key)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_nodes) return;
    const unsigned long long h = nodes[r].hash;
    if (h == TABLE_EMPTY) {   // the one hash value that cannot be stored lives beside the table
        *rank_of_empty_key = (uint32_t)r;
        return;
    }
    uint64_t s = table_mix(h) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&slots[s].x, TABLE_EMPTY, h);
        if (old == TABLE_EMPTY) {   // claimed (node hashes are distinct: `old == h` cannot happen)
            slots[s].y = r;
            return;
        }
        s = (s + 1) & mask;
    }
}

__global__ void k_table_lookup(const uint64_t *__restrict__ hash, uint64_t n, const ulonglong2 *__restrict__ slots, uint64_t mask,
                               const uint32_t *__restrict__ rank_of_empty_key, uint32_t *__restrict__ rank)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h = hash[i];
    if (h == TABLE_EMPTY) {
        rank[i] = *rank_of_empty_key;
        return;
    }
    uint64_t s = table_mix(h) & mask;
    for (;;) {
        const ulonglong2 e = slots[s];
        if (e.x == h) {
            rank[i] = (uint32_t)e.y;
            return;
        }
        if (e.x == TABLE_EMPTY) {   // cannot happen for a hash of this index; never spin
            rank[i] = 0xFFFFFFFFu;
            return;
        }
        s = (s + 1) & mask;
    }
}

// ---- unsort: node rank of every occurrence, back in (record_idx, pos) order --------------------------------------
// (index, rank) pairs arrive in hash order, i.e. with random indices: scattering 4 B to rank[index] costs a 128-B HBM
// line each (26 ms for 745 M on MI355X).  Instead the pairs -- one 64-bit word, index << 32 | rank -- are first brought
// into buckets of 2^14 consecutive indices (a keys-only radix sort on the index's high bits only; the indices are a
// permutation of [0, n), so bucket b is exactly positions [b * 2^14, (b + 1) * 2^14)), and one workgroup per bucket
// scatters its pairs inside LDS and writes the 64 KiB of ranks out in order.
__global__ __launch_bounds__(1024) void k_unsort_bucket(const uint64_t *__restrict__ uval, uint64_t n, uint32_t *__restrict__ rank)
{
    __shared__ uint32_t sr[UNSORT_RANGE];
    const uint64_t b0 = (uint64_t)blockIdx.x * UNSORT_RANGE;
    const uint32_t cnt = (uint32_t)min((uint64_t)UNSORT_RANGE, n - b0);
    // (eight loads in flight per thread: written as a plain loop the compiler waits for every word before it requests the next)
#pragma unroll
    for (uint32_t h = 0; h < UNSORT_RANGE / 1024u; h += 8) {
        uint64_t v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t t = threadIdx.x + (h + j) * 1024u;
            v[j] = t < cnt ? uval[b0 + t] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
            if (threadIdx.x + (h + j) * 1024u < cnt) sr[(uint32_t)(v[j] >> 32) & (UNSORT_RANGE - 1u)] = (uint32_t)v[j];
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < cnt; t += 1024) rank[b0 + t] = sr[t];
}

// ---- multi-GPU merge helpers ---------------------------------------------------------------------
// rows[s] = (hash of the node owning occurrence s, pos | (record_idx + rec_offset) << 32)
__global__ void k_node_starts(const sw_node *__restrict__ nodes, uint64_t n_nodes, uint32_t *__restrict__ start)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) start[i] = (uint32_t)nodes[i].start;
}

__global__ void k_occ_rows(const sw_kmer *__restrict__ kmers, const sw_node *__restrict__ nodes,
                           const uint32_t *__restrict__ start, uint64_t n_nodes, uint64_t n_kmers, uint64_t rec_offset,
                           uint64_t *__restrict__ rows)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_kmers) return;
    uint64_t lo = 0, hi = n_nodes;  // last node with start <= s (compact 4-byte starts: cache resident)
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (start[mid] <= s) lo = mid; else hi = mid;
    }
    rows[2 * s] = nodes[lo].hash;
    rows[2 * s + 1] = (uint64_t)kmers[s].pos | (((uint64_t)kmers[s].record_idx + rec_offset) << 32);
}

__global__ void k_strided_copy(const uint64_t *__restrict__ src, uint32_t stride, uint32_t col, uint64_t n,
                               uint64_t *__restrict__ dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i * stride + col];
}

__global__ void k_gather_col(const uint64_t *__restrict__ rows, uint32_t stride, uint32_t col,
                             const uint32_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = rows[(uint64_t)idx[i] * stride + col];
}

__global__ void k_rebase_nodes(sw_node *__restrict__ nodes, uint64_t n_nodes, uint64_t base)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) {
        nodes[i].start += base;
        nodes[i].stop += base;
    }
}

struct PairHeadFlag {
    const uint64_t *a, *b;
    __host__ __device__ uint32_t operator()(uint64_t s) const
    {
        return (s == 0 || a[s] != a[s - 1] || b[s] != b[s - 1]) ? 1u : 0u;
    }
};

__global__ void k_merge_edge_heads(const uint64_t *__restrict__ f, const uint64_t *__restrict__ sd,
                                   const uint32_t *__restrict__ ecum, uint64_t n, uint64_t *__restrict__ edge_start)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (s == 0 || f[s] != f[s - 1] || sd[s] != sd[s - 1]) edge_start[ecum[s] - 1] = s;
}

__global__ void k_merge_edges(const uint64_t *__restrict__ f, const uint64_t *__restrict__ sd,
                              const uint64_t *__restrict__ wcum, const uint64_t *__restrict__ edge_start,
                              uint64_t n_edges, uint64_t n, sw_edge *__restrict__ edges)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint64_t s = edge_start[e];
    const uint64_t s1 = (e + 1 < n_edges) ? edge_start[e + 1] : n;
    edges[e].first = f[s];
    edges[e].second = sd[s];
    edges[e].weight = wcum[s1 - 1] - (s ? wcum[s - 1] : 0ull);   // build_internals.cpp:283-285
}

__global__ void k_lower_bounds(const sw_node *__restrict__ nodes, uint64_t n_nodes, uint64_t n_kmers,
                               const sw_edge *__restrict__ edges, uint64_t n_edges, const uint64_t *__restrict__ nb,
                               const uint64_t *__restrict__ eb, uint32_t n_bounds, uint64_t *__restrict__ occ_split,
                               uint64_t *__restrict__ edge_split)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_bounds) return;
    {
        uint64_t lo = 0, hi = n_nodes;  // first node with hash >= nb[j]
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (nodes[mid].hash < nb[j]) lo = mid + 1; else hi = mid;
        }
        occ_split[j] = (lo < n_nodes) ? nodes[lo].start : n_kmers;
    }
    {
        uint64_t lo = 0, hi = n_edges;  // first edge with first >= eb[j]
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (edges[mid].first < eb[j]) lo = mid + 1; else hi = mid;
        }
        edge_split[j] = lo;
    }
}

// ---- get_penalty ------------------------------------------------------------------------------------
// X[s] = (target-assembly change << 32) | non-target-assembly change, between occurrence s-1 and s.
// Y[s] = (record_idx decreased << 32) | record_idx out of range.
// (For occurrences that come from outside: the C-ABI get_penalty, the multi-GPU merges, SEQWIN_AMD_CHECK_ORDER=1.  The
//  single-GPU build's own counts come from the bitmaps k_nodes writes: k_pen_bits.)
__global__ void k_pen_flags(const sw_kmer *__restrict__ kmers, uint64_t n, const uint32_t *__restrict__ rec_asm,
                            uint64_t n_records, const uint8_t *__restrict__ is_target, uint64_t *__restrict__ X,
                            uint64_t *__restrict__ Y)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t r = kmers[s].record_idx;
    const bool oob = r >= n_records;
    uint64_t x = 0, y = oob ? 1ull : 0ull;
    if (s) {
        const uint32_t rp = kmers[s - 1].record_idx;
        if (r < rp) y |= 1ull << 32;
        if (!oob) {
            const uint32_t a = rec_asm[r];
            const uint32_t ap = (rp < n_records) ? rec_asm[rp] : 0xFFFFFFFFu;
            if (a != ap) x = is_target[a] ? (1ull << 32) : 1ull;
        }
    }
    X[s] = x;
    Y[s] = y;
}

__global__ void k_pen_nodes(const sw_kmer *__restrict__ kmers, uint64_t n_kmers, sw_node *__restrict__ nodes,
                            uint64_t n_nodes, const uint32_t *__restrict__ rec_asm, uint64_t n_records,
                            const uint8_t *__restrict__ is_target, const uint64_t *__restrict__ XS,
                            const uint64_t *__restrict__ YS, double inv_tar, double inv_neg,
                            uint32_t *__restrict__ err)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint64_t start = nodes[i].start, stop = nodes[i].stop;
    if (start == stop) {  // filter.cpp:95-100
        nodes[i].n_tar = 0;
        nodes[i].n_neg = 0;
        nodes[i].penalty = 1.0;
        return;
    }
    if (start > stop || stop > n_kmers) {
        atomicOr(err, 1u);
        return;
    }
    {
        const uint64_t y = YS[stop - 1] - (start ? YS[start - 1] : 0ull);
        // out-of-range anywhere in [start, stop); decreasing strictly inside (start, stop)
        if ((uint32_t)y) {
            atomicOr(err, 2u);
            return;
        }
        const uint64_t dec = (YS[stop - 1] - YS[start]) >> 32;
        if (dec) {
            atomicOr(err, 4u);
            return;
        }
    }
    const uint32_t a0 = rec_asm[kmers[start].record_idx];
    const uint64_t x = XS[stop - 1] - XS[start];
    const uint32_t t0 = is_target[a0] ? 1u : 0u;
    const uint32_t n_tar = t0 + (uint32_t)(x >> 32);
    const uint32_t n_neg = (1u - t0) + (uint32_t)x;
    nodes[i].n_tar = n_tar;
    nodes[i].n_neg = n_neg;
    {
// filter.cpp:132-134 evaluated with separate IEEE multiply / add / sqrt (no FMA contraction)
#pragma clang fp contract(off)
        const double ft = (double)n_tar * inv_tar;
        const double fn = (double)n_neg * inv_neg;
        const double omf = 1.0 - ft;
        const double a = omf * omf;
        const double b = fn * fn;
        const double sum = a + b;
        nodes[i].penalty = __dsqrt_rn(sum);
    }
}

// ---- edges --------------------------------------------------------------------------------------------
// Adjacency keys of consecutive minimizers of a record (build.cpp:177-189) from the node rank and the record of every
// occurrence, both in (record_idx, pos) order: key = (rank_lo << nb) | rank_hi, record boundaries -> sentinel (sorts last);
// the assembly is a 32-bit value next to the key (the (pair, assembly) form: SEQWIN_AMD_NO_PACKED_EDGES=1, 2^31 nodes or more).
// Four consecutive occurrences per thread: 16-B loads of ranks and records, 2 x 16-B stores of keys.
__global__ __launch_bounds__(256) void k_adj_keys(const uint32_t *__restrict__ rec, const uint32_t *__restrict__ rank,
                                                  const uint32_t *__restrict__ rec_asm, uint64_t n, unsigned nb, uint64_t sentinel,
                                                  uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 + 1 >= n) return;
    uint32_t r[5], k[5];
    if (i0 + 4 < n) {
        const uint4 rv = *reinterpret_cast<const uint4 *>(rec + i0), kv = *reinterpret_cast<const uint4 *>(rank + i0);
        r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w; r[4] = rec[i0 + 4];
        k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w; k[4] = rank[i0 + 4];
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            r[j] = (i0 + j < n) ? rec[i0 + j] : 0xFFFFFFFFu;
            k[j] = (i0 + j < n) ? rank[i0 + j] : 0u;
        }
    }
    uint64_t out[4];
    uint32_t asmv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool pair = i0 + j + 1 < n && r[j] == r[j + 1];
        uint32_t u = k[j], v = k[j + 1];
        if (v < u) { const uint32_t t = u; u = v; v = t; }
        const uint32_t a = pair ? rec_asm[r[j]] : 0xFFFFFFFFu;
        const uint64_t pk = ((uint64_t)u << nb) | v;
        out[j] = pair ? pk : sentinel;
        asmv[j] = a;
    }
    if (i0 + 4 < n) {   // all four keys exist (m = n - 1 keys): vector stores
        *reinterpret_cast<ulonglong2 *>(key + i0) = make_ulonglong2(out[0], out[1]);
        *reinterpret_cast<ulonglong2 *>(key + i0 + 2) = make_ulonglong2(out[2], out[3]);
        *reinterpret_cast<uint4 *>(val + i0) = make_uint4(asmv[0], asmv[1], asmv[2], asmv[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j + 1 < n) {
                key[i0 + j] = out[j];
                val[i0 + j] = asmv[j];
            }
    }
}

// The same keys without any assembly: the weight of a pair is the NUMBER of its adjacency records minus the records that
// repeat the pair inside one assembly, and only records touching an occurrence marked RANK_REP can do that (two records
// of one pair in one assembly contain two occurrences of one of its nodes there).  Those candidates -- none on random
// genomes, a few per cent on real ones (repeats) -- are also appended to a side list with their assembly
// (subtract_repeats); everything else needs no assembly at all: keys-only sort, run lengths.
struct RecArray {    // record of every occurrence as its own array (single-GPU build)
    const uint32_t *rec;
    __device__ void load4(uint64_t i0, uint32_t *r) const
    {
        const uint4 rv = *reinterpret_cast<const uint4 *>(rec + i0);
        r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
    }
    __device__ uint32_t at(uint64_t i) const { return rec[i]; }
};
struct RecOfKmer {   // record = top half of pos | record << 32 (exchange form)
    const uint64_t *kmer;
    __device__ void load4(uint64_t i0, uint32_t *r) const
    {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(kmer + i0), b = *reinterpret_cast<const ulonglong2 *>(kmer + i0 + 2);
        r[0] = (uint32_t)(a.x >> 32); r[1] = (uint32_t)(a.y >> 32); r[2] = (uint32_t)(b.x >> 32); r[3] = (uint32_t)(b.y >> 32);
    }
    __device__ uint32_t at(uint64_t i) const { return (uint32_t)(kmer[i] >> 32); }
};

// With `hist` (radix.hip sorts the keys next: radix_layout): the workgroup takes `iters` consecutive blocks of 1024 records and
// counts the digits of the keys it writes -- hist[pass][digit], hbits wide, key bits [0, end_bit) -- so that the sort needs
// no counting sweep over the 6 GB of keys.
template <class Rec>
__global__ __launch_bounds__(256) void k_adj_pairs(const Rec rec, const uint32_t *__restrict__ rank,
                                                   const uint32_t *__restrict__ rec_asm, uint32_t asm_base, uint64_t n, unsigned nb,
                                                   uint64_t sentinel, uint64_t *__restrict__ key, uint64_t *__restrict__ cand_key,
                                                   uint32_t *__restrict__ cand_asm, unsigned long long *__restrict__ n_cand,
                                                   unsigned long long *__restrict__ hist, unsigned hbits, unsigned hpasses,
                                                   unsigned end_bit, uint32_t iters)
{
    __shared__ uint32_t sh_hist[8 * 512];
    if (hist) {
        for (uint32_t i = threadIdx.x; i < (hpasses << hbits); i += 256) sh_hist[i] = 0;
        __syncthreads();
    }
    for (uint32_t it = 0; it < iters; ++it) {
    const uint64_t i0 = (((uint64_t)blockIdx.x * iters + it) * blockDim.x + threadIdx.x) * 4;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t r[5], k[5];
    uint64_t out[4];
    uint32_t cm = 0;   // bit j: record i0 + j is a candidate
    if (i0 + 1 < n) {
        if (i0 + 4 < n) {
            const uint4 kv = *reinterpret_cast<const uint4 *>(rank + i0);
            rec.load4(i0, r);
            r[4] = rec.at(i0 + 4);
            k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w; k[4] = rank[i0 + 4];
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                r[j] = (i0 + j < n) ? rec.at(i0 + j) : 0xFFFFFFFFu;
                k[j] = (i0 + j < n) ? rank[i0 + j] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pair = i0 + j + 1 < n && r[j] == r[j + 1];
            uint32_t u = k[j] & ~RANK_REP, v = k[j + 1] & ~RANK_REP;
            if (v < u) { const uint32_t t = u; u = v; v = t; }
            out[j] = pair ? (((uint64_t)u << nb) | v) : sentinel;
            if (pair && ((k[j] | k[j + 1]) & RANK_REP)) cm |= 1u << j;
        }
        if (i0 + 4 < n) {   // all four keys exist (m = n - 1 keys): vector stores
            *reinterpret_cast<ulonglong2 *>(key + i0) = make_ulonglong2(out[0], out[1]);
            *reinterpret_cast<ulonglong2 *>(key + i0 + 2) = make_ulonglong2(out[2], out[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j + 1 < n) key[i0 + j] = out[j];
        }
        if (hist) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j + 1 < n)
                    for (unsigned p = 0; p < hpasses; ++p) {
                        const unsigned shp = hbits * p, wd = min(hbits, end_bit - shp);
                        atomicAdd(&sh_hist[(p << hbits) + ((uint32_t)(out[j] >> shp) & ((1u << wd) - 1u))], 1u);
                    }
        }
    }
    if (__any(cm != 0)) {   // rare: one atomic per wave that has candidates
        const uint32_t c = (uint32_t)__popc(cm);
        uint32_t incl = c;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        unsigned long long base = 0;
        if (lane == 63) base = atomicAdd(n_cand, (unsigned long long)total);
        base = __shfl(base, 63, 64) + (incl - c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((cm >> j) & 1u) {
                cand_key[base] = out[j];
                cand_asm[base] = asm_base + rec_asm[r[j]];
                ++base;
            }
    }
    }   // (iters)
    if (hist) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (hpasses << hbits); i += 256)
            if (sh_hist[i]) atomicAdd(&hist[i], (unsigned long long)sh_hist[i]);
    }
}

// The unsort's last step and k_adj_pairs in one kernel (single-GPU build, ranks with repeat marks): a workgroup scatters a bucket
// of 2^14 consecutive indices inside LDS as k_unsort_bucket does and writes the adjacency keys of those occurrences from
// there -- the rank array (4 B written and read again per occurrence) is never made.  The pair that straddles two buckets is
// left to k_adj_bounds (the first and last rank of every bucket go to edge_rank).
template <class Rec>
__global__ __launch_bounds__(1024, 8) void k_unsort_adj(const uint64_t *__restrict__ uval, uint64_t n, const Rec rec,
                                                     const uint32_t *__restrict__ rec_asm, uint32_t asm_base, unsigned nb, uint64_t sentinel,
                                                     uint64_t *__restrict__ key, uint64_t *__restrict__ cand_key,
                                                     uint32_t *__restrict__ cand_asm, unsigned long long *__restrict__ n_cand,
                                                     unsigned long long *__restrict__ hist, unsigned hbits, unsigned hpasses,
                                                     unsigned end_bit, uint32_t *__restrict__ edge_rank, uint32_t per_wg)
{
    __shared__ uint32_t sr[UNSORT_RANGE];
    extern __shared__ uint32_t sh_hist[];   // (hpasses << hbits counters, dynamic: 12 KB at 15 000 genomes lets two workgroups share a CU)
    // r05: a bucket's candidates take their place in the list with ONE global atomic per workgroup and bucket: the pair loop only
    // notes which of a thread's pairs are candidates (16 bits), the waves' counts meet in LDS behind a barrier, and the threads
    // that have any form those keys again from the ranks still in LDS.  One atomic per wave that had any (r02-r04) is rare on the
    // pan-genome sets; on 12 500 iid genomes at k = 15 -- 2e7 nodes for 6e8 occurrences, 0.5-1 % of the pairs touch a node that
    // recurs in its assembly -- it was over a million atomics on one address, serialised across the XCDs: 18 of the kernel's
    // 21 ms (profiles/r05_random100k_k15_kernel_stats.txt; a 128-entry staging area in LDS overflowed on that set and gained 2 ms).
    __shared__ uint32_t s_wtot[16];
    __shared__ unsigned long long s_cbase;
    if (hist)
        for (uint32_t i = threadIdx.x; i < (hpasses << hbits); i += 1024) sh_hist[i] = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t q = 0; q < per_wg; ++q) {
        const uint64_t bkt = (uint64_t)blockIdx.x * per_wg + q, b0 = bkt * UNSORT_RANGE;
        if (b0 >= n) break;   // (workgroup-uniform)
        const uint32_t cnt = (uint32_t)min((uint64_t)UNSORT_RANGE, n - b0);
        __syncthreads();      // the previous bucket's ranks have been read (first round: the counters are zero)
        // the bucket's words and the records of its occurrences are requested in batches (as a plain loop the compiler waits
        // for every word before it requests the next: 16 + 4 serialized round trips per bucket, 5.0 ms of the kernel's 5.8)
        constexpr uint32_t ITERS = UNSORT_RANGE / 4096u;
        uint32_t rr[ITERS][5];
#pragma unroll
        for (uint32_t h = 0; h < UNSORT_RANGE / 1024u; h += 8) {
            uint64_t v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t t = threadIdx.x + (h + j) * 1024u;
                v[j] = t < cnt ? uval[b0 + t] : 0ull;
            }
            if (h == 0) {
#pragma unroll
                for (uint32_t it = 0; it < ITERS; ++it) {
                    const uint32_t t0 = (it * 1024u + threadIdx.x) * 4u;
                    if (t0 + 4 < cnt) {
                        rec.load4(b0 + t0, rr[it]);
                        rr[it][4] = rec.at(b0 + t0 + 4);
                    } else {
#pragma unroll
                        for (int j = 0; j < 5; ++j) rr[it][j] = (t0 + j < cnt) ? rec.at(b0 + t0 + j) : 0xFFFFFFFFu;
                    }
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                if (threadIdx.x + (h + j) * 1024u < cnt) sr[(uint32_t)(v[j] >> 32) & (UNSORT_RANGE - 1u)] = (uint32_t)v[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            edge_rank[2 * bkt] = sr[0];
            edge_rank[2 * bkt + 1] = sr[cnt - 1];
        }
        uint32_t cmask = 0;   // bit 4 it + j: the pair (t0 + j, t0 + j + 1) of iteration `it` is a candidate
#pragma unroll
        for (uint32_t it = 0; it < ITERS; ++it) {
            const uint32_t t0 = (it * 1024u + threadIdx.x) * 4u;   // pairs (t0 + j, t0 + j + 1), j < 4, that lie inside the bucket
            const uint64_t i0 = b0 + t0;
            uint32_t r[5], k[5];
            uint64_t out[4];
            uint32_t cm = 0;   // bit j: record i0 + j is a candidate
            if (t0 + 1 < cnt) {
#pragma unroll
                for (int j = 0; j < 5; ++j) r[j] = rr[it][j];
                if (t0 + 4 < cnt) {
                    const uint4 kv = *reinterpret_cast<const uint4 *>(&sr[t0]);
                    k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w; k[4] = sr[t0 + 4];
                } else {
#pragma unroll
                    for (int j = 0; j < 5; ++j) k[j] = (t0 + j < cnt) ? sr[t0 + j] : 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool pair = t0 + j + 1 < cnt && r[j] == r[j + 1];
                    uint32_t u = k[j] & ~RANK_REP, v = k[j + 1] & ~RANK_REP;
                    if (v < u) { const uint32_t t = u; u = v; v = t; }
                    out[j] = pair ? (((uint64_t)u << nb) | v) : sentinel;
                    if (pair && ((k[j] | k[j + 1]) & RANK_REP)) cm |= 1u << j;
                }
                if (t0 + 4 < cnt) {
                    *reinterpret_cast<ulonglong2 *>(key + i0) = make_ulonglong2(out[0], out[1]);
                    *reinterpret_cast<ulonglong2 *>(key + i0 + 2) = make_ulonglong2(out[2], out[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (t0 + j + 1 < cnt) key[i0 + j] = out[j];
                }
                if (hist) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (t0 + j + 1 < cnt)
                            for (unsigned p = 0; p < hpasses; ++p) {
                                const unsigned shp = hbits * p, wd = min(hbits, end_bit - shp);
                                atomicAdd(&sh_hist[(p << hbits) + ((uint32_t)(out[j] >> shp) & ((1u << wd) - 1u))], 1u);
                            }
                }
            }
            cmask |= cm << (4u * it);
        }
        // the bucket's candidates
        const uint32_t c = (uint32_t)__popc(cmask);
        uint32_t incl = c;
        if (__any(cmask != 0)) {
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
        }
        if (lane == 63) s_wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t wg_total = 0, before = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t t = s_wtot[w];
            wg_total += t;
            before += w < (threadIdx.x >> 6) ? t : 0u;
        }
        if (wg_total) {   // (workgroup-uniform)
            if (threadIdx.x == 0) s_cbase = atomicAdd(n_cand, (unsigned long long)wg_total);
            __syncthreads();
            if (cmask) {
                unsigned long long at = s_cbase + before + (incl - c);
#pragma unroll
                for (uint32_t it = 0; it < ITERS; ++it)
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j)
                        if ((cmask >> (4u * it + j)) & 1u) {
                            const uint32_t t = (it * 1024u + threadIdx.x) * 4u + j;
                            uint32_t u = sr[t] & ~RANK_REP, v = sr[t + 1] & ~RANK_REP;
                            if (v < u) { const uint32_t x = u; u = v; v = x; }
                            cand_key[at] = ((uint64_t)u << nb) | v;
                            cand_asm[at] = asm_base + rec_asm[rr[it][j]];
                            ++at;
                        }
            }
        }
    }
    if (hist) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (hpasses << hbits); i += 1024)
            if (sh_hist[i]) atomicAdd(&hist[i], (unsigned long long)sh_hist[i]);
    }
}

// the pairs (last occurrence of bucket b, first of bucket b + 1)
template <class Rec>
__global__ void k_adj_bounds(const uint32_t *__restrict__ edge_rank, uint32_t n_buckets, const Rec rec, const uint32_t *__restrict__ rec_asm,
                             uint32_t asm_base, unsigned nb, uint64_t sentinel, uint64_t *__restrict__ key, uint64_t *__restrict__ cand_key,
                             uint32_t *__restrict__ cand_asm, unsigned long long *__restrict__ n_cand, unsigned long long *__restrict__ hist,
                             unsigned hbits, unsigned hpasses, unsigned end_bit)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b + 1 >= n_buckets) return;
    const uint64_t i = (uint64_t)(b + 1) * UNSORT_RANGE - 1;   // (i + 1 < n: bucket b + 1 exists)
    const uint32_t ka = edge_rank[2 * b + 1], kb = edge_rank[2 * (b + 1)], ra = rec.at(i), rb = rec.at(i + 1);
    uint32_t u = ka & ~RANK_REP, v = kb & ~RANK_REP;
    if (v < u) { const uint32_t t = u; u = v; v = t; }
    const bool pair = ra == rb;
    const uint64_t out = pair ? (((uint64_t)u << nb) | v) : sentinel;
    key[i] = out;
    if (hist)
        for (unsigned p = 0; p < hpasses; ++p) {
            const unsigned shp = hbits * p, wd = min(hbits, end_bit - shp);
            atomicAdd(&hist[(p << hbits) + ((uint32_t)(out >> shp) & ((1u << wd) - 1u))], 1ull);
        }
    if (pair && ((ka | kb) & RANK_REP)) {
        const unsigned long long at = atomicAdd(n_cand, 1ull);
        cand_key[at] = out;
        cand_asm[at] = asm_base + rec_asm[ra];
    }
}

// ---- multi-GPU form: the ranks come back slice-LOCAL (with the repeat mark), together with the owner of every tuple ------
// Global rank = node_base[owner] + local rank, up to 2^hi_bits - 1 -- more than 32 bits for the 5e9 distinct minimizers of
// 100 000 random genomes (the reference indexes nodes with size_t, cpp/include/seqwin/graph.hpp:28-41).  An edge belongs to
// the owner of its rank_lo's range (quantile splitters, dist.rank_bounds), so its key holds rank_lo RELATIVE to that range:
//     key = (rank_lo - lo_base[owner]) << hi_bits | rank_hi          lo_bits + hi_bits <= 64
// and the owner of every key travels next to it (one byte), because it no longer follows from the key's value.
struct RankSpace {
    uint64_t node_base[17];     // prefix of the node counts of the slice owners (n_owners + 1 entries)
    uint64_t lo_base[17];       // first rank of every edge owner's range (n_edge_owners entries), then the total
    uint32_t n_owners, n_edge_owners;
    unsigned hi_bits;
};

template <class Rec>
__global__ __launch_bounds__(256) void k_adj_pairs_dist(const Rec rec, const uint32_t *__restrict__ rank, const uint8_t *__restrict__ own,
                                                        const uint32_t *__restrict__ rec_asm, uint32_t asm_base, uint64_t n,
                                                        const RankSpace S, uint64_t *__restrict__ key, uint8_t *__restrict__ key_own,
                                                        uint64_t *__restrict__ cand_key, uint32_t *__restrict__ cand_asm,
                                                        uint8_t *__restrict__ cand_own, unsigned long long *__restrict__ n_cand)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t r[5];
    uint64_t g[5];            // global ranks
    uint32_t mk = 0;          // bit j: occurrence i0 + j carries the repeat mark
    uint64_t out[4];
    uint8_t oo[4];
    uint32_t cm = 0;
    if (i0 + 1 < n) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const bool live = i0 + j < n;
            r[j] = live ? rec.at(i0 + j) : 0xFFFFFFFFu;
            const uint32_t k = live ? rank[i0 + j] : 0u;
            g[j] = live ? S.node_base[own[i0 + j]] + (k & ~RANK_REP) : 0ull;
            if (k & RANK_REP) mk |= 1u << j;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pair = i0 + j + 1 < n && r[j] == r[j + 1];
            uint64_t u = g[j], v = g[j + 1];
            if (v < u) { const uint64_t t = u; u = v; v = t; }
            uint32_t e = 0;
            for (uint32_t q = 1; q < S.n_edge_owners; ++q) e += (S.lo_base[q] <= u) ? 1u : 0u;
            out[j] = ((u - S.lo_base[e]) << S.hi_bits) | v;
            oo[j] = pair ? (uint8_t)e : OWNER_DROP;
            if (pair && ((mk >> j) & 3u)) cm |= 1u << j;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j + 1 < n) {
                key[i0 + j] = out[j];
                key_own[i0 + j] = oo[j];
            }
    }
    if (__any(cm != 0)) {   // rare: one atomic per wave that has candidates
        const uint32_t c = (uint32_t)__popc(cm);
        uint32_t incl = c;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        unsigned long long base = 0;
        if (lane == 63) base = atomicAdd(n_cand, (unsigned long long)total);
        base = __shfl(base, 63, 64) + (incl - c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((cm >> j) & 1u) {
                cand_key[base] = out[j];
                cand_asm[base] = asm_base + rec_asm[r[j]];
                cand_own[base] = oo[j];
                ++base;
            }
    }
}

// rank -> hash through the job-wide table as the all-gather leaves it: owner o's hashes at table[o * pad ...]
struct RankHash {
    const uint64_t *table;
    uint64_t node_base[17];
    uint64_t pad;
    uint32_t n_owners;
    __device__ uint64_t operator()(uint64_t rank) const
    {
        if (!table) return rank;                 // no table: the edges keep global ranks (edge_hash_requests / _attach)
        uint32_t o = 0;
        for (uint32_t q = 1; q < n_owners; ++q) o += (node_base[q] <= rank) ? 1u : 0u;
        return table[(uint64_t)o * pad + (rank - node_base[o])];
    }
};

// one thread per run of equal wide keys (see k_edges_runs)
template <bool RUNS = false>
__global__ void k_edges_runs_wide(const uint64_t *__restrict__ ukeys, const uint32_t *__restrict__ usum, uint64_t n_edges,
                                  unsigned hi_bits, uint64_t lo_base, const RankHash H, sw_edge *__restrict__ edges)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint64_t key = ukeys[e];
    const uint64_t u = (hi_bits >= 64 ? 0ull : (key >> hi_bits)) + lo_base;
    const uint64_t v = hi_bits >= 64 ? key : (key & ((1ull << hi_bits) - 1ull));
    edges[e].first = H(u);
    edges[e].second = H(v);
    edges[e].weight = RUNS ? usum[e + 1] - usum[e] : usum[e];
}

struct RepeatFlag {   // 1 where a sorted candidate row repeats the (pair, assembly) of its predecessor
    const uint64_t *keys;
    const uint32_t *vals;
    __host__ __device__ uint32_t operator()(uint64_t s) const
    {
        return (s && keys[s] == keys[s - 1] && vals[s] == vals[s - 1]) ? 1u : 0u;
    }
};

// one thread per candidate pair with repeats: find its edge (the run-length keys are sorted) and take the repeats off
__global__ void k_subtract_repeats(const uint64_t *__restrict__ rkeys, const uint32_t *__restrict__ rdups, uint32_t n_runs,
                                   const uint64_t *__restrict__ ukeys, uint64_t n_edges, sw_edge *__restrict__ edges)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs || rdups[i] == 0) return;
    const uint64_t key = rkeys[i];
    uint64_t lo = 0, hi = n_edges;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < n_edges && ukeys[lo] == key) edges[lo].weight -= rdups[i];   // (every candidate is also a record of the main sort)
}

// 1 at the first row of every (pair, assembly) combination of the sorted adjacency rows (first row: 1)
struct AsmChangeAny {
    const uint64_t *keys;
    const uint32_t *vals;
    __host__ __device__ uint32_t operator()(uint64_t s) const
    {
        return (s == 0 || keys[s] != keys[s - 1] || vals[s] != vals[s - 1]) ? 1u : 0u;
    }
};

// order guard of the keys-only sorts where no kernel of ours streams over the sorted keys anyway (rocPRIM run lengths)
__global__ void k_check_ascending(const uint64_t *__restrict__ keys, uint64_t m, uint32_t *__restrict__ order_bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i && i < m && keys[i] < keys[i - 1]) atomicAdd(order_bad, 1u);
}

__global__ void k_drop_sentinel_run(const uint64_t *__restrict__ ukeys, uint64_t sentinel, uint32_t *__restrict__ count)
{
    const uint32_t n = *count;
    if (n && ukeys[n - 1] == sentinel) *count = n - 1;   // the run of record boundaries (all keys == sentinel) is no edge
}

// ---- run lengths of the sorted pair keys (r04; rocprim::run_length_encode before) ------------------------------------------------
// One streaming pass: a thread holds eight consecutive keys (one 64-byte load), a key that differs from its predecessor heads a
// run; the number of heads before a tile comes from a chained look-back over per-tile totals (tiles numbered by a ticket, so a
// predecessor is always running -- as in k_nodes); every head writes its key and its POSITION: the length of run e is
// start[e + 1] - start[e] (start[n_runs] = m is written by the last tile), taken by the kernel that turns runs into edges.
constexpr int RLE_THREADS = 1024, RLE_ITEMS = 8;
constexpr uint32_t RLE_TILE = RLE_THREADS * RLE_ITEMS;
__global__ __launch_bounds__(RLE_THREADS) void k_rle_keys(const uint64_t *__restrict__ keys, uint64_t m, uint64_t *__restrict__ ukeys,
                                                           uint32_t *__restrict__ ustart, unsigned long long *__restrict__ tile_state,
                                                           uint32_t *__restrict__ ticket, uint32_t *__restrict__ n_runs_out,
                                                           uint32_t *__restrict__ order_bad)
{
    __shared__ uint32_t s_tile, s_excl, s_wave[RLE_THREADS / 64];
    __shared__ uint64_t s_last[RLE_THREADS / 64];        // last key of every wave (the predecessor of the next wave's first)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    // wave w takes keys [t0 + w * 512, + 512), item j of lane l the key w * 512 + j * 64 + l: every load is 512 contiguous bytes,
    // and a key's predecessor sits in the lane below (lane 0: lane 63 of the item before)
    const uint64_t t0 = (uint64_t)tile * RLE_TILE, w0 = t0 + (uint64_t)wave * (64 * RLE_ITEMS);
    uint64_t k[RLE_ITEMS];
#pragma unroll
    for (int j = 0; j < RLE_ITEMS; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        k[j] = i < m ? keys[i] : 0ull;
    }
    if (lane == 63) s_last[wave] = k[RLE_ITEMS - 1];
    __syncthreads();
    uint64_t carry = wave ? s_last[wave - 1] : (t0 ? keys[t0 - 1] : 0ull);   // the key before the wave's first (unused at position 0)
    uint32_t cnt[RLE_ITEMS], below[RLE_ITEMS], total_w = 0, headm = 0;
    unsigned long long misorder = 0;   // (wave-uniform: lanes that saw a key below its predecessor)
#pragma unroll
    for (int j = 0; j < RLE_ITEMS; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        uint64_t prev = __shfl_up(k[j], 1, 64);
        if (lane == 0) prev = carry;
        const bool head = i < m && (i == 0 || k[j] != prev);
#ifndef SW_NO_ORDER_GUARD
        // always-on order guard (r05): the keys must arrive ascending.  One compare into a scalar mask + one scalar OR per key (the
        // first form kept a flag per lane: k_rle_keys 2.31 -> 2.65 ms at 745 M keys)
        misorder |= __ballot(i < m && i && k[j] < prev);
#endif
        const unsigned long long b = __ballot(head);
        below[j] = total_w + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));   // heads of the wave before this key
        cnt[j] = (uint32_t)__popcll(b);
        total_w += cnt[j];
        if (head) headm |= 1u << j;
        carry = __shfl(k[j], 63, 64);
    }
    if (lane == 0) s_wave[wave] = total_w;
    if (misorder && lane == 0) atomicAdd(order_bad, (uint32_t)__popcll(misorder));
    __syncthreads();
    if (wave == 0) {
        const uint32_t wc = lane < RLE_THREADS / 64 ? s_wave[lane] : 0u;
        uint32_t winc = wc;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(winc, d, 64);
            if (lane >= d) winc += up;
        }
        const uint32_t total = __shfl(winc, RLE_THREADS / 64 - 1, 64);
        if (lane < RLE_THREADS / 64) s_wave[lane] = winc - wc;            // exclusive offsets of the waves
        if (lane == 0)
            __hip_atomic_store(&tile_state[tile], (tile == 0 ? TS_INC : TS_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile) {
            int64_t look = (int64_t)tile - 1;
            for (;;) {
                const int64_t idx = look - lane;   // lane 0 reads the nearest predecessor
                const unsigned long long st = idx >= 0 ? __hip_atomic_load(&tile_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : TS_INC;   // before the first tile: nothing
                const unsigned long long inc = __ballot((st >> 62) == 2), none = __ballot((st >> 62) == 0);
                const uint32_t first_inc = inc ? (uint32_t)__builtin_ctzll(inc) : 64u;
                const unsigned long long needed = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
                if (none & needed) {   // a predecessor in reach has not published yet (it is running: tickets)
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t v = (lane <= first_inc) ? (uint32_t)st : 0u;
                for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
                excl += v;
                if (first_inc < 64) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&tile_state[tile], TS_INC | (unsigned long long)(excl + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            if (t0 + RLE_TILE >= m) {            // the last tile: the number of runs, and the end of the last run
                *n_runs_out = excl + total;
                ustart[excl + total] = (uint32_t)m;
            }
        }
    }
    __syncthreads();
    const uint32_t base = s_excl + s_wave[wave];
#pragma unroll
    for (int j = 0; j < RLE_ITEMS; ++j)
        if ((headm >> j) & 1u) {
            ukeys[base + below[j]] = k[j];
            ustart[base + below[j]] = (uint32_t)(w0 + (uint64_t)j * 64 + lane);
        }
    (void)cnt;
}

// one thread per run of equal pairs: pair = (ukeys >> pshift) & pmask, weight = distinct assemblies in the run
// (RUNS: usum holds the START of every run and of the end, k_rle_keys: weight = the run's length)
template <bool RUNS = false>
__global__ void k_edges_runs(const uint64_t *__restrict__ ukeys, const uint32_t *__restrict__ usum, unsigned pshift, uint64_t pmask,
                             uint64_t n_edges, unsigned nb, const sw_node *__restrict__ nodes,
                             const uint64_t *__restrict__ rank_hash, sw_edge *__restrict__ edges)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint64_t pair = (ukeys[e] >> pshift) & pmask;
    const uint32_t u = (uint32_t)(pair >> nb), v = (uint32_t)(pair & ((1ull << nb) - 1ull));
    // (r05: the two hashes by non-temporal loads -- random 8-byte reads that are used once -- measured the same, NOTES.md)
    edges[e].first = rank_hash ? rank_hash[u] : nodes[u].hash;
    edges[e].second = rank_hash ? rank_hash[v] : nodes[v].hash;
    edges[e].weight = RUNS ? usum[e + 1] - usum[e] : usum[e];
}

// ---- tuple-exchange form of the multi-GPU build (dist.py) ---------------------------------------------


__global__ void k_unpermute(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ by_row, uint64_t n,
                            uint32_t *__restrict__ orig)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) orig[perm[j]] = by_row[j];
}

// (key, assembly) of consecutive minimizers of a record, from GLOBAL node ranks; sentinel otherwise
__global__ void k_adj_rows(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ rank,
                           const uint32_t *__restrict__ rec_asm, uint64_t n, unsigned nb, uint64_t sentinel,
                           uint64_t asm_base, uint64_t *__restrict__ rows)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const uint32_t r0 = (uint32_t)(kmer[i] >> 32), r1 = (uint32_t)(kmer[i + 1] >> 32);
    if (r0 == r1) {
        uint32_t u = rank[i], v = rank[i + 1];
        if (v < u) { const uint32_t t = u; u = v; v = t; }
        rows[2 * i] = ((uint64_t)u << nb) | v;
        rows[2 * i + 1] = asm_base + rec_asm[r0];
    } else {
        rows[2 * i] = sentinel;
        rows[2 * i + 1] = 0;
    }
}


__global__ void k_adj_rows_packed(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ rank,
                                  const uint32_t *__restrict__ rec_asm, uint64_t n, unsigned nb, unsigned ab, uint64_t sentinel,
                                  uint64_t asm_base, uint64_t *__restrict__ rows)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const uint32_t r0 = (uint32_t)(kmer[i] >> 32), r1 = (uint32_t)(kmer[i + 1] >> 32);
    if (r0 == r1) {
        uint32_t u = rank[i], v = rank[i + 1];
        if (v < u) { const uint32_t t = u; u = v; v = t; }
        rows[i] = (((((uint64_t)u << nb) | v)) << ab) | (asm_base + rec_asm[r0]);
    } else {
        rows[i] = sentinel;
    }
}

__global__ void k_split_rows2(const uint64_t *__restrict__ rows, uint64_t n, uint64_t *__restrict__ key,
                              uint32_t *__restrict__ val)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        key[i] = rows[2 * i];
        val[i] = (uint32_t)rows[2 * i + 1];
    }
}

__global__ void k_node_hashes(const sw_node *__restrict__ nodes, uint64_t n_nodes, uint64_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) out[i] = nodes[i].hash;
}

// ---- next rows: threshold sums, edge / node filter on device --------------------------------
// sums[0] = sum n_tar, sums[1] = sum n_tar^2, sums[2] = sum n_tar * n_neg  (kmers.py:426-429, exact integers)
__global__ void k_threshold_sums(const sw_node *__restrict__ nodes, uint64_t n, unsigned long long *__restrict__ sums)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a = 0, b = 0, c = 0;
    if (i < n) {
        const uint64_t t = nodes[i].n_tar, g = nodes[i].n_neg;
        a = t;
        b = t * t;
        c = t * g;
    }
    for (int d = 32; d; d >>= 1) {
        a += __shfl_down(a, d, 64);
        b += __shfl_down(b, d, 64);
        c += __shfl_down(c, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&sums[0], (unsigned long long)a);
        if (b) atomicAdd(&sums[1], (unsigned long long)b);
        if (c) atomicAdd(&sums[2], (unsigned long long)c);
    }
}

struct EdgeKeepFlag {   // edges['weight'] > th  (kmers.py:152-153)
    const sw_edge *edges;
    uint64_t th;
    __host__ __device__ uint32_t operator()(uint64_t e) const { return edges[e].weight > th ? 1u : 0u; }
};

__device__ __forceinline__ uint64_t node_index_of(const sw_node *nodes, uint64_t n_nodes, uint64_t h)
{
    uint64_t lo = 0, hi = n_nodes;   // np.searchsorted(nodes['hash'], h)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (nodes[mid].hash < h) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_filter_edges(const sw_edge *__restrict__ edges, uint64_t n_edges, uint64_t th,
                               const uint32_t *__restrict__ ecum, const sw_node *__restrict__ nodes, uint64_t n_nodes,
                               sw_edge *__restrict__ out, uint32_t *__restrict__ node_keep)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges || !(edges[e].weight > th)) return;
    out[ecum[e] - 1] = edges[e];
    node_keep[node_index_of(nodes, n_nodes, edges[e].first)] = 1u;     // endpoints survive (kmers.py:157-161)
    node_keep[node_index_of(nodes, n_nodes, edges[e].second)] = 1u;
}

__global__ void k_compact_nodes(const sw_node *__restrict__ nodes, uint64_t n_nodes, const uint32_t *__restrict__ keep,
                                const uint32_t *__restrict__ kcum, sw_node *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes && keep[i]) out[kcum[i] - 1] = nodes[i];
}

// ---- filter_kmers -----------------------------------------------------------------------------------
__global__ void k_filter_keep(const sw_node *__restrict__ nodes, uint64_t n_nodes, const uint64_t *__restrict__ used,
                              uint64_t n_used, uint32_t *__restrict__ keep, uint64_t *__restrict__ size)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint64_t h = nodes[i].hash;
    uint64_t lo = 0, hi = n_used;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (used[mid] < h) lo = mid + 1; else hi = mid;
    }
    const bool k = lo < n_used && used[lo] == h;
    keep[i] = k ? 1u : 0u;
    size[i] = k ? nodes[i].stop - nodes[i].start : 0ull;
}

__global__ void k_filter_nodes(const sw_node *__restrict__ nodes, uint64_t n_nodes, const uint32_t *__restrict__ keep,
                               const uint32_t *__restrict__ kcum, const uint64_t *__restrict__ scum,
                               sw_node *__restrict__ nodes_out, uint64_t *__restrict__ src_start,
                               uint64_t *__restrict__ dst_start)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || !keep[i]) return;
    const uint32_t o = kcum[i] - 1;
    const uint64_t sz = nodes[i].stop - nodes[i].start;
    const uint64_t ns = scum[i] - sz;
    sw_node nd = nodes[i];
    src_start[o] = nd.start;
    dst_start[o] = ns;
    nd.start = ns;
    nd.stop = ns + sz;
    nodes_out[o] = nd;
}

__global__ void k_filter_kmers(const sw_kmer *__restrict__ kmers, const uint64_t *__restrict__ src_start,
                               const uint64_t *__restrict__ dst_start, uint64_t n_kept, uint64_t n_out,
                               sw_kmer *__restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    uint64_t lo = 0, hi = n_kept;  // last kept node with dst_start <= j
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (dst_start[mid] <= j) lo = mid; else hi = mid;
    }
    out[j] = kmers[src_start[lo] + (j - dst_start[lo])];
}

// ---- checksums (the per-element terms: device.hpp) ------------------------------------------------------
// element i of a slice is element base + i of the whole (concatenated) array: the sums of the slices of a sharded
// index add up (mod 2^64) to the checksums of the unsharded one
__global__ void k_checksum(const sw_kmer *kmers, uint64_t nk, const sw_node *nodes, uint64_t nn, const sw_edge *edges,
                           uint64_t ne, uint64_t kbase, uint64_t nbase, uint64_t ebase, unsigned long long *sums)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a = 0, b = 0, c = 0;
    if (i < nk) a = ck_kmer(kbase + i, kmers[i]);
    if (i < nn) b = ck_node(nbase + i, nodes[i]);
    if (i < ne) c = ck_edge(ebase + i, edges[i]);
    for (int d = 32; d; d >>= 1) {
        a += __shfl_down(a, d, 64);
        b += __shfl_down(b, d, 64);
        c += __shfl_down(c, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&sums[0], (unsigned long long)a);
        if (b) atomicAdd(&sums[1], (unsigned long long)b);
        if (c) atomicAdd(&sums[2], (unsigned long long)c);
    }
}

// ---- self-check of a resident index (size-independent properties, for sets too large to compare on the host) ----
// out[0] nodes not strictly ascending by hash     out[1] node ranges not a partition of [0, n_kmers) (empty nodes count too)
// out[2] (record_idx, pos) not strictly ascending inside a node   out[3] edges not strictly ascending by (first, second)
// out[4] edges with first > second                out[5] edge weight outside [1, n_assemblies]
// out[6] edge endpoints that are no node hash     out[7] nodes with n_tar + n_neg outside [1, min(size, n_assemblies)] (scored only)
// out[8] sum of edge weights                      out[9] number of records that hold at least one occurrence
__global__ void k_verify_nodes(const sw_node *__restrict__ nodes, uint64_t n_nodes, uint64_t n_kmers, uint64_t base,
                               uint64_t n_assemblies, int scored, uint32_t *__restrict__ start_bits,
                               unsigned long long *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const sw_node nd = nodes[i];
    if (i && !(nodes[i - 1].hash < nd.hash)) atomicAdd(&out[0], 1ull);
    const uint64_t prev_stop = i ? nodes[i - 1].stop : base;
    bool bad = nd.start != prev_stop || nd.stop <= nd.start || nd.stop - base > n_kmers;
    if (i == n_nodes - 1 && nd.stop - base != n_kmers) bad = true;
    if (bad) atomicAdd(&out[1], 1ull);
    else atomicOr(&start_bits[(nd.start - base) >> 5], 1u << ((nd.start - base) & 31u));
    if (scored) {
        const uint64_t c = (uint64_t)nd.n_tar + nd.n_neg, size = nd.stop - nd.start;
        if (c < 1 || c > size || c > n_assemblies) atomicAdd(&out[7], 1ull);
    }
}

__global__ void k_verify_kmers(const sw_kmer *__restrict__ kmers, uint64_t n, const uint32_t *__restrict__ start_bits,
                               unsigned long long *__restrict__ out)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0, first_of_record = 0;
    if (s < n) {
        const uint64_t cur = ((uint64_t)kmers[s].record_idx << 32) | kmers[s].pos;
        const bool head = (start_bits[s >> 5] >> (s & 31u)) & 1u;
        if (s && !head) {
            const uint64_t prev = ((uint64_t)kmers[s - 1].record_idx << 32) | kmers[s - 1].pos;
            if (!(prev < cur)) bad = 1;
        }
    }
    (void)first_of_record;
    for (int d = 32; d; d >>= 1) bad += __shfl_down(bad, d, 64);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(&out[2], bad);
}

__global__ void k_verify_edges(const sw_edge *__restrict__ edges, uint64_t n_edges, const sw_node *__restrict__ nodes,
                               uint64_t n_nodes, const uint64_t *__restrict__ rank_hash, uint64_t n_rank_hash,
                               uint64_t n_assemblies, unsigned long long *__restrict__ out)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long wsum = 0;
    if (e < n_edges) {
        const sw_edge ed = edges[e];
        if (e) {
            const sw_edge pe = edges[e - 1];
            if (!(pe.first < ed.first || (pe.first == ed.first && pe.second < ed.second))) atomicAdd(&out[3], 1ull);
        }
        if (ed.first > ed.second) atomicAdd(&out[4], 1ull);
        if (ed.weight < 1 || ed.weight > n_assemblies) atomicAdd(&out[5], 1ull);
        wsum = ed.weight;
        for (int side = 0; side < 2; ++side) {
            const uint64_t h = side ? ed.second : ed.first;
            bool found;
            if (rank_hash) {   // slice of a sharded index: endpoints may be nodes of another slice
                uint64_t lo = 0, hi = n_rank_hash;
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (rank_hash[mid] < h) lo = mid + 1; else hi = mid;
                }
                found = lo < n_rank_hash && rank_hash[lo] == h;
            } else {
                const uint64_t lo = node_index_of(nodes, n_nodes, h);
                found = lo < n_nodes && nodes[lo].hash == h;
            }
            if (!found) atomicAdd(&out[6], 1ull);
        }
    }
    for (int d = 32; d; d >>= 1) wsum += __shfl_down(wsum, d, 64);
    if ((threadIdx.x & 63u) == 0 && wsum) atomicAdd(&out[8], wsum);
}

// ---- stable sort of the occurrences by their 64-bit hash, in two phases ----------------------------------
// Phase 1: 4 radix passes over the TOP 32 bits only (key32), carrying OccPay (low half, pos, record, original index):
// 20 B per element and pass for 4 passes instead of 24 B for 8.  The result stays in this split form; consumers
// rebuild hash = key32 << 32 | pay.low on the fly.
// Phase 2: a run of equal top halves whose low halves are out of order (two different hashes sharing 32 bits:
// ~n_nodes^2 / 2^33 runs) is repaired in place; what the in-place pass does not take is pulled into a side array,
// sorted by the full hash (stable) and put back into the same positions -- positions ascend with the top half, so
// the outcome is exactly the stable 64-bit sort.
// SEQWIN_AMD_SORT_KEYBITS=b (tests) makes phase 1 look at the top b bits only.
__global__ void k_rot_keys(uint32_t *__restrict__ key32, uint64_t n, unsigned left)   // rotate left by `left` (1..31)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t k = key32[i];
        key32[i] = (k << left) | (k >> (32u - left));
    }
}

// descents: q > 0 in the same phase-1 run as q-1 with a smaller full hash.  Two passes without global atomics (a
// single counter serialises: ~15 ns per descent, 9 ms for the 6e5 descents of 1e8 unique hashes): blocks of 1024
// consecutive positions count theirs, an exclusive scan places them, and only blocks that have any list them --
// in ascending position, so the key list comes out sorted.
constexpr uint32_t DESC_BLOCK = 1024;   // positions per workgroup (256 threads x 4 consecutive positions)

// The two-phase sort works on a "view" of n elements ordered by (key, low): key(q) is the phase-1 key (View::Key: 32 bits for
// the occurrences, 64 for the edge keys), low(q) what orders elements of equal key, load / store move a whole element.
struct PayView {   // key32[q] = top half of the hash, pay[q].low = its low half
    using Key = uint32_t;
    using Elem = OccPay;
    uint32_t *key32;
    OccPay *pay;
    __device__ uint32_t key(uint64_t q) const { return key32[q]; }
    __device__ uint64_t low(uint64_t q) const { return pay[q].low; }
    __device__ static uint64_t low_of(const OccPay &v) { return v.low; }
    __device__ void load(uint64_t q, uint32_t &k, OccPay &v) const { k = key32[q]; v = pay[q]; }
    __device__ void store(uint64_t q, uint32_t k, const OccPay &v) const { key32[q] = k; pay[q] = v; }
};
struct LowView {   // the descent sweeps' view of the occurrences when the sort's last pass left the low halves in an array of
                   // their own (4 B per element instead of the 16-byte payloads' lines): key32[q], low32[q]
    using Key = uint32_t;
    const uint32_t *key32;
    const uint32_t *low32;
    __device__ uint32_t key(uint64_t q) const { return key32[q]; }
    __device__ uint64_t low(uint64_t q) const { return low32[q]; }
};
struct EdgeKeyView {   // 64-bit adjacency keys sorted on bits [low_bits, 64) only: phase-1 key = those bits, the whole key orders the rest
    using Key = uint64_t;
    using Elem = uint64_t;
    uint64_t *keys;
    unsigned low_bits;
    __device__ uint64_t key(uint64_t q) const { return keys[q] >> low_bits; }
    __device__ uint64_t low(uint64_t q) const { return keys[q]; }
    __device__ static uint64_t low_of(const uint64_t &v) { return v; }
    __device__ void load(uint64_t q, uint64_t &k, uint64_t &v) const { v = keys[q]; k = v >> low_bits; }
    __device__ void store(uint64_t q, uint64_t, const uint64_t &v) const { keys[q] = v; }
};

template <class View>
__device__ __forceinline__ uint32_t descents_of_thread(const View &V, typename View::Key kmask, uint64_t n, uint64_t q0, uint32_t *heads = nullptr)
{
    uint32_t m = 0, hm = 0;   // bit i: position q0 + i is a descent / differs from its predecessor (a run head of this order)
    if (heads) *heads = 0;
    if (q0 >= n) return 0;
    typename View::Key kp = q0 ? V.key(q0 - 1) : 0;
    uint64_t lp = q0 ? V.low(q0 - 1) : 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint64_t q = q0 + i;
        if (q >= n) break;
        const typename View::Key kq = V.key(q);
        const uint64_t lq = V.low(q);
        if (q && (kq & kmask) == (kp & kmask) && (kq < kp || (kq == kp && lq < lp))) m |= 1u << i;
        if (q == 0 || kq != kp || lq != lp) hm |= 1u << i;
        kp = kq;
        lp = lq;
    }
    if (heads) *heads = (uint32_t)__popc(hm);
    return m;
}

// the same for the occurrence arrays, with 16-B loads (four consecutive keys; the payloads whole: their lines are read anyway)
__device__ __forceinline__ uint32_t descents_of_thread(const PayView &V, uint32_t kmask, uint64_t n, uint64_t q0, uint32_t *heads = nullptr)
{
    uint32_t m = 0, hm = 0;
    if (heads) *heads = 0;
    if (q0 >= n) return 0;
    uint32_t kq[4];
    uint64_t lq[4];
    if (q0 + 4 <= n) {
        const uint4 kv = *reinterpret_cast<const uint4 *>(V.key32 + q0);
        kq[0] = kv.x; kq[1] = kv.y; kq[2] = kv.z; kq[3] = kv.w;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) lq[i] = reinterpret_cast<const uint4 *>(V.pay + q0 + i)->x;   // OccPay::low
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            kq[i] = (q0 + i < n) ? V.key32[q0 + i] : 0u;
            lq[i] = (q0 + i < n) ? V.pay[q0 + i].low : 0u;
        }
    }
    uint32_t kp = q0 ? V.key32[q0 - 1] : 0;
    uint64_t lp = q0 ? V.pay[q0 - 1].low : 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint64_t q = q0 + i;
        if (q >= n) break;
        if (q && (kq[i] & kmask) == (kp & kmask) && (kq[i] < kp || (kq[i] == kp && lq[i] < lp))) m |= 1u << i;
        if (q == 0 || kq[i] != kp || lq[i] != lp) hm |= 1u << i;
        kp = kq[i];
        lp = lq[i];
    }
    if (heads) *heads = (uint32_t)__popc(hm);
    return m;
}

__device__ __forceinline__ uint32_t descents_of_thread(const LowView &V, uint32_t kmask, uint64_t n, uint64_t q0, uint32_t *heads = nullptr)
{
    uint32_t m = 0, hm = 0;
    if (heads) *heads = 0;
    if (q0 >= n) return 0;
    uint32_t kq[4], lq[4];
    if (q0 + 4 <= n) {
        const uint4 kv = *reinterpret_cast<const uint4 *>(V.key32 + q0), lv = *reinterpret_cast<const uint4 *>(V.low32 + q0);
        kq[0] = kv.x; kq[1] = kv.y; kq[2] = kv.z; kq[3] = kv.w;
        lq[0] = lv.x; lq[1] = lv.y; lq[2] = lv.z; lq[3] = lv.w;
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            kq[i] = (q0 + i < n) ? V.key32[q0 + i] : 0u;
            lq[i] = (q0 + i < n) ? V.low32[q0 + i] : 0u;
        }
    }
    uint32_t kp = q0 ? V.key32[q0 - 1] : 0, lp = q0 ? V.low32[q0 - 1] : 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint64_t q = q0 + i;
        if (q >= n) break;
        if (q && (kq[i] & kmask) == (kp & kmask) && (kq[i] < kp || (kq[i] == kp && lq[i] < lp))) m |= 1u << i;
        if (q == 0 || kq[i] != kp || lq[i] != lp) hm |= 1u << i;
        kp = kq[i];
        lp = lq[i];
    }
    if (heads) *heads = (uint32_t)__popc(hm);
    return m;
}

// cnt[block] = run heads << 32 | descents.  The heads of the order BEFORE the repair bound the number of distinct hashes
// from above (a repair only merges runs of equal hashes that a descent had split), closely: the nodes array is sized by it.
// (A block with 1..DESC_SLOT descents -- the rule: 7e5 descents over 7e5 blocks -- also leaves them, in position order, in
// its slot, so that the listing pass need not read the block's 20 KB again.)
constexpr uint32_t DESC_SLOT = 4;
template <class View>
__global__ __launch_bounds__(256) void k_count_descents(const View V, typename View::Key kmask, uint64_t n, unsigned long long *__restrict__ cnt,
                                                        uint32_t *__restrict__ slot_q, uint32_t *__restrict__ slot_k)
{
    const uint64_t q0 = (uint64_t)blockIdx.x * DESC_BLOCK + threadIdx.x * 4u;
    uint32_t heads = 0;
    const uint32_t m = descents_of_thread(V, kmask, n, q0, &heads);
    const uint32_t c = (uint32_t)__popc(m);
    __shared__ unsigned long long s;
    __shared__ uint32_t s_n, s_q[DESC_SLOT], s_kk[DESC_SLOT];
    if (threadIdx.x == 0) {
        s = 0;
        s_n = 0;
    }
    __syncthreads();
    unsigned long long w = ((unsigned long long)heads << 32) | c;
    for (int d = 32; d; d >>= 1) w += __shfl_down(w, d, 64);
    if ((threadIdx.x & 63u) == 0 && w) atomicAdd(&s, w);
    __syncthreads();
    const uint32_t dc = (uint32_t)s;          // (workgroup-uniform)
    if (threadIdx.x == 0) cnt[blockIdx.x] = s;
    if (dc == 0 || dc > DESC_SLOT) return;
    uint32_t mm = m;
    while (mm) {
        const uint32_t i = (uint32_t)__builtin_ctz(mm);
        mm &= mm - 1;
        const uint32_t at = atomicAdd(&s_n, 1u);
        s_q[at] = (uint32_t)(q0 + i);
        s_kk[at] = (uint32_t)(V.key(q0 + i) & kmask);   // (only read by a general repair: 32-bit keys)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t i = 1; i < dc; ++i)     // ascending positions (at most DESC_SLOT entries)
            for (uint32_t j = i; j > 0 && s_q[j] < s_q[j - 1]; --j) {
                const uint32_t tq = s_q[j], tk = s_kk[j];
                s_q[j] = s_q[j - 1];
                s_kk[j] = s_kk[j - 1];
                s_q[j - 1] = tq;
                s_kk[j - 1] = tk;
            }
        for (uint32_t i = 0; i < dc; ++i) {
            slot_q[(size_t)blockIdx.x * DESC_SLOT + i] = s_q[i];
            slot_k[(size_t)blockIdx.x * DESC_SLOT + i] = s_kk[i];
        }
    }
}

template <class View>
__global__ __launch_bounds__(256) void k_list_descents(const View V, typename View::Key kmask, uint64_t n,
                                                       const unsigned long long *__restrict__ cnt,
                                                       const unsigned long long *__restrict__ off, uint32_t n_blocks,
                                                       uint32_t *__restrict__ bad, uint32_t cap, uint32_t *__restrict__ bad_q,
                                                       uint32_t cap_q, unsigned long long *__restrict__ n_desc,
                                                       const uint32_t *__restrict__ slot_q, const uint32_t *__restrict__ slot_k)
{
    const uint32_t b = blockIdx.x;
    const uint32_t c_blk = (uint32_t)cnt[b];
    if (b == n_blocks - 1 && threadIdx.x == 0) {
        const unsigned long long tot = off[b] + cnt[b];
        n_desc[0] = tot & 0xFFFFFFFFull;   // descents
        n_desc[1] = tot >> 32;             // run heads before the repair
    }
    if (c_blk == 0) return;
    if (c_blk <= DESC_SLOT) {              // the counting pass left them in the block's slot
        if (threadIdx.x < c_blk) {
            const uint32_t at = (uint32_t)off[b] + threadIdx.x;
            if (bad && at < cap) bad[at] = slot_k[(size_t)b * DESC_SLOT + threadIdx.x];
            if (at < cap_q) bad_q[at] = slot_q[(size_t)b * DESC_SLOT + threadIdx.x];
        }
        return;
    }
    const uint64_t q0 = (uint64_t)b * DESC_BLOCK + threadIdx.x * 4u;
    const uint32_t m = descents_of_thread(V, kmask, n, q0);
    const uint32_t c = (uint32_t)__popc(m);
    // exclusive prefix of c over the 256 threads (position order)
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = c;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = (uint32_t)off[b] + (incl - c);
    for (uint32_t i = 0; i < wave; ++i) base += wsum[i];
    uint32_t mm = m;
    while (mm) {
        const uint32_t i = (uint32_t)__builtin_ctz(mm);
        mm &= mm - 1;
        if (base < cap) bad[base] = (uint32_t)(V.key(q0 + i) & kmask);
        if (base < cap_q) bad_q[base] = (uint32_t)(q0 + i);
        ++base;
    }
}

// ---- in-place repair (the common case: short runs) -- no host round trip -----------------------------------
// One WAVE per descent (k_repair_wave).  The descents are listed in position order, so the first descent of a run --
// its owner -- is the one whose predecessor in the list lies in another run; the test reads only the masked keys, which
// no repair changes, so owners of other runs may be writing meanwhile.  The owner finds its run [a, e), 64 positions
// per step, and rank-sorts it by (key, low, position) -- a total order, so the result is the stable sort: a run of up
// to 64 elements in registers (one element per lane, the others read with v_readlane), a longer one is listed for
// k_repair_sort (a workgroup per run, in LDS).
// Anything the fast path does not take (more than REPAIR_MAX_DESC descents, a run longer than REPAIR_MAX_RUN)
// raises `status` and is left to the general repair of the caller.
// r05: one THREAD per descent first (k_repair_thread) -- a run of up to REPAIR_THREAD_RUN elements (two different hashes that
// share their top half: the rule) is delimited, ranked and rewritten by one lane; only the owners of longer runs are listed
// (`mid`) for the wave form.  With that the in-place route takes 2^26 descents: one GPU's share of 100 000 iid genomes
// (5.4e8 nodes -> n^2 / 2^33 = 3.4e7 shared top halves, 1.7e7 descents) went through the general repair before -- k_bad_runs +
// gather + an 8-pass rocPRIM pair sort + scatter, 12 ms per build (profiles/r04d_random100k_k19_kernel_stats.txt).
constexpr uint32_t REPAIR_MAX_DESC = 1u << 26;   // 15k genomes: 7.3e5 descents (79 M nodes); 5.4e8 nodes: 1.7e7
constexpr uint32_t REPAIR_MAX_MID = 1u << 22;    // owners of runs longer than REPAIR_THREAD_RUN (15k genomes: a node has up to 500 occurrences)
constexpr uint32_t REPAIR_THREAD_RUN = 6;
constexpr uint32_t REPAIR_MAX_RUN = 2048;
constexpr uint32_t REPAIR_GRID = 2048;

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t lane)   // `lane` is wave-uniform
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)lane);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t readlane_key(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint64_t readlane_key(uint64_t v, uint32_t lane) { return readlane64(v, lane); }

template <class View>
__global__ __launch_bounds__(256) void k_repair_thread(const View V, typename View::Key kmask, uint64_t n, const uint32_t *__restrict__ bad_q,
                                                       const unsigned long long *__restrict__ n_desc, uint32_t *__restrict__ mid,
                                                       uint32_t *__restrict__ n_mid, uint32_t *__restrict__ status)
{
    constexpr uint32_t RT = REPAIR_THREAD_RUN;
    const unsigned long long D = *n_desc;
    if (D > REPAIR_MAX_DESC) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *status = 1u;
        return;
    }
    const uint64_t n_threads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < D; b += n_threads) {
        const uint32_t q = bad_q[b];
        const typename View::Key k = V.key(q) & kmask;
        if (b && (V.key(bad_q[b - 1]) & kmask) == k) continue;   // an earlier descent of the same run owns it
        uint64_t a = q, e = (uint64_t)q + 1;                     // [a, e) belongs to the run
        while (a > 0 && e - a <= RT && (V.key(a - 1) & kmask) == k) --a;
        while (e < n && e - a <= RT && (V.key(e) & kmask) == k) ++e;
        const uint32_t len = (uint32_t)(e - a);
        if (len > RT) {                                          // the wave form's (k_repair_wave walks this list)
            const uint32_t s = atomicAdd(n_mid, 1u);
            if (s < REPAIR_MAX_MID) mid[s] = (uint32_t)b;
            else *status = 1u;
            continue;
        }
        typename View::Key ki[RT];
        typename View::Elem vi[RT];
#pragma unroll
        for (uint32_t i = 0; i < RT; ++i)
            if (i < len) V.load(a + i, ki[i], vi[i]);
        uint32_t rank[RT];
#pragma unroll
        for (uint32_t i = 0; i < RT; ++i) {
            rank[i] = 0;
#pragma unroll
            for (uint32_t j = 0; j < RT; ++j)
                if (i < len && j < len && j != i) {
                    const uint64_t li = View::low_of(vi[i]), lj = View::low_of(vi[j]);
                    rank[i] += (ki[j] < ki[i] || (ki[j] == ki[i] && (lj < li || (lj == li && j < i)))) ? 1u : 0u;
                }
        }
#pragma unroll
        for (uint32_t i = 0; i < RT; ++i)
            if (i < len && rank[i] != i) V.store(a + rank[i], ki[i], vi[i]);   // (all of the run's loads have returned: the ranks depend on them)
    }
}

template <class View>
__global__ __launch_bounds__(256) void k_repair_wave(const View V, typename View::Key kmask, uint64_t n, const uint32_t *__restrict__ bad_q,
                                                     const unsigned long long *__restrict__ n_desc, const uint32_t *__restrict__ mid,
                                                     const uint32_t *__restrict__ n_mid, uint32_t *__restrict__ big,
                                                     uint32_t *__restrict__ n_big, uint32_t *__restrict__ status)
{
    if (*n_desc > REPAIR_MAX_DESC) return;                       // (k_repair_thread has raised status)
    const uint32_t M = min(*n_mid, REPAIR_MAX_MID);              // owners of runs the thread form left (wave-uniform)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t bi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); bi < M; bi += n_waves) {   // (wave-uniform)
        const uint32_t q = bad_q[mid[bi]];
        const typename View::Key k = V.key(q) & kmask;
        uint64_t a = q, e = (uint64_t)q + 1;                     // [a, e) belongs to the run
        bool too_long = false;
        for (;;) {                                               // leftwards: positions a-1, a-2, ..
            const bool match = a >= 1 + (uint64_t)lane && (V.key(a - 1 - lane) & kmask) == k;
            const unsigned long long miss = ~__ballot(match);
            if (miss) {
                a -= (uint64_t)__builtin_ctzll(miss);
                break;
            }
            a -= 64;
            if (e - a > REPAIR_MAX_RUN) { too_long = true; break; }
        }
        while (!too_long) {                                      // rightwards: positions e, e+1, ..
            const bool match = e + lane < n && (V.key(e + lane) & kmask) == k;
            const unsigned long long miss = ~__ballot(match);
            if (miss) {
                e += (uint64_t)__builtin_ctzll(miss);
                break;
            }
            e += 64;
            if (e - a > REPAIR_MAX_RUN) too_long = true;
        }
        const uint32_t len = (uint32_t)(e - a);
        if (too_long || len > REPAIR_MAX_RUN) {
            if (lane == 0) *status = 1u;
            continue;
        }
        if (len > 64) {
            if (lane == 0) {
                const uint32_t s = atomicAdd(n_big, 1u);         // (at most one entry per listed owner: <= REPAIR_MAX_MID, the list's size)
                big[2 * s] = (uint32_t)a;
                big[2 * s + 1] = len;
            }
            continue;
        }
        typename View::Key ki = 0;
        typename View::Elem vi;
        uint64_t li = 0;
        const bool have = lane < len;
        if (have) {
            V.load(a + lane, ki, vi);
            li = View::low_of(vi);
        }
        uint32_t rank = 0;
        for (uint32_t j = 0; j < len; ++j) {
            const typename View::Key kj = readlane_key(ki, j);
            const uint64_t lj = readlane64(li, j);
            rank += (kj < ki || (kj == ki && (lj < li || (lj == li && j < lane)))) ? 1u : 0u;
        }
        if (have) V.store(a + rank, ki, vi);   // (every load of the run has returned: the ranks depend on them)
    }
}

// A long run is almost always the occurrences of two or three hashes interleaved in input order.  Its stable sort is found
// value by value: the smallest (key, low) not yet placed (a workgroup minimum), then its elements in input order (a
// workgroup scan over contiguous chunks of the run) -- a handful of reductions instead of len^2 comparisons (1.9 -> 0.x ms at
// 15k genomes, where a node has up to 500 occurrences); a run with more than REPAIR_MAX_VALUES distinct values falls back to
// ranking every element against every other.
constexpr uint32_t REPAIR_MAX_VALUES = 12;
template <class View>
__global__ __launch_bounds__(256) void k_repair_sort(const View V, const uint32_t *__restrict__ n_big, const uint32_t *__restrict__ big)
{
    __shared__ typename View::Elem sv[REPAIR_MAX_RUN];
    __shared__ typename View::Key sk[REPAIR_MAX_RUN];
    __shared__ uint16_t srank[REPAIR_MAX_RUN];
    __shared__ typename View::Key w_k[4];
    __shared__ uint32_t w_cnt[4];
    __shared__ uint64_t w_l[4];
    const uint32_t B = *n_big;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t b = blockIdx.x; b < B; b += gridDim.x) {
        const uint32_t a = big[2 * b], len = big[2 * b + 1];
        __syncthreads();                      // the previous run has been written out
        for (uint32_t i = tid; i < len; i += blockDim.x) V.load(a + i, sk[i], sv[i]);
        __syncthreads();
        const uint32_t per = (len + 255u) / 256u, i0 = min(len, tid * per), i1 = min(len, i0 + per);   // this thread's chunk
        uint32_t placed = 0;
        typename View::Key last_k = 0;
        uint64_t last_l = 0;
        bool have_last = false;
        for (uint32_t it = 0; it < REPAIR_MAX_VALUES && placed < len; ++it) {
            // the smallest (key, low) above the last one placed
            typename View::Key mk = ~(typename View::Key)0;
            uint64_t ml = ~0ull;
            bool any = false;
            for (uint32_t i = i0; i < i1; ++i) {
                const typename View::Key k = sk[i];
                const uint64_t l = View::low_of(sv[i]);
                if (have_last && !(k > last_k || (k == last_k && l > last_l))) continue;
                if (!any || k < mk || (k == mk && l < ml)) {
                    mk = k;
                    ml = l;
                    any = true;
                }
            }
            for (int d = 32; d; d >>= 1) {
                const typename View::Key ok = __shfl_xor(mk, d, 64);
                const uint64_t ol = __shfl_xor(ml, d, 64);
                const bool oany = __shfl_xor((int)any, d, 64) != 0;
                if (oany && (!any || ok < mk || (ok == mk && ol < ml))) {
                    mk = ok;
                    ml = ol;
                    any = true;
                }
            }
            __syncthreads();                  // (w_* of the previous round have been read)
            if (lane == 0) {
                w_k[wave] = mk;
                w_l[wave] = ml;
                w_cnt[wave] = any ? 1u : 0u;
            }
            __syncthreads();
            any = false;
            for (uint32_t w = 0; w < 4; ++w)
                if (w_cnt[w] && (!any || w_k[w] < mk || (w_k[w] == mk && w_l[w] < ml))) {
                    mk = w_k[w];
                    ml = w_l[w];
                    any = true;
                }
            // (placed < len: some element is above the last value, so `any` holds for the workgroup)
            uint32_t c = 0;
            for (uint32_t i = i0; i < i1; ++i) c += (sk[i] == mk && View::low_of(sv[i]) == ml) ? 1u : 0u;
            uint32_t incl = c;
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            __syncthreads();                  // (w_* have been read)
            if (lane == 63) w_cnt[wave] = incl;
            __syncthreads();
            uint32_t before = placed + incl - c, total = 0;
            for (uint32_t w = 0; w < 4; ++w) {
                if (w < wave) before += w_cnt[w];
                total += w_cnt[w];
            }
            for (uint32_t i = i0; i < i1; ++i)
                if (sk[i] == mk && View::low_of(sv[i]) == ml) srank[i] = (uint16_t)before++;
            placed += total;
            last_k = mk;
            last_l = ml;
            have_last = true;
        }
        __syncthreads();
        if (placed == len) {                  // (workgroup-uniform)
            for (uint32_t i = tid; i < len; i += blockDim.x) V.store(a + srank[i], sk[i], sv[i]);
        } else {
            for (uint32_t i = tid; i < len; i += blockDim.x) {
                const typename View::Key ki = sk[i];
                const typename View::Elem vi = sv[i];
                const uint64_t li = View::low_of(vi);
                uint32_t rank = 0;
                for (uint32_t j = 0; j < len; ++j) {
                    const typename View::Key kj = sk[j];
                    const uint64_t lj = View::low_of(sv[j]);
                    rank += (kj < ki || (kj == ki && (lj < li || (lj == li && j < i)))) ? 1u : 0u;
                }
                V.store(a + rank, ki, vi);
            }
        }
    }
}

// Buffers of the in-place repair (kept by the caller until the stream has been synchronised).
struct RepairState {
    DevArray<unsigned long long> n_desc;   // [0] descents, [1] run heads before the repair
    DevArray<unsigned long long> blk_cnt, blk_off;
    DevArray<uint32_t> slot_q, slot_k;   // per block of the descent passes: its first DESC_SLOT descents (position, masked key)
    DevArray<uint32_t> bad_q, mid, big, status;   // status: [0] leftovers for the general repair, [1] listed long runs (> 64), [2] listed
                                                  // owners of runs the thread form left (> REPAIR_THREAD_RUN)
};

// enqueue: list the descents of the phase-1 order, repair short runs in place.  `bad` (may be null) also receives the
// masked keys of the descents, ascending, for a general repair.
// W: the view the two sweeps read (the same elements as V; LowView for the occurrences where the compact low halves exist)
template <class View, class Sweep>
void enqueue_repair(const View &V, const Sweep &W, typename View::Key kmask, uint64_t n, uint32_t *bad, uint32_t cap, RepairState &r,
                    hipStream_t stream)
{
    const uint32_t n_blocks = (uint32_t)((n + DESC_BLOCK - 1) / DESC_BLOCK);
    const uint32_t max_desc = (uint32_t)std::min<uint64_t>(REPAIR_MAX_DESC, std::max<uint64_t>(n, 1));
    const uint32_t max_mid = std::min(max_desc, REPAIR_MAX_MID);
    r.bad_q.alloc(max_desc);
    r.mid.alloc(max_mid);
    r.big.alloc(2 * (size_t)max_mid);
    r.n_desc.alloc(2);
    r.status.alloc(3);
    r.blk_cnt.alloc(n_blocks);
    r.blk_off.alloc(n_blocks);
    r.slot_q.alloc((size_t)n_blocks * DESC_SLOT);
    r.slot_k.alloc((size_t)n_blocks * DESC_SLOT);
    SW_HIP(hipMemsetAsync(r.status.p, 0, 12, stream));
    hipLaunchKernelGGL(k_count_descents<Sweep>, dim3(n_blocks), dim3(256), 0, stream, W, kmask, n, r.blk_cnt.p, r.slot_q.p, r.slot_k.p);
    SW_HIP(hipGetLastError());
    exclusive_sum(r.blk_cnt.p, r.blk_off.p, n_blocks, 0ull, stream);   // both halves at once: neither sum reaches 2^32
    hipLaunchKernelGGL(k_list_descents<Sweep>, dim3(n_blocks), dim3(256), 0, stream, W, kmask, n, r.blk_cnt.p, r.blk_off.p,
                       n_blocks, bad, cap, r.bad_q.p, max_desc, r.n_desc.p, r.slot_q.p, r.slot_k.p);
    hipLaunchKernelGGL(k_repair_thread<View>, dim3(REPAIR_GRID), dim3(256), 0, stream, V, kmask, n, r.bad_q.p, r.n_desc.p, r.mid.p,
                       r.status.p + 2, r.status.p);
    hipLaunchKernelGGL(k_repair_wave<View>, dim3(REPAIR_GRID), dim3(256), 0, stream, V, kmask, n, r.bad_q.p, r.n_desc.p,
                       (const uint32_t *)r.mid.p, (const uint32_t *)(r.status.p + 2), r.big.p, r.status.p + 1, r.status.p);
    hipLaunchKernelGGL(k_repair_sort<View>, dim3(REPAIR_GRID), dim3(256), 0, stream, V, r.status.p + 1, r.big.p);
    SW_HIP(hipGetLastError());
}

// one thread per (sorted) bad key: the first of equal entries looks its run up in the sorted phase-1 keys
__global__ void k_bad_runs(const uint32_t *__restrict__ bad, uint32_t n_bad, const uint32_t *__restrict__ key32,
                           uint32_t kmask, uint64_t n, uint32_t *__restrict__ run_start, uint32_t *__restrict__ run_len)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bad) return;
    const uint32_t k = bad[i];
    if (i && bad[i - 1] == k) {
        run_start[i] = 0;
        run_len[i] = 0;
        return;
    }
    uint64_t lo = 0, hi = n;   // first q with (key32[q] & kmask) >= k
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((key32[mid] & kmask) < k) lo = mid + 1; else hi = mid;
    }
    const uint64_t a = lo;
    hi = n;                    // first q with (key32[q] & kmask) > k
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((key32[mid] & kmask) <= k) lo = mid + 1; else hi = mid;
    }
    run_start[i] = (uint32_t)a;
    run_len[i] = (uint32_t)(lo - a);
}

// sub[t] = element t of the concatenated bad runs (run_off = exclusive sum of run_len; null: identity, everything)
__global__ void k_gather_sub(const uint32_t *__restrict__ key32, const OccPay *__restrict__ pay,
                             const uint32_t *__restrict__ run_start, const uint64_t *__restrict__ run_off, uint32_t n_bad,
                             uint64_t n_sub, uint64_t *__restrict__ sub_h, OccPay *__restrict__ sub_v,
                             uint32_t *__restrict__ pos)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_sub) return;
    uint64_t q = t;
    if (run_off) {
        uint32_t lo = 0, hi = n_bad;   // last entry with run_off <= t (zero-length entries share an offset: take the last)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (run_off[mid] <= t) lo = mid; else hi = mid;
        }
        q = (uint64_t)run_start[lo] + (t - run_off[lo]);
        pos[t] = (uint32_t)q;
    }
    const OccPay v = pay[q];
    sub_h[t] = ((uint64_t)key32[q] << 32) | v.low;
    sub_v[t] = v;
}

__global__ void k_scatter_sub(const uint64_t *__restrict__ sub_h, const OccPay *__restrict__ sub_v,
                              const uint32_t *__restrict__ pos, uint64_t n_sub, uint32_t *__restrict__ key32,
                              OccPay *__restrict__ pay)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_sub) return;
    const uint64_t q = pos ? pos[t] : t;
    key32[q] = (uint32_t)(sub_h[t] >> 32);
    pay[q] = sub_v[t];   // (its low half is the hash's: it travelled with the element)
}

struct PaySort {
    DevArray<uint32_t> key_a, key_b;   // phase-1 double buffers; key_a / pay_a are handed in filled
    DevArray<OccPay> pay_a, pay_b;
    uint32_t *key32 = nullptr;         // result: top halves, ascending
    OccPay *pay = nullptr;             // result: low half, pos, record, original index (stable)
    uint64_t n_repaired = 0;
    // repair bookkeeping (device): keys of the descents for the general repair, and the in-place repair's state
    DevArray<uint32_t> bad;
    RepairState rep;
    uint32_t cap = 0, kmask = ~0u;
    uint64_t n = 0;
    DevArray<uint32_t> fail;           // radix.hip's passes: non-zero if one gave up waiting (read in settle_sort)
    DevArray<uint32_t> low;            // OccPay::low in sorted order, from the sort's last pass (radix.hip): what the descent sweeps read
    const uint64_t *rows_src = nullptr;   // the occurrences are rows of (out_hash, pos | record << 32) (a slice's received tuples): the
                                          // sort's first pass reads them itself; key_a / pay_a are allocated by the sort
    OrderedOcc *staged = nullptr;      // the occurrences still lie in the sketch stage (order_tuples, take_stage): key_a / pay_a are
                                       // allocated by the sort once its first pass has read the stage, which is released there
};

// Phases 1 and 2 are enqueued without any host round trip; the caller must call sort_pay_settle() once the
// stream is synchronised (it reads `status`) before it relies on the order.
void sort_pay(uint64_t n, hipStream_t stream, PaySort &o)
{
    unsigned bits = 32;
    if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_SORT_KEYBITS")) {
        const int b = atoi(e);
        if (b >= 1 && b <= 32) bits = (unsigned)b;
    }
    o.kmask = ~0u << (32 - bits);
    o.n = n;
    o.key_b.alloc(n);
    o.pay_b.alloc(n);
    uint32_t *keys = o.key_a.p, *keys_alt = o.key_b.p;
    OccPay *vals = o.pay_a.p, *vals_alt = o.pay_b.p;
    bool have_low = false;   // the last pass left the low halves in o.low (SEQWIN_AMD_DESC_LOW=0: A/B, the sweeps read the payloads)
    const bool want_low = bits == 32 && !(SW_AB_GETENV("SEQWIN_AMD_DESC_LOW") && atoi(SW_AB_GETENV("SEQWIN_AMD_DESC_LOW")) == 0);   // (=0: A/B, -DSW_AB)
    if (o.staged) {   // (order_tuples made sure: 32 key bits, radix.hip's pair passes)
        OrderedOcc &occ = *o.staged;
        StageSource S{};
        S.stage_hash = occ.stage.stage_hash.p;
        S.stage_kmer = occ.stage.stage_kmer.p;
        S.tile_count = occ.stage.tile_count.p;
        S.tile_offset = occ.stage.tile_offset.p;
        S.dst_off = occ.dst_off.p;
        S.n_tiles = occ.stage_tiles;
        S.mult = occ.stage_mult;
        S.rec_out = occ.rec.p;
        o.fail.alloc(1);
        SW_HIP(hipMemsetAsync(o.fail.p, 0, 4, stream));
        if (want_low) o.low.alloc(n);
        have_low = want_low && n != 0;
        radix_sort_pairs32(keys, keys_alt, vals, vals_alt, n, 32, stream, o.fail.p, &S, [&] {
            occ.stage = SketchOut();           // (stream-ordered pool: the blocks' next users follow the pass on this stream)
            occ.dst_off.release();
            occ.staged = false;
            o.key_a.alloc(n);
            o.pay_a.alloc(n);
            keys = o.key_a.p;
            vals = o.pay_a.p;
        }, o.low.p);
        o.staged = nullptr;
    } else if (o.rows_src) {   // (merge_build made sure: 32 key bits, radix.hip's pair passes)
        StageSource S{};
        S.rows = o.rows_src;
        o.fail.alloc(1);
        SW_HIP(hipMemsetAsync(o.fail.p, 0, 4, stream));
        if (want_low) o.low.alloc(n);
        have_low = want_low && n != 0;
        radix_sort_pairs32(keys, keys_alt, vals, vals_alt, n, 32, stream, o.fail.p, &S, [&] {
            o.key_a.alloc(n);
            o.pay_a.alloc(n);
            keys = o.key_a.p;
            vals = o.pay_a.p;
        }, o.low.p);
    } else {
        if (bits < 32) {   // test knob: the top `bits` bits rotated down to bit 0, sorted there, rotated back (no bit is lost)
            hipLaunchKernelGGL(k_rot_keys, dim3(blocks_for(n)), dim3(TPB), 0, stream, keys, n, bits);
            SW_HIP(hipGetLastError());
        }
        o.fail.alloc(1);
        SW_HIP(hipMemsetAsync(o.fail.p, 0, 4, stream));
        if (want_low && sort_pairs_is_own(n, bits)) o.low.alloc(n);
        have_low = sort_pairs32(keys, keys_alt, vals, vals_alt, n, bits, stream, o.fail.p, o.low.p);
        if (bits < 32) {
            hipLaunchKernelGGL(k_rot_keys, dim3(blocks_for(n)), dim3(TPB), 0, stream, keys, n, 32 - bits);
            SW_HIP(hipGetLastError());
        }
    }
    o.key32 = keys;
    o.pay = vals;
    o.cap = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>(1u << 16, n / 16));
    o.bad.alloc(o.cap);
    if (have_low) enqueue_repair(PayView{keys, vals}, LowView{keys, o.low.p}, o.kmask, n, o.bad.p, o.cap, o.rep, stream);
    else enqueue_repair(PayView{keys, vals}, PayView{keys, vals}, o.kmask, n, o.bad.p, o.cap, o.rep, stream);
}

// General repair, for what the in-place pass left (status != 0).  Returns true if the order changed.
bool sort_pay_settle(PaySort &o, unsigned long long D, uint32_t status, hipStream_t stream)
{
    if (D == 0 || status == 0) {
        o.n_repaired = D;
        return false;
    }
    const uint64_t n = o.n;
    const uint32_t kmask = o.kmask, cap = o.cap;
    uint32_t *keys = o.key32;
    OccPay *vals = o.pay;
    DevArray<uint32_t> &bad = o.bad;
    uint64_t n_sub = n;
    DevArray<uint32_t> run_start, run_len, pos;
    DevArray<uint64_t> run_off;
    const bool listed = D <= cap;   // otherwise: more descents than the list holds -> treat everything as one subset
    if (listed) {
        const uint32_t nb = (uint32_t)D;
        // (the key list is ascending: the descents were listed in position order)
        run_start.alloc(nb);
        run_len.alloc(nb);
        run_off.alloc((size_t)nb + 1);
        hipLaunchKernelGGL(k_bad_runs, dim3((unsigned)blocks_for(nb)), dim3(TPB), 0, stream, bad.p, nb, keys, kmask, n,
                           run_start.p, run_len.p);
        SW_HIP(hipGetLastError());
        exclusive_sum(rocprim::make_transform_iterator(run_len.p, U32ToU64()), run_off.p, nb, (uint64_t)0, stream);
        uint64_t last_off = 0;
        uint32_t last_len = 0;
        SW_HIP(hipMemcpyAsync(&last_off, run_off.p + (nb - 1), 8, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipMemcpyAsync(&last_len, run_len.p + (nb - 1), 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));
        n_sub = last_off + last_len;
        pos.alloc(n_sub);
    }
    DevArray<uint64_t> sh0(n_sub), sh1(n_sub);
    DevArray<OccPay> sp0(n_sub), sp1(n_sub);
    hipLaunchKernelGGL(k_gather_sub, dim3(blocks_for(n_sub)), dim3(TPB), 0, stream, keys, vals, run_start.p,
                       listed ? run_off.p : (const uint64_t *)nullptr, (uint32_t)(listed ? D : 0), n_sub, sh0.p, sp0.p, pos.p);
    SW_HIP(hipGetLastError());
    uint64_t *sk = sh0.p, *sk_alt = sh1.p;
    OccPay *sv = sp0.p, *sv_alt = sp1.p;
    sort_pairs(sk, sk_alt, sv, sv_alt, n_sub, 0, 64, stream);
    hipLaunchKernelGGL(k_scatter_sub, dim3(blocks_for(n_sub)), dim3(TPB), 0, stream, sk, sv,
                       listed ? pos.p : (const uint32_t *)nullptr, n_sub, keys, vals);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));   // side arrays are released on return
    o.n_repaired = n_sub;
    return true;
}

// wait for the sort and its in-place repair, run the general repair if something was left; returns an upper bound of the
// number of distinct hashes (the run heads counted before the repair)
uint64_t settle_sort(PaySort &ps, hipStream_t stream)
{
    uint32_t status = 0, failed = 0;
    unsigned long long dh[2] = {0, 0};
    SW_HIP(hipMemcpyAsync(&status, ps.rep.status.p, 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipMemcpyAsync(dh, ps.rep.n_desc.p, 16, hipMemcpyDeviceToHost, stream));
    if (ps.fail.p) SW_HIP(hipMemcpyAsync(&failed, ps.fail.p, 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    check_sort_failed(failed);
    sort_pay_settle(ps, dh[0], status, stream);
    return dh[1];
}

// Sorted occurrences -> kmers, nodes (hash, start, stop; counts zero), node rank of every occurrence in ORIGINAL order
// (rank_out, may be null), and -- with rec_flag -- the first-of-assembly bitmaps for the counts (tbits / nbits non-null)
// and / or the repeated-in-assembly mark in bit 31 of the rank words (*rep_marked, if the node count allows it).
// ps.key_a / ps.pay_a hold the n occurrences on entry.  Returns the number of nodes.
// hold (single-GPU build): when the ranks come back through the bucketed unsort AND carry the repeat marks, its last step is
// left to the caller, who writes the adjacency keys from the buckets (k_unsort_adj) instead of a rank array: hold->sorted then
// points at the (index << 32 | rank) words grouped by bucket and rank_out stays unwritten.
struct UnsortHold {
    DevArray<uint64_t> a, b;
    const uint64_t *sorted = nullptr;
};
// what k_nodes leaves of the nodes: dense hashes and first-occurrence positions (relative to the sorted range); k_finish_nodes
// makes the node array of them (finish_nodes), the edges read the hashes
struct NodeParts {
    DevArray<uint64_t> hash;
    DevArray<uint32_t> start;
};
void finish_nodes(sw_index &ix, const NodeParts &np, uint64_t base, uint64_t end, const unsigned long long *tbits,
                  const unsigned long long *nbits, double inv_tar, double inv_neg, hipStream_t stream)
{
    if (!ix.n_nodes) return;
    const dim3 grid((unsigned)((ix.n_nodes + FIN_THREADS - 1) / FIN_THREADS));
    if (tbits && nbits)
        hipLaunchKernelGGL(k_finish_nodes<true>, grid, dim3(FIN_THREADS), 0, stream, ix.nodes.p, ix.n_nodes, (const uint64_t *)np.hash.p,
                           (const uint32_t *)np.start.p, base, end, tbits, nbits, inv_tar, inv_neg);
    else
        hipLaunchKernelGGL(k_finish_nodes<false>, grid, dim3(FIN_THREADS), 0, stream, ix.nodes.p, ix.n_nodes, (const uint64_t *)np.hash.p,
                           (const uint32_t *)np.start.p, base, end, (const unsigned long long *)nullptr,
                           (const unsigned long long *)nullptr, 0.0, 0.0);
    SW_HIP(hipGetLastError());
}

// ---- recovery when k_nodes' order guard trips (r05) ---------------------------------------------------------------------------
// The sorted arrays still hold every occurrence with its place in the (record_idx, pos) stream (OccPay::idx), so the stream can
// be restored without the sketch stage (released by the sort's first pass): scatter every element to its index, then sort by the
// whole 64-bit hash with rocPRIM's radix sort (ballot-free match-any ranking, stable by construction) -- exactly
// lsd_radix_sort (build_internals.cpp:76-144) on the original stream.  Slow (8 passes of 24 B) and never taken on a device whose
// LDS unit behaves as the start-up self-check saw it; taken in the suite through SEQWIN_AMD_FAULT_INJECT=rank.
__global__ void k_restore_stream(const uint32_t *__restrict__ key32, const OccPay *__restrict__ pay, uint64_t n, uint64_t *__restrict__ key64,
                                 OccPay *__restrict__ out, uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const OccPay q = pay[i];
    if (q.idx >= n) {
        atomicAdd(bad, 1u);
        return;
    }
    key64[q.idx] = ((uint64_t)key32[i] << 32) | q.low;
    out[q.idx] = q;
}
__global__ void k_split_restored(const uint64_t *__restrict__ key64, const OccPay *__restrict__ pay, uint64_t n, uint32_t *__restrict__ key32,
                                 uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key32[i] = (uint32_t)(key64[i] >> 32);
    if (pay[i].idx == 0xFFFFFFFFu) atomicAdd(bad, 1u);   // a place of the stream nothing was scattered to: the passes lost an element
}
void resort_pay_stable(PaySort &ps, uint64_t n, hipStream_t stream)
{
    OccPay *cur = ps.pay, *other = (ps.pay == ps.pay_a.p) ? ps.pay_b.p : ps.pay_a.p;
    if (!other || !cur) raise(SW_ERR_RUNTIME, "internal error: the node sort's second buffer is gone");
    DevArray<uint64_t> k0(n), k1(n);
    DevArray<uint32_t> bad(1);
    SW_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
    SW_HIP(hipMemsetAsync(other, 0xFF, n * sizeof(OccPay), stream));
    hipLaunchKernelGGL(k_restore_stream, dim3(blocks_for(n)), dim3(TPB), 0, stream, ps.key32, cur, n, k0.p, other, bad.p);
    SW_HIP(hipGetLastError());
    uint64_t *k = k0.p, *k_alt = k1.p;
    OccPay *v = other, *v_alt = cur;
    sort_pairs(k, k_alt, v, v_alt, n, 0, 64, stream);
    hipLaunchKernelGGL(k_split_restored, dim3(blocks_for(n)), dim3(TPB), 0, stream, k, v, n, ps.key32, bad.p);
    SW_HIP(hipGetLastError());
    uint32_t nbad = 0;
    SW_HIP(hipMemcpyAsync(&nbad, bad.p, 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));   // (k0 / k1 are released on return)
    if (nbad) raise(SW_ERR_RUNTIME, "internal error: the node sort lost %u occurrences (not a permutation of its input)", nbad);
    ps.pay = v;
}
// parts: the dense hash / start arrays of the nodes, kept for the caller -- who then also completes the nodes (finish_nodes:
// with the bitmaps, on the stream of its choice); without `parts` the nodes are completed here, counts zero.
uint32_t group_occurrences(PaySort &ps, uint64_t n, uint64_t base, const uint32_t *rec_flag, hipStream_t stream, sw_index &ix,
                           uint32_t *rank_out, DevArray<unsigned long long> *tbits, DevArray<unsigned long long> *nbits,
                           bool *rep_marked = nullptr, UnsortHold *hold = nullptr, NodeParts *parts = nullptr)
{
    sort_pay(n, stream, ps);
    const uint64_t node_cap = settle_sort(ps, stream);   // >= the number of nodes, within ~2 descents of it
    ix.nodes.alloc(node_cap);
    NodeParts own_parts;
    NodeParts &np = parts ? *parts : own_parts;
    np.hash.alloc(node_cap);
    np.start.alloc(node_cap);
    uint64_t direct_max = UNSORT_DIRECT_MAX;
    if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_UNSORT_DIRECT")) direct_max = 1ull << std::min(40, std::max(0, atoi(e)));   // A/B, tests
    const bool direct = rank_out && n <= direct_max;
    DevArray<uint64_t> uv0, uv1;
    if (rank_out && !direct) uv0.alloc(n);
    const bool bits = rec_flag && tbits && nbits;
    const bool rep = rec_flag && rep_marked && rank_out && node_cap < (1ull << 31);
    if (rep_marked) *rep_marked = rep;
    if (bits) {
        tbits->alloc((n + 63) / 64);
        nbits->alloc((n + 63) / 64);
    }
    const unsigned blocks = (unsigned)((n + NODES_TILE - 1) / NODES_TILE);
    DevArray<unsigned long long> tile_state(blocks);
    DevArray<uint32_t> words(4);   // [0] tile tickets, [1] the number of nodes, [2] the unsort's radix passes gave up, [3] order-guard violations
    uint32_t back[3] = {0, 0, 0};  // words 1 .. 3 (read after the unsort has been enqueued)
    if (rank_out && !direct) uv1.alloc(n);
    for (int attempt = 0;; ++attempt) {
        SW_HIP(hipMemsetAsync(tile_state.p, 0, (size_t)blocks * 8, stream));
        SW_HIP(hipMemsetAsync(words.p, 0, 16, stream));
        {
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(NODES_THREADS), 0, stream, ps.key32, ps.pay, n, base, rec_flag, ix.kmers.p,
                                   np.hash.p, np.start.p, direct ? rank_out : (uint32_t *)nullptr, uv0.p, bits ? tbits->p : (unsigned long long *)nullptr,
                                   bits ? nbits->p : (unsigned long long *)nullptr, tile_state.p, words.p, words.p + 1, words.p + 3);
            };
            if (bits && rep) launch(k_nodes<true, true>);
            else if (bits) launch(k_nodes<true, false>);
            else if (rep) launch(k_nodes<false, true>);
            else launch(k_nodes<false, false>);
        }
        SW_HIP(hipGetLastError());
        if (rank_out && !direct) {
            unsigned nbit = 1;
            while (nbit < 32 && (1ull << nbit) < n) ++nbit;          // indices < 2^nbit
            uint64_t *v = uv0.p;
            if (nbit > UNSORT_BITS) {   // buckets of 2^14 consecutive indices: sort on the index's bits above those
                // (begin_bit > 0 of rocPRIM's radix sort is checked on this stack by scripts/micro/sort_beginbit.hip)
                uint64_t *v_alt = uv1.p;
                if (!(sort_keys64_is_own(n) && radix_unsort_perm(v, v_alt, n, UNSORT_BITS, nbit, stream, words.p + 2)))
                    sort_keys64(v, v_alt, n, 32 + UNSORT_BITS, 32 + nbit, stream, words.p + 2, true);   // (the indices are a permutation)
            }
            if (hold && rep && nbit > UNSORT_BITS && !SW_TEST_GETENV("SEQWIN_AMD_ADJ_SEPARATE")) {
                hold->sorted = v;
            } else {
                hipLaunchKernelGGL(k_unsort_bucket, dim3((unsigned)((n + UNSORT_RANGE - 1) / UNSORT_RANGE)), dim3(1024), 0, stream, v, n,
                                   rank_out);
                SW_HIP(hipGetLastError());
            }
        }
        SW_HIP(hipMemcpyAsync(back, words.p + 1, 12, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));   // n_nodes has arrived
        if (back[2] == 0) break;
        // k_nodes saw occurrences out of (hash, stream index) order: the LDS-atomic ranking of the radix passes did not keep
        // the lane order on this device (or SEQWIN_AMD_FAULT_INJECT=rank).  Nothing of this attempt is kept.
        if (attempt)
            raise(SW_ERR_RUNTIME, "internal error: the node sort is still out of order after the rocPRIM re-sort (%u places)", back[2]);
        order_guard_tripped(0, back[2]);
        radix_demote_rank();
        resort_pay_stable(ps, n, stream);
    }
    const uint32_t n_nodes = back[0];
    check_sort_failed(back[1]);
    if (n_nodes > node_cap) raise(SW_ERR_RUNTIME, "internal error: %u nodes exceed the bound %llu", n_nodes, (unsigned long long)node_cap);
    ix.n_nodes = n_nodes;
    if (!parts) {   // (a caller that takes the parts completes the nodes itself: finish_nodes)
        finish_nodes(ix, np, base, base + n, nullptr, nullptr, 0.0, 0.0, stream);
        SW_HIP(hipStreamSynchronize(stream));   // (own_parts goes back to the pool on return)
    }
    if (hold && hold->sorted) {
        hold->a = std::move(uv0);
        hold->b = std::move(uv1);
    }
    // (the sort buffers go back to the pool here; later users are ordered after these kernels on this stream, or fenced)
    return n_nodes;
}

// node rank of every occurrence through the open-addressing table (A/B alternative to the unsort)
void ranks_from_table(const sw_index &ix, const uint64_t *hash, uint64_t n, hipStream_t stream, uint32_t *rank)
{
    uint64_t cap = 1024;
    while (cap < 2 * ix.n_nodes) cap <<= 1;   // load <= 0.5
    DevArray<ulonglong2> slots(cap);
    DevArray<uint32_t> special(1);
    SW_HIP(hipMemsetAsync(slots.p, 0xFF, cap * sizeof(ulonglong2), stream));
    SW_HIP(hipMemsetAsync(special.p, 0xFF, 4, stream));
    if (ix.n_nodes)
        hipLaunchKernelGGL(k_table_build, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, slots.p, cap - 1,
                           special.p);
    hipLaunchKernelGGL(k_table_lookup, dim3(blocks_for(n)), dim3(TPB), 0, stream, hash, n, slots.p, cap - 1, special.p, rank);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));   // the table is released on return
}

// chunk c of a low-memory build: its occurrences take places [occ_base, occ_base + n) of the whole stream and its
// records follow rec_base earlier ones (record re-basing: build_internals.cpp:334-355)
__global__ void k_append_chunk(const uint32_t *__restrict__ key32, const OccPay *__restrict__ pay, const uint32_t *__restrict__ rec,
                               uint64_t n, uint32_t occ_base, uint32_t rec_base, uint32_t *__restrict__ key32_out,
                               OccPay *__restrict__ pay_out, uint32_t *__restrict__ rec_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    OccPay p = pay[i];
    p.rec += rec_base;
    p.idx += occ_base;
    key32_out[occ_base + i] = key32[i];
    pay_out[occ_base + i] = p;
    rec_out[occ_base + i] = rec[i] + rec_base;
}

}  // namespace

void concat_occ(std::vector<OrderedOcc> &chunks, const std::vector<uint64_t> &rec_base, hipStream_t stream, OrderedOcc &out)
{
    uint64_t n = 0;
    for (const OrderedOcc &c : chunks) n += c.n;
    if (n > occ_cap()) raise_occ_cap(n, "minimizer occurrences");
    out.n = n;
    out.key32.alloc(n);
    out.pay.alloc(n);
    out.rec.alloc(n);
    uint64_t at = 0;
    for (size_t c = 0; c < chunks.size(); ++c) {
        OrderedOcc &ch = chunks[c];
        if (ch.n) {
            hipLaunchKernelGGL(k_append_chunk, dim3(blocks_for(ch.n)), dim3(TPB), 0, stream, ch.key32.p, ch.pay.p, ch.rec.p, ch.n,
                               (uint32_t)at, (uint32_t)rec_base[c], out.key32.p, out.pay.p, out.rec.p);
            SW_HIP(hipGetLastError());
        }
        at += ch.n;
        ch = OrderedOcc();   // (released blocks are reused in stream order)
    }
    // The chunk-sized blocks are too small for the full-size sort buffers that follow (the pool hands out blocks within 25 %
    // of the request): back to the device with them now, instead of peaking 24 B per minimizer above the one-shot build until
    // a failed hipMalloc trims the pool.
    if (chunks.size() > 1) {
        SW_HIP(hipStreamSynchronize(stream));
        dev_pool_trim();
    }
}

namespace {

// ---- windows above SW_MAX_WINDOW -------------------------------------------------------------------------
// The tile kernels ran with a smaller window w' (get_plan); their output C -- the minimizers of w', canonical hashes, in
// (record, position) order -- contains every minimizer of w, and the rightmost minimum of a w-window is the rightmost
// minimum of the members of C inside it.  So candidate i (valid-k-mer index g_i of its record, n valid k-mers) is a
// minimizer of w iff some window [x, x + w - 1], 0 <= x <= n - w, holds it with no strictly smaller candidate to its
// left and no smaller-or-equal one to its right (minimizer.cpp:36,40: '<=', the rightmost wins):
//     max(Lg + 1, g_i - w + 1, 0) <= min(Rg - w, g_i, n - w),
// Lg = index of the nearest candidate to the left with a smaller hash (-1: none), Rg = of the nearest to the right
// with a smaller or equal hash (n: none), both within the record.  The two neighbours come from a 64-ary tree of
// minima over C (level t + 1 = minima of 64 consecutive entries of level t): climb until a sibling's minimum
// qualifies, descend to the nearest element -- at most 128 reads per level whatever the input (a homopolymer makes
// every k-mer a candidate and every comparison a tie).
struct MinTree {
    const uint64_t *lvl[7];
    uint64_t size[7];
    int levels;
};
constexpr uint64_t LW_NONE = ~0ull;

// nodes of level t + 1 from level t: one wave per node
__global__ void k_lw_level(const uint64_t *__restrict__ in, uint64_t n_in, uint64_t *__restrict__ out, uint64_t n_out)
{
    const uint64_t node = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (node >= n_out) return;
    const uint64_t i = node * 64 + lane;
    uint64_t v = i < n_in ? in[i] : ~0ull;
    for (int d = 32; d; d >>= 1) {
        const uint64_t o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    if (lane == 0) out[node] = v;
}

// largest j in [lo, i) with hash[j] < h
__device__ uint64_t lw_nearest_left(const MinTree &T, uint64_t i, uint64_t lo, uint64_t h)
{
    uint64_t pos = i, q = 0;
    int t = 0;
    for (;;) {   // climb: the siblings left of `pos` at level t; nodes below lo >> 6t lie entirely before lo
        const uint64_t gstart = pos & ~63ull, lo_t = lo >> (6 * t), stop = gstart > lo_t ? gstart : lo_t;
        bool found = false;
        for (q = pos; q > stop;) {
            --q;
            if (T.lvl[t][q] < h) {
                found = true;
                break;
            }
        }
        if (found) break;
        if (gstart <= lo_t || t + 1 >= T.levels) return LW_NONE;
        pos >>= 6;
        ++t;
    }
    while (t > 0) {   // descend: the rightmost child that qualifies (only the leftmost admissible child can straddle lo,
                      // so a failure below it means there is nothing)
        const uint64_t cfirst = q << 6, clo = lo >> (6 * (t - 1));
        uint64_t c = cfirst + 64 < T.size[t - 1] ? cfirst + 64 : T.size[t - 1];   // one past the last child
        const uint64_t stop = cfirst > clo ? cfirst : clo;
        bool found = false;
        while (c > stop) {
            --c;
            if (T.lvl[t - 1][c] < h) {
                found = true;
                break;
            }
        }
        if (!found) return LW_NONE;
        q = c;
        --t;
    }
    return q;
}

// smallest j in (i, hi) with hash[j] <= h
__device__ uint64_t lw_nearest_right(const MinTree &T, uint64_t i, uint64_t hi, uint64_t h)
{
    uint64_t pos = i, q = 0;
    int t = 0;
    for (;;) {   // nodes from (hi + 64^t - 1) >> 6t on lie entirely at or behind hi
        const uint64_t gend = (pos | 63ull) + 1, hi_t = (hi + ((1ull << (6 * t)) - 1)) >> (6 * t);
        uint64_t stop = gend < hi_t ? gend : hi_t;
        if (T.size[t] < stop) stop = T.size[t];
        bool found = false;
        for (q = pos + 1; q < stop; ++q)
            if (T.lvl[t][q] <= h) {
                found = true;
                break;
            }
        if (found) break;
        if (gend >= hi_t || gend >= T.size[t] || t + 1 >= T.levels) return LW_NONE;
        pos >>= 6;
        ++t;
    }
    while (t > 0) {
        const uint64_t cfirst = q << 6, chi = (hi + ((1ull << (6 * (t - 1))) - 1)) >> (6 * (t - 1));
        uint64_t stop = cfirst + 64 < T.size[t - 1] ? cfirst + 64 : T.size[t - 1];
        if (chi < stop) stop = chi;
        uint64_t c = cfirst;
        bool found = false;
        for (; c < stop; ++c)
            if (T.lvl[t - 1][c] <= h) {
                found = true;
                break;
            }
        if (!found) return LW_NONE;
        q = c;
        --t;
    }
    return q;
}

// valid-k-mer index of every candidate: its segment is the last one starting at or before its position
__global__ void k_lw_index(const uint64_t *__restrict__ kmer, uint64_t c, const uint32_t *__restrict__ rec_seg_off,
                           const uint32_t *__restrict__ seg_pos, const uint32_t *__restrict__ seg_idx, uint32_t *__restrict__ g)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const uint32_t pos = (uint32_t)kmer[i], rec = (uint32_t)(kmer[i] >> 32);
    uint32_t a = rec_seg_off[rec], b = rec_seg_off[rec + 1];
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (seg_pos[mid] <= pos) a = mid; else b = mid;
    }
    g[i] = seg_idx[a] + (pos - seg_pos[a]);
}

__global__ void k_lw_select(const MinTree T, const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ g, uint64_t c,
                            const uint32_t *__restrict__ rec_nvalid, uint32_t w, uint32_t *__restrict__ keep)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const uint64_t h = T.lvl[0][i];
    const uint32_t rec = (uint32_t)(kmer[i] >> 32);
    const int64_t n = rec_nvalid[rec], gi = g[i];
    uint32_t k = 0;
    if (n >= (int64_t)w && h != ~0ull) {   // (minimizer.cpp:44-45: a window whose minimum is 2^64 - 1 emits nothing)
        uint64_t lo = 0, hi = i;           // the record's candidates are [lo, hi): first entry of rec / first of a later record
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint32_t)(kmer[mid] >> 32) < rec) lo = mid + 1; else hi = mid;
        }
        uint64_t a = i + 1;
        hi = c;
        while (a < hi) {
            const uint64_t mid = (a + hi) >> 1;
            if ((uint32_t)(kmer[mid] >> 32) <= rec) a = mid + 1; else hi = mid;
        }
        const uint64_t jl = lw_nearest_left(T, i, lo, h), jr = lw_nearest_right(T, i, hi, h);
        const int64_t Lg = jl == LW_NONE ? -1 : (int64_t)g[jl], Rg = jr == LW_NONE ? n : (int64_t)g[jr];
        int64_t x_lo = Lg + 1, x_hi = Rg - (int64_t)w;
        if (gi - (int64_t)w + 1 > x_lo) x_lo = gi - (int64_t)w + 1;
        if (x_lo < 0) x_lo = 0;
        if (gi < x_hi) x_hi = gi;
        if (n - (int64_t)w < x_hi) x_hi = n - (int64_t)w;
        k = x_lo <= x_hi ? 1u : 0u;
    }
    keep[i] = k;
}

template <bool INDEX>
__global__ void k_lw_emit(const uint64_t *__restrict__ canon, const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ keep,
                          const uint64_t *__restrict__ dst, uint64_t c, uint64_t mult, uint64_t *__restrict__ hash,
                          uint64_t *__restrict__ kmer_out, uint32_t *__restrict__ key32, OccPay *__restrict__ pay,
                          uint32_t *__restrict__ rec, unsigned long long *__restrict__ total)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    if (i == c - 1) *total = dst[i] + keep[i];
    if (!keep[i]) return;
    const uint64_t d = dst[i], km = kmer[i];
    uint64_t h = canon[i] * mult;   // extend_hashes, hashing_internals.hpp:89-103
    h ^= h >> 27;
    if (INDEX) {
        key32[d] = (uint32_t)(h >> 32);
        OccPay p;
        p.low = (uint32_t)h;
        p.pos = (uint32_t)km;
        p.rec = (uint32_t)(km >> 32);
        p.idx = (uint32_t)d;
        pay[d] = p;
        rec[d] = p.rec;
        if (hash) hash[d] = h;
    } else {
        hash[d] = h;
        kmer_out[d] = km;
    }
}

}  // namespace

// Staged index form: the node sort's first pass reads the stage (radix.hip, k_rs_pair_pass<true>), so the ordered copy
// (k_order: 40 B per tuple, and the pass's own 20 B read of it) is never made.  Taken when that pass is this library's, the
// tiles hold enough tuples for its tile window to cover a wave's share (STAGE_WIN), and nothing asks for the arrays themselves.
// SEQWIN_AMD_ORDER=copy keeps k_order (A/B, tests).
static bool stage_fits_sort(uint64_t n, uint32_t n_tiles)
{
    if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_ORDER")) {
        if (!strcmp(e, "copy")) return false;
        if (!strcmp(e, "stage")) return n > 0 && radix_pairs_available() && !SW_TEST_GETENV("SEQWIN_AMD_SORT_KEYBITS");
    }
    return n > 0 && sort_pairs_is_own(n, 32) && !SW_TEST_GETENV("SEQWIN_AMD_SORT_KEYBITS") && n / 16 >= n_tiles;
}

__global__ void k_fill_u64(uint64_t *p, uint32_t n, uint64_t v)
{
    if (threadIdx.x < n) p[threadIdx.x] = v;
}

void order_tuples(SketchOut &sk, const Plan &plan, hipStream_t stream, OrderedOcc &out, bool index_form, bool take_stage)
{
    const bool large = plan.w_full > plan.w;   // the stage holds a superset: the minimizers of plan.w (see above)
    out.n = sk.n_occ;
    out.staged = false;
    if (!large && out.n > occ_cap()) raise_occ_cap(out.n, "minimizer occurrences");
    const bool table_ranks = index_form && ranks_by_table();
    if (take_stage && index_form && !large && !table_ranks && plan.n_tiles && stage_fits_sort(sk.n_occ, plan.n_tiles)) {
        out.rec.alloc(out.n);
        out.dst_off.alloc((size_t)plan.n_tiles + STAGE_PAD);
        exclusive_sum(rocprim::make_transform_iterator(sk.tile_count.p, U32ToU64()), out.dst_off.p, plan.n_tiles, (uint64_t)0, stream);
        hipLaunchKernelGGL(k_fill_u64, dim3(1), dim3(64), 0, stream, out.dst_off.p + plan.n_tiles, STAGE_PAD, out.n);
        SW_HIP(hipGetLastError());
        out.stage = std::move(sk);
        out.stage_tiles = plan.n_tiles;
        out.stage_mult = plan.mult;
        out.staged = true;
        return;
    }
    auto alloc_out = [&](uint64_t n) {
        if (index_form) {
            out.key32.alloc(n);
            out.pay.alloc(n);
            out.rec.alloc(n);
            if (table_ranks) out.hash.alloc(n);
        } else {
            out.hash.alloc(n);
            out.kmer.alloc(n);
        }
    };
    if (!large) alloc_out(out.n);
    if (plan.n_tiles == 0 || sk.n_occ == 0) {
        out.n = 0;
        if (large) alloc_out(0);
        return;
    }
    DevArray<uint64_t> dst_off(plan.n_tiles);
    exclusive_sum(rocprim::make_transform_iterator(sk.tile_count.p, U32ToU64()), dst_off.p, plan.n_tiles,
                  (uint64_t)0, stream);
    const uint64_t threads = (uint64_t)plan.n_tiles * 64;
    if (large) {
        const uint64_t c = sk.n_occ;
        if (c > occ_cap()) raise_occ_cap(c, "window candidates");
        DevArray<uint64_t> canon(c), kmer(c), keep_off(c);
        DevArray<uint32_t> g(c), keep(c);
        DevArray<unsigned long long> total(1);
        hipLaunchKernelGGL((k_order<false, true>), dim3(blocks_for(threads)), dim3(TPB), 0, stream, sk.stage_hash.p, sk.stage_kmer.p,
                           sk.tile_count.p, sk.tile_offset.p, dst_off.p, plan.n_tiles, plan.mult, canon.p, kmer.p,
                           (uint32_t *)nullptr, (OccPay *)nullptr, (uint32_t *)nullptr);
        hipLaunchKernelGGL(k_lw_index, dim3(blocks_for(c)), dim3(TPB), 0, stream, kmer.p, c, plan.rec_seg_off.p, plan.seg_pos.p,
                           plan.seg_idx.p, g.p);
        SW_HIP(hipGetLastError());
        MinTree T;
        std::vector<DevArray<uint64_t>> levels;
        levels.reserve(7);
        T.lvl[0] = canon.p;
        T.size[0] = c;
        T.levels = 1;
        while (T.size[T.levels - 1] > 64) {   // c < 2^32 candidates: at most 6 levels
            const int t = T.levels;
            if (t >= 7) raise(SW_ERR_RUNTIME, "internal error: more than 2^36 window candidates");
            const uint64_t n_in = T.size[t - 1], n_out = (n_in + 63) / 64;
            levels.emplace_back(n_out);
            hipLaunchKernelGGL(k_lw_level, dim3(blocks_for(n_out * 64)), dim3(TPB), 0, stream, T.lvl[t - 1], n_in, levels.back().p,
                               n_out);
            SW_HIP(hipGetLastError());
            T.lvl[t] = levels.back().p;
            T.size[t] = n_out;
            T.levels = t + 1;
        }
        for (int t = T.levels; t < 7; ++t) {
            T.lvl[t] = nullptr;
            T.size[t] = 0;
        }
        hipLaunchKernelGGL(k_lw_select, dim3(blocks_for(c)), dim3(TPB), 0, stream, T, kmer.p, g.p, c, plan.rec_nvalid.p, plan.w_full,
                           keep.p);
        SW_HIP(hipGetLastError());
        exclusive_sum(rocprim::make_transform_iterator(keep.p, U32ToU64()), keep_off.p, c, (uint64_t)0, stream);
        // the survivors are at most the candidates: emit into arrays of that size, shrink the count afterwards
        alloc_out(c);
        if (index_form)
            hipLaunchKernelGGL(k_lw_emit<true>, dim3(blocks_for(c)), dim3(TPB), 0, stream, canon.p, kmer.p, keep.p, keep_off.p, c,
                               plan.mult, table_ranks ? out.hash.p : (uint64_t *)nullptr, (uint64_t *)nullptr, out.key32.p,
                               out.pay.p, out.rec.p, total.p);
        else
            hipLaunchKernelGGL(k_lw_emit<false>, dim3(blocks_for(c)), dim3(TPB), 0, stream, canon.p, kmer.p, keep.p, keep_off.p, c,
                               plan.mult, out.hash.p, out.kmer.p, (uint32_t *)nullptr, (OccPay *)nullptr, (uint32_t *)nullptr,
                               total.p);
        SW_HIP(hipGetLastError());
        unsigned long long n_keep = 0;
        SW_HIP(hipMemcpyAsync(&n_keep, total.p, 8, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));   // (the side arrays are released on return)
        out.n = n_keep;
        if (out.n > occ_cap()) raise_occ_cap(out.n, "minimizer occurrences");
        return;
    }
    if (index_form)
        hipLaunchKernelGGL(k_order<true>, dim3(blocks_for(threads)), dim3(TPB), 0, stream, sk.stage_hash.p, sk.stage_kmer.p,
                           sk.tile_count.p, sk.tile_offset.p, dst_off.p, plan.n_tiles, plan.mult,
                           table_ranks ? out.hash.p : (uint64_t *)nullptr, (uint64_t *)nullptr, out.key32.p, out.pay.p, out.rec.p);
    else
        hipLaunchKernelGGL(k_order<false>, dim3(blocks_for(threads)), dim3(TPB), 0, stream, sk.stage_hash.p, sk.stage_kmer.p,
                           sk.tile_count.p, sk.tile_offset.p, dst_off.p, plan.n_tiles, plan.mult, out.hash.p, out.kmer.p,
                           (uint32_t *)nullptr, (OccPay *)nullptr, (uint32_t *)nullptr);
    SW_HIP(hipGetLastError());
    // (dst_off goes back to the pool here; its next user is ordered after k_order on this stream)
}

namespace {
struct PenaltyJob {   // buffers of an in-flight get_penalty (asynchronous on `stream`)
    DevArray<uint64_t> X, Y;
    DevArray<uint32_t> err;
    DevArray<unsigned char> tmp_x, tmp_y;   // scan temp storage: held until the job is finished, because the main
                                            // stream allocates from the same pool while this job runs
    hipStream_t stream = nullptr;
    bool active = false;
    ~PenaltyJob() { if (active) (void)hipStreamSynchronize(stream); }   // never release buffers of running kernels
};

void penalty_launch(const sw_kmer *d_kmers, uint64_t n_kmers, sw_node *d_nodes, uint64_t n_nodes,
                    const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target, uint64_t n_targets,
                    uint64_t n_non_targets, hipStream_t stream, PenaltyJob &job)
{
    job.stream = stream;
    job.active = n_nodes != 0;
    if (!job.active) return;
    job.X.alloc(n_kmers);
    job.Y.alloc(n_kmers);
    job.err.alloc(1);
    DevArray<uint64_t> &X = job.X, &Y = job.Y;
    DevArray<uint32_t> &err = job.err;
    SW_HIP(hipMemsetAsync(err.p, 0, 4, stream));
    const double inv_tar = 1.0 / (double)n_targets;        // filter.cpp:89-90
    const double inv_neg = 1.0 / (double)n_non_targets;
    if (n_kmers) {
        hipLaunchKernelGGL(k_pen_flags, dim3(blocks_for(n_kmers)), dim3(TPB), 0, stream, d_kmers, n_kmers, d_rec_asm, n_records,
                           d_is_target, X.p, Y.p);
        SW_HIP(hipGetLastError());
        inclusive_sum_keep(X.p, X.p, n_kmers, (uint64_t)0, stream, job.tmp_x);
        inclusive_sum_keep(Y.p, Y.p, n_kmers, (uint64_t)0, stream, job.tmp_y);
    }
    hipLaunchKernelGGL(k_pen_nodes, dim3(blocks_for(n_nodes)), dim3(TPB), 0, stream, d_kmers, n_kmers, d_nodes, n_nodes, d_rec_asm,
                       n_records, d_is_target, X.p, Y.p, inv_tar, inv_neg, err.p);
    SW_HIP(hipGetLastError());
}

uint64_t penalty_finish(PenaltyJob &job)
{
    if (!job.active) return 0;
    uint32_t e = 0;
    SW_HIP(hipMemcpyAsync(&e, job.err.p, 4, hipMemcpyDeviceToHost, job.stream));
    SW_HIP(hipStreamSynchronize(job.stream));
    job.active = false;
    return e;
}

hipStream_t side_stream()   // one per (thread, current device); never destroyed
{
    static thread_local std::map<int, hipStream_t> streams;
    int dev = 0;
    SW_HIP(hipGetDevice(&dev));
    hipStream_t &s = streams[dev];
    if (!s) SW_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
}  // namespace

namespace {
// ---- packed edge keys of the multi-GPU slices: pair and assembly in ONE 64-bit key when 2 nb + ab <= 64 ----
// (keys-only radix sort: 16 B instead of 24 B per element and pass; equal pairs of one assembly become adjacent
// duplicates, so weight = number of distinct keys inside a pair's run)
struct PackedChangeAny {   // 1 where the whole key differs from its predecessor (first element: 1)
    const uint64_t *keys;
    __host__ __device__ uint32_t operator()(uint64_t s) const { return (s == 0 || keys[s] != keys[s - 1]) ? 1u : 0u; }
};
struct PackedPairEq {      // same (rank_lo, rank_hi) pair
    unsigned pshift;
    uint64_t pmask;
    __host__ __device__ bool operator()(uint64_t a, uint64_t b) const { return ((a >> pshift) & pmask) == ((b >> pshift) & pmask); }
};
// keys = (pair << ab) | assembly (rows of the multi-GPU exchange, routed by key value)
void edges_from_packed(uint64_t *keys, uint64_t *keys_alt, uint64_t m, uint64_t sentinel, unsigned nb, unsigned ab,
                       const uint64_t *rank_hash, hipStream_t stream, sw_index &ix)
{
    ix.n_edges = 0;
    if (m == 0) return;
    const unsigned pshift = ab;
    const uint64_t pmask = ~0ull;
    DevArray<uint32_t> sort_fail(1);
    SW_HIP(hipMemsetAsync(sort_fail.p, 0, 4, stream));
    {
        const unsigned sort_bits = 2 * nb + ab;
        sort_keys64(keys, keys_alt, m, 0, sort_bits, stream, sort_fail.p);
        if (sort_keys64_is_own(m) && radix_rank_mode() == 1) {   // order guard (r05; this legacy form synchronises here, the default forms do not)
            DevArray<uint32_t> bad(1);
            uint32_t descents = 0;
            SW_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
            hipLaunchKernelGGL(k_check_ascending, dim3(blocks_for(m)), dim3(TPB), 0, stream, (const uint64_t *)keys, m, bad.p);
            SW_HIP(hipMemcpyAsync(&descents, bad.p, 4, hipMemcpyDeviceToHost, stream));
            SW_HIP(hipStreamSynchronize(stream));
            if (descents) {
                order_guard_tripped(1, descents);
                radix_demote_rank();
                sort_keys64(keys, keys_alt, m, 0, sort_bits, stream, sort_fail.p);
            }
        }
    }
    // One run-length pass over the sorted keys (rocprim::reduce_by_key, decoupled look-back): runs = equal pairs, value of
    // an element = 1 where the whole key (pair, assembly) differs from its predecessor, so a run's sum is the number of
    // distinct assemblies of the pair = the edge weight (build.cpp:170-196).  Replaces two prefix sums over all keys, a
    // head scatter and their re-reads.  Record boundaries were written as `sentinel`; they sort last and form the final run.
    DevArray<uint64_t> ukeys(m);
    DevArray<uint32_t> usum(m);
    DevArray<uint32_t> ucount(1);
    {
        auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), PackedChangeAny{keys});
        const PackedPairEq eq{pshift, pmask};
        size_t tmp_bytes = 0;
        SW_HIP(rocprim::reduce_by_key(nullptr, tmp_bytes, keys, flags, m, ukeys.p, usum.p, ucount.p, rocprim::plus<uint32_t>(), eq,
                                      stream));
        DevArray<unsigned char> tmp(tmp_bytes);
        SW_HIP(rocprim::reduce_by_key(tmp.p, tmp_bytes, keys, flags, m, ukeys.p, usum.p, ucount.p, rocprim::plus<uint32_t>(), eq,
                                      stream));
        hipLaunchKernelGGL(k_drop_sentinel_run, dim3(1), dim3(1), 0, stream, ukeys.p, sentinel, ucount.p);
        SW_HIP(hipGetLastError());
        uint32_t n_edges = 0, failed = 0;
        SW_HIP(hipMemcpyAsync(&n_edges, ucount.p, 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipMemcpyAsync(&failed, sort_fail.p, 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));   // (the pass's scratch is released after this)
        check_sort_failed(failed);
        ix.n_edges = n_edges;
    }
    if (ix.n_edges == 0) return;
    ix.edges.alloc(ix.n_edges);
    hipLaunchKernelGGL(k_edges_runs<false>, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ukeys.p, usum.p, pshift, pmask,
                       (uint64_t)ix.n_edges, nb, ix.nodes.p, rank_hash, ix.edges.p);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));
}
}  // namespace

namespace {
// keys[m] = (rank_lo << nb) | rank_hi of every adjacency record (sentinels sort last), no assemblies: a keys-only sort and
// the run lengths give the number of records of every pair; the candidates (records that may repeat their pair inside one
// assembly: ck / ca, *d_n_cand of them, unordered) are sorted by (pair, assembly) and the repeats taken off.
// (d_n_cand == nullptr: the caller knows the number of candidates, host_n_cand.)
struct WideKeys {   // multi-GPU slices: key = (rank_lo - lo_base) << hi_bits | rank_hi, hashes through the per-owner table
    unsigned lo_bits, hi_bits;
    uint64_t lo_base;
    RankHash hash;
};
void edges_from_pairs(uint64_t *keys, uint64_t *keys_alt, uint64_t m, uint64_t sentinel, unsigned nb, unsigned ab,
                      uint64_t *ck, uint32_t *ca, const unsigned long long *d_n_cand, uint64_t host_n_cand,
                      const uint64_t *rank_hash, hipStream_t stream, sw_index &ix, hipEvent_t rank_hash_ready = nullptr,
                      unsigned long long *d_hist = nullptr, const WideKeys *wide = nullptr)
{
    ix.n_edges = 0;
    if (m == 0) return;
    const unsigned key_bits = wide ? wide->lo_bits + wide->hi_bits : 2 * nb;
    DevArray<uint32_t> sort_fail(1);
    SW_HIP(hipMemsetAsync(sort_fail.p, 0, 4, stream));
    // Two phases, like the node sort: the radix passes leave out the lowest digit of rank_hi, and the few runs of equal upper
    // bits that hold two different pairs out of order are repaired in place (one sweep over the keys instead of a pass: 15k
    // genomes, 745 M keys: 18 334 descents with 9 bits left out, edges 34.9 -> 32.5 ms; with 18 bits 8.4 M descents -- more
    // than the repair takes).  Anything the in-place repair leaves (more than 2^22 descents, two pairs of one run with more
    // than 2048 records) makes the whole sort run again on all bits.
    unsigned low_bits = 0, skip = 0, digit_bits = 0, n_passes = 0;
    if (sort_keys64_is_own(m)) {
        radix_layout(key_bits, &digit_bits, &n_passes);
        const unsigned hb = wide ? wide->hi_bits : nb;
        skip = hb >= digit_bits + 8 ? 1 : 0;
        // r05: a SECOND digit is left to the repair where few keys repeat -- at most four occurrences per node on average.  With the
        // thread form of the repair (2^26 descents) the second digit's ~500 x more descents are cheap while the runs that hold them
        // are short: one GPU's share of 100 000 iid genomes 118 177 descents, no run above 64 keys, -2.3 ms per build; salmonella500 at
        // w = 10 973 217 descents, -0.7 ms; the 15 000-genome set (9.4 occurrences per node, up to 500) 8.4 M descents in 358 k runs
        // above 64 keys, +2.1 ms: stays at one (gpurun_out/r5g).  A third digit (4.9e7 descents on the iid share) loses everywhere.
        if (skip && hb >= 2 * digit_bits + 8 && ix.n_nodes && m <= 4 * ix.n_nodes) skip = 2;
        if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_EDGE_SKIP_PASSES")) skip = (unsigned)atoi(e);   // A/B, tests (0: all bits by radix passes)
        skip = std::min(skip, n_passes - 1);
        low_bits = skip * digit_bits;
    }
    sort_keys64(keys, keys_alt, m, low_bits, key_bits, stream, sort_fail.p, false,
                d_hist ? d_hist + ((size_t)skip << digit_bits) : nullptr, key_bits);   // (d_hist: the digit counts k_adj_pairs took)
    RepairState rep;
    if (low_bits) enqueue_repair(EdgeKeyView{keys, low_bits}, EdgeKeyView{keys, low_bits}, ~0ull, m, nullptr, 0, rep, stream);
    DevArray<uint64_t> ukeys(m);
    DevArray<uint32_t> ucnt(m + 1), ucount(2);   // ucount[1]: places where the sorted keys descend (order guard, r05)
    unsigned long long n_cand = host_n_cand;
    // run lengths by this library's streaming pass (k_rle_keys: ucnt then holds the START of every run, and m behind the last);
    // SEQWIN_AMD_RLE=rocprim: rocprim::run_length_encode (ucnt = the lengths) -- A/B, and what rounds 1-3 ran
    const char *rle_env = SW_AB_GETENV("SEQWIN_AMD_RLE");   // (-DSW_AB builds only)
    const bool own_rle = !(rle_env && !strcmp(rle_env, "rocprim"));
    bool demoted = false;
    for (int attempt = 0;; ++attempt) {
        SW_HIP(hipMemsetAsync(ucount.p + 1, 0, 4, stream));
        if (own_rle) {
            const unsigned blocks = (unsigned)((m + RLE_TILE - 1) / RLE_TILE);
            DevArray<unsigned long long> tile_state(blocks);
            DevArray<uint32_t> ticket(1);
            SW_HIP(hipMemsetAsync(tile_state.p, 0, (size_t)blocks * 8, stream));
            SW_HIP(hipMemsetAsync(ticket.p, 0, 4, stream));
            hipLaunchKernelGGL(k_rle_keys, dim3(blocks), dim3(RLE_THREADS), 0, stream, (const uint64_t *)keys, m, ukeys.p, ucnt.p, tile_state.p,
                               ticket.p, ucount.p, ucount.p + 1);
            SW_HIP(hipGetLastError());
        } else {
#ifdef SW_AB
            hipLaunchKernelGGL(k_check_ascending, dim3(blocks_for(m)), dim3(TPB), 0, stream, (const uint64_t *)keys, m, ucount.p + 1);
            size_t tmp_bytes = 0;
            SW_HIP(rocprim::run_length_encode(nullptr, tmp_bytes, keys, m, ukeys.p, ucnt.p, ucount.p, stream));
            DevArray<unsigned char> tmp(tmp_bytes);
            SW_HIP(rocprim::run_length_encode(tmp.p, tmp_bytes, keys, m, ukeys.p, ucnt.p, ucount.p, stream));
            SW_HIP(hipStreamSynchronize(stream));   // (tmp is released here)
#endif
        }
        hipLaunchKernelGGL(k_drop_sentinel_run, dim3(1), dim3(1), 0, stream, ukeys.p, sentinel, ucount.p);
        SW_HIP(hipGetLastError());
        uint32_t n_edges = 0, failed = 0, left = 0, descents = 0;
        SW_HIP(hipMemcpyAsync(&n_edges, ucount.p, 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipMemcpyAsync(&descents, ucount.p + 1, 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipMemcpyAsync(&failed, sort_fail.p, 4, hipMemcpyDeviceToHost, stream));
        if (low_bits && attempt == 0) SW_HIP(hipMemcpyAsync(&left, rep.status.p, 4, hipMemcpyDeviceToHost, stream));
        if (d_n_cand) SW_HIP(hipMemcpyAsync(&n_cand, d_n_cand, 8, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));   // (the pass's scratch is released after this)
        check_sort_failed(failed);
        if (low_bits && attempt == 0 && getenv("SEQWIN_AMD_DEBUG_EDGE_REPAIR")) {
            unsigned long long nd[2] = {0, 0};
            uint32_t st[2] = {0, 0};
            SW_HIP(hipMemcpy(nd, rep.n_desc.p, 16, hipMemcpyDeviceToHost));
            SW_HIP(hipMemcpy(st, rep.status.p, 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "[edge repair] %llu keys, %u low bits: %llu descents, %u long runs, leftovers %u\n", (unsigned long long)m, low_bits,
                    nd[0], st[1], st[0]);
        }
        if (left) {   // the in-place repair left runs unsorted: the keys are still the same multiset, sort them on all bits
            sort_keys64(keys, keys_alt, m, 0, key_bits, stream, sort_fail.p);
            continue;
        }
        if (descents) {
            // the sorted keys descend somewhere although sort and repair reported nothing: the LDS-atomic ranking of the radix
            // passes did not keep the lane order (or SEQWIN_AMD_FAULT_INJECT=rank).  The keys are still the same multiset: sort
            // them again, on all bits, ranking by ballots.
            if (demoted)
                raise(SW_ERR_RUNTIME, "internal error: the edge keys are still out of order after the ballot-ranked re-sort (%u places)", descents);
            order_guard_tripped(1, descents);
            radix_demote_rank();
            demoted = true;
            sort_keys64(keys, keys_alt, m, 0, key_bits, stream, sort_fail.p);
            continue;
        }
        ix.n_edges = n_edges;
        break;
    }
    if (ix.n_edges == 0) return;
    ix.edges.alloc(ix.n_edges);
    if (rank_hash_ready) SW_HIP(hipStreamWaitEvent(stream, rank_hash_ready, 0));   // (rank_hash is written on another stream)
    if (wide && own_rle)
        hipLaunchKernelGGL(k_edges_runs_wide<true>, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ukeys.p, ucnt.p, (uint64_t)ix.n_edges,
                           wide->hi_bits, wide->lo_base, wide->hash, ix.edges.p);
    else if (wide)
        hipLaunchKernelGGL(k_edges_runs_wide<false>, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ukeys.p, ucnt.p, (uint64_t)ix.n_edges,
                           wide->hi_bits, wide->lo_base, wide->hash, ix.edges.p);
    else if (own_rle)
        hipLaunchKernelGGL(k_edges_runs<true>, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ukeys.p, ucnt.p, 0u, ~0ull,
                           (uint64_t)ix.n_edges, nb, ix.nodes.p, rank_hash, ix.edges.p);
    else
        hipLaunchKernelGGL(k_edges_runs<false>, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ukeys.p, ucnt.p, 0u, ~0ull,
                           (uint64_t)ix.n_edges, nb, ix.nodes.p, rank_hash, ix.edges.p);
    SW_HIP(hipGetLastError());
    if (n_cand) {
        const uint64_t c = n_cand;
        // (pair, assembly) order: least significant key first, both sorts stable
        DevArray<uint64_t> ck1(c);
        DevArray<uint32_t> ca1(c);
        uint32_t *a = ca, *a_alt = ca1.p;
        uint64_t *k = ck, *k_alt = ck1.p;
        sort_pairs(a, a_alt, k, k_alt, c, 0, ab, stream);
        sort_pairs(k, k_alt, a, a_alt, c, 0, key_bits, stream);
        DevArray<uint64_t> rkeys(c);
        DevArray<uint32_t> rdups(c), rcount(1);
        auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RepeatFlag{k, a});
        size_t tmp_bytes = 0;
        SW_HIP(rocprim::reduce_by_key(nullptr, tmp_bytes, k, flags, c, rkeys.p, rdups.p, rcount.p, rocprim::plus<uint32_t>(),
                                      rocprim::equal_to<uint64_t>(), stream));
        DevArray<unsigned char> tmp(tmp_bytes);
        SW_HIP(rocprim::reduce_by_key(tmp.p, tmp_bytes, k, flags, c, rkeys.p, rdups.p, rcount.p, rocprim::plus<uint32_t>(),
                                      rocprim::equal_to<uint64_t>(), stream));
        uint32_t n_runs = 0;
        SW_HIP(hipMemcpyAsync(&n_runs, rcount.p, 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));
        if (n_runs) {
            hipLaunchKernelGGL(k_subtract_repeats, dim3(blocks_for(n_runs)), dim3(TPB), 0, stream, rkeys.p, rdups.p, n_runs, ukeys.p,
                               (uint64_t)ix.n_edges, ix.edges.p);
            SW_HIP(hipGetLastError());
        }
    }
    SW_HIP(hipStreamSynchronize(stream));
}

// keys[m] = (rank_lo << nb) | rank_hi (sentinels sort last), vals[m] = assembly; stable sort keeps equal
// pairs in assembly order, so weight = number of assembly changes inside a run (+1).
void edges_from_adjacency(uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, uint64_t m,
                          uint64_t sentinel, unsigned nb, const uint64_t *rank_hash, hipStream_t stream, sw_index &ix)
{
    ix.n_edges = 0;
    if (m == 0) return;
    sort_pairs(keys, keys_alt, vals, vals_alt, m, 0, 2 * nb, stream);
    // one run-length pass (see edges_from_packed): runs = equal pairs, a run's sum of (pair, assembly) changes = its weight
    DevArray<uint64_t> ukeys(m);
    DevArray<uint32_t> usum(m);
    DevArray<uint32_t> ucount(1);
    {
        auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), AsmChangeAny{keys, vals});
        size_t tmp_bytes = 0;
        SW_HIP(rocprim::reduce_by_key(nullptr, tmp_bytes, keys, flags, m, ukeys.p, usum.p, ucount.p, rocprim::plus<uint32_t>(),
                                      rocprim::equal_to<uint64_t>(), stream));
        DevArray<unsigned char> tmp(tmp_bytes);
        SW_HIP(rocprim::reduce_by_key(tmp.p, tmp_bytes, keys, flags, m, ukeys.p, usum.p, ucount.p, rocprim::plus<uint32_t>(),
                                      rocprim::equal_to<uint64_t>(), stream));
        hipLaunchKernelGGL(k_drop_sentinel_run, dim3(1), dim3(1), 0, stream, ukeys.p, sentinel, ucount.p);
        SW_HIP(hipGetLastError());
        uint32_t n_edges = 0;
        SW_HIP(hipMemcpyAsync(&n_edges, ucount.p, 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));   // (tmp is released here, after the pass has finished)
        ix.n_edges = n_edges;
    }
    if (ix.n_edges == 0) return;
    ix.edges.alloc(ix.n_edges);
    hipLaunchKernelGGL(k_edges_runs<false>, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ukeys.p, usum.p, 0u, ~0ull,
                       (uint64_t)ix.n_edges, nb, ix.nodes.p, rank_hash, ix.edges.p);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));
}
}  // namespace

// sw_graph_export's packed forms (api.hip): a node as {hash, stop - start}, the start of every `per`-th node, an edge with a 32-bit
// weight; *flag becomes non-zero if the forms cannot carry the arrays (see there).
__global__ void k_pack_export_nodes(const sw_node *__restrict__ nodes, uint64_t n, uint64_t n_kmers, uint64_t per, uint32_t *__restrict__ out,
                                    uint64_t *__restrict__ bases, uint32_t *__restrict__ flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const sw_node nd = nodes[i];
    const uint64_t cnt = nd.stop - nd.start, next = i + 1 < n ? nodes[i + 1].start : n_kmers;
    uint64_t pen;
    memcpy(&pen, &nd.penalty, 8);
    if (nd.stop < nd.start || cnt > 0xFFFFFFFFull || next != nd.stop || (i == 0 && nd.start != 0) || nd.n_tar || nd.n_neg || pen) atomicOr(flag, 1u);
    out[3 * i] = (uint32_t)nd.hash;
    out[3 * i + 1] = (uint32_t)(nd.hash >> 32);
    out[3 * i + 2] = (uint32_t)cnt;
    if (i % per == 0) bases[i / per] = nd.start;
}
__global__ void k_pack_export_edges(const sw_edge *__restrict__ edges, uint64_t n, uint32_t *__restrict__ out, uint32_t *__restrict__ flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const sw_edge e = edges[i];
    if (e.weight > 0xFFFFFFFFull) atomicOr(flag, 1u);
    out[5 * i] = (uint32_t)e.first;
    out[5 * i + 1] = (uint32_t)(e.first >> 32);
    out[5 * i + 2] = (uint32_t)e.second;
    out[5 * i + 3] = (uint32_t)(e.second >> 32);
    out[5 * i + 4] = (uint32_t)e.weight;
}
void pack_export(const sw_node *nodes, uint64_t n_nodes, uint64_t n_kmers, const sw_edge *edges, uint64_t n_edges, uint64_t per, uint32_t *pn,
                 uint64_t *bases, uint32_t *pe, uint32_t *flag, hipStream_t stream)
{
    hipLaunchKernelGGL(k_pack_export_nodes, dim3(blocks_for(n_nodes)), dim3(TPB), 0, stream, nodes, n_nodes, n_kmers, per, pn, bases, flag);
    SW_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_pack_export_edges, dim3(blocks_for(n_edges)), dim3(TPB), 0, stream, edges, n_edges, pe, flag);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));
}

void device_get_penalty(const sw_kmer *d_kmers, uint64_t n_kmers, sw_node *d_nodes, uint64_t n_nodes,
                        const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target,
                        uint64_t n_targets, uint64_t n_non_targets, hipStream_t stream, uint64_t *err_flags_host)
{
    PenaltyJob job;
    penalty_launch(d_kmers, n_kmers, d_nodes, n_nodes, d_rec_asm, n_records, d_is_target, n_targets, n_non_targets,
                   stream, job);
    *err_flags_host = penalty_finish(job);
}

// get_penalty on ONE slice of a multi-device graph: the slice's nodes hold global occurrence ranges [kmer_base + ...), its kmers
// are local -- the ranges are taken down to the slice, scored, and put back.
void slice_get_penalty(sw_index &ix, uint64_t kmer_base, const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target,
                       uint64_t n_targets, uint64_t n_non_targets, hipStream_t stream, uint64_t *err_flags_host)
{
    *err_flags_host = 0;
    if (!ix.n_nodes) return;
    hipLaunchKernelGGL(k_rebase_nodes, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, (uint64_t)0 - kmer_base);
    SW_HIP(hipGetLastError());
    struct Restore {   // the resident slice keeps GLOBAL ranges: they are put back on every way out (an exception between the
                       // two re-basings left the slice corrupted for later calls)
        sw_index &ix;
        uint64_t base;
        hipStream_t st;
        ~Restore()
        {
            hipLaunchKernelGGL(k_rebase_nodes, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, st, ix.nodes.p, ix.n_nodes, base);
            (void)hipGetLastError();
            (void)hipStreamSynchronize(st);
        }
    } restore{ix, kmer_base, stream};
    device_get_penalty(ix.kmers.p, ix.n_kmers, ix.nodes.p, ix.n_nodes, d_rec_asm, n_records, d_is_target, n_targets, n_non_targets,
                       stream, err_flags_host);
}

void build_index(const uint32_t *d_rec_asm, uint64_t n_records, uint64_t n_assemblies, OrderedOcc &occ,
                 const uint8_t *d_is_target, uint64_t n_targets, uint64_t n_non_targets, hipStream_t stream, sw_index &ix)
{
    const uint64_t n = occ.n;
    if (n > occ_cap()) raise_occ_cap(n, "minimizer occurrences");
    Event ev[6];
    SW_HIP(hipEventRecord(ev[0], stream));

    ix.n_kmers = n;
    ix.kmers.alloc(n);
    DevArray<uint32_t> rank(n);
    // The counts come from two first-of-assembly bitmaps written while the sorted occurrences stream by (k_nodes); the
    // occurrence order is this library's own stable sort, so the validation flags of filter.cpp:103-123 are only produced
    // under SEQWIN_AMD_CHECK_ORDER=1 (then the C-ABI get_penalty's flag / prefix-sum form runs instead).
    const bool want_counts = d_is_target != nullptr;
    const bool check_order = want_counts && SW_TEST_GETENV("SEQWIN_AMD_CHECK_ORDER") != nullptr;
    // The edges are counted without assemblies (keys-only pair sort + a side list of the records that can repeat a pair
    // inside one assembly, k_adj_pairs) unless SEQWIN_AMD_NO_PACKED_EDGES=1 asks for the (pair, assembly) sort that the
    // multi-GPU slices use, or the rank words have no spare bit (2^31 nodes or more).
    const bool by_table = occ.hash.p != nullptr;   // SEQWIN_AMD_RANKS=table (order_tuples then kept the hashes)
    const bool pair_edges = !SW_TEST_GETENV("SEQWIN_AMD_NO_PACKED_EDGES") && !by_table;
    bool rep_marked = false;
    DevArray<uint32_t> rec_flag;
    DevArray<unsigned long long> tbits, nbits;
    NodeParts parts;                // dense node hashes / first-occurrence positions from k_nodes: the edges read the hashes, finish_nodes both
    UnsortHold unsort_hold;         // the unsort's buckets, when the adjacency keys are written straight from them
    // -- nodes: stable radix sort of the occurrences by hash, run-length heads, ranks back in stream order ------
    if (n) {
        const bool bits = want_counts && !check_order;
        if (bits || pair_edges) {
            rec_flag.alloc(n_records);
            hipLaunchKernelGGL(k_rec_flag, dim3(blocks_for(n_records)), dim3(TPB), 0, stream, d_rec_asm, bits ? d_is_target : nullptr,
                               n_records, rec_flag.p);
            SW_HIP(hipGetLastError());
        }
        PaySort ps;   // the sort input written by k_order is consumed in place (staged: the sort reads the sketch stage itself)
        ps.key_a = std::move(occ.key32);
        ps.pay_a = std::move(occ.pay);
        if (occ.staged) ps.staged = &occ;
        group_occurrences(ps, n, 0, rec_flag.p, stream, ix, by_table ? nullptr : rank.p, bits ? &tbits : nullptr,
                          bits ? &nbits : nullptr, pair_edges ? &rep_marked : nullptr, pair_edges ? &unsort_hold : nullptr, &parts);
        if (!bits || by_table) {   // no bitmaps to wait for (or the table route reads the nodes next): complete the nodes here, counts zero
            finish_nodes(ix, parts, 0, n, nullptr, nullptr, 0.0, 0.0, stream);
            if (by_table) ranks_from_table(ix, occ.hash.p, n, stream, rank.p);
        }
    } else {
        ix.n_nodes = 0;
        ix.nodes.alloc(0);
    }
    SW_HIP(hipEventRecord(ev[1], stream));

    // -- per-node target / non-target assembly counts + penalty (filter.cpp:62-136), on a second stream
    //    so that it overlaps the edge stage (nodes fields are disjoint) -----
    PenaltyJob pen;
    hipStream_t side = side_stream();
    bool forked = false;
    if (want_counts && ix.n_nodes) {
        alloc_fork(side);   // blocks released from here on are fenced against the counts stream as well
        forked = true;
        SW_HIP(hipStreamWaitEvent(side, ev[1], 0));
        SW_HIP(hipEventRecord(ev[4], side));
        if (check_order) {
            penalty_launch(ix.kmers.p, n, ix.nodes.p, ix.n_nodes, d_rec_asm, n_records, d_is_target, n_targets, n_non_targets,
                           side, pen);
        } else {
            // (with SEQWIN_AMD_RANKS=table the nodes were completed above with zero counts: this pass writes them again, whole)
            finish_nodes(ix, parts, 0, n, tbits.p, nbits.p, 1.0 / (double)n_targets, 1.0 / (double)n_non_targets, side);   // filter.cpp:89-90
        }
        SW_HIP(hipEventRecord(ev[5], side));
    }
    SW_HIP(hipEventRecord(ev[2], stream));

    // -- edges -----------------------------------------------------------------------------------------
    ix.n_edges = 0;
    if (n >= 2) {
        const uint64_t m = n - 1;
        unsigned nb = 1;
        while (((1ull << nb) - 1ull) < ix.n_nodes) ++nb;  // n_nodes <= 2^nb - 1, so (2^nb-1, 2^nb-1) is free
        unsigned ab = 1;
        while ((1ull << ab) < n_assemblies) ++ab;            // assembly index < 2^ab
        const unsigned adj_blocks = (unsigned)((n + 1023) / 1024);   // 4 occurrences per thread
        const uint64_t sentinel = (nb == 32) ? ~0ull : ((1ull << (2 * nb)) - 1ull);
        DevArray<uint64_t> k0(m), k1(m);
        if (rep_marked) {
            // weight = records of the pair - records that repeat it inside one assembly
            DevArray<uint64_t> ck(m);
            DevArray<uint32_t> ca(m);
            DevArray<unsigned long long> n_cand(1);
            SW_HIP(hipMemsetAsync(n_cand.p, 0, 8, stream));
            // when radix.hip sorts the keys, k_adj_pairs counts their digits on the way (no counting sweep over the keys)
            DevArray<unsigned long long> ehist;
            unsigned hbits = 0, hpasses = 0;
            uint32_t iters = 1;
            if (sort_keys64_is_own(m) && !SW_AB_GETENV("SEQWIN_AMD_NO_ADJ_HIST")) {
                radix_layout(2 * nb, &hbits, &hpasses);
                ehist.alloc((size_t)hpasses << hbits);
                SW_HIP(hipMemsetAsync(ehist.p, 0, ehist.bytes(), stream));
                iters = 32;
            }
            if (unsort_hold.sorted) {
                const uint32_t n_buckets = (uint32_t)((n + UNSORT_RANGE - 1) / UNSORT_RANGE);
                const uint32_t per_wg = 4;   // buckets per workgroup between two flushes of its digit counts (1 / 2 / 4: 5.85 / 5.74 / 5.69 ms)
                DevArray<uint32_t> edge_rank(2 * (size_t)n_buckets);
                hipLaunchKernelGGL(k_unsort_adj<RecArray>, dim3((n_buckets + per_wg - 1) / per_wg), dim3(1024),
                                   ehist.p ? ((size_t)hpasses << hbits) * 4 : 0, stream, unsort_hold.sorted,
                                   n, RecArray{occ.rec.p}, d_rec_asm, 0u, nb, sentinel, k0.p, ck.p, ca.p, n_cand.p,
                                   ehist.p, hbits, hpasses, 2 * nb, edge_rank.p, per_wg);
                hipLaunchKernelGGL(k_adj_bounds<RecArray>, dim3((n_buckets + 255) / 256), dim3(256), 0, stream, edge_rank.p, n_buckets,
                                   RecArray{occ.rec.p}, d_rec_asm, 0u, nb, sentinel, k0.p, ck.p, ca.p, n_cand.p, ehist.p, hbits, hpasses,
                                   2 * nb);
                unsort_hold.a.release();   // (stream-ordered pool, as edge_rank at the end of this block)
                unsort_hold.b.release();
            } else {
                hipLaunchKernelGGL(k_adj_pairs<RecArray>, dim3((adj_blocks + iters - 1) / iters), dim3(256), 0, stream, RecArray{occ.rec.p},
                                   rank.p, d_rec_asm, 0u, n, nb, sentinel, k0.p, ck.p, ca.p, n_cand.p, ehist.p, hbits, hpasses, 2 * nb, iters);
            }
            SW_HIP(hipGetLastError());
            edges_from_pairs(k0.p, k1.p, m, sentinel, nb, ab, ck.p, ca.p, n_cand.p, 0, parts.hash.p, stream, ix, (hipEvent_t) nullptr,
                             ehist.p);   // (the dense hashes are k_nodes' own, on this stream: nothing to wait for)
        } else {
            DevArray<uint32_t> v0(m), v1(m);
            hipLaunchKernelGGL(k_adj_keys, dim3(adj_blocks), dim3(256), 0, stream, occ.rec.p, rank.p, d_rec_asm, n, nb,
                               sentinel, k0.p, v0.p);
            SW_HIP(hipGetLastError());
            edges_from_adjacency(k0.p, k1.p, v0.p, v1.p, m, sentinel, nb, parts.hash.p, stream, ix);
        }
    }
    if (ix.n_edges == 0) ix.edges.alloc(0);
    if (forked) {
        const uint64_t err = pen.active ? penalty_finish(pen) : 0;
        SW_HIP(hipStreamWaitEvent(stream, ev[5], 0));   // the build is complete on `stream` only after the counts
        alloc_join(side);
        if (err) raise(SW_ERR_RUNTIME, "internal error: inconsistent occurrence order in device index (%llu)",
                       (unsigned long long)err);
    }
    SW_HIP(hipEventRecord(ev[3], stream));
    SW_HIP(hipEventSynchronize(ev[3]));   // (the bitmaps and the rank array are released after this)
    float ms = 0.f;
    SW_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    ix.timings.nodes_ms = ms;
    ix.timings.counts_ms = 0;
    if (forked) {
        SW_HIP(hipEventElapsedTime(&ms, ev[4], ev[5]));
        ix.timings.counts_ms = ms;                       // overlaps edges_ms
    }
    SW_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
    ix.timings.edges_ms = ms;
}

void index_occ_rows(const sw_index &ix, uint64_t rec_offset, uint64_t *d_rows, hipStream_t stream)
{
    if (ix.n_kmers == 0) return;
    DevArray<uint32_t> start(ix.n_nodes);
    hipLaunchKernelGGL(k_node_starts, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, start.p);
    hipLaunchKernelGGL(k_occ_rows, dim3(blocks_for(ix.n_kmers)), dim3(TPB), 0, stream, ix.kmers.p, ix.nodes.p, start.p,
                       ix.n_nodes, ix.n_kmers, rec_offset, d_rows);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));   // `start` is released on return
}

void index_splits(const sw_index &ix, const uint64_t *node_bounds, const uint64_t *edge_bounds, uint32_t n_bounds,
                  uint64_t *occ_split, uint64_t *edge_split, hipStream_t stream)
{
    if (n_bounds == 0) return;
    DevArray<uint64_t> nb(n_bounds), eb(n_bounds), os(n_bounds), es(n_bounds);
    SW_HIP(hipMemcpyAsync(nb.p, node_bounds, n_bounds * 8, hipMemcpyHostToDevice, stream));
    SW_HIP(hipMemcpyAsync(eb.p, edge_bounds, n_bounds * 8, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_lower_bounds, dim3(blocks_for(n_bounds)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, ix.n_kmers,
                       ix.edges.p, ix.n_edges, nb.p, eb.p, n_bounds, os.p, es.p);
    SW_HIP(hipGetLastError());
    SW_HIP(hipMemcpyAsync(occ_split, os.p, n_bounds * 8, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipMemcpyAsync(edge_split, es.p, n_bounds * 8, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
}

// Union of partial graphs (the GPU counterpart of merge_thread_graphs, build_internals.cpp:295-392):
// occurrence rows arrive concatenated in source-rank order (each source sorted by (hash, record, pos)
// with globally rebased record indices), edge rows as (first, second, partial weight).
void merge_build(const uint64_t *d_occ_rows, uint64_t n, const uint64_t *d_edge_rows, uint64_t m, uint64_t kmer_base,
                 const uint32_t *d_rec_asm, uint64_t n_records, const uint8_t *d_is_target, uint64_t n_targets,
                 uint64_t n_non_targets, hipStream_t stream, sw_index &ix, uint32_t *d_rank_out)
{
    if (n > occ_cap() || m > occ_cap()) raise_occ_cap(std::max<uint64_t>(n, m), "rows");
    ix.n_kmers = n;
    ix.kmers.alloc(n);
    ix.n_nodes = 0;
    ix.n_edges = 0;
    // Slice build of the tuple-exchange form (ranks wanted, no partial edges): the rows are this library's own tuples,
    // concatenated in source-rank = (record, pos) order, so the counts come from k_nodes' bitmaps as in the single-GPU build
    // (no flag arrays, no prefix sums, nothing left running) and the ranks carry the repeat marks.  SEQWIN_AMD_CHECK_ORDER=1
    // keeps the validating form below.
    const bool slice = d_rank_out && m == 0 && d_rec_asm && n_records && !SW_TEST_GETENV("SEQWIN_AMD_CHECK_ORDER");
    if (n) {
        PaySort ps;
        const char *order = SW_TEST_GETENV("SEQWIN_AMD_ORDER");   // ("copy": A/B, as for the sketch stage of the single-GPU build)
        if (sort_pairs_is_own(n, 32) && !SW_TEST_GETENV("SEQWIN_AMD_SORT_KEYBITS") && !(order && !strcmp(order, "copy"))) {
            ps.rows_src = d_occ_rows;   // the sort's first pass reads the rows
        } else {
            ps.key_a.alloc(n);
            ps.pay_a.alloc(n);
            hipLaunchKernelGGL(k_rows_to_pay, dim3(blocks_for(n)), dim3(TPB), 0, stream, d_occ_rows, n, ps.key_a.p, ps.pay_a.p);
            SW_HIP(hipGetLastError());
        }
        // stable: ties keep source-rank order; d_rank_out[j] = node rank of received row j -- with RANK_REP in bit 31 where the
        // node recurs in the occurrence's assembly (record table given, fewer than 2^31 nodes)
        DevArray<uint32_t> rec_flag;
        DevArray<unsigned long long> tbits, nbits;
        bool marked = false;
        if (d_rank_out && d_rec_asm && n_records) {
            rec_flag.alloc(n_records);
            hipLaunchKernelGGL(k_rec_flag, dim3(blocks_for(n_records)), dim3(TPB), 0, stream, d_rec_asm,
                               slice ? d_is_target : (const uint8_t *)nullptr, n_records, rec_flag.p);
            SW_HIP(hipGetLastError());
        }
        const bool bits = slice && d_is_target;
        NodeParts parts;
        const uint64_t base = slice ? kmer_base : 0;
        group_occurrences(ps, n, base, rec_flag.p, stream, ix, d_rank_out, bits ? &tbits : nullptr,
                          bits ? &nbits : nullptr, rec_flag.p ? &marked : nullptr, nullptr, &parts);
        ix.ranks_marked = marked;
        if (bits) finish_nodes(ix, parts, base, base + n, tbits.p, nbits.p, 1.0 / (double)n_targets, 1.0 / (double)n_non_targets, stream);   // filter.cpp:89-90
        else finish_nodes(ix, parts, base, base + n, nullptr, nullptr, 0.0, 0.0, stream);
        SW_HIP(hipStreamSynchronize(stream));   // (the bitmaps and the parts are released here)
    } else {
        ix.nodes.alloc(0);
        ix.ranks_marked = d_rec_asm && n_records;   // an empty slice returns no rank at all: it must not switch the job's pairs form off
    }
    if (slice) {
        ix.edges.alloc(0);
        return;
    }
    // counts on slice-local ranges, on the side stream so that they overlap the partial-edge merge below;
    // ranges are re-based after both are done
    PenaltyJob pen;
    Event ev_nodes(false), ev_pen(false);
    hipStream_t side = nullptr;
    if (d_is_target && ix.n_nodes) {
        side = side_stream();
        alloc_fork(side);
        SW_HIP(hipEventRecord(ev_nodes, stream));
        SW_HIP(hipStreamWaitEvent(side, ev_nodes, 0));
        penalty_launch(ix.kmers.p, n, ix.nodes.p, ix.n_nodes, d_rec_asm, n_records, d_is_target, n_targets, n_non_targets,
                       side, pen);
        SW_HIP(hipEventRecord(ev_pen, side));
    }
    ix.n_edges = 0;
    if (m) {
        DevArray<uint64_t> k0(m), k1(m), f(m), sd(m), wt(m);
        DevArray<uint32_t> v0(m), v1(m);
        hipLaunchKernelGGL(k_strided_copy, dim3(blocks_for(m)), dim3(TPB), 0, stream, d_edge_rows, 3u, 1u, m, k0.p);
        hipLaunchKernelGGL(k_iota, dim3(blocks_for(m)), dim3(TPB), 0, stream, v0.p, m);
        uint64_t *keys = k0.p, *keys_alt = k1.p;
        uint32_t *vals = v0.p, *vals_alt = v1.p;
        sort_pairs(keys, keys_alt, vals, vals_alt, m, 0, 64, stream);   // LSD: second first ...
        hipLaunchKernelGGL(k_gather_col, dim3(blocks_for(m)), dim3(TPB), 0, stream, d_edge_rows, 3u, 0u, vals, m, keys);
        sort_pairs(keys, keys_alt, vals, vals_alt, m, 0, 64, stream);   // ... then first (build_internals.cpp:261)
        hipLaunchKernelGGL(k_gather_col, dim3(blocks_for(m)), dim3(TPB), 0, stream, d_edge_rows, 3u, 0u, vals, m, f.p);
        hipLaunchKernelGGL(k_gather_col, dim3(blocks_for(m)), dim3(TPB), 0, stream, d_edge_rows, 3u, 1u, vals, m, sd.p);
        hipLaunchKernelGGL(k_gather_col, dim3(blocks_for(m)), dim3(TPB), 0, stream, d_edge_rows, 3u, 2u, vals, m, wt.p);
        SW_HIP(hipGetLastError());
        DevArray<uint32_t> ecum(m);
        inclusive_sum(rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), PairHeadFlag{f.p, sd.p}),
                      ecum.p, m, (uint32_t)0, stream);
        inclusive_sum(wt.p, wt.p, m, (uint64_t)0, stream);
        uint32_t n_edges = 0;
        SW_HIP(hipMemcpyAsync(&n_edges, ecum.p + (m - 1), 4, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));
        ix.n_edges = n_edges;
        ix.edges.alloc(n_edges);
        DevArray<uint64_t> edge_start(n_edges);
        hipLaunchKernelGGL(k_merge_edge_heads, dim3(blocks_for(m)), dim3(TPB), 0, stream, f.p, sd.p, ecum.p, m, edge_start.p);
        hipLaunchKernelGGL(k_merge_edges, dim3(blocks_for(n_edges)), dim3(TPB), 0, stream, f.p, sd.p, wt.p, edge_start.p,
                           (uint64_t)n_edges, m, ix.edges.p);
        SW_HIP(hipGetLastError());
        SW_HIP(hipStreamSynchronize(stream));
    } else {
        ix.edges.alloc(0);
    }
    if (pen.active) {
        const uint64_t err = penalty_finish(pen);
        SW_HIP(hipStreamWaitEvent(stream, ev_pen, 0));
        alloc_join(side);
        if (err) raise(SW_ERR_RUNTIME, "internal error: inconsistent occurrence order in merged index (%llu)",
                       (unsigned long long)err);
    }
    if (kmer_base && ix.n_nodes) {
        hipLaunchKernelGGL(k_rebase_nodes, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes,
                           kmer_base);
        SW_HIP(hipGetLastError());
        SW_HIP(hipStreamSynchronize(stream));
    }
}

// ---- stable multi-way partition (tuple exchange) ----------------------------------------------------
// One wave owns PART_ROWS x 64 consecutive elements and walks them row by row; inside a row the lanes
// with the same owner are found with 5 ballots (owner < 32), so the position of every element inside its
// (wave, owner) group is exact and the partition is stable.  Pass A counts, a scan turns the
// (owner-major, wave-minor) counts into offsets, pass B recomputes the owners and writes the rows.
namespace {
constexpr int PART_ROWS = 32;
constexpr uint32_t PART_BUCKETS = 32;

struct PartArgs;
__device__ __forceinline__ uint32_t owner_of(const PartArgs &P, uint64_t kx);
struct TupleSrc {   // rows from the ordered tuple stream
    const uint64_t *hash, *kmer;
    uint64_t rec_off;
    __device__ uint32_t owner(const PartArgs &P, uint64_t i) const { return owner_of(P, hash[i]); }
    __device__ void row(uint64_t i, uint64_t &r0, uint64_t &r1) const { r0 = hash[i]; r1 = kmer[i] + (rec_off << 32); }
};
struct RowSrc {     // rows[n][2], key = column 0
    const uint64_t *rows;
    __device__ uint32_t owner(const PartArgs &P, uint64_t i) const { return owner_of(P, rows[2 * i]); }
    __device__ void row(uint64_t i, uint64_t &r0, uint64_t &r1) const { r0 = rows[2 * i]; r1 = rows[2 * i + 1]; }
};

struct KeySrc {     // rows[n] of packed keys (one column); the second output column is unused
    const uint64_t *rows;
    __device__ uint32_t owner(const PartArgs &P, uint64_t i) const { return owner_of(P, rows[i]); }
    __device__ void row(uint64_t i, uint64_t &r0, uint64_t &r1) const { r0 = rows[i]; r1 = 0; }
};

struct PartArgs {
    uint64_t n;
    uint64_t bounds[16];
    uint32_t n_bounds;
    uint64_t drop_key;
    uint32_t has_drop;
    uint32_t n_waves;
    uint32_t cols;       // output columns per row: 2, or 1 for packed keys
};

}  // namespace
struct PartState {   // what a tuple partition leaves behind for the way back (OrderedOcc::part)
    PartArgs args;
    DevArray<uint32_t> offs;
    uint64_t rec_off = 0;
    bool valid = false;
};
void part_state_delete(PartState *p) { delete p; }
uint32_t occ_partition_owners(const OrderedOcc &occ) { return occ.part && occ.part->valid ? occ.part->args.n_bounds + 1 : 0u; }
namespace {

// keys whose owner does not follow from their value (the edge keys of the multi-GPU adjacency hold rank_lo relative to their
// owner's range): the producer wrote the owner of every key next to it (OWNER_DROP: not a row, lands behind the last owner)
struct OwnedKeySrc {
    const uint64_t *rows;
    const uint8_t *own;
    uint32_t n_owners;
    __device__ uint32_t owner(const PartArgs &, uint64_t i) const { const uint8_t o = own[i]; return o == OWNER_DROP ? n_owners : o; }
    __device__ void row(uint64_t i, uint64_t &r0, uint64_t &r1) const { r0 = rows[i]; r1 = 0; }
};
struct OwnedPairSrc {   // candidate rows {key, assembly} with their owner
    const uint64_t *key_;
    const uint32_t *asm_;
    const uint8_t *own;
    __device__ uint32_t owner(const PartArgs &, uint64_t i) const { return own[i]; }
    __device__ void row(uint64_t i, uint64_t &r0, uint64_t &r1) const { r0 = key_[i]; r1 = asm_[i]; }
};

__device__ __forceinline__ uint32_t owner_of(const PartArgs &P, uint64_t kx)
{
    uint32_t o = 0;
    for (uint32_t j = 0; j < P.n_bounds; ++j) o += (P.bounds[j] <= kx) ? 1u : 0u;
    if (P.has_drop && kx == P.drop_key) o = P.n_bounds + 1;
    return o;
}

// mask of lanes (among `active`) whose 5-bit value equals this lane's
__device__ __forceinline__ uint64_t match5(uint32_t v, uint64_t active)
{
    uint64_t m = active;
    for (int b = 0; b < 5; ++b) {
        const uint64_t bal = __ballot((v >> b) & 1u);
        m &= ((v >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

// MODE 0 counts, 1 writes the rows (and perm), 2 walks the same positions again and brings a 32-bit value per partitioned row
// back to the original order: gather_out[i] = gather_in[dst(i)] -- one sequential read cursor per owner instead of a
// random scatter through perm (26 ms for 745 M rows).
template <class Src, int MODE>
__global__ __launch_bounds__(256) void k_partition(const Src src, const PartArgs P, uint32_t *__restrict__ hist,
                                                   const uint32_t *__restrict__ offsets, uint64_t *__restrict__ rows_out,
                                                   uint32_t *__restrict__ perm_out, const uint32_t *__restrict__ gather_in = nullptr,
                                                   uint32_t *__restrict__ gather_out = nullptr, uint8_t *__restrict__ owner_out = nullptr)
{
    constexpr bool WRITE = MODE != 0;
    __shared__ uint32_t cnt[4][PART_BUCKETS];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + wv;
    if (lane < PART_BUCKETS) cnt[wv][lane] = (WRITE && wave < P.n_waves) ? offsets[(uint64_t)lane * P.n_waves + wave] : 0u;
    __syncthreads();
    if (wave >= P.n_waves) return;
    const uint64_t base = wave * (uint64_t)(PART_ROWS * 64);
    for (int r = 0; r < PART_ROWS; ++r) {
        const uint64_t i = base + (uint64_t)r * 64 + lane;
        const bool live = i < P.n;
        const uint64_t active = __ballot(live);
        if (!active) break;
        uint32_t o = 0;
        if (live) o = src.owner(P, i);
        const uint64_t same = match5(o, active);
        if (live) {
            const uint32_t before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            const uint32_t start = cnt[wv][o];               // all lanes of a group read before the leader adds
            if (MODE == 2) {
                gather_out[i] = gather_in[(uint64_t)start + before];
                if (owner_out) owner_out[i] = (uint8_t)o;
            } else if (MODE == 1) {
                const uint64_t dst = (uint64_t)start + before;
                uint64_t r0, r1;
                src.row(i, r0, r1);
                if (P.cols == 2) {
                    rows_out[2 * dst] = r0;
                    rows_out[2 * dst + 1] = r1;
                } else {
                    rows_out[dst] = r0;
                }
                if (perm_out) perm_out[dst] = (uint32_t)i;
            }
            __builtin_amdgcn_wave_barrier();
            if (before == 0) cnt[wv][o] = start + (uint32_t)__popcll(same);   // leader of the group
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!WRITE && lane < PART_BUCKETS) hist[(uint64_t)lane * P.n_waves + wave] = cnt[wv][lane];
}

__global__ void k_part_counts(const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ hist, uint32_t n_waves,
                              unsigned long long *__restrict__ counts)
{
    const uint32_t o = threadIdx.x;
    if (o >= PART_BUCKETS) return;
    const uint64_t last = (uint64_t)o * n_waves + (n_waves - 1);
    const uint64_t first = (uint64_t)o * n_waves;
    counts[o] = (unsigned long long)(offsets[last] + hist[last]) - offsets[first];
}

template <class Src>
void stable_partition(const Src &src, uint64_t n, const uint64_t *bounds, uint32_t n_bounds, bool has_drop, uint64_t drop_key,
                      uint64_t *d_rows_out, uint32_t *d_perm_out, uint64_t *counts_host, hipStream_t stream, uint32_t cols = 2,
                      PartState *keep = nullptr)
{
    if (n_bounds > 15) raise(SW_ERR_VALUE, "at most 16 owners are supported");
    for (uint32_t j = 0; j < n_bounds + 2; ++j) counts_host[j] = 0;
    if (n > occ_cap()) raise_occ_cap(n, "rows");
    PartArgs P{};
    P.n = n;
    P.n_bounds = n_bounds;
    for (uint32_t j = 0; j < n_bounds && bounds; ++j) P.bounds[j] = bounds[j];   // (null: the source knows its rows' owners itself)
    P.drop_key = drop_key;
    P.has_drop = has_drop ? 1u : 0u;
    P.cols = cols;
    P.n_waves = (uint32_t)((n + PART_ROWS * 64 - 1) / (PART_ROWS * 64));
    if (n == 0) {
        if (keep) {   // an empty shard is partitioned too: nothing to walk on the way back, but the owners are known
            keep->args = P;
            keep->offs.alloc(0);
            keep->valid = true;
        }
        return;
    }
    const uint64_t nh = (uint64_t)PART_BUCKETS * P.n_waves;
    DevArray<uint32_t> hist(nh), offs(nh);
    DevArray<unsigned long long> counts(PART_BUCKETS);
    const unsigned blocks = (P.n_waves + 3) / 4;
    hipLaunchKernelGGL((k_partition<Src, 0>), dim3(blocks), dim3(256), 0, stream, src, P, hist.p, (const uint32_t *)nullptr,
                       (uint64_t *)nullptr, (uint32_t *)nullptr);
    SW_HIP(hipGetLastError());
    exclusive_sum(hist.p, offs.p, nh, (uint32_t)0, stream);
    hipLaunchKernelGGL((k_partition<Src, 1>), dim3(blocks), dim3(256), 0, stream, src, P, (uint32_t *)nullptr, offs.p,
                       d_rows_out, d_perm_out);
    hipLaunchKernelGGL(k_part_counts, dim3(1), dim3(64), 0, stream, offs.p, hist.p, P.n_waves, counts.p);
    SW_HIP(hipGetLastError());
    unsigned long long h[PART_BUCKETS];
    SW_HIP(hipMemcpyAsync(h, counts.p, sizeof h, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    for (uint32_t j = 0; j < n_bounds + 2; ++j) counts_host[j] = h[j];
    if (keep) {   // the way back (MODE 2) walks the same offsets
        keep->args = P;
        keep->offs = std::move(offs);
        keep->valid = true;
    }
}
}  // namespace

void occ_partition(const OrderedOcc &occ, const uint64_t *bounds, uint32_t n_bounds, uint64_t rec_offset, uint64_t *d_rows,
                   uint32_t *d_perm, uint64_t *counts_host, hipStream_t stream)
{
    uint64_t counts[18];
    OrderedOcc &o = const_cast<OrderedOcc &>(occ);
    if (!o.part) o.part = new PartState;
    o.part->valid = false;
    o.part->rec_off = rec_offset;
    stable_partition(TupleSrc{occ.hash.p, occ.kmer.p, rec_offset}, occ.n, bounds, n_bounds, false, 0, d_rows, d_perm, counts,
                     stream, 2, o.part);
    for (uint32_t j = 0; j <= n_bounds; ++j) counts_host[j] = counts[j];
}

// ab == 0: rows are {key, assembly} pairs; ab > 0: one packed 64-bit key per row, (key << ab) | assembly
void occ_adjacency(const OrderedOcc &occ, const uint32_t *d_rec_asm, const uint32_t *d_perm, const uint32_t *d_rank_by_row,
                   unsigned nb, unsigned ab, uint64_t asm_base, const uint64_t *rank_bounds, uint32_t n_bounds,
                   uint64_t *d_rows_out, uint64_t *counts_host, hipStream_t stream)
{
    for (uint32_t j = 0; j <= n_bounds; ++j) counts_host[j] = 0;
    const uint64_t n = occ.n;
    if (n < 2) return;
    const uint64_t m = n - 1;
    const unsigned tb = 2 * nb + ab;
    const uint64_t sentinel = (tb >= 64) ? ~0ull : ((1ull << tb) - 1ull);
    DevArray<uint32_t> rank(n);
    DevArray<uint64_t> rows(ab ? m : 2 * m);
    if (occ.part && occ.part->valid) {
        // ranks arrive in partitioned-row order; the partition was stable, so walking it again reads them back in order
        const PartArgs &P = occ.part->args;
        hipLaunchKernelGGL((k_partition<TupleSrc, 2>), dim3((P.n_waves + 3) / 4), dim3(256), 0, stream,
                           TupleSrc{occ.hash.p, occ.kmer.p, occ.part->rec_off}, P, (uint32_t *)nullptr, occ.part->offs.p,
                           (uint64_t *)nullptr, (uint32_t *)nullptr, d_rank_by_row, rank.p);
    } else {
        if (!d_perm) raise(SW_ERR_VALUE, "sw_occ_adjacency needs perm when the tuples were not partitioned by sw_occ_partition");
        hipLaunchKernelGGL(k_unpermute, dim3(blocks_for(n)), dim3(TPB), 0, stream, d_perm, d_rank_by_row, n, rank.p);
    }
    if (ab)
        hipLaunchKernelGGL(k_adj_rows_packed, dim3(blocks_for(m)), dim3(TPB), 0, stream, occ.kmer.p, rank.p, d_rec_asm, n, nb, ab,
                           sentinel, asm_base, rows.p);
    else
        hipLaunchKernelGGL(k_adj_rows, dim3(blocks_for(m)), dim3(TPB), 0, stream, occ.kmer.p, rank.p, d_rec_asm, n, nb, sentinel,
                           asm_base, rows.p);
    SW_HIP(hipGetLastError());
    std::vector<uint64_t> kb(n_bounds);
    for (uint32_t j = 0; j < n_bounds; ++j) kb[j] = rank_bounds[j] << (nb + ab);   // key is monotone in rank_lo
    uint64_t counts[18];
    // dropped (sentinel) rows land after the last owner; the output buffer holds all m rows
    if (ab)
        stable_partition(KeySrc{rows.p}, m, kb.data(), n_bounds, true, sentinel, d_rows_out, (uint32_t *)nullptr, counts, stream, 1);
    else
        stable_partition(RowSrc{rows.p}, m, kb.data(), n_bounds, true, sentinel, d_rows_out, (uint32_t *)nullptr, counts, stream);
    for (uint32_t j = 0; j <= n_bounds; ++j) counts_host[j] = counts[j];
}

void slice_edges(sw_index &ix, const uint64_t *d_adj_rows, uint64_t m, unsigned nb, unsigned ab, const uint64_t *d_rank_hash,
                 hipStream_t stream)
{
    ix.n_edges = 0;
    if (m > occ_cap()) raise_occ_cap(m, "adjacency rows");
    if (m && ab) {
        DevArray<uint64_t> k0(m), k1(m);
        SW_HIP(hipMemcpyAsync(k0.p, d_adj_rows, m * 8, hipMemcpyDeviceToDevice, stream));
        edges_from_packed(k0.p, k1.p, m, ~0ull, nb, ab, d_rank_hash, stream, ix);   // rows carry no sentinels
    } else if (m) {
        DevArray<uint64_t> k0(m), k1(m);
        DevArray<uint32_t> v0(m), v1(m);
        hipLaunchKernelGGL(k_split_rows2, dim3(blocks_for(m)), dim3(TPB), 0, stream, d_adj_rows, m, k0.p, v0.p);
        SW_HIP(hipGetLastError());
        edges_from_adjacency(k0.p, k1.p, v0.p, v1.p, m, ~0ull, nb, d_rank_hash, stream, ix);
    }
    if (ix.n_edges == 0) ix.edges.alloc(0);
}

namespace {
struct PairSrc {   // candidate rows as two arrays: key, assembly
    const uint64_t *key_;
    const uint32_t *asm_;
    __device__ uint32_t owner(const PartArgs &P, uint64_t i) const { return owner_of(P, key_[i]); }
    __device__ void row(uint64_t i, uint64_t &r0, uint64_t &r1) const { r0 = key_[i]; r1 = asm_[i]; }
};
}  // namespace

// Pairs form of the adjacency exchange: d_keys_out[<= n - 1] = one key per adjacency record, grouped by edge owner; the
// candidates (records that may repeat a pair inside one assembly) as {key, global assembly} rows, grouped by owner, stay
// in occ.cand_rows.  d_rank_by_row: slice-LOCAL ranks with RANK_REP, in the order of the partitioned rows; node_base[n_owners
// + 1]: prefix of the slice owners' node counts (n_owners = owners of the tuple partition).  counts_host / cand_counts_host
// [n_bounds + 1]; key_bits_host[2] = {lo_bits, hi_bits} of the keys (see RankSpace).
void occ_adjacency_pairs(OrderedOcc &occ, const uint32_t *d_rec_asm, const uint32_t *d_rank_by_row, const uint64_t *node_base,
                         uint64_t asm_base, const uint64_t *rank_bounds, uint32_t n_bounds, uint64_t *d_keys_out,
                         uint64_t *counts_host, uint64_t *cand_counts_host, uint64_t *key_bits_host, hipStream_t stream)
{
    for (uint32_t j = 0; j <= n_bounds; ++j) counts_host[j] = cand_counts_host[j] = 0;
    occ.cand_rows.alloc(0);
    if (!occ.part || !occ.part->valid) raise(SW_ERR_VALUE, "sw_occ_adjacency_pairs needs the tuples partitioned by sw_occ_partition");
    const PartArgs &P = occ.part->args;
    if (n_bounds > 15) raise(SW_ERR_VALUE, "at most 16 owners are supported");
    RankSpace S{};
    S.n_owners = P.n_bounds + 1;
    S.n_edge_owners = n_bounds + 1;
    for (uint32_t q = 0; q <= S.n_owners; ++q) S.node_base[q] = node_base[q];
    const uint64_t total = node_base[S.n_owners];
    uint64_t widest = 0;
    for (uint32_t q = 0; q <= n_bounds; ++q) {
        S.lo_base[q] = q ? rank_bounds[q - 1] : 0;
        const uint64_t end = q < n_bounds ? rank_bounds[q] : total;
        if (end < S.lo_base[q] || end > total) raise(SW_ERR_VALUE, "rank bounds must ascend and stay within the number of nodes");
        widest = std::max(widest, end - S.lo_base[q]);
    }
    S.lo_base[n_bounds + 1] = total;
    unsigned hi_bits = 1, lo_bits = 1;
    while (hi_bits < 64 && (total >> hi_bits)) ++hi_bits;      // rank_hi < total < 2^hi_bits
    while (lo_bits < 64 && (widest >> lo_bits)) ++lo_bits;
    if (lo_bits + hi_bits > 64)
        raise(SW_ERR_RUNTIME, "%llu nodes over %u edge owners do not fit 64-bit edge keys (%u + %u bits): use more GPUs",
              (unsigned long long)total, n_bounds + 1, lo_bits, hi_bits);
    S.hi_bits = hi_bits;
    key_bits_host[0] = lo_bits;
    key_bits_host[1] = hi_bits;
    const uint64_t n = occ.n;
    if (n < 2) return;
    const uint64_t m = n - 1;
    DevArray<uint32_t> rank(n), ca(m);
    DevArray<uint8_t> own(n), kown(m), cown(m);
    DevArray<uint64_t> keys(m), ck(m);
    DevArray<unsigned long long> n_cand(1);
    SW_HIP(hipMemsetAsync(n_cand.p, 0, 8, stream));
    // the ranks arrive in partitioned-row order; the partition was stable, so walking it again reads them back in stream
    // order, together with the owner every tuple went to
    hipLaunchKernelGGL((k_partition<TupleSrc, 2>), dim3((P.n_waves + 3) / 4), dim3(256), 0, stream,
                       TupleSrc{occ.hash.p, occ.kmer.p, occ.part->rec_off}, P, (uint32_t *)nullptr, occ.part->offs.p,
                       (uint64_t *)nullptr, (uint32_t *)nullptr, d_rank_by_row, rank.p, own.p);
    hipLaunchKernelGGL(k_adj_pairs_dist<RecOfKmer>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, RecOfKmer{occ.kmer.p},
                       rank.p, own.p, d_rec_asm, (uint32_t)asm_base, n, S, keys.p, kown.p, ck.p, ca.p, cown.p, n_cand.p);
    SW_HIP(hipGetLastError());
    uint64_t counts[18];
    // dropped rows (record boundaries) land after the last owner; the output buffer holds all m keys
    stable_partition(OwnedKeySrc{keys.p, kown.p, n_bounds + 1}, m, nullptr, n_bounds, true, 0, d_keys_out, (uint32_t *)nullptr, counts,
                     stream, 1);
    for (uint32_t j = 0; j <= n_bounds; ++j) counts_host[j] = counts[j];
    unsigned long long c = 0;
    SW_HIP(hipMemcpyAsync(&c, n_cand.p, 8, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    if (c) {
        occ.cand_rows.alloc(2 * c);
        stable_partition(OwnedPairSrc{ck.p, ca.p, cown.p}, c, nullptr, n_bounds, false, 0, occ.cand_rows.p, (uint32_t *)nullptr, counts,
                         stream);
        for (uint32_t j = 0; j <= n_bounds; ++j) cand_counts_host[j] = counts[j];
    }
}

// Owner: edges of its rank range from the received keys and candidate rows.  rank_hash: the job-wide table, owner o's node
// hashes at d_rank_hash[o * pad ...] (pad >= every owner's node count; what all_gather_into_tensor leaves).
void slice_edges_pairs(sw_index &ix, uint64_t *d_keys, uint64_t m, const uint64_t *d_cand_rows, uint64_t c, unsigned lo_bits,
                       unsigned hi_bits, uint64_t lo_base, unsigned ab, const uint64_t *d_rank_hash, const uint64_t *node_base,
                       uint32_t n_owners, uint64_t pad, hipStream_t stream)
{
    ix.n_edges = 0;
    ix.edges_hold_ranks = d_rank_hash == nullptr;   // (null table: first / second stay global ranks for now)
    if (m > occ_cap() || c > occ_cap()) raise_occ_cap(std::max<uint64_t>(m, c), "adjacency rows");
    if (n_owners == 0 || n_owners > 16) raise(SW_ERR_VALUE, "1 .. 16 owners are supported");
    if (lo_bits + hi_bits > 64 || lo_bits == 0 || hi_bits == 0) raise(SW_ERR_VALUE, "edge keys: 1 <= lo_bits, hi_bits and lo_bits + hi_bits <= 64");
    if (m) {
        WideKeys wk{};
        wk.lo_bits = lo_bits;
        wk.hi_bits = hi_bits;
        wk.lo_base = lo_base;
        wk.hash.table = d_rank_hash;
        wk.hash.pad = pad;
        wk.hash.n_owners = n_owners;
        for (uint32_t q = 0; q <= n_owners; ++q) wk.hash.node_base[q] = node_base[q];
        // the received keys are sorted where they lie (the caller's buffer is one half of the double buffer: it is clobbered)
        DevArray<uint64_t> k1(m), ck(c);
        DevArray<uint32_t> ca(c);
        if (c) {
            hipLaunchKernelGGL(k_split_rows2, dim3(blocks_for(c)), dim3(TPB), 0, stream, d_cand_rows, c, ck.p, ca.p);
            SW_HIP(hipGetLastError());
        }
        edges_from_pairs(d_keys, k1.p, m, ~0ull, 0, ab, ck.p, ca.p, nullptr, c, nullptr, stream, ix, nullptr, nullptr, &wk);   // rows carry no sentinels
    }
    if (ix.n_edges == 0) ix.edges.alloc(0);
}

void index_node_hashes(const sw_index &ix, uint64_t *d_out, hipStream_t stream)
{
    if (ix.n_nodes == 0) return;
    hipLaunchKernelGGL(k_node_hashes, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, d_out);
    SW_HIP(hipGetLastError());
}

// ---- rank -> hash by request (multi-GPU slices built without the job-wide table) ---------------------------------
// The edges of a slice name up to 2 E distinct nodes by global rank.  Sorted by rank (a pair sort: rank, slot = edge * 2 +
// side) the distinct ones are the requests: owner-local ranks, ascending, so the requests to one node owner are one
// contiguous piece and that owner reads its hashes in order.  The replies come back in request order and go into the edges
// through the same sorted list.  Volume: 12 B per distinct endpoint instead of 8 B per node of the whole job per GPU.
struct EdgeHashJob {
    DevArray<uint32_t> slot, uniq;   // [2 E], in rank order: where the hash goes / which request answers it
    DevArray<uint32_t> req;          // [n_unique]
    uint64_t n_unique = 0, n_slots = 0;
};
}  // namespace sw
sw_index::~sw_index() { delete hash_job; }
namespace sw {
namespace {
__global__ void k_edge_ranks(const sw_edge *__restrict__ edges, uint64_t n_edges, uint64_t *__restrict__ rank, uint32_t *__restrict__ slot)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    rank[2 * e] = edges[e].first;
    rank[2 * e + 1] = edges[e].second;
    slot[2 * e] = (uint32_t)(2 * e);
    slot[2 * e + 1] = (uint32_t)(2 * e + 1);
}
struct RankHeadFlag {   // 1 where a sorted rank differs from its predecessor
    const uint64_t *rank;
    __host__ __device__ uint32_t operator()(uint64_t j) const { return (j == 0 || rank[j] != rank[j - 1]) ? 1u : 0u; }
};
struct OwnerBases {
    uint64_t base[18];
    uint32_t n_owners;
};
// cum[j] = number of distinct ranks up to and including sorted position j; the head of every distinct rank writes its request
__global__ void k_edge_requests(const uint64_t *__restrict__ rank, const uint32_t *__restrict__ cum, uint64_t n, const OwnerBases B,
                                uint32_t *__restrict__ uniq, uint32_t *__restrict__ req, uint32_t *__restrict__ bad)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t u = cum[j] - 1u;
    uniq[j] = u;
    if (j == 0 || rank[j] != rank[j - 1]) {
        const uint64_t r = rank[j];
        uint32_t o = 0;
        for (uint32_t q = 1; q < B.n_owners; ++q) o += (B.base[q] <= r) ? 1u : 0u;
        if (r >= B.base[B.n_owners] || r - B.base[o] > 0xFFFFFFFFull) atomicOr(bad, 1u);
        req[u] = (uint32_t)(r - B.base[o]);
    }
}
// cnt[o] = distinct ranks below base[o] (o = 0 .. n_owners): one thread each, a binary search in the sorted ranks
__global__ void k_owner_cuts(const uint64_t *__restrict__ rank, const uint32_t *__restrict__ cum, uint64_t n, const OwnerBases B,
                             unsigned long long *__restrict__ below)
{
    const uint32_t o = threadIdx.x;
    if (o > B.n_owners) return;
    uint64_t lo = 0, hi = n;        // first j with rank[j] >= base[o]
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (rank[mid] < B.base[o]) lo = mid + 1; else hi = mid;
    }
    below[o] = lo ? cum[lo - 1] : 0u;
}
__global__ void k_lookup_hashes(const sw_node *__restrict__ nodes, uint64_t n_nodes, const uint32_t *__restrict__ local, uint64_t n,
                                uint64_t *__restrict__ out, uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = local[i];
    if (r >= n_nodes) {
        atomicOr(bad, 1u);
        out[i] = 0;
        return;
    }
    out[i] = nodes[r].hash;
}
__global__ void k_attach_hashes(sw_edge *__restrict__ edges, const uint32_t *__restrict__ slot, const uint32_t *__restrict__ uniq,
                                const uint64_t *__restrict__ replies, uint64_t n)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t s = slot[j];
    uint64_t *e = reinterpret_cast<uint64_t *>(edges + (s >> 1));
    e[s & 1u] = replies[uniq[j]];
}
}  // namespace

uint64_t edge_hash_requests(sw_index &ix, const uint64_t *node_base, uint32_t n_owners, uint64_t *counts_host, hipStream_t stream)
{
    if (!ix.edges_hold_ranks) raise(SW_ERR_VALUE, "the edges of this index hold hashes already (it was built with the rank -> hash table)");
    if (n_owners == 0 || n_owners > 16) raise(SW_ERR_VALUE, "1 to 16 node owners are supported");
    for (uint32_t o = 0; o < n_owners; ++o) counts_host[o] = 0;
    delete ix.hash_job;
    ix.hash_job = new EdgeHashJob;
    EdgeHashJob &J = *ix.hash_job;
    const uint64_t n = 2 * ix.n_edges;
    if (n >= 0xFFFFFFFFull) raise(SW_ERR_RUNTIME, "more than 2^31-1 edges on one device");
    J.n_slots = n;
    if (n == 0) return 0;
    OwnerBases B{};
    B.n_owners = n_owners;
    for (uint32_t o = 0; o <= n_owners; ++o) B.base[o] = node_base[o];
    unsigned bits = 1;
    while (bits < 64 && (node_base[n_owners] >> bits)) ++bits;
    DevArray<uint64_t> r0(n), r1(n);
    DevArray<uint32_t> s0(n), s1(n), cum(n), bad(1);
    DevArray<unsigned long long> below(n_owners + 1);
    SW_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
    hipLaunchKernelGGL(k_edge_ranks, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ix.edges.p, (uint64_t)ix.n_edges, r0.p, s0.p);
    SW_HIP(hipGetLastError());
    uint64_t *rk = r0.p, *rk_alt = r1.p;
    uint32_t *sl = s0.p, *sl_alt = s1.p;
    sort_pairs(rk, rk_alt, sl, sl_alt, n, 0, bits, stream);
    inclusive_sum(rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), RankHeadFlag{rk}), cum.p, n, (uint32_t)0,
                  stream);
    uint32_t n_unique = 0;
    SW_HIP(hipMemcpyAsync(&n_unique, cum.p + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    J.n_unique = n_unique;
    J.uniq.alloc(n);
    J.req.alloc(n_unique);
    hipLaunchKernelGGL(k_edge_requests, dim3(blocks_for(n)), dim3(TPB), 0, stream, rk, cum.p, n, B, J.uniq.p, J.req.p, bad.p);
    hipLaunchKernelGGL(k_owner_cuts, dim3(1), dim3(32), 0, stream, rk, cum.p, n, B, below.p);
    SW_HIP(hipGetLastError());
    unsigned long long h[18];
    uint32_t hb = 0;
    SW_HIP(hipMemcpyAsync(h, below.p, (n_owners + 1) * 8, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipMemcpyAsync(&hb, bad.p, 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    if (hb) raise(SW_ERR_VALUE, "an edge names a rank outside the node bases");
    for (uint32_t o = 0; o < n_owners; ++o) counts_host[o] = h[o + 1] - h[o];
    // the slots in rank order are kept (the sort's current buffer)
    if (sl == s0.p) J.slot = std::move(s0); else J.slot = std::move(s1);
    return n_unique;
}

void edge_hash_request_rows(const sw_index &ix, uint32_t *d_out, hipStream_t stream)
{
    if (!ix.hash_job) raise(SW_ERR_VALUE, "sw_index_edge_hash_requests has not been called on this index");
    if (ix.hash_job->n_unique)
        SW_HIP(hipMemcpyAsync(d_out, ix.hash_job->req.p, ix.hash_job->n_unique * 4, hipMemcpyDeviceToDevice, stream));
}

void node_hash_lookup(const sw_index &ix, const uint32_t *d_local_ranks, uint64_t n, uint64_t *d_out, hipStream_t stream)
{
    if (n == 0) return;
    DevArray<uint32_t> bad(1);
    SW_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
    hipLaunchKernelGGL(k_lookup_hashes, dim3(blocks_for(n)), dim3(TPB), 0, stream, ix.nodes.p, (uint64_t)ix.n_nodes, d_local_ranks, n, d_out,
                       bad.p);
    SW_HIP(hipGetLastError());
    uint32_t hb = 0;
    SW_HIP(hipMemcpyAsync(&hb, bad.p, 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    if (hb) raise(SW_ERR_VALUE, "a requested rank is not a node of this slice");
}

void edge_hash_attach(sw_index &ix, const uint64_t *d_replies, uint64_t n, hipStream_t stream)
{
    if (!ix.hash_job) raise(SW_ERR_VALUE, "sw_index_edge_hash_requests has not been called on this index");
    EdgeHashJob &J = *ix.hash_job;
    if (n != J.n_unique) raise(SW_ERR_VALUE, "%llu replies for %llu requests", (unsigned long long)n, (unsigned long long)J.n_unique);
    if (J.n_slots) {
        hipLaunchKernelGGL(k_attach_hashes, dim3(blocks_for(J.n_slots)), dim3(TPB), 0, stream, ix.edges.p, J.slot.p, J.uniq.p, d_replies,
                           J.n_slots);
        SW_HIP(hipGetLastError());
        SW_HIP(hipStreamSynchronize(stream));   // (the job's arrays go back to the pool below)
    }
    delete ix.hash_job;
    ix.hash_job = nullptr;
    ix.edges_hold_ranks = false;
}

void index_threshold_sums(const sw_index &ix, hipStream_t stream, uint64_t *sums3)
{
    DevArray<unsigned long long> sums(3);
    SW_HIP(hipMemsetAsync(sums.p, 0, 24, stream));
    if (ix.n_nodes) {
        hipLaunchKernelGGL(k_threshold_sums, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, sums.p);
        SW_HIP(hipGetLastError());
    }
    unsigned long long h[3];
    SW_HIP(hipMemcpyAsync(h, sums.p, 24, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < 3; ++i) sums3[i] = h[i];
}

// kmers._filter_edges_and_nodes (src/seqwin/kmers.py:132-173) on device: edges with weight > th and the
// nodes that are an endpoint of a surviving edge; kmers are shared with `ix` (not copied).
void index_filter_graph(const sw_index &ix, uint64_t weight_th, hipStream_t stream, sw_index &out)
{
    out.device = ix.device;
    out.n_kmers = 0;
    out.kmers.alloc(0);
    out.n_nodes = 0;
    out.n_edges = 0;
    if (ix.n_edges == 0 || ix.n_nodes == 0) {
        out.nodes.alloc(0);
        out.edges.alloc(0);
        return;
    }
    DevArray<uint32_t> ecum(ix.n_edges), keep(ix.n_nodes), kcum(ix.n_nodes);
    SW_HIP(hipMemsetAsync(keep.p, 0, ix.n_nodes * 4, stream));
    inclusive_sum(rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), EdgeKeepFlag{ix.edges.p, weight_th}),
                  ecum.p, ix.n_edges, (uint32_t)0, stream);
    uint32_t ne = 0;
    SW_HIP(hipMemcpyAsync(&ne, ecum.p + (ix.n_edges - 1), 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    out.n_edges = ne;
    out.edges.alloc(ne);
    hipLaunchKernelGGL(k_filter_edges, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ix.edges.p, ix.n_edges, weight_th,
                       ecum.p, ix.nodes.p, ix.n_nodes, out.edges.p, keep.p);
    SW_HIP(hipGetLastError());
    inclusive_sum(keep.p, kcum.p, ix.n_nodes, (uint32_t)0, stream);
    uint32_t nn = 0;
    SW_HIP(hipMemcpyAsync(&nn, kcum.p + (ix.n_nodes - 1), 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    out.n_nodes = nn;
    out.nodes.alloc(nn);
    hipLaunchKernelGGL(k_compact_nodes, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, keep.p, kcum.p,
                       out.nodes.p);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));
}

void device_filter_kmers(const sw_kmer *d_kmers, uint64_t n_kmers, const sw_node *d_nodes, uint64_t n_nodes,
                         const uint64_t *d_used_sorted, uint64_t n_used, hipStream_t stream,
                         DevArray<sw_kmer> &kmers_out, DevArray<sw_node> &nodes_out, uint64_t *n_kmers_out,
                         uint64_t *n_nodes_out)
{
    (void)n_kmers;
    *n_kmers_out = 0;
    *n_nodes_out = 0;
    if (n_nodes == 0 || n_used == 0) {
        kmers_out.alloc(0);
        nodes_out.alloc(0);
        return;
    }
    DevArray<uint32_t> keep(n_nodes), kcum(n_nodes);
    DevArray<uint64_t> size(n_nodes), scum(n_nodes);
    hipLaunchKernelGGL(k_filter_keep, dim3(blocks_for(n_nodes)), dim3(TPB), 0, stream, d_nodes, n_nodes, d_used_sorted,
                       n_used, keep.p, size.p);
    SW_HIP(hipGetLastError());
    inclusive_sum(keep.p, kcum.p, n_nodes, (uint32_t)0, stream);
    inclusive_sum(size.p, scum.p, n_nodes, (uint64_t)0, stream);
    uint32_t n_kept = 0;
    uint64_t n_out = 0;
    SW_HIP(hipMemcpyAsync(&n_kept, kcum.p + (n_nodes - 1), 4, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipMemcpyAsync(&n_out, scum.p + (n_nodes - 1), 8, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    nodes_out.alloc(n_kept);
    kmers_out.alloc(n_out);
    *n_nodes_out = n_kept;
    *n_kmers_out = n_out;
    if (n_kept == 0) return;
    DevArray<uint64_t> src_start(n_kept), dst_start(n_kept);
    hipLaunchKernelGGL(k_filter_nodes, dim3(blocks_for(n_nodes)), dim3(TPB), 0, stream, d_nodes, n_nodes, keep.p, kcum.p,
                       scum.p, nodes_out.p, src_start.p, dst_start.p);
    if (n_out)
        hipLaunchKernelGGL(k_filter_kmers, dim3(blocks_for(n_out)), dim3(TPB), 0, stream, d_kmers, src_start.p,
                           dst_start.p, (uint64_t)n_kept, n_out, kmers_out.p);
    SW_HIP(hipGetLastError());
    SW_HIP(hipStreamSynchronize(stream));
}

void index_verify(const sw_index &ix, uint64_t n_assemblies, bool scored, hipStream_t stream, uint64_t *out10)
{
    DevArray<unsigned long long> out(10);
    SW_HIP(hipMemsetAsync(out.p, 0, 80, stream));
    DevArray<uint32_t> bits(ix.n_kmers / 32 + 2);
    SW_HIP(hipMemsetAsync(bits.p, 0, bits.bytes(), stream));
    uint64_t base = 0;   // a slice's node ranges are offset by the occurrences of lower ranks
    if (ix.n_nodes) SW_HIP(hipMemcpyAsync(&base, &ix.nodes.p[0].start, 8, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    if (ix.n_nodes)
        hipLaunchKernelGGL(k_verify_nodes, dim3(blocks_for(ix.n_nodes)), dim3(TPB), 0, stream, ix.nodes.p, ix.n_nodes, ix.n_kmers, base,
                           n_assemblies, scored ? 1 : 0, bits.p, out.p);
    if (ix.n_kmers)
        hipLaunchKernelGGL(k_verify_kmers, dim3(blocks_for(ix.n_kmers)), dim3(TPB), 0, stream, ix.kmers.p, ix.n_kmers, bits.p, out.p);
    if (ix.n_edges)
        hipLaunchKernelGGL(k_verify_edges, dim3(blocks_for(ix.n_edges)), dim3(TPB), 0, stream, ix.edges.p, ix.n_edges, ix.nodes.p,
                           ix.n_nodes, (const uint64_t *)nullptr, (uint64_t)0, n_assemblies, out.p);
    SW_HIP(hipGetLastError());
    unsigned long long h[10];
    SW_HIP(hipMemcpyAsync(h, out.p, 80, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    if (ix.n_nodes == 0 && ix.n_kmers) h[1] += 1;
    for (int i = 0; i < 10; ++i) out10[i] = h[i];
}

// identity of an index's immutable part (kmers; nodes' hash / start / stop): what sw_get_penalty compares a caller's host
// arrays with before it reuses the still-resident index of the last sw_build (api.hip: host_identity is the same sums)
__global__ void k_identity(const sw_kmer *kmers, uint64_t nk, const sw_node *nodes, uint64_t nn, unsigned long long *sums,
                           uint64_t kbase, uint64_t nbase)   // (bases: the slice's place in the whole, multi-device graphs)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a = 0, b = 0;
    if (i < nk) a = ck_kmer(kbase + i, kmers[i]);
    if (i < nn) b = ck_node_identity(nbase + i, nodes[i]);
    for (int d = 32; d; d >>= 1) {
        a += __shfl_down(a, d, 64);
        b += __shfl_down(b, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&sums[0], (unsigned long long)a);
        if (b) atomicAdd(&sums[1], (unsigned long long)b);
    }
}

void device_identity(const sw_index &ix, hipStream_t stream, uint64_t *sums2, uint64_t kbase, uint64_t nbase)
{
    DevArray<unsigned long long> sums(2);
    SW_HIP(hipMemsetAsync(sums.p, 0, 16, stream));
    const uint64_t n = std::max(ix.n_kmers, ix.n_nodes);
    if (n) {
        hipLaunchKernelGGL(k_identity, dim3(blocks_for(n)), dim3(TPB), 0, stream, ix.kmers.p, ix.n_kmers, ix.nodes.p, ix.n_nodes, sums.p,
                           kbase, nbase);
        SW_HIP(hipGetLastError());
    }
    unsigned long long h[2];
    SW_HIP(hipMemcpyAsync(h, sums.p, 16, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    sums2[0] = h[0];
    sums2[1] = h[1];
}

void device_checksums(const sw_index &ix, hipStream_t stream, uint64_t *sums3, uint64_t kbase, uint64_t nbase, uint64_t ebase)
{
    DevArray<unsigned long long> sums(3);
    SW_HIP(hipMemsetAsync(sums.p, 0, 24, stream));
    const uint64_t n = std::max(ix.n_kmers, std::max(ix.n_nodes, ix.n_edges));
    if (n) {
        hipLaunchKernelGGL(k_checksum, dim3(blocks_for(n)), dim3(TPB), 0, stream, ix.kmers.p, ix.n_kmers, ix.nodes.p,
                           ix.n_nodes, ix.edges.p, ix.n_edges, kbase, nbase, ebase, sums.p);
        SW_HIP(hipGetLastError());
    }
    unsigned long long h[3];
    SW_HIP(hipMemcpyAsync(h, sums.p, 24, hipMemcpyDeviceToHost, stream));
    SW_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < 3; ++i) sums3[i] = h[i];
}

}  // namespace sw
