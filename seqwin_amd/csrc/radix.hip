// radix.hip -- stable LSD radix sort of 64-bit keys (keys only), 8- or 9-bit digits, for gfx950.
//
// The index stage sorts 8-byte keys eight times per build (the edge pairs over 2 ceil(log2 n_nodes) bits, the unsort words over
// the index bits above 2^14: DESIGN.md 3.2) -- the reference's lsd_radix_sort_key (cpp/src/seqwin/build_internals.cpp:76-144)
// on the device.  A pass follows the onesweep scheme -- the first digit's histogram in one sweep (every pass counts the next
// digit of the keys it holds anyway), then per pass one kernel that ranks a tile's keys, learns the tile's global digit
// offsets by decoupled look-back and writes the keys out through LDS in digit order -- with a ballot-based rank (no LDS
// atomics in the ranking, so the order inside a digit is the input order: stable).  745 M 54-bit keys: 25.0 ms against 29.9 (r03: tests/tools/sort_time.py; r02: 26.3 against 31.2)
// for rocPRIM's onesweep (tests/tools/sort_time.py); with loads, stores and look-back switched off a pass still takes 70 % of
// its time (ranking, LDS traffic, five barriers per tile with one 150 KiB workgroup per CU): the passes are bound on chip.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <mutex>

#include "device.hpp"

namespace sw {
namespace {

constexpr int RS_ITEMS = 16;
#ifdef SW_RS_STAMPS   // timing builds: every 64th tile of a pass writes the shader clock at its phase boundaries (thread 0)
__device__ unsigned long long *g_rs_stamps = nullptr;
#define RS_STAMP(i)                                                                                        \
    do {                                                                                                   \
        if (g_rs_stamps && (tile & 63u) == 0 && threadIdx.x == 0) g_rs_stamps[(size_t)(tile >> 6) * 16 + (i)] = clock64(); \
    } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif
#ifndef RS_LOOK
#define RS_LOOK 4   // predecessors read per look-back step (independent loads; 8: more registers and state traffic, 29.0 against 25.1 ms)
#endif

// digit histograms of ALL passes in one sweep over the keys (in LDS; the 16 LDS atomics per thread and tile that counted the
// next pass's digit inside a pass cost more there: the passes are bound by their LDS traffic, this sweep by HBM)
template <int BITS>
__global__ __launch_bounds__(256) void k_rs_hist(const uint64_t *__restrict__ keys, uint64_t n, unsigned begin_bit, unsigned end_bit,
                                                 unsigned n_passes, unsigned long long *__restrict__ hist)
{
    constexpr uint32_t RADIX = 1u << BITS, MAXP = (64 + BITS - 1) / BITS;
    __shared__ uint32_t h[MAXP * RADIX];
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256) h[i] = 0;
    __syncthreads();
    const uint64_t chunk = 32768;   // a workgroup's share, small enough for 32-bit counters
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    auto count = [&](uint64_t k) {
        for (unsigned p = 0; p < n_passes; ++p) {
            const unsigned sh = begin_bit + BITS * p, bits = min((unsigned)BITS, end_bit - sh);
            atomicAdd(&h[p * RADIX + ((uint32_t)(k >> sh) & ((1u << bits) - 1u))], 1u);
        }
    };
    for (uint64_t i = i0 + 2 * threadIdx.x; i < i1; i += 512) {
        if (i + 1 < i1) {
            const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(keys + i);
            count(kk.x);
            count(kk.y);
        } else {
            count(keys[i]);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// The same histograms without reading a key, for keys whose upper halves are a permutation of 0 .. n-1 sorted on bits >= 32:
// the values of a digit at index bits [sh, sh + b) repeat with period 2^(sh+b), 2^sh times each per period.
template <int BITS>
__global__ void k_rs_perm_hist(uint64_t n, unsigned begin_bit, unsigned end_bit, unsigned long long *__restrict__ hist)
{
    constexpr uint32_t RADIX = 1u << BITS;
    const unsigned p = blockIdx.x, at = begin_bit + BITS * p, sh = at - 32u, bits = min((unsigned)BITS, end_bit - at);
    const uint64_t v = threadIdx.x, block = 1ull << sh, period = block << bits;
    unsigned long long c = 0;
    if (v < (1ull << bits)) {
        const uint64_t q = n / period, r = n % period, lo = v * block;
        c = q * block + (r > lo ? (r - lo < block ? r - lo : block) : 0ull);
    }
    hist[(size_t)p * RADIX + threadIdx.x] = c;
}

// exclusive scan of a pass's digit counts (one workgroup of RADIX threads)
constexpr uint32_t RS_CURSOR_STRIDE = 32;   // an unstable pass's digit cursors lie 256 B apart (atomics spread over the L2 channels)
template <int BITS> __global__ void k_rs_scan(unsigned long long *__restrict__ h, unsigned long long *__restrict__ cursor)
{
    constexpr uint32_t RADIX = 1u << BITS;
    __shared__ unsigned long long s[RADIX];
    const unsigned long long v = h[threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < RADIX; d <<= 1) {
        const unsigned long long add = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    h[threadIdx.x] = s[threadIdx.x] - v;
    if (cursor) cursor[(size_t)threadIdx.x * RS_CURSOR_STRIDE] = s[threadIdx.x] - v;
}

#ifdef SW_AB   // (records of the one-tile-per-workgroup pass)
constexpr unsigned long long RS_AGG = 1ull << 62, RS_INC = 2ull << 62, RS_VAL = (1ull << 62) - 1ull;
#endif

// Lanes of the wave whose digit equals this lane's (among the live ones), as two 32-bit halves: per digit bit one ballot and,
// per half, one three-input bit operation  m & ~(ballot ^ -bit)  (the compiler's own form of  m &= bit ? bal : ~bal  was nine
// VALU instructions per bit on 64-bit lane masks -- 107 per key at 9 bits, a third of a pass).
template <int BITS>
__device__ __forceinline__ void match_digit(uint32_t d, bool live, uint32_t &mlo, uint32_t &mhi)
{
    const uint64_t lv = __ballot(live);
    mlo = (uint32_t)lv;
    mhi = (uint32_t)(lv >> 32);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t sx = (uint32_t)(-(int32_t)((d >> b) & 1u));   // all ones / zero
        const uint64_t bal = __ballot(sx != 0);
        const uint32_t blo = (uint32_t)bal, bhi = (uint32_t)(bal >> 32);
        uint32_t nlo, nhi;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x90" : "=v"(nlo) : "v"(mlo), "s"(blo), "v"(sx));   // src0 & ~(src1 ^ src2)
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x90" : "=v"(nhi) : "v"(mhi), "s"(bhi), "v"(sx));
        mlo = nlo;
        mhi = nhi;
    }
}

// The property RANK = 1 rests on, checked on the device itself: 16 waves per workgroup (the occupancy of a pass) rank 96 rounds
// of contended digit patterns -- one word for all lanes, both halves of one word, pairs, triples, the two halves of the wave
// on one word, random digits, dead lanes -- once with the LDS atomic and once with ballots; any difference is counted.
__global__ __launch_bounds__(1024) void k_rs_rank_selftest(uint32_t *__restrict__ mismatches)
{
    constexpr int BITS = 9;
    constexpr uint32_t RADIX = 1u << BITS, WAVES = 16;
    __shared__ uint16_t wa[WAVES][RADIX], wb[WAVES][RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < WAVES * RADIX; i += 1024) (&wa[0][0])[i] = (&wb[0][0])[i] = 0;
    __syncthreads();
    uint32_t bad = 0;
    for (uint32_t r = 0; r < 96; ++r) {
        uint32_t x = (lane * 0x9E3779B1u) ^ ((r + 1u) * 0x85EBCA6Bu) ^ ((wave + blockIdx.x * 16u + 1u) * 0xC2B2AE35u);
        x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
        uint32_t d;
        switch (r & 7u) {
        case 0: d = 7; break;
        case 1: d = 10 + (lane & 1u); break;
        case 2: d = lane >> 1; break;
        case 3: d = 100 + lane % 3u; break;
        case 4: d = (lane & 32u) ? 5u : 4u; break;
        case 5: d = x & 15u; break;
        case 6: d = 511u - lane; break;
        default: d = x & 511u; break;
        }
        const bool live = (r % 5u != 4u) || ((x >> 20) & 3u) != 0;      // every fifth round a quarter of the lanes sits out
        const uint32_t sh = (d & 1u) << 4;
        uint32_t old = 0;
        if (live) old = atomicAdd(reinterpret_cast<uint32_t *>(&wa[wave][d & ~1u]), 1u << sh);
        const uint32_t rank_a = (old >> sh) & 0xFFFFu;
        uint32_t mlo, mhi;
        match_digit<BITS>(d, live, mlo, mhi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        uint32_t prior = 0;
        if (live) prior = wb[wave][d];
        const uint32_t rank_b = prior + below;
        __builtin_amdgcn_wave_barrier();
        if (live && below == 0) wb[wave][d] = (uint16_t)(prior + (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi));
        __builtin_amdgcn_wave_barrier();
        if (live && rank_a != rank_b) ++bad;
    }
    __syncthreads();
    for (uint32_t i = tid; i < WAVES * RADIX; i += 1024) bad += (&wa[0][0])[i] != (&wb[0][0])[i];
    if (bad) atomicAdd(mismatches, bad);
}

// 1: rank by LDS atomics, 0: by ballots.  SEQWIN_AMD_RADIX_RANK=ballot|atomic overrides ("atomic" still runs the check and
// fails loudly if the device does not pass it).
bool ballot_forced()
{
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_RADIX_RANK");
    return e && !strcmp(e, "ballot");
}
std::mutex &rank_mu()
{
    static std::mutex &mu = *new std::mutex;
    return mu;
}
std::map<int, int> &rank_modes()
{
    static std::map<int, int> &modes = *new std::map<int, int>;
    return modes;
}
int rank_mode()
{
    std::mutex &mu = rank_mu();
    std::map<int, int> &modes = rank_modes();
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_RADIX_RANK");
    if (e && !strcmp(e, "ballot")) return 0;
    int dev = 0;
    SW_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = modes.find(dev);
    if (it == modes.end()) {
        uint32_t *d_bad = nullptr, bad = 1;
        SW_HIP(hipMalloc(&d_bad, 4));
        SW_HIP(hipMemset(d_bad, 0, 4));
        hipLaunchKernelGGL(k_rs_rank_selftest, dim3(512), dim3(1024), 0, nullptr, d_bad);
        SW_HIP(hipGetLastError());
        SW_HIP(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        (void)hipFree(d_bad);
        if (bad) fprintf(stderr, "[seqwin_amd] radix sort: the LDS-atomic ranking self-check found %u differences on device %d; ranking by ballots\n", bad, dev);
        it = modes.emplace(dev, bad ? 0 : 1).first;
    }
    if (e && !strcmp(e, "atomic") && it->second != 1)
        raise(SW_ERR_RUNTIME, "SEQWIN_AMD_RADIX_RANK=atomic, but this device does not serve the lanes of an LDS atomic in lane order");
    return it->second;
}

// SEQWIN_AMD_FAULT_INJECT=rank (tests): the first wave of every tile of the LDS-atomic passes swaps the ranks of neighbouring
// lanes that hold the same digit -- exactly the failure the stability of these passes rests on NOT happening.  The ballot
// passes and rocPRIM's have no such hook, so a build recovers once the guards have demoted the device.
static uint32_t fault_rank()
{
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_FAULT_INJECT");
    return (e && !strcmp(e, "rank")) ? 1u : 0u;
}

// One pass: tile t = workgroup t takes THREADS x 16 consecutive keys, wave w of it the w-th 1024 of them, lane l item i the
// key w * 1024 + i * 64 + l.  Two shapes: 512 threads with 8-bit digits (8192-key tiles, 72 KiB of LDS, two workgroups per CU)
// and 1024 threads with 9-bit digits (16384-key tiles, 150 KiB: the same 32 keys per digit and tile, one pass fewer for the
// 54 bits of the edge pairs).  This one-tile-per-workgroup form (SEQWIN_AMD_RADIX_KERNEL=classic; the default is the
// persistent form below) waits for lower-numbered workgroups: like rocPRIM's onesweep it relies on the dispatcher starting
// workgroups in index order.  Should a wait ever outlast RS_SPIN_LIMIT polls, the workgroup gives up and raises *fail --
// the caller then reports an error -- instead of hanging the device.
constexpr uint32_t RS_SPIN_LIMIT = 1u << 22;
#ifdef SW_AB
template <int THREADS, int BITS>
__global__ __launch_bounds__(THREADS) void k_rs_pass(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n,
                                                     unsigned shift, unsigned bits, const unsigned long long *__restrict__ digit_base,
                                                     unsigned long long *__restrict__ state, uint32_t *__restrict__ fail)
{
    constexpr uint32_t RADIX = 1u << BITS, WAVES = THREADS / 64, TILE = THREADS * RS_ITEMS;
    static_assert(RADIX <= (uint32_t)THREADS, "one thread per digit");
    __shared__ uint64_t sk[TILE];
    __shared__ uint16_t whist[WAVES][RADIX];         // per wave: running digit counts (<= 1024), then the wave's base inside the tile's digit
    __shared__ uint32_t lstart[RADIX];               // first local position of a digit in the tile
    __shared__ unsigned long long goff[RADIX];       // global position of local position 0 of a digit: out[goff[d] + local]
    __shared__ uint32_t wsum[RADIX / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint64_t t0 = (uint64_t)tile * TILE;
    const uint32_t cnt_tile = (uint32_t)min((uint64_t)TILE, n - t0);
    const uint32_t dmask = (1u << bits) - 1u;

    uint64_t key[RS_ITEMS];
    uint32_t rank[RS_ITEMS];   // position of the item among the items of its digit in its wave
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t li = wave * (64 * RS_ITEMS) + i * 64 + lane;
        key[i] = li < cnt_tile ? in[t0 + li] : ~0ull;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t li = wave * (64 * RS_ITEMS) + i * 64 + lane;
        const bool live = li < cnt_tile;
        const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
        uint32_t mlo, mhi;                                 // lanes with the same digit (among the live ones)
        match_digit<BITS>(d, live, mlo, mhi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // ... in lower lanes
        uint32_t prior = 0;
        if (live) prior = whist[wave][d];                  // every lane of a group reads before its leader adds
        rank[i] = prior + below;
        __builtin_amdgcn_wave_barrier();
        if (live && below == 0) whist[wave][d] = (uint16_t)(prior + (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi));   // the group's first lane
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // thread d < RADIX: the digit's counts per wave -> bases, total; publish; scan over the digits; look back
    uint32_t total = 0, incl = 0;
    unsigned long long *st = state + (size_t)tile * RADIX;
    if (tid < RADIX) {
        const uint32_t d = tid;
#pragma unroll
        for (uint32_t w = 0; w < WAVES; ++w) {
            const uint32_t c = whist[w][d];
            whist[w][d] = (uint16_t)total;
            total += c;
        }
        __hip_atomic_store(&st[d], (tile == 0 ? RS_INC : RS_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        incl = total;
        for (uint32_t dd = 1; dd < 64; dd <<= 1) {
            const uint32_t up = __shfl_up(incl, dd, 64);
            if (lane >= dd) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
    }
    __syncthreads();
    if (tid < RADIX) {
        const uint32_t d = tid;
        uint32_t before = incl - total;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        lstart[d] = before;
        unsigned long long excl = 0;
        if (tile) {
            for (int64_t t = (int64_t)tile - 1; t >= 0; --t) {
                unsigned long long v;
                uint32_t spins = 0;
                for (;;) {
                    v = __hip_atomic_load(&state[(size_t)t * RADIX + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 62) != 0) break;           // the predecessor started earlier: it will publish
                    if (++spins > RS_SPIN_LIMIT) {         // (never seen; see above)
                        atomicOr(fail, 1u);
                        v = RS_INC;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                excl += v & RS_VAL;
                if ((v >> 62) == 2) break;
            }
            __hip_atomic_store(&st[d], RS_INC | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        goff[d] = digit_base[d] + excl - before;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t li = wave * (64 * RS_ITEMS) + i * 64 + lane;
        if (li < cnt_tile) {
            const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
            sk[lstart[d] + whist[wave][d] + rank[i]] = key[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const uint32_t t = j * THREADS + tid;
        if (t < cnt_tile) {
            const uint64_t k = sk[t];
            out[goff[(uint32_t)(k >> shift) & dmask] + t] = k;
        }
    }
}
#endif   // SW_AB (the one-tile-per-workgroup pass)

// Look-back records of the persistent passes carry a 16-bit epoch (the pass number of their state buffer) above flag and count:
// a record of an earlier pass -- or of an earlier sort -- reads as "not published", so the buffer is never cleared between passes
// (r03: twelve passes per build cleared 3.5 GB of state: 30 fills, 1 ms, 10.8 GB of counted writes); it is cleared when it is
// made and once per 65 535 passes (state_buf / next_epoch below).
constexpr unsigned long long RSE_VAL = (1ull << 46) - 1ull;
__device__ __forceinline__ unsigned long long rse_pack(uint32_t epoch, uint32_t flag, unsigned long long v)
{
    return ((unsigned long long)epoch << 48) | ((unsigned long long)flag << 46) | v;
}
__device__ __forceinline__ uint32_t rse_flag(unsigned long long w, uint32_t epoch)   // 0: not published (in this epoch), 1: aggregate, 2: inclusive
{
    return (uint32_t)(w >> 48) == epoch ? (uint32_t)(w >> 46) & 3u : 0u;
}

// The same pass as a PERSISTENT kernel: a workgroup takes tiles from a ticket counter until none are left, and the keys of
// its next tile are already on their way (registers) while it ranks, scans and writes the current one -- with one workgroup
// of 150 KiB per CU the one-tile-per-workgroup form leaves the memory pipe idle during ranking and look-back and the ALUs idle
// during the loads (2.7 TB/s of moved bytes against 4.5 for rocPRIM's 20-byte pair passes).  Tickets also make the look-back
// independent of the dispatch order: the lowest unfinished tile is always the CURRENT tile of a running workgroup (every
// workgroup takes its tickets in increasing order and holds at most the current and the next one), so the waits terminate
// whatever the residency.  The next tile's loads are issued once the current keys sit in LDS -- into the same registers -- and
// are in flight during the look-back and the write-out; the look-back reads RS_LOOK predecessors per step (independent loads).
//
// RANK = 1 (default where the device passes k_rs_rank_selftest): a key's place among the keys of its digit in its wave comes
// from ONE LDS atomic with return on the wave's counter of that digit (two 16-bit counters per word: a wave holds 1024 keys)
// instead of eight or nine ballots -- ~6 instead of ~45 vector instructions per key; ranking was 42 % of a tile (VALU issue).
// Stability then rests on two properties of the LDS unit: the instructions of one wave are served in order, and the lanes of
// one instruction that hit the same word are served in ascending lane order.  The second is not an architectural promise, so
// it is CHECKED on the device before the first sort (contended patterns under full occupancy, every rank compared with the
// ballot form's); a device that fails keeps RANK = 0.  The unstable passes (atomic cursors) need neither property.
template <int THREADS, int BITS, int RANK>
__global__ __launch_bounds__(THREADS) void k_rs_pass_p(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n,
                                                       uint32_t n_tiles, unsigned shift, unsigned bits,
                                                       const unsigned long long *__restrict__ digit_base,
                                                       unsigned long long *__restrict__ state, uint32_t *__restrict__ ticket,
                                                       uint32_t *__restrict__ fail, uint32_t dbg,
                                                       unsigned long long *__restrict__ cursor, uint32_t cursor_stride,
                                                       uint32_t group_shift, uint32_t epoch)
{
    constexpr uint32_t RADIX = 1u << BITS, WAVES = THREADS / 64, TILE = THREADS * RS_ITEMS;
    static_assert(RADIX <= (uint32_t)THREADS, "one thread per digit");
    __shared__ uint64_t sk[TILE];
    __shared__ uint16_t whist[WAVES][RADIX];
    __shared__ uint32_t lstart[RADIX];
    __shared__ unsigned long long goff[RADIX];
    __shared__ uint32_t wsum[RADIX / 64];
    __shared__ uint32_t s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t dmask = (1u << bits) - 1u;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    uint32_t tile = s_tile;
    uint64_t key[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint64_t g = (uint64_t)tile * TILE + wave * (64 * RS_ITEMS) + i * 64 + lane;
        key[i] = (tile < n_tiles && g < n) ? in[g] : ~0ull;
    }
    const uint32_t tid0 = threadIdx.x;
    // The ticket of the tile AFTER the next is taken while the next tile's keys are requested (r04: taken at the top of an
    // iteration, its ~2 us round trip sat in front of the ranking -- the wait for the tile's keys waits for every earlier
    // global operation of the wave).  A workgroup so holds up to three tickets -- current, next, the one after -- in increasing
    // order; the lowest unfinished tile is still some workgroup's current one or becomes it without waiting for anything.
    uint32_t fut = 0;
    if (tid0 == 0) fut = atomicAdd(ticket, 1u);
    while (tile < n_tiles) {
        // (opaque per iteration: otherwise the 16 + 16 + 16 load / LDS / store addresses derived from the thread index are
        // hoisted out of the loop and kept in ~100 registers across it)
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        RS_STAMP(0);
        for (uint32_t i = tid; i < WAVES * RADIX / 2; i += THREADS) (reinterpret_cast<uint32_t *>(&whist[0][0]))[i] = 0;
        __syncthreads();
        const uint64_t t0 = (uint64_t)tile * TILE;
        const uint32_t cnt_tile = (uint32_t)min((uint64_t)TILE, n - t0);
        if constexpr (RANK == 1) {
            // all sixteen atomics are issued before the first result is looked at (the LDS unit serves a wave's instructions in
            // order, so the results are the ranks whatever the wave waits for): one exposed LDS round trip instead of sixteen
#pragma unroll
            for (int i = 0; i < RS_ITEMS; ++i) {
                const uint32_t li = wave * (64 * RS_ITEMS) + i * 64 + lane;
                const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
                rank[i] = 0;
                if (li < cnt_tile) rank[i] = atomicAdd(reinterpret_cast<uint32_t *>(&whist[wave][d & ~1u]), 1u << ((d & 1u) << 4));   // ds_add_rtn_u32
            }
#pragma unroll
            for (int i = 0; i < RS_ITEMS; ++i) {
                const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
                rank[i] = (rank[i] >> ((d & 1u) << 4)) & 0xFFFFu;
            }
            if ((dbg & 16u) && !cursor && wave == 0) {   // SEQWIN_AMD_FAULT_INJECT=rank (tests): see fault_rank()
#pragma unroll
                for (int i = 0; i < RS_ITEMS; ++i) {
                    const uint32_t li = i * 64 + lane;
                    const uint32_t d = li < cnt_tile ? ((uint32_t)(key[i] >> shift) & dmask) : ~lane;
                    const uint32_t od = __shfl_xor(d, 1, 64), orank = __shfl_xor(rank[i], 1, 64);
                    if (od == d) rank[i] = orank;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i) {
            const uint32_t li = wave * (64 * RS_ITEMS) + i * 64 + lane;
            const bool live = li < cnt_tile;
            const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
            if constexpr (RANK == 1) {
                (void)live;
                (void)d;
            } else {
                uint32_t mlo = 1u << (lane & 31u), mhi = 0;
                if (!(dbg & 8u)) match_digit<BITS>(d, live, mlo, mhi);
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                uint32_t prior = 0;
                if (live) prior = whist[wave][d];
                rank[i] = prior + below;
                __builtin_amdgcn_wave_barrier();
                if (live && below == 0) whist[wave][d] = (uint16_t)(prior + (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi));
                __builtin_amdgcn_wave_barrier();
            }
        }
        RS_STAMP(1);   // this wave's keys ranked
        __syncthreads();
        RS_STAMP(2);
        uint32_t total = 0, incl = 0;
        unsigned long long pre[RS_LOOK];
        unsigned long long *st = state + (size_t)tile * RADIX;
        if (tid < RADIX) {
            const uint32_t d = tid;
#pragma unroll
            for (uint32_t w = 0; w < WAVES; ++w) {
                const uint32_t c = whist[w][d];
                whist[w][d] = (uint16_t)total;
                total += c;
            }
            if (cursor) {
                // UNSTABLE pass (the order inside a digit's range is free: the unsort): the tile claims its places in every
                // digit's range with one atomic add -- no states, no look-back; the answer is due after the keys sit in LDS
                // (group_shift < 64: the cursors of the range of 2^group_shift positions the tile lies in -- a pass inside the buckets
                //  an earlier pass made, see radix_unsort_perm)
                const uint64_t group = group_shift < 64u ? (((uint64_t)tile * TILE) >> group_shift) : 0ull;
                pre[0] = total ? atomicAdd(&cursor[(group * RADIX + d) * cursor_stride], (unsigned long long)total) : 0ull;
            } else {
                __hip_atomic_store(&st[d], rse_pack(epoch, tile == 0 ? 2u : 1u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < RS_LOOK; ++j)   // the first look-back step: requested now, examined after the keys are placed in LDS
                    pre[j] = (int64_t)tile - 1 - j >= 0
                                 ? __hip_atomic_load(&state[(size_t)(tile - 1 - j) * RADIX + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : rse_pack(epoch, 2u, 0);
            }
            incl = total;
            for (uint32_t dd = 1; dd < 64; dd <<= 1) {
                const uint32_t up = __shfl_up(incl, dd, 64);
                if (lane >= dd) incl += up;
            }
            if (lane == 63) wsum[wave] = incl;
        }
        RS_STAMP(3);   // digit totals published, scan
        __syncthreads();
        uint32_t before = 0;
        if (tid < RADIX) {
            before = incl - total;
            for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
            lstart[tid] = before;
        }
        if (tid == 0) s_tile = fut;
        __syncthreads();
        const uint32_t ntile = s_tile;
        // the tile in digit order, in LDS; then the next tile's keys are requested into the same registers (in flight during the
        // write-out) and thread 0 takes the ticket after that one
        auto place_and_fetch = [&]() {
#pragma unroll
            for (int i = 0; i < RS_ITEMS; ++i) {
                const uint32_t li = wave * (64 * RS_ITEMS) + i * 64 + lane;
                if (li < cnt_tile) {
                    const uint32_t d = (uint32_t)(key[i] >> shift) & dmask;
                    sk[lstart[d] + whist[wave][d] + rank[i]] = key[i];
                }
            }
#pragma unroll
            for (int i = 0; i < RS_ITEMS; ++i) {
                const uint64_t g = (uint64_t)ntile * TILE + wave * (64 * RS_ITEMS) + i * 64 + lane;
                key[i] = (ntile < n_tiles && g < n && !(dbg & 4u)) ? in[g] : (dbg & 4u ? g * 0x9E3779B97F4A7C15ull : ~0ull);
            }
            if (tid == 0) fut = atomicAdd(ticket, 1u);
        };
        // (r04, measured on 745 M keys with the counters of the timing build: a look-back step of four 8-byte records takes ~2 800
        //  cycles -- the records are read at agent scope and queue behind the pass's own streams -- and the walk is ~16 records
        //  long, i.e. one step's latency over the ~180 cycles between two tiles' starts: 5.9 steps, 16 500 of a tile's 45 600 cycles.
        //  Eight records per step (4 100 cycles a step, a walk of 20), 16-byte loads for two digits per thread, and 4-byte aggregate
        //  records with one inclusive-sum probe per step of eight were all measured within 2 % of this form or slower: NOTES.md.)
        // (r04) The waves that own no digit place their keys at once; the digit waves look back FIRST and place theirs afterwards:
        // a tile's inclusive record is published a whole scatter phase earlier, so every later tile finds the end of its walk
        // nearer (the walk is as long as the tiles that have published an aggregate but not yet their inclusive sum), and the
        // look-back of these waves runs beside the LDS traffic of the others.
        const bool digit_wave = tid < RADIX;            // (wave-uniform: RADIX is a multiple of 64)
        if (!digit_wave || cursor) place_and_fetch();
        RS_STAMP(4);   // keys in LDS in digit order (waves without a digit)
        if (tid < RADIX && cursor) {
            goff[tid] = pre[0] - before;                // (cursor[d] started at digit_base[d])
        } else if (tid < RADIX) {                       // look-back, four predecessors per step
            const uint32_t d = tid;
            unsigned long long excl = 0;
            if (tile) {
                int64_t t = (int64_t)tile - 1;
                uint32_t spins = 0;
                bool done = false;
                bool first_step = true;
#ifdef SW_RS_STAMPS
                uint32_t n_steps = 0, n_polls = 0;
#endif
                while (!done && t >= 0 && !(dbg & 1u)) {
#ifdef SW_RS_STAMPS
                    ++n_steps;
#endif
                    unsigned long long v[RS_LOOK];
#pragma unroll
                    for (int j = 0; j < RS_LOOK; ++j)
                        v[j] = first_step ? pre[j]
                               : t - j >= 0 ? __hip_atomic_load(&state[(size_t)(t - j) * RADIX + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : rse_pack(epoch, 2u, 0);   // before the first tile: nothing
                    first_step = false;
                    int j = 0;
#pragma unroll
                    for (; j < RS_LOOK; ++j) {
                        const uint32_t f = rse_flag(v[j], epoch);
                        if (f == 0) break;              // not published yet (in this pass): poll again from here
                        excl += v[j] & RSE_VAL;
                        if (f == 2) {
                            done = true;
                            break;
                        }
                    }
                    if (done) break;
                    t -= j;
                    if (j < RS_LOOK) {
#ifdef SW_RS_STAMPS
                        ++n_polls;
#endif
                        if (++spins > RS_SPIN_LIMIT) {  // (never seen; the tile's owner is running, see above)
                            atomicOr(fail, 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
#ifdef SW_RS_STAMPS
                if (g_rs_stamps && (tile & 63u) == 0 && tid == 0) {   // digit 0: steps, polls of an unpublished record, entries walked
                    g_rs_stamps[(size_t)(tile >> 6) * 16 + 8] = n_steps;
                    g_rs_stamps[(size_t)(tile >> 6) * 16 + 9] = n_polls;
                    g_rs_stamps[(size_t)(tile >> 6) * 16 + 10] = (unsigned long long)((int64_t)tile - 1 - t);
                }
#endif
                __hip_atomic_store(&st[d], rse_pack(epoch, 2u, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            goff[d] = digit_base[d] + excl - before;
        }
        RS_STAMP(5);   // digit 0's look-back done
        if (digit_wave && !cursor) place_and_fetch();
        __syncthreads();
        RS_STAMP(6);   // every digit's
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const uint32_t t = j * THREADS + tid;
            if (t < cnt_tile) {
                const uint64_t k = sk[t];
                if (!(dbg & 2u)) out[goff[(uint32_t)(k >> shift) & dmask] + t] = k;
            }
        }
        RS_STAMP(7);   // stores issued
        // (no barrier here: the next iteration touches sk / goff / s_tile only behind its own barriers)
        tile = ntile;
    }
}


// (place in the stage - dense index) of the tile that holds dense index g, searched from tile `lo` on: dst_off[lo] <= g < n
__device__ __forceinline__ unsigned long long stage_search(const StageSource &S, uint32_t lo, uint64_t g)
{
    uint32_t hi = S.n_tiles;   // dst_off[hi] = n > g; the answer is the last tile whose first dense index is <= g
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (S.dst_off[mid] <= g) lo = mid; else hi = mid;
    }
    return S.tile_offset[lo] - S.dst_off[lo];
}

// ---- pairs: 32-bit keys with a 16-byte payload (the node sort: key32 = top half of the hash, OccPay) ---------------------------
// lsd_radix_sort of cpp/src/seqwin/build_internals.cpp:76-144 on the device, for the occurrences.  Same scheme as the keys-only
// pass (tickets, ranking per wave, decoupled look-back, write-out through LDS in digit order) in a shape made for 20-byte
// elements: 1024 threads x 7 elements (7168-element tiles), 8-bit digits, keys and payloads staged through LDS together (140 KiB:
// one placing phase and one write-out phase per tile) -- one workgroup per CU (r04, first form: 4096-element tiles, two workgroups per CU: twice the tiles in
// flight, so a look-back twice as long for half the elements -- 8.3 ms per pass of 745 M pairs against 7.0 for rocPRIM).  The
// four digit waves look back while the other twelve place their keys.  A thread keeps the global place of "its" output slots
// from the key round for the payload round.
// Ranking is by the LDS atomic only (k_rs_pass_p, RANK = 1): on a device that fails the self-check the pair sort is rocPRIM's.
// Look-back records carry a 16-bit epoch (the pass number of this state buffer) above flag and count: a record of an earlier
// pass -- or of an earlier sort -- reads as "not published", so the buffer is never cleared between passes (twelve passes per
// build cleared 3.5 GB; the buffer is cleared once per 65 535 passes).
// STAGE (the first pass of the node sort): the tile's elements are read from the sketch stage -- what k_order did in a pass
// of its own (16 B read + 24 B written per tuple, and 20 B read again here); how an element finds its place in the stage is
// described at `fetch` below.  The registers hold the raw tuple (canonical hash, pos | record << 32) until the tile is ranked:
// extend_hashes (hashing_internals.hpp:89-103), the split into key and payload and the store of the record index happen there.
constexpr int RP_THREADS = 1024, RP_ITEMS = 7, RP_BITS = 8;
// SRC = 2 (the first pass of a slice's node sort, multi-GPU forms): the elements are the received tuples themselves, rows of
// (out_hash, pos | record << 32) -- the copy k_rows_to_pay made of them (16 B read + 20 B written per tuple) is not needed.
template <int SRC>
__global__ __launch_bounds__(RP_THREADS) void k_rs_pair_pass(const uint32_t *__restrict__ kin, const uint4 *__restrict__ pin,
                                                             uint32_t *__restrict__ kout, uint4 *__restrict__ pout, uint64_t n,
                                                             uint32_t n_tiles, unsigned shift,
                                                             const unsigned long long *__restrict__ digit_base,
                                                             unsigned long long *__restrict__ state, uint32_t epoch,
                                                             uint32_t *__restrict__ ticket, uint32_t *__restrict__ fail,
                                                             const StageSource S, uint32_t *__restrict__ lowout, uint32_t fault)
{
    constexpr bool STAGE = SRC == 1, ROWS = SRC == 2;
    constexpr uint32_t THREADS = RP_THREADS, ITEMS = RP_ITEMS, RADIX = 1u << RP_BITS, WAVES = THREADS / 64, TILE = THREADS * ITEMS;
    __shared__ uint4 sp[TILE];                        // the tile in digit order: payloads (112 KiB) ...
    __shared__ uint32_t skey[TILE];                   // ... and keys (28 KiB): one placing phase, one write-out phase
    __shared__ uint16_t whist[WAVES][RADIX];
    __shared__ uint32_t lstart[RADIX];
    __shared__ unsigned long long goff[RADIX];
    __shared__ uint32_t wsum[RADIX / 64];
    __shared__ uint32_t s_tile;
    const uint32_t tid0 = threadIdx.x;
    if (tid0 == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    uint32_t tile = s_tile;
    uint32_t key[ITEMS];
    uint4 pay[ITEMS];
    // STAGE: a wave takes 448 consecutive dense indices (a "chunk"); k_rs_stage_prepare has left one directory row per chunk
    // (STAGE_ROW words: lane j requests word j -- the digit waves before their look-back, so that it has arrived when they come
    // back): words 0 .. STAGE_WIN-1 = (place in the stage - first dense index) of the chunk's tiles, in order; the next seven =
    // a 448-bit map with a mark at the first index of every tile but the first; then the first dense index behind the row's
    // tiles (n: none) and the first tile.  An element's tile is the number of marks up to its own index; elements behind the
    // row (more than STAGE_WIN tiles in 448 tuples) search dst_off.
    struct Window {
        unsigned long long word;
        bool ok;                     // (wave-uniform) the wave has elements in the tile
    };
    auto fetch_window = [&](uint32_t tl, uint32_t lane, uint32_t wave) __attribute__((always_inline)) {
        Window w{0ull, false};
        if constexpr (STAGE) {
            const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
            const uint64_t chunk = (uint64_t)tl * WAVES + wv;               // (wave-uniform; TILE = WAVES chunks)
            w.ok = tl < n_tiles && chunk * (64 * ITEMS) < n;
            if (w.ok && lane < STAGE_ROW) w.word = S.chunk_dir[chunk * STAGE_ROW + lane];
        }
        return w;
    };
    // requests the elements of tile `tl` into key / pay (STAGE: the raw tuples into pay, see above)
    auto fetch = [&](uint32_t tl, uint32_t lane, uint32_t wave, const Window &w) __attribute__((always_inline)) {
        if constexpr (ROWS) {
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) {   // (raw row: out_hash low / high, pos, record)
                const uint64_t g = (uint64_t)tl * TILE + wave * (64 * ITEMS) + i * 64 + lane;
                pay[i] = (tl < n_tiles && g < n) ? reinterpret_cast<const uint4 *>(S.rows)[g] : uint4{0, 0, 0, 0};
            }
        } else if constexpr (!STAGE) {
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) {
                const uint64_t g = (uint64_t)tl * TILE + wave * (64 * ITEMS) + i * 64 + lane;
                const bool in_range = tl < n_tiles && g < n;
                key[i] = in_range ? kin[g] : ~0u;
                pay[i] = in_range ? pin[g] : uint4{0, 0, 0, 0};
            }
        } else {
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) pay[i] = uint4{0, 0, 0, 0};
            const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
            const uint64_t c0 = ((uint64_t)tl * WAVES + wv) * (64 * ITEMS);   // (wave-uniform)
            if (w.ok) {
                const unsigned long long d_end = __shfl(w.word, (int)(STAGE_WIN + 7), 64);
                uint32_t before = 0;
#pragma unroll
                for (int i = 0; i < (int)ITEMS; ++i) {
                    const unsigned long long m = __shfl(w.word, (int)STAGE_WIN + i, 64);   // (constant lane: v_readlane)
                    const uint64_t g = c0 + i * 64 + lane;
                    const uint32_t r = before + (uint32_t)__popcll(m & ((2ull << lane) - 1ull));   // marks at or before this index
                    before += (uint32_t)__popcll(m);
                    unsigned long long delta = __shfl(w.word, (int)min(r, STAGE_WIN - 1u), 64);
                    if (g < n) {
                        if (g >= d_end) delta = stage_search(S, (uint32_t)__shfl(w.word, (int)(STAGE_WIN + 8), 64) + STAGE_WIN, g);
                        const uint2 h = *reinterpret_cast<const uint2 *>(S.stage_hash + (g + delta));
                        const uint2 km = *reinterpret_cast<const uint2 *>(S.stage_kmer + (g + delta));
                        pay[i] = uint4{h.x, h.y, km.x, km.y};
                    }
                }
            }
        }
    };
    fetch(tile, tid0 & 63u, tid0 >> 6, fetch_window(tile, tid0 & 63u, tid0 >> 6));
    uint32_t fut = 0;                                 // the ticket after the next (taken a tile ahead: see k_rs_pass_p)
    if (tid0 == 0) fut = atomicAdd(ticket, 1u);
    while (tile < n_tiles) {
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));     // (keeps the per-item addresses from being hoisted out of the loop into registers)
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        for (uint32_t i = tid; i < WAVES * RADIX / 2; i += THREADS) (reinterpret_cast<uint32_t *>(&whist[0][0]))[i] = 0;
        const uint64_t t0 = (uint64_t)tile * TILE;
        const uint32_t cnt_tile = (uint32_t)min((uint64_t)TILE, n - t0);
        if constexpr (ROWS) {
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) {   // row -> (key, payload)
                const uint32_t li = wave * (64 * ITEMS) + i * 64 + lane;
                key[i] = li < cnt_tile ? pay[i].y : ~0u;
                pay[i] = uint4{pay[i].x, pay[i].z, pay[i].w, (uint32_t)(t0 + li)};   // OccPay: low, pos, rec, idx
            }
        }
        if constexpr (STAGE) {
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) {   // raw tuple -> (key, payload); the record index goes to the adjacency's array
                const uint32_t li = wave * (64 * ITEMS) + i * 64 + lane;
                unsigned long long h = ((unsigned long long)pay[i].y << 32) | pay[i].x;
                h *= S.mult;
                h ^= h >> 27;
                const uint32_t rec = pay[i].w;
                key[i] = li < cnt_tile ? (uint32_t)(h >> 32) : ~0u;
                pay[i] = uint4{(uint32_t)h, pay[i].z, rec, (uint32_t)(t0 + li)};   // OccPay: low, pos, rec, idx
                if (li < cnt_tile) S.rec_out[t0 + li] = rec;
            }
        }
        __syncthreads();
        uint32_t pos[ITEMS];              // rank inside (wave, digit), later the element's place in the tile's digit order
#pragma unroll
        for (int i = 0; i < (int)ITEMS; ++i) {
            const uint32_t li = wave * (64 * ITEMS) + i * 64 + lane;
            const uint32_t d = (key[i] >> shift) & (RADIX - 1u);
            pos[i] = 0;                             // (ranking by the LDS atomic: see k_rs_pass_p, RANK = 1; all issued, then all read)
            if (li < cnt_tile) pos[i] = atomicAdd(reinterpret_cast<uint32_t *>(&whist[wave][d & ~1u]), 1u << ((d & 1u) << 4));
        }
#pragma unroll
        for (int i = 0; i < (int)ITEMS; ++i) {
            const uint32_t d = (key[i] >> shift) & (RADIX - 1u);
            pos[i] = (pos[i] >> ((d & 1u) << 4)) & 0xFFFFu;
        }
        if (fault && wave == 0) {   // SEQWIN_AMD_FAULT_INJECT=rank (tests): see fault_rank()
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) {
                const uint32_t li = i * 64 + lane;
                const uint32_t d = li < cnt_tile ? ((key[i] >> shift) & (RADIX - 1u)) : ~lane;
                const uint32_t od = __shfl_xor(d, 1, 64), opos = __shfl_xor(pos[i], 1, 64);
                if (od == d) pos[i] = opos;
            }
        }
        __syncthreads();
        uint32_t total = 0, incl = 0;
        unsigned long long *st = state + (size_t)tile * RADIX;
        if (tid < RADIX) {
            const uint32_t d = tid;
#pragma unroll
            for (uint32_t w = 0; w < WAVES; ++w) {
                const uint32_t c = whist[w][d];
                whist[w][d] = (uint16_t)total;
                total += c;
            }
            __hip_atomic_store(&st[d], rse_pack(epoch, tile == 0 ? 2u : 1u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            incl = total;
            for (uint32_t dd = 1; dd < 64; dd <<= 1) {
                const uint32_t up = __shfl_up(incl, dd, 64);
                if (lane >= dd) incl += up;
            }
            if (lane == 63) wsum[wave] = incl;
        }
        __syncthreads();
        uint32_t before = 0;
        if (tid < RADIX) {
            before = incl - total;
            for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
            lstart[tid] = before;
        }
        if (tid == 0) s_tile = fut;
        __syncthreads();
        const uint32_t ntile = s_tile;
        // the tile in digit order, in LDS; the next tile's keys and payloads are requested into the same registers (in flight during
        // the look-back and the write-out)
        auto place_keys = [&](const Window &win) {
#pragma unroll
            for (int i = 0; i < (int)ITEMS; ++i) {
                const uint32_t li = wave * (64 * ITEMS) + i * 64 + lane;
                const uint32_t d = (key[i] >> shift) & (RADIX - 1u);
                const uint32_t at = pos[i] + lstart[d] + whist[wave][d];
                if (li < cnt_tile) {
                    skey[at] = key[i];
                    sp[at] = pay[i];
                }
            }
            fetch(ntile, lane, wave, win);
            if (tid == 0) fut = atomicAdd(ticket, 1u);
        };
        // the four digit waves look back while the other twelve place their keys, and place theirs afterwards (k_rs_pass_p)
        const bool look_wave = tid < RADIX;
        const Window win = fetch_window(ntile, lane, wave);
        if (!look_wave) place_keys(win);
        if (look_wave) {                                  // look-back, RS_LOOK predecessors per step
            const uint32_t d = tid;
            unsigned long long excl = 0;
            if (tile) {
                int64_t t = (int64_t)tile - 1;
                uint32_t spins = 0;
                bool done = false;
                while (!done && t >= 0) {
                    unsigned long long v[RS_LOOK];
#pragma unroll
                    for (int j = 0; j < RS_LOOK; ++j)
                        v[j] = t - j >= 0 ? __hip_atomic_load(&state[(size_t)(t - j) * RADIX + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : rse_pack(epoch, 2u, 0);
                    int j = 0;
#pragma unroll
                    for (; j < RS_LOOK; ++j) {
                        const uint32_t f = rse_flag(v[j], epoch);
                        if (f == 0) break;              // not published yet: poll again from here
                        excl += v[j] & RSE_VAL;
                        if (f == 2) {
                            done = true;
                            break;
                        }
                    }
                    if (done) break;
                    t -= j;
                    if (j < RS_LOOK) {
                        if (++spins > RS_SPIN_LIMIT) {  // (the tile's owner is running -- tickets; see k_rs_pass_p)
                            atomicOr(fail, 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __hip_atomic_store(&st[d], rse_pack(epoch, 2u, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            goff[d] = digit_base[d] + excl - before;
            place_keys(win);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (int)ITEMS; ++j) {            // keys and payloads out, consecutive lanes to consecutive places of a digit
            const uint32_t t = j * THREADS + tid;
            if (t < cnt_tile) {
                const uint32_t k = skey[t];
                const uint64_t dst = goff[(k >> shift) & (RADIX - 1u)] + t;
                const uint4 pv = sp[t];
                kout[dst] = k;
                pout[dst] = pv;
                if (lowout) lowout[dst] = pv.x;   // (the sort's last pass: OccPay::low in an array of its own for the descent sweeps)
            }
        }
        // (the next iteration writes sp / goff / s_tile only behind its own barriers; its first barrier also orders these reads)
        tile = ntile;
    }
}

// digit histograms of all four passes of a 32-bit key array in one sweep (3 GB at 15 000 genomes)
__global__ __launch_bounds__(256) void k_rs_hist32(const uint32_t *__restrict__ keys, uint64_t n, unsigned n_passes,
                                                   unsigned long long *__restrict__ hist)
{
    constexpr uint32_t RADIX = 1u << RP_BITS;
    __shared__ uint32_t h[4 * RADIX];
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256) h[i] = 0;
    __syncthreads();
    const uint64_t chunk = 65536;
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    auto count = [&](uint32_t k) {
        for (unsigned p = 0; p < n_passes; ++p) atomicAdd(&h[p * RADIX + ((k >> (RP_BITS * p)) & (RADIX - 1u))], 1u);
    };
    for (uint64_t i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
        if (i + 3 < i1) {
            const uint4 kk = *reinterpret_cast<const uint4 *>(keys + i);
            count(kk.x); count(kk.y); count(kk.z); count(kk.w);
        } else {
            for (uint64_t j = i; j < i1; ++j) count(keys[j]);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// ... and from rows of (out_hash, pos | record << 32) (SRC = 2)
__global__ __launch_bounds__(256) void k_rs_hist32_rows(const uint4 *__restrict__ rows, uint64_t n, unsigned n_passes,
                                                        unsigned long long *__restrict__ hist)
{
    constexpr uint32_t RADIX = 1u << RP_BITS;
    __shared__ uint32_t h[4 * RADIX];
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256) h[i] = 0;
    __syncthreads();
    const uint64_t chunk = 16384;
    const uint64_t i0 = (uint64_t)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += 1024) {   // four rows in flight per thread
        uint32_t k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = i + 256u * j < i1 ? rows[i + 256u * j].y : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + 256u * j < i1)
                for (unsigned p = 0; p < n_passes; ++p) atomicAdd(&h[p * RADIX + ((k[j] >> (RP_BITS * p)) & (RADIX - 1u))], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// ---- the sketch stage as the first pair pass's input: directory + digit counts -------------------------------------------------
// chunk_tile[c] = the tile that holds dense index 448 c (every chunk start lies in exactly one non-empty tile)
__global__ void k_rs_chunk_tiles(const uint32_t *__restrict__ tile_count, const uint64_t *__restrict__ dst_off, uint32_t n_tiles,
                                 uint32_t *__restrict__ chunk_tile)
{
    constexpr uint64_t CHUNK = 64 * RP_ITEMS;
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    if (T >= n_tiles) return;
    const uint32_t c = tile_count[T];
    const uint64_t d = dst_off[T];
    for (uint64_t m = (d + CHUNK - 1) / CHUNK; m * CHUNK < d + c; ++m) chunk_tile[m] = T;
}

// One wave per chunk of 448 dense indices: lane j looks at tile T0 + j (T0 = chunk_tile: a window of STAGE_WIN tiles); a
// non-empty tile that starts inside the chunk marks its first index in the chunk's 448-bit map and leaves (its place in the
// stage - its first dense index) at its number among the marking tiles -- the directory row k_rs_pair_pass<true> reads
// (layout: see its `fetch`).  The chunk's hashes are read on the way for the digit counts of all passes (k_rs_hist32's job:
// the keys do not exist yet).
__global__ __launch_bounds__(256) void k_rs_stage_prepare(const StageSource S, const uint32_t *__restrict__ chunk_tile, uint64_t n,
                                                          unsigned n_passes, unsigned long long *__restrict__ chunk_dir,
                                                          unsigned long long *__restrict__ hist)
{
    constexpr uint32_t RADIX = 1u << RP_BITS, CHUNK = 64 * RP_ITEMS;
    __shared__ uint32_t h[4 * RADIX];
    __shared__ unsigned long long s_row[4][STAGE_ROW];
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_chunks = (n + CHUNK - 1) / CHUNK;
    for (uint64_t chunk = (uint64_t)blockIdx.x * 4u + wv; chunk < n_chunks; chunk += (uint64_t)gridDim.x * 4u) {
        const uint64_t c0 = chunk * CHUNK;
        const uint32_t T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_tile[chunk]);
        const uint32_t Tj = T0 + lane;
        const bool in_win = lane < STAGE_WIN && Tj < S.n_tiles;
        const unsigned long long dj = lane <= STAGE_WIN ? S.dst_off[min(Tj, S.n_tiles)] : ~0ull;   // (dst_off[n_tiles] = n)
        const unsigned long long oj = in_win ? S.tile_offset[Tj] - dj : 0ull;
        const uint32_t cj = in_win ? S.tile_count[Tj] : 0u;
        const bool marks = in_win && lane >= 1u && cj != 0u && dj < c0 + CHUNK;   // (dj > c0: tile T0 holds c0)
        const unsigned long long bal = __ballot(marks);
        const uint32_t rank = 1u + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane < STAGE_ROW) s_row[wv][lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0u) {
            s_row[wv][0] = oj;
            s_row[wv][STAGE_WIN + 8] = T0;
        }
        if (lane == STAGE_WIN) s_row[wv][STAGE_WIN + 7] = dj;   // first dense index behind the window (n: none)
        if (marks) {
            const uint32_t rel = (uint32_t)(dj - c0);
            atomicOr(&s_row[wv][STAGE_WIN + (rel >> 6)], 1ull << (rel & 63u));
            s_row[wv][rank] = oj;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long word = lane < STAGE_ROW ? s_row[wv][lane] : 0ull;
        if (lane < STAGE_ROW) chunk_dir[chunk * STAGE_ROW + lane] = word;
        __builtin_amdgcn_wave_barrier();   // (the row is zeroed again only after every lane has read it)
        const unsigned long long d_end = __shfl(word, (int)(STAGE_WIN + 7), 64);
        uint32_t before = 0;
        unsigned long long hv[RP_ITEMS];   // (all seven hashes requested before the first is counted)
#pragma unroll
        for (int i = 0; i < RP_ITEMS; ++i) {
            const unsigned long long m = __shfl(word, (int)STAGE_WIN + i, 64);
            const uint64_t g = c0 + i * 64 + lane;
            const uint32_t r = before + (uint32_t)__popcll(m & ((2ull << lane) - 1ull));
            before += (uint32_t)__popcll(m);
            unsigned long long delta = __shfl(word, (int)min(r, STAGE_WIN - 1u), 64);
            hv[i] = 0;
            if (g < n) {
                if (g >= d_end) delta = stage_search(S, T0 + STAGE_WIN, g);
                hv[i] = S.stage_hash[g + delta];
            }
        }
#pragma unroll
        for (int i = 0; i < RP_ITEMS; ++i) {
            if (c0 + i * 64 + lane < n) {
                unsigned long long v = hv[i] * S.mult;       // extend_hashes, hashing_internals.hpp:89-103
                v ^= v >> 27;
                const uint32_t k = (uint32_t)(v >> 32);
                for (unsigned p = 0; p < n_passes; ++p) atomicAdd(&h[p * RADIX + ((k >> (RP_BITS * p)) & (RADIX - 1u))], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_passes * RADIX; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// Look-back state that outlives the sorts: one buffer per (device, stream) -- sorts on one stream follow one another on the
// device --, cleared when it is made and whenever its 16-bit epoch wraps.  r05: a sort holds the buffer's mutex
// from its first epoch to its last launch, so two host threads that sort on the same stream (the NULL stream from two Python
// threads) get disjoint epochs and never see a regrow in between; and the buffers are known to the pool's trim
// (radix_trim_state, called by dev_pool_trim: sw_pool_trim and the out-of-memory retry of dev_alloc give them back too).
struct StateBuf {
    std::mutex mu;
    unsigned long long *p = nullptr;
    size_t words = 0;
    uint32_t epoch = 0;
};
std::mutex &state_registry_mu()
{
    static std::mutex &mu = *new std::mutex;
    return mu;
}
std::map<std::pair<int, hipStream_t>, StateBuf *> &state_registry()
{
    static auto &bufs = *new std::map<std::pair<int, hipStream_t>, StateBuf *>;   // (leaked on purpose, like the pool)
    return bufs;
}
thread_local int g_state_uses = 0;   // state buffers this thread holds right now (radix_trim_state must not try its own mutex)
struct StateUse {   // the buffer, locked for the duration of one sort's enqueue
    StateBuf &b;
    std::unique_lock<std::mutex> lock;
    StateUse(StateBuf &buf, std::unique_lock<std::mutex> &&l) : b(buf), lock(std::move(l)) { ++g_state_uses; }
    StateUse(StateUse &&o) : b(o.b), lock(std::move(o.lock)) { ++g_state_uses; }
    StateUse(const StateUse &) = delete;
    StateUse &operator=(const StateUse &) = delete;
    ~StateUse() { --g_state_uses; }
};
StateUse state_buf(hipStream_t stream, size_t words)
{
    int dev = 0;
    SW_HIP(hipGetDevice(&dev));
    StateBuf *b = nullptr;
    {
        std::lock_guard<std::mutex> lock(state_registry_mu());
        StateBuf *&slot = state_registry()[std::make_pair(dev, stream)];
        if (!slot) slot = new StateBuf;
        b = slot;
    }
    StateUse u(*b, std::unique_lock<std::mutex>(b->mu));
    if (b->words < words) {
        if (b->p) (void)hipFree(b->p);        // (waits for the device: nothing is reading it any more)
        b->p = nullptr;
        b->words = 0;
        const size_t want = words + words / 4;
        SW_HIP(hipMalloc(&b->p, want * 8));
        b->words = want;
        b->epoch = 0;
        SW_HIP(hipMemsetAsync(b->p, 0, want * 8, stream));
    }
    return u;
}
uint32_t next_epoch(StateBuf &b, hipStream_t stream)   // (b.mu held: StateUse)
{
    if (++b.epoch >= 0xFFFFu) {             // every record in the buffer could alias a new epoch: start over
        SW_HIP(hipMemsetAsync(b.p, 0, b.words * 8, stream));
        b.epoch = 1;
    }
    return b.epoch;
}

template <int THREADS, int BITS>
void sort_passes(uint64_t *&keys, uint64_t *&alt, uint64_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream, uint32_t *d_fail,
                 bool perm_hi32, unsigned long long *d_hist_given)
{
    constexpr uint32_t RADIX = 1u << BITS, TILE = THREADS * RS_ITEMS;
    const unsigned n_passes = (end_bit - begin_bit + BITS - 1) / BITS;
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    const char *kind = SW_AB_GETENV("SEQWIN_AMD_RADIX_KERNEL");   // A/B (-DSW_AB): "classic" = one tile per workgroup
    const bool persistent = !(kind && !strcmp(kind, "classic"));
    uint32_t dbg = 0;
#ifdef SW_RADIX_ABLATION     // timing experiments only (tests/tools/sort_time.py with SEQWIN_AMD_RADIX_DEBUG; the output is NOT sorted): 1 no look-back, 2 no stores, 4 no loads, 8 no ranking
    if (const char *e = getenv("SEQWIN_AMD_RADIX_DEBUG")) dbg = (uint32_t)atoi(e) & 15u;
#endif
    if (fault_rank()) dbg |= 16u;
    // resident workgroups of the persistent form (per template instance; taken from the first device that sorts -- any number
    // is correct, tiles are handed out by tickets)
    int grid_p = 0;
    if (persistent) {
        static std::mutex &mu = *new std::mutex;                          // leaked on purpose (see api.hip: pool())
        static std::map<int, int> &grids = *new std::map<int, int>;       // per device (and per template instance)
        int dev = 0;
        SW_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        int &g = grids[dev];
        if (!g) {
            int per_cu = 0;
            hipDeviceProp_t prop;
            SW_HIP(hipGetDeviceProperties(&prop, dev));
            SW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rs_pass_p<THREADS, BITS, 0>, THREADS, 0));
            g = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
        }
        grid_p = g;
    }
    const bool atomic_rank = persistent && rank_mode() == 1;
    DevArray<unsigned long long> hist_own(d_hist_given ? 0 : (size_t)n_passes * RADIX), state(persistent ? 0 : (size_t)n_tiles * RADIX);
    std::unique_ptr<StateUse> su(persistent ? new StateUse(state_buf(stream, (size_t)n_tiles * RADIX)) : nullptr);   // (epoch-tagged records: never cleared per pass)
    StateBuf *sb = su ? &su->b : nullptr;
    struct { unsigned long long *p; } hist{d_hist_given ? d_hist_given : hist_own.p};   // ([pass][digit] counts; scanned in place below)
    const bool unstable = perm_hi32 && persistent && !SW_AB_GETENV("SEQWIN_AMD_RADIX_STABLE_UNSORT");   // (A/B, -DSW_AB: the look-back form)
    DevArray<unsigned long long> cursor(unstable ? (size_t)RADIX * RS_CURSOR_STRIDE : 0);
    DevArray<uint32_t> tickets(n_passes);
    SW_HIP(hipMemsetAsync(tickets.p, 0, tickets.bytes(), stream));
    if (d_hist_given) {
        // (the producer of the keys counted the digits while it wrote them: radix_layout)
    } else if (perm_hi32 && begin_bit >= 32) {
        hipLaunchKernelGGL(k_rs_perm_hist<BITS>, dim3(n_passes), dim3(RADIX), 0, stream, n, begin_bit, end_bit, hist.p);
    } else {
        SW_HIP(hipMemsetAsync(hist.p, 0, hist_own.bytes(), stream));
        hipLaunchKernelGGL(k_rs_hist<BITS>, dim3((unsigned)((n + 32767) / 32768)), dim3(256), 0, stream, keys, n, begin_bit, end_bit,
                           n_passes, hist.p);
    }
    SW_HIP(hipGetLastError());
#ifdef SW_RS_STAMPS
    DevArray<unsigned long long> stamps;
    const size_t n_st = (size_t)(n_tiles / 64 + 1) * 16;
    if (getenv("SEQWIN_AMD_STAMPS") && persistent) stamps.alloc(n_st);
#endif
    for (unsigned p = 0; p < n_passes; ++p) {
        const unsigned sh = begin_bit + BITS * p, bits = std::min<unsigned>(BITS, end_bit - sh);
#ifdef SW_RS_STAMPS
        if (stamps.p) {
            SW_HIP(hipMemsetAsync(stamps.p, 0, stamps.bytes(), stream));
            unsigned long long *ptr = stamps.p;
            SW_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rs_stamps), &ptr, sizeof ptr, 0, hipMemcpyHostToDevice, stream));
        }
#endif
        // (only the FIRST pass may be unstable: the later ones must keep the order the earlier ones made)
        unsigned long long *cur = unstable && p == 0 ? cursor.p : nullptr;
        hipLaunchKernelGGL(k_rs_scan<BITS>, dim3(1), dim3(RADIX), 0, stream, hist.p + (size_t)p * RADIX, cur);
        if (!cur && !persistent) SW_HIP(hipMemsetAsync(state.p, 0, state.bytes(), stream));
        if (persistent) {
            const uint32_t epoch = cur ? 0u : next_epoch(*sb, stream);
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>(n_tiles, (uint64_t)grid_p)), dim3(THREADS), 0, stream, keys, alt, n,
                                   (uint32_t)n_tiles, sh, bits, hist.p + (size_t)p * RADIX, sb->p, tickets.p + p, d_fail, dbg, cur,
                                   RS_CURSOR_STRIDE, 64u, epoch);
            };
            // (an unstable pass may rank by atomics on any device: the order inside a digit is free there)
            if (cur ? !ballot_forced() : atomic_rank) launch(k_rs_pass_p<THREADS, BITS, 1>);
            else launch(k_rs_pass_p<THREADS, BITS, 0>);
        } else {
#ifdef SW_AB
            hipLaunchKernelGGL((k_rs_pass<THREADS, BITS>), dim3((unsigned)n_tiles), dim3(THREADS), 0, stream, keys, alt, n, sh, bits,
                               hist.p + (size_t)p * RADIX, state.p, d_fail);
#endif
        }
        SW_HIP(hipGetLastError());
#ifdef SW_RS_STAMPS
        if (stamps.p) {
            std::vector<unsigned long long> hs(n_st);
            SW_HIP(hipStreamSynchronize(stream));
            SW_HIP(hipMemcpy(hs.data(), stamps.p, stamps.bytes(), hipMemcpyDeviceToHost));
            double acc[8] = {0}, look[3] = {0};
            size_t cnt = 0;
            for (size_t t = 0; t + 16 <= n_st; t += 16) {
                if (!hs[t] || !hs[t + 7]) continue;
                for (int i = 1; i <= 7; ++i) acc[i] += (double)(hs[t + i] - hs[t + i - 1]);
                for (int i = 0; i < 3; ++i) look[i] += (double)hs[t + 8 + i];
                ++cnt;
            }
            static const char *nm[] = {"", "rank", "barrier", "totals+scan", "barrier+scatter", "lookback(d0)", "barrier", "write-out"};
            fprintf(stderr, "[rs stamps] %d x %d bits, pass %u, %zu tiles sampled; mean clocks:", THREADS, BITS, p, cnt);
            double sum = 0;
            for (int i = 1; i <= 7; ++i) { fprintf(stderr, " %s=%.0f", nm[i], cnt ? acc[i] / cnt : 0.0); sum += cnt ? acc[i] / cnt : 0.0; }
            fprintf(stderr, " | tile %.0f | look-back of digit 0: %.1f steps, %.1f polls of an unpublished record, %.1f records behind\n", sum,
                    cnt ? look[0] / cnt : 0.0, cnt ? look[1] / cnt : 0.0, cnt ? look[2] / cnt : 0.0);
        }
#endif
        std::swap(keys, alt);
    }
    // (hist / state go back to the pool here; their next user is ordered after these kernels on this stream)
}

}  // namespace

namespace {
// cursor[(g * RADIX + d) * stride] = first position of digit d inside group g of a permutation sorted so far on the bits above
// this digit: (g << group_bits) + (d << digit_shift)   (one group: group_bits = 64)
__global__ void k_rs_perm_cursors(unsigned long long *__restrict__ cursor, uint32_t n_groups, uint32_t radix, uint32_t stride,
                                  unsigned group_bits, unsigned digit_shift)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * radix) return;
    const uint64_t g = i / radix, d = i % radix;
    cursor[(size_t)i * stride] = (group_bits < 64 ? g << group_bits : 0ull) + (d << digit_shift);
}
}  // namespace

// The unsort's partial sort as TWO unstable passes, most significant digit first: keys[i] >> 32 are a permutation of 0 .. n-1
// and only their grouping by the index bits [low_bits, nbit) matters (the order inside a group is free).  So every bucket's
// place is known in advance -- digit d of the first pass owns positions [d << (low_bits + 8), ...), and inside it digit e of
// the second pass owns [.. + (e << low_bits), ...) -- and a tile claims its places with one atomic add per digit: no counting,
// no scan, no look-back in either pass (an LSD pair of passes needs its second one stable).  Takes two digits of up to 8
// bits (8 < nbit - low_bits <= 16, tiles of 8192 keys inside one first-pass bucket); returns false otherwise.
bool radix_unsort_perm(uint64_t *&keys, uint64_t *&alt, uint64_t n, unsigned low_bits, unsigned nbit, hipStream_t stream, uint32_t *d_fail)
{
    constexpr int THREADS = 512, BITS = 8;
    constexpr uint32_t RADIX = 1u << BITS, TILE = THREADS * RS_ITEMS;
    if (n == 0 || nbit <= low_bits + 8 || nbit > low_bits + 16 || SW_AB_GETENV("SEQWIN_AMD_RADIX_STABLE_UNSORT")) return false;
    if ((1ull << (low_bits + 8)) % TILE) return false;
    const char *kind = SW_AB_GETENV("SEQWIN_AMD_RADIX_KERNEL");
    if (kind && !strcmp(kind, "classic")) return false;
    const unsigned hi_bits = nbit - low_bits - 8;                         // 1 .. 8
    const uint32_t n_groups = (uint32_t)((n + (1ull << (low_bits + 8)) - 1) >> (low_bits + 8));
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    int dev = 0, per_cu = 0;
    SW_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SW_HIP(hipGetDeviceProperties(&prop, dev));
    SW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rs_pass_p<THREADS, BITS, 0>, THREADS, 0));
    const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)std::max(1, per_cu) * std::max(1, prop.multiProcessorCount));
    DevArray<unsigned long long> cur_a((size_t)RADIX * RS_CURSOR_STRIDE), cur_b((size_t)n_groups * RADIX);
    DevArray<uint32_t> tickets(2);
    SW_HIP(hipMemsetAsync(tickets.p, 0, 8, stream));
    hipLaunchKernelGGL(k_rs_perm_cursors, dim3(1), dim3(RADIX), 0, stream, cur_a.p, 1u, RADIX, RS_CURSOR_STRIDE, 64u, low_bits + 8);
    hipLaunchKernelGGL(k_rs_perm_cursors, dim3((n_groups * RADIX + 255) / 256), dim3(256), 0, stream, cur_b.p, n_groups, RADIX, 1u,
                       low_bits + 8, low_bits);
    // (both passes are unstable -- the order inside a bucket is free --, so they rank by LDS atomics on any device)
    auto pass = ballot_forced() ? k_rs_pass_p<THREADS, BITS, 0> : k_rs_pass_p<THREADS, BITS, 1>;
    hipLaunchKernelGGL(pass, dim3(grid), dim3(THREADS), 0, stream, (const uint64_t *)keys, alt, n, (uint32_t)n_tiles,
                       32u + low_bits + 8, hi_bits, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, tickets.p, d_fail,
                       0u, cur_a.p, RS_CURSOR_STRIDE, 64u, 0u);
    std::swap(keys, alt);
    hipLaunchKernelGGL(pass, dim3(grid), dim3(THREADS), 0, stream, (const uint64_t *)keys, alt, n, (uint32_t)n_tiles, 32u + low_bits,
                       8u, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, tickets.p + 1, d_fail, 0u, cur_b.p, 1u,
                       low_bits + 8, 0u);
    std::swap(keys, alt);
    SW_HIP(hipGetLastError());
    return true;
}

bool radix_pairs_available() { return rank_mode() == 1; }   // (the pair passes rank by LDS atomics only)
int radix_rank_mode() { return rank_mode(); }               // 1: LDS atomics (the device passed the self-check), 0: ballots

// Gives the look-back buffers of idle (device, stream) pairs back to the driver; a buffer whose sort is being enqueued right now
// (its mutex is held -- possibly by this very thread, whose allocation failed inside a sort) is skipped.  hipFree waits for the
// device, so passes already enqueued have finished with it.
uint64_t radix_trim_state()
{
    uint64_t freed = 0;
    if (g_state_uses) return 0;   // (called from inside a sort of this thread: an allocation of the sort failed)
    std::lock_guard<std::mutex> lock(state_registry_mu());
    for (auto &kv : state_registry()) {
        StateBuf &b = *kv.second;
        std::unique_lock<std::mutex> use(b.mu, std::try_to_lock);
        if (!use.owns_lock() || !b.p) continue;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != kv.first.first) (void)hipSetDevice(kv.first.first);
        (void)hipFree(b.p);
        if (cur != kv.first.first && cur >= 0) (void)hipSetDevice(cur);
        freed += b.words * 8;
        b.p = nullptr;
        b.words = 0;
        b.epoch = 0;
    }
    return freed;
}

// The always-on order guards of the consumers (index.hip: k_nodes, k_rle_keys, k_check_ascending) found a result of the
// LDS-atomic ranking out of order: the current device ranks by ballots from here on (pairs: rocPRIM), for the rest of the process.
void radix_demote_rank()
{
    int dev = 0;
    SW_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(rank_mu());
    rank_modes()[dev] = 0;
}

// Stable sort of (key32, 16-byte payload) pairs by bits [0, end_bit) of the keys, end_bit a multiple of 8 up to 32 (the node
// sort: all 32).  Double buffers; on return keys / vals point at the sorted data.
void radix_sort_pairs32(uint32_t *&keys, uint32_t *&keys_alt, OccPay *&vals, OccPay *&vals_alt, uint64_t n, unsigned end_bit,
                        hipStream_t stream, uint32_t *d_fail, const StageSource *src, const std::function<void()> &after_first,
                        uint32_t *low_out)
{
    constexpr uint32_t RADIX = 1u << RP_BITS, TILE = RP_THREADS * RP_ITEMS;
    if (n == 0 || end_bit == 0) return;
    if (end_bit > 32 || end_bit % RP_BITS) raise(SW_ERR_RUNTIME, "radix_sort_pairs32: key bits must be a multiple of 8 up to 32");
    if (n >= 0xFFFFFFFFull) raise(SW_ERR_RUNTIME, "radix_sort_pairs32: more than 2^32-2 elements");
    const unsigned n_passes = end_bit / RP_BITS;
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    int grid_p = 0;
    {
        static std::mutex &mu = *new std::mutex;
        static std::map<int, int> &grids = *new std::map<int, int>;
        int dev = 0;
        SW_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        int &g = grids[dev];
        if (!g) {
            int per_cu = 0;
            hipDeviceProp_t prop;
            SW_HIP(hipGetDeviceProperties(&prop, dev));
            SW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rs_pair_pass<0>, RP_THREADS, 0));
            g = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
        }
        grid_p = g;
    }
    DevArray<unsigned long long> hist((size_t)n_passes * RADIX);
    DevArray<uint32_t> tickets(n_passes);
    SW_HIP(hipMemsetAsync(tickets.p, 0, tickets.bytes(), stream));
    SW_HIP(hipMemsetAsync(hist.p, 0, hist.bytes(), stream));
    StageSource stage{};
    DevArray<uint32_t> chunk_tile;                 // (released on return: their next users follow the passes on this stream)
    DevArray<unsigned long long> chunk_dir;
    if (src && src->rows) {
        stage = *src;
        hipLaunchKernelGGL(k_rs_hist32_rows, dim3((unsigned)((n + 16383) / 16384)), dim3(256), 0, stream,
                           reinterpret_cast<const uint4 *>(src->rows), n, n_passes, hist.p);
    } else if (src) {
        constexpr uint64_t CHUNK = 64 * RP_ITEMS;
        static_assert(TILE == (RP_THREADS / 64) * CHUNK, "a tile is one chunk per wave");
        const uint64_t n_chunks = (n + CHUNK - 1) / CHUNK;
        stage = *src;
        chunk_tile.alloc(n_chunks);
        chunk_dir.alloc(n_chunks * STAGE_ROW);
        stage.chunk_dir = chunk_dir.p;
        hipLaunchKernelGGL(k_rs_chunk_tiles, dim3((stage.n_tiles + 255u) / 256u), dim3(256), 0, stream, stage.tile_count, stage.dst_off,
                           stage.n_tiles, chunk_tile.p);
        hipLaunchKernelGGL(k_rs_stage_prepare, dim3((unsigned)std::min<uint64_t>(2048, (n_chunks + 3) / 4)), dim3(256), 0, stream, stage,
                           (const uint32_t *)chunk_tile.p, n, n_passes, chunk_dir.p, hist.p);
    } else {
        hipLaunchKernelGGL(k_rs_hist32, dim3((unsigned)((n + 65535) / 65536)), dim3(256), 0, stream, keys, n, n_passes, hist.p);
    }
    SW_HIP(hipGetLastError());
    StateUse su = state_buf(stream, (size_t)n_tiles * RADIX);
    StateBuf &sb = su.b;
    const StageSource none{};
    const uint32_t fault = fault_rank();
    for (unsigned p = 0; p < n_passes; ++p) {
        hipLaunchKernelGGL(k_rs_scan<RP_BITS>, dim3(1), dim3(RADIX), 0, stream, hist.p + (size_t)p * RADIX, (unsigned long long *)nullptr);
        const uint32_t epoch = next_epoch(sb, stream);
        const dim3 grid((unsigned)std::min<uint64_t>(n_tiles, (uint64_t)grid_p));
        if (p == 0 && src) {
            auto first = src->rows ? k_rs_pair_pass<2> : k_rs_pair_pass<1>;
            hipLaunchKernelGGL(first, grid, dim3(RP_THREADS), 0, stream, (const uint32_t *)nullptr, (const uint4 *)nullptr,
                               keys_alt, reinterpret_cast<uint4 *>(vals_alt), n, (uint32_t)n_tiles, 0u,
                               (const unsigned long long *)hist.p, sb.p, epoch, tickets.p, d_fail, stage,
                               n_passes == 1 ? low_out : (uint32_t *)nullptr, fault);
            SW_HIP(hipGetLastError());
            if (after_first) after_first();
        } else {
            hipLaunchKernelGGL(k_rs_pair_pass<0>, grid, dim3(RP_THREADS), 0, stream, (const uint32_t *)keys,
                               reinterpret_cast<const uint4 *>(vals), keys_alt, reinterpret_cast<uint4 *>(vals_alt), n, (uint32_t)n_tiles,
                               RP_BITS * p, (const unsigned long long *)(hist.p + (size_t)p * RADIX), sb.p, epoch, tickets.p + p, d_fail,
                               none, p + 1 == n_passes ? low_out : (uint32_t *)nullptr, fault);
            SW_HIP(hipGetLastError());
        }
        std::swap(keys, keys_alt);
        std::swap(vals, vals_alt);
    }
}

// which shape a sort of `bits` key bits takes: 0 = 512 threads x 8 bits, 1 = 1024 x 9, 2 = 1024 x 8, 3 = 512 x 9, 4 = 256 x 8 (the last three: A/B)
static int pick_shape(unsigned bits)
{
    const char *e = SW_TEST_GETENV("SEQWIN_AMD_RADIX_BITS");   // A/B: 8 or 9
    const bool nine = e ? atoi(e) == 9 : (bits + 8) / 9 < (bits + 7) / 8;   // 9-bit digits where they save a pass (54 bits: 6 for 7)
    const char *shape = SW_AB_GETENV("SEQWIN_AMD_RADIX_SHAPE");   // A/B (-DSW_AB)
    if (shape && !strcmp(shape, "1024x8")) return 2;
    if (shape && !strcmp(shape, "512x9")) return 3;
    if (shape && !strcmp(shape, "256x8")) return 4;
    return nine ? 1 : 0;
}

// Digit width and number of passes of a sort on `bits` key bits -- for a producer that counts the digits of its keys while
// it writes them (hist[pass][digit], 2^digit_bits counters per pass, pass p = key bits [begin + p * digit_bits, ...)) and
// hands the counts to radix_sort_keys64 instead of the counting sweep.
void radix_layout(unsigned bits, unsigned *digit_bits, unsigned *n_passes)
{
    const int sh = pick_shape(bits);
    *digit_bits = (sh == 1 || sh == 3) ? 9u : 8u;
    *n_passes = (bits + *digit_bits - 1) / *digit_bits;
}

// Stable sort of keys[0, n) by bits [begin_bit, end_bit); keys / alt are a double buffer, on return `keys` points at the
// sorted data and `alt` at the other buffer.  *d_fail (device word, zeroed by the caller) becomes non-zero if a pass gave
// up waiting (the caller checks it at its next host synchronisation: check_sort_failed in index.hip).
void radix_sort_keys64(uint64_t *&keys, uint64_t *&alt, uint64_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream,
                       uint32_t *d_fail, bool perm_hi32, unsigned long long *d_hist_given, unsigned layout_bits)
{
    if (n == 0 || end_bit <= begin_bit) return;
    if (end_bit - begin_bit > 64) raise(SW_ERR_RUNTIME, "radix_sort_keys64: more than 64 key bits");
    // (layout_bits: the upper passes of a radix_layout(layout_bits) sort -- same digit width, begin_bit on a digit boundary)
    switch (pick_shape(layout_bits ? layout_bits : end_bit - begin_bit)) {
#ifdef SW_AB   // (the shapes that lost: 1024 x 8 bits 28.3 ms, 512 x 9 bits 35.5 ms against 25.1 on 745 M 54-bit keys, NOTES.md)
    case 2: sort_passes<1024, 8>(keys, alt, n, begin_bit, end_bit, stream, d_fail, perm_hi32, d_hist_given); break;
    case 3: sort_passes<512, 9>(keys, alt, n, begin_bit, end_bit, stream, d_fail, perm_hi32, d_hist_given); break;
    case 4: sort_passes<256, 8>(keys, alt, n, begin_bit, end_bit, stream, d_fail, perm_hi32, d_hist_given); break;
#endif
    case 1: sort_passes<1024, 9>(keys, alt, n, begin_bit, end_bit, stream, d_fail, perm_hi32, d_hist_given); break;
    default: sort_passes<512, 8>(keys, alt, n, begin_bit, end_bit, stream, d_fail, perm_hi32, d_hist_given); break;
    }
}

}  // namespace sw
