// sketch.hip -- fused ntHash + sliding-window-minimizer kernel for gfx950 (wave64, LDS-staged).
//
// Replaces, for a whole batch of 2-bit packed records resident in HBM, the reference's per-record
//   btllib::minimize_sequence   cpp/vendor/btllib/minimizer.cpp:53-90  (calc_minimizer :14-49)
//   btllib::NtHash::roll/init   cpp/vendor/btllib/nthash_kmer.hpp:315-333, 491-511
//   next_forward/reverse_hash   cpp/vendor/btllib/nthash_kmer.hpp:65-75, 145-155
//   srol / sror / extend_hashes cpp/vendor/btllib/hashing_internals.hpp:29-35, 69-74, 89-103
// and emits (out_hash, pos, record_idx) of every minimizer, the tuples build_worker consumes
// (cpp/src/seqwin/build.cpp:151-168).
//
// Formulation.  Valid k-mers of a record are numbered idx = 0..n_valid-1 in
// increasing position; windows are w consecutive idx values (they span N gaps exactly as the
// reference's ring buffer does).  The reference emits the rightmost minimum of every window whenever its
// position advances; because that position is monotone in the window index this equals "the set of
// elements that are the rightmost minimum of at least one window", in position order.  A workgroup
// owns one TILE = TW consecutive window ends of one record plus a halo of w earlier elements (the generic kernel
// below is the plain form of the scheme; the fast kernel further down keeps hashes in registers):
//   phase 1  each of the 256 threads rolls ntHash over its own run of L consecutive k-mers
//            (k-1 warm-up steps, then one LDS LUT lookup + ~20 integer VALU ops per base) and stores
//            the 64-bit canonical hash of every element in LDS; it also keeps its run minimum and,
//            walking back, the offset of the suffix minimum of its run from every element (1 byte).
//   phase 2  every window [x, e] is (suffix of an earlier run from x) + (whole runs) + (prefix of the
//            owning thread's run up to e): the thread streams its prefix minimum in registers, takes
//            the whole-run part from <= w/L run minima and the suffix part from two LDS reads, and
//            sets a bit for the winner (ds_or, idempotent -> duplicates vanish).
//   phase 3  the bit of the minimizer of the window just before the tile is cleared (the previous
//            tile owns it), bits are counted (workgroup scan) and the tuples are written in position order
//            into the tile's own slot of the stage arrays (tile * slot_cap; a tile with more winners than
//            its slot takes a range of a shared overflow area with one atomicAdd).
// Tiles land slot by slot; index.hip's order pass packs them into (record_idx, pos) order.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device.hpp"

namespace sw {

namespace {

constexpr int BLOCK = 256;   // threads of the generic kernel and of the large fast tile class
constexpr uint32_t MAX_TILES_PER_LAUNCH = 0xFFFFFFu;  // x 256 threads stays below the 2^32 work-item grid limit (15k genomes: 10.2 M tiles, one launch)
constexpr uint32_t L_MAX = 33;  // 256*33 elements * 8 B = 66 KiB of hashes -> two workgroups per CU

// hashing_internals.hpp:128-131
constexpr uint64_t SEED_A = 0x3c8bfbb395c60474ULL, SEED_C = 0x3193c18562a02b4cULL,
                   SEED_G = 0x20323ed082572324ULL, SEED_T = 0x295549f54be24456ULL;
constexpr uint64_t MULTISEED = 0x90b45d39fb6da1faULL;  // :79

uint64_t host_srol1(uint64_t x)  // hashing_internals.hpp:29-35
{
    uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}
uint64_t host_srol(uint64_t x, unsigned d)
{
    for (d %= 1023u; d; --d) x = host_srol1(x);  // periods 33 and 31
    return x;
}

struct SketchArgs {
    const uint32_t *packed;
    const uint64_t *rec_base;
    const uint32_t *rec_seg_off;
    const uint32_t *rec_nvalid;
    const uint32_t *rec_tile_off;   // [R+1] global tile numbering (record-major)
    const uint32_t *cls_tile_off;   // [R+1] tile numbering of the class this launch covers
    const uint32_t *seg_pos;
    const uint32_t *seg_idx;
    const uint64_t *lut;
    const uint64_t *t4;           // [256][2] warm-up table: 4 bases per step (forward, reverse), fast kernel
    uint32_t n_records, k, w, L, TW, n_tiles;
    uint32_t tile_base;           // class tile id of blockIdx.x == 0 (launches are chunked: 2^32 work-item grid limit)
    uint64_t packed_words;
    uint64_t mult;
    uint32_t halo;   // fast class: elements a tile holds before its first owned window end (w rounded up to whole runs)
    uint64_t *stage_hash;
    uint64_t *stage_kmer;
    unsigned long long *cursor;   // entries taken from the shared overflow area (tiles with more than slot_cap winners)
    uint64_t cap;                 // total entries of the stage arrays (slots + overflow area)
    uint64_t ovf_base;            // first entry of the overflow area = n_tiles * slot_cap
    uint32_t slot_cap;            // entries of a tile's own slot at tile * slot_cap
    const uint32_t *cls_tile_rec; // generic kernel, own class: [class tiles] record of every tile
    const TileDesc *cls_desc;     // fast classes (and the generic kernel's list mode over them): [class tiles] descriptors
    uint32_t *tile_count;
    uint64_t *tile_offset;
    uint32_t *ovf_count;          // fast kernel: number of tiles handed over to the generic kernel
    uint32_t *ovf_list;           // fast kernel: their class tile ids; generic kernel in list mode reads it
    uint32_t rc_limit;            // fast kernel: records per run that may be consumed (RC; lower only for tests)
    const uint32_t *list;         // generic kernel: nullptr = own class, else class tile ids of the FAST class
    unsigned long long *stamps;   // timing builds (-DSW_SK_STAMPS): [sampled tile][wave][16] shader-clock stamps, else nullptr
};

// 2-bit base reader over the packed stream (16 bases per 32-bit word).
struct BaseStream {
    const uint32_t *wp;
    uint32_t win;
    uint32_t nleft;
    __device__ __forceinline__ void init(const uint32_t *packed, uint64_t base)
    {
        wp = packed + (base >> 4);
        const uint32_t ph = (uint32_t)base & 15u;
        win = *wp++ >> (2u * ph);
        nleft = 16u - ph;
    }
    __device__ __forceinline__ uint32_t next()
    {
        if (nleft == 0) {
            win = *wp++;
            nleft = 16;
        }
        const uint32_t c = win & 3u;
        win >>= 2;
        --nleft;
        return c;
    }
};

// Split-rotate the 64-bit value (hi:lo) left by one: bits [0,32] and [33,63] rotate separately
// (hashing_internals.hpp:29-35), in 32-bit halves.  The instruction choice is pinned with inline assembly: on gfx950 the
// three-register VOP3 forms the compiler fuses these into (v_bitop3 / v_or3 / v_and_or with three VGPRs, v_lshlrev) issue
// at ~4.2 cycles per wave, v_bitop3 with an inline constant, v_add and v_lshrrev at ~2.4-2.9 (scripts/micro/valu_kinds.hip).
// bitop3:0xe2 with src1 = constant M is the bit-field insert (src0 & M) | (src2 & ~M).
__device__ __forceinline__ void srol1(uint32_t &lo, uint32_t &hi)
{
    uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 31);   // (hi << 1) | (lo >> 31)
    uint32_t u = hi >> 30, l2, nhi, nlo;                   // bit 1 of u = old bit 63
    asm("v_bitop3_b32 %0, %1, 2, %2 bitop3:0xe2" : "=v"(nhi) : "v"(u), "v"(t));      // old bit 63 -> bit 33
    asm("v_add_u32_e32 %0, %1, %1" : "=v"(l2) : "v"(lo));                            // lo << 1
    asm("v_bitop3_b32 %0, %1, 1, %2 bitop3:0xe2" : "=v"(nlo) : "v"(hi), "v"(l2));    // old bit 32 -> bit 0
    lo = nlo;
    hi = nhi;
}
// Inverse (hashing_internals.hpp:69-74).
__device__ __forceinline__ void sror1(uint32_t &lo, uint32_t &hi)
{
    const uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 1);   // (lo >> 1) | (hi << 31): old bit 32 -> bit 31
    const uint32_t t1 = hi >> 1;
    const uint32_t t2 = __builtin_amdgcn_alignbit(t1, hi, 1);    // (hi >> 1) | (old bit 33 << 31): old bit 33 -> bit 63
    uint32_t nhi;
    asm("v_bitop3_b32 %0, %1, 1, %2 bitop3:0xe2" : "=v"(nhi) : "v"(lo), "v"(t2));    // old bit 0 -> bit 32
    lo = nlo;
    hi = nhi;
}

// srol / sror applied N times (1 <= N <= 30) in one go: rotate the low 33 and the high 31 bits by N.
template <int N> __device__ __forceinline__ void srolN(uint32_t &lo, uint32_t &hi)
{
    const uint32_t b32 = hi & 1u, h31 = hi >> 1;
    const uint32_t nlo = (lo << N) | (lo >> (33 - N)) | (b32 << (N - 1));
    const uint32_t nb32 = (lo >> (32 - N)) & 1u;
    const uint32_t nh = ((h31 << N) | (h31 >> (31 - N))) & 0x7FFFFFFFu;
    lo = nlo;
    hi = (nh << 1) | nb32;
}
template <int N> __device__ __forceinline__ void srorN(uint32_t &lo, uint32_t &hi)
{
    const uint32_t b32 = hi & 1u, h31 = hi >> 1;
    const uint32_t nlo = (lo >> N) | (b32 << (32 - N)) | (lo << (33 - N));
    const uint32_t nb32 = (lo >> (N - 1)) & 1u;
    const uint32_t nh = ((h31 >> N) | (h31 << (31 - N))) & 0x7FFFFFFFu;
    lo = nlo;
    hi = (nh << 1) | nb32;
}

// srol^4 / sror^4 for the 4-base warm-up steps, in funnel shifts (same instruction-cost reasoning as srol1 / sror1).
// Bits: lo = [31:0], hi bit 0 = bit 32 (top of the 33-bit word), hi[31:1] = the 31-bit word.
__device__ __forceinline__ void srol4(uint32_t &lo, uint32_t &hi)
{
    const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, 1);     // [bit 32, lo[31:1]]
    const uint32_t nlo = __builtin_amdgcn_alignbit(lo, x, 28);   // (lo << 4) | [bit 32, lo[31:29]]
    const uint32_t r4 = __builtin_amdgcn_alignbit(hi, hi, 28);   // rotl32(hi, 4): bits [31:5] = hi[27:1] are in place
    const uint32_t low5 = __builtin_amdgcn_alignbit(hi >> 28, lo << 3, 31);   // (hi[31:28] << 1) | lo[28]
    uint32_t nhi;
    asm("v_bitop3_b32 %0, %1, 31, %2 bitop3:0xe2" : "=v"(nhi) : "v"(low5), "v"(r4));   // (low5 & 31) | (r4 & ~31)
    lo = nlo;
    hi = nhi;
}
__device__ __forceinline__ void sror4(uint32_t &lo, uint32_t &hi)
{
    uint32_t l2, y, n;
    asm("v_add_u32_e32 %0, %1, %1" : "=v"(l2) : "v"(lo));                          // lo << 1
    asm("v_bitop3_b32 %0, %1, 1, %2 bitop3:0xe2" : "=v"(y) : "v"(hi), "v"(l2));    // [lo[30:0], bit 32]
    const uint32_t nlo = __builtin_amdgcn_alignbit(y, lo, 4);                      // (lo >> 4) | [lo[2:0], bit 32] << 28
    asm("v_bitop3_b32 %0, %1, 16, %2 bitop3:0xe2" : "=v"(n) : "v"(l2), "v"(hi));   // hi with bit 4 := lo[3]
    const uint32_t nhi = __builtin_amdgcn_alignbit(hi >> 1, n, 4);                 // [hi[4:1], hi[31:5], lo[3]]
    lo = nlo;
    hi = nhi;
}

__device__ __forceinline__ uint64_t make64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

__global__ __launch_bounds__(BLOCK, 2) void sketch_generic_kernel(const SketchArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t L = A.L, NE = BLOCK * L, w = A.w, k = A.k;
    uint64_t *H = reinterpret_cast<uint64_t *>(smem);         // [NE]    canonical hash of every element
    uint64_t *RMh = H + NE;                                    // [BLOCK] run minimum (hash)
    uint64_t *LUT = RMh + BLOCK;                               // [40]    roll tables, {F,R} interleaved
    uint32_t *EM = reinterpret_cast<uint32_t *>(LUT + 40);     // [NE/32 + 2] emit bitmap
    uint32_t *MISC = EM + (NE / 32 + 2);                       // [16]
    uint16_t *RMp = reinterpret_cast<uint16_t *>(MISC + 16);   // [BLOCK] run minimum (element index)
    uint8_t *SP = reinterpret_cast<uint8_t *>(RMp + BLOCK);    // [NE]    offset of the run-suffix minimum

    const uint32_t tid = threadIdx.x;

    // ---- which record / which window range (uniform; scalar loads) ---------------------------
    // list mode: redo tiles of the fast class (same window ranges: TW = the fast class's TW <= NE - w)
    if (A.list && A.tile_base + blockIdx.x >= *A.ovf_count) return;   // list length is only known on the device
    const uint32_t ctile = A.list ? A.list[A.tile_base + blockIdx.x] : A.tile_base + blockIdx.x;
    // own class: tiles of a record are numbered t = 0, 1, ...; tiles of the fast classes (list mode) carry descriptors
    uint32_t rec, tile, I0;
    if (A.cls_desc) {
        const TileDesc d = A.cls_desc[ctile];
        rec = d.rec;
        tile = d.gid;                                         // global tile id (order pass)
        I0 = d.i0;                                            // first window end (idx space)
    } else {
        rec = A.cls_tile_rec[ctile];   // (one load instead of a 15-step search: the chain is pure latency)
        const uint32_t t = ctile - A.cls_tile_off[rec];
        tile = A.rec_tile_off[rec] + t;
        I0 = (w - 1) + t * A.TW;
    }
    const uint32_t nvalid = A.rec_nvalid[rec];
    const bool first = I0 == w - 1;                           // first tile of its record
    const uint32_t I1 = (uint32_t)min((uint64_t)I0 + A.TW, (uint64_t)nvalid);   // no wrap near 2^32 k-mers               // one past the last window end
    const uint32_t E0 = first ? 0u : I0 - w;                  // first element held by this tile
    const uint32_t ne = I1 - E0;                              // elements held (<= NE)
    const uint32_t e_first = I0 - E0;                         // tile-local index of the first owned window end
    const uint32_t slo = A.rec_seg_off[rec], shi = A.rec_seg_off[rec + 1];
    const uint64_t rbase = A.rec_base[rec];

    for (uint32_t i = tid; i < NE / 32 + 2; i += BLOCK) EM[i] = 0;
    if (tid < 40) LUT[tid] = A.lut[tid];
    __syncthreads();

    // ---- phase 1: roll ntHash over this thread's run of L elements ------------------------------
    const uint32_t e0 = tid * L;
    const uint32_t n = (e0 < ne) ? min(L, ne - e0) : 0u;
    uint64_t rmin_h = ~0ull;
    uint32_t rmin_e = e0;
    if (n) {
        uint32_t g = E0 + e0;
        uint32_t s = slo;
        {
            uint32_t a = slo, b = shi;  // last segment with seg_idx <= g
            while (b - a > 1) {
                const uint32_t mid = (a + b) >> 1;
                if (A.seg_idx[mid] <= g) a = mid; else b = mid;
            }
            s = a;
        }
        uint32_t e = e0, rem = n;
        while (rem) {
            const uint32_t sidx = A.seg_idx[s];
            const uint32_t send = (s + 1 < shi) ? A.seg_idx[s + 1] : nvalid;
            const uint32_t pos = A.seg_pos[s] + (g - sidx);
            const uint32_t cnt = min(rem, send - g);
            if (cnt == 0) break;  // cannot happen with a well-formed plan; never spin
            const uint64_t b0 = rbase + pos;
            BaseStream in;
            in.init(A.packed, b0);
            uint32_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
            // warm-up: k steps without an outgoing base (base_forward/reverse_hash, nthash_kmer.hpp:22-54,104-133)
            for (uint32_t j = 0; j < k; ++j) {
                const uint32_t idx = 16u | in.next();
                const uint64_t lf = LUT[2 * idx], lr = LUT[2 * idx + 1];
                srol1(flo, fhi);
                flo ^= (uint32_t)lf;
                fhi ^= (uint32_t)(lf >> 32);
                rlo ^= (uint32_t)lr;
                rhi ^= (uint32_t)(lr >> 32);
                sror1(rlo, rhi);
            }
            {
                const uint64_t h = make64(flo, fhi) + make64(rlo, rhi);  // canonical(), hashing_internals.hpp:12-17
                H[e] = h;
                if (h <= rmin_h) { rmin_h = h; rmin_e = e; }
            }
            if (cnt > 1) {
                BaseStream out;
                out.init(A.packed, b0);
                for (uint32_t j = 1; j < cnt; ++j) {
                    // next_forward_hash / next_reverse_hash (nthash_kmer.hpp:65-75,145-155) with the
                    // (in, out) seed pair folded into one 16-entry LUT by XOR-linearity
                    const uint32_t idx = in.next() | (out.next() << 2);
                    const uint64_t lf = LUT[2 * idx], lr = LUT[2 * idx + 1];
                    srol1(flo, fhi);
                    flo ^= (uint32_t)lf;
                    fhi ^= (uint32_t)(lf >> 32);
                    rlo ^= (uint32_t)lr;
                    rhi ^= (uint32_t)(lr >> 32);
                    sror1(rlo, rhi);
                    const uint64_t h = make64(flo, fhi) + make64(rlo, rhi);
                    H[e + j] = h;
                    if (h <= rmin_h) { rmin_h = h; rmin_e = e + j; }
                }
            }
            g += cnt;
            e += cnt;
            rem -= cnt;
            ++s;
        }
        // suffix minima of the run, rightmost on ties (strict '<' walking right to left)
        uint64_t cur = 0;
        uint32_t off = 0;
        for (uint32_t j = n; j-- > 0;) {
            const uint64_t h = H[e0 + j];
            if (j == n - 1 || h < cur) { cur = h; off = j; }
            SP[e0 + j] = (uint8_t)off;
        }
    }
    RMh[tid] = rmin_h;
    RMp[tid] = (uint16_t)rmin_e;
    __syncthreads();

    // ---- phase 2: rightmost minimum of every window ending in this thread's run ------------------
    if (n && e0 + n > w - 1) {
        const uint32_t j0 = (e0 >= w - 1) ? 0u : (w - 1 - e0);
        const uint32_t x0 = e0 + j0 - (w - 1);
        const uint32_t rxA = x0 / L;
        const uint32_t bnd = (rxA + 1) * L;
        uint64_t mA_h = ~0ull, mB_h = ~0ull;
        uint32_t mA_e = 0, mB_e = 0;
        if (rxA < tid) {
            for (uint32_t r = tid; r-- > rxA + 2;) {  // whole runs rxA+2 .. tid-1, right to left
                const uint64_t h = RMh[r];
                if (h < mB_h) { mB_h = h; mB_e = RMp[r]; }
            }
            mA_h = mB_h;
            mA_e = mB_e;
            if (rxA + 1 < tid) {
                const uint64_t h = RMh[rxA + 1];
                if (h < mA_h) { mA_h = h; mA_e = RMp[rxA + 1]; }
            }
        }
        uint64_t pre_h = ~0ull;
        uint32_t pre_e = e0;
        uint32_t prev_arg = 0xFFFFFFFFu;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t e = e0 + j;
            const uint64_t h = H[e];
            if (h <= pre_h) { pre_h = h; pre_e = e; }   // '<=': rightmost wins (minimizer.cpp:36,40)
            if (j < j0) continue;
            const uint32_t x = e - (w - 1);
            uint64_t ch = pre_h;
            uint32_t ce = pre_e;
            if (x < e0) {
                const bool inB = x >= bnd;
                const uint64_t mh = inB ? mB_h : mA_h;
                const uint32_t me = inB ? mB_e : mA_e;
                if (mh < ch) { ch = mh; ce = me; }       // parts further left win only if strictly smaller
                const uint32_t se = (inB ? bnd : bnd - L) + SP[x];
                const uint64_t sh = H[se];
                if (sh < ch) { ch = sh; ce = se; }
            }
            if (e < e_first) {
                MISC[0] = ce;  // minimizer of the window just before this tile: owned by the previous tile
            } else if (ce != prev_arg && ch != ~0ull) {  // minimizer.cpp:44-45
                atomicOr(&EM[ce >> 5], 1u << (ce & 31u));
            }
            prev_arg = ce;
        }
    }
    __syncthreads();
    if (tid == 0 && !first) {
        const uint32_t sarg = MISC[0];
        EM[sarg >> 5] &= ~(1u << (sarg & 31u));
    }
    __syncthreads();

    // ---- phase 3: compact the set bits in position order ------------------------------------------
    uint64_t bits = 0;
    if (n) {
        const uint32_t wd = e0 >> 5, sh = e0 & 31u;
        const uint64_t two = (uint64_t)EM[wd] | ((uint64_t)EM[wd + 1] << 32);
        bits = (two >> sh) & ((n >= 64) ? ~0ull : ((1ull << n) - 1ull));
    }
    const uint32_t cnt = (uint32_t)__popcll(bits);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t incl = cnt;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) MISC[4 + wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
    for (uint32_t i = 0; i < BLOCK / 64; ++i) {
        const uint32_t v = MISC[4 + i];
        if (i < wave) wave_off += v;
        total += v;
    }
    // Output range: the tile's own slot (no atomic, no round trip); only a tile with more winners than the slot holds
    // takes a range of the shared overflow area.  (One atomic per tile on one address caps the whole launch at
    // ~80 M tiles/s on MI355X, measured: that was the sketch kernel's real bound.)
    uint64_t base = (uint64_t)tile * A.slot_cap;
    if (__builtin_amdgcn_readfirstlane(total) > A.slot_cap) {   // workgroup-uniform
        if (tid == 0) {
            const unsigned long long got = atomicAdd(A.cursor, (unsigned long long)total);
            MISC[8] = (uint32_t)got;
            MISC[9] = (uint32_t)(got >> 32);
        }
        __syncthreads();
        base = A.ovf_base + make64(MISC[8], MISC[9]);
    }
    if (tid == 0) {
        A.tile_count[tile] = total;
        A.tile_offset[tile] = base;
    }
    if (cnt && base + total <= A.cap) {
        uint64_t o = base + wave_off + (incl - cnt);
        while (bits) {
            const uint32_t j = (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            const uint32_t e = e0 + j;
            const uint32_t g = E0 + e;
            uint32_t a = slo, b = shi;
            while (b - a > 1) {
                const uint32_t mid = (a + b) >> 1;
                if (A.seg_idx[mid] <= g) a = mid; else b = mid;
            }
            const uint32_t pos = A.seg_pos[a] + (g - A.seg_idx[a]);
            A.stage_hash[o] = H[e];   // canonical hash; k_order applies extend_hashes (hashing_internals.hpp:89-103)
            A.stage_kmer[o] = (uint64_t)pos | ((uint64_t)rec << 32);
            ++o;
        }
    }
}

// ================================================================================================
// Fast path: tiles of records that are ONE valid segment (no invalid base inside the tile's reach),
// k <= KF, run length L a power of two (32, or 16 for 16 <= w < 32).
//  * the tile's 2-bit words are staged in LDS with coalesced loads; every lane starts on the same
//    bit phase (L is a multiple of 16 bases = one word), so word refills are WAVE-UNIFORM (scalar
//    branch, no divergence) and are fetched one word ahead of use;
//  * the roll LUT read of step j+1 is issued before the arithmetic of step j (software pipeline);
//  * the L hashes of a lane stay in REGISTERS.  What other lanes need from a run is only its suffix
//    minima, a step function with ~ln(L) steps ("suffix records": elements smaller than everything to
//    their right in the run).  Each lane publishes a 32-bit position mask of its records and the
//    hashes of the first RC records from the right (RC * 8 B instead of L * 8 B of LDS per lane);
//    the suffix minimum from offset ox is record number popc(mask >> ox) - 1, at offset
//    ox + ctz(mask >> ox).  A tile in which some run has more than RC records (probability ~1e-4 per
//    run on random hashes) is not finished here: its id is appended to an overflow list and the tile
//    is redone, exactly, by sketch_generic_kernel in list mode (run_sketch).
//    LDS per workgroup drops from 79 KiB to ~30 KiB -> 5 workgroups (20 waves) per CU instead of 2.
// ================================================================================================
constexpr uint32_t KF = 256;
constexpr uint32_t KPAIR = 32;   // k up to which the warm-up reads the pair table (16 pair positions x 16 rows x 16 B = the 4 KiB of t4)
constexpr uint32_t RC = 12;   // published suffix records per run (12 x 8 B: the most that keeps 5 workgroups per CU)
// A/B switches (timing builds: tests/tools/build_variant.sh <name> -DSW_SK_AB=bits; 0 = the shipped kernel)
//   1 the suffix-record pass tests the slot address against the end of the lane's area (r02) instead of clamping it
//   2 wave scans of the emit counts through ds_bpermute (__shfl_up, r02) instead of DPP row shifts
//   4 whole-run minima of the window pass read in a counted loop (r02) instead of five loads in flight
//   8 the winner of the window before the tile cleared in the bitmap by one thread between two barriers (r02)
//  16 the masked moves of the window pass behind v_cmp + s_and_saveexec (r02) instead of v_cmpx (EXEC written by the compare)
//  32 the suffix-record pass as the compiler lays it out (compare, EXEC round trip, branch) instead of v_cmpx statements
//  64 the suffix-record pass clamps its slot address (v_min per element, r03a) instead of writing on into the neighbours' slots
// 128 the emit loop tests every position in every lane instead of skipping, by a scalar test, those no lane of the wave emits
#ifndef SW_SK_AB
#define SW_SK_AB 0
#endif
#ifndef SW_SK_SLEEP
#define SW_SK_SLEEP 0      // experiment: every wave sleeps ~64 x this many cycles once per tile (is the kernel bound by wave latency?)
#endif
constexpr bool SK_CLAMP = !(SW_SK_AB & 1), SK_DPP_SCAN = !(SW_SK_AB & 2), SK_RUNMIN_UNROLLED = !(SW_SK_AB & 4),
               SK_TWO_BARRIERS = (SW_SK_AB & 8) != 0, SK_CMPX = !(SW_SK_AB & 16),
               SK_SUFFIX_ASM = !(SW_SK_AB & 32), SK_SUFFIX_MIN = (SW_SK_AB & 64) != 0, SK_EMIT_GUARD = !(SW_SK_AB & 128);

#ifdef SW_SK_STAMPS
constexpr uint32_t STAMP_EVERY = 512, STAMP_SLOTS = 16;   // every 512th tile of a launch writes its waves' phase times
#define SK_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (A.stamps && (ctile % STAMP_EVERY) == 0) {   /* (uniform) */                                          \
            const uint32_t wv_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));              \
            unsigned long long *p_ = A.stamps + ((size_t)(ctile / STAMP_EVERY) * (B / 64) + wv_) * STAMP_SLOTS + (i); \
            const unsigned long long c_ = __builtin_readcyclecounter();                                          \
            if ((threadIdx.x & 63u) == 0) __builtin_nontemporal_store(c_, p_);                                   \
        }                                                                                                        \
    } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

// inclusive prefix sum over the 64 lanes of a wave in DPP row shifts / broadcasts (six dependent VALU instructions, no trip
// through the LDS crossbar: a __shfl_up scan is six dependent ds_bpermute round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    if (SK_DPP_SCAN) {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1 (zeros come in from below the row)
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8: inclusive within rows of 16
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
        return v;
    }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}

// OR over the 64 lanes of a wave (DPP row shifts / broadcasts as in wave_incl_scan; the result is read from lane 63): wave-uniform
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int L, int B> struct FastCfg {
    static constexpr int NE = B * L;
    static constexpr int NSTG = NE / 16 + KF / 16 + 4;
    static constexpr int NEM = NE / 32 + 2;
    static constexpr size_t off_REC = 0;
    static constexpr size_t off_RMh = off_REC + (size_t)B * RC * 8;
    static constexpr size_t off_LUT = off_RMh + B * 8;
    static constexpr size_t off_STG = off_LUT + 40 * 8;
    static constexpr size_t off_EM = off_STG + (size_t)NSTG * 4;
    static constexpr size_t off_MASK = off_EM + (size_t)NEM * 4;
    static constexpr size_t off_MISC = off_MASK + B * 4;
    static constexpr size_t off_RMp = off_MISC + 16 * 4;
    static constexpr size_t bytes = off_RMp + B * 2;
};

template <int L, int B> __global__ __launch_bounds__(B, 4) void sketch_fast_kernel(const SketchArgs A)
{
    using C = FastCfg<L, B>;
    constexpr uint32_t LM = L - 1;
    constexpr uint32_t LSH = (L == 32) ? 5 : (L == 16) ? 4 : (L == 8) ? 3 : 2;
    // (r04) L = 8 for 8 <= w < 16 and L = 4 for 4 <= w < 8: the window logic needs w >= L (a window then holds the start of the
    // lane's run, or lies to the left of it), so small windows take short runs; a lane's first base is then no longer on a word
    // boundary of the staged stream for every lane (bit phase per lane instead of per wave: the funnel shifts take a register).
    static_assert(L == 32 || L == 16 || L == 8 || L == 4, "run length must be 4, 8, 16 or 32");
    constexpr int NWL = (L + 15) / 16;                // staged words that hold a run's own bases
    static_assert((size_t)B * RC * 8 >= 4096, "the warm-up table shares the suffix-record area");
    __shared__ __align__(16) unsigned char smem[C::bytes];   // static: LDS addresses are compile-time constants
    uint64_t *REC = reinterpret_cast<uint64_t *>(smem + C::off_REC);
    uint64_t *RMh = reinterpret_cast<uint64_t *>(smem + C::off_RMh);
    uint64_t *LUT = reinterpret_cast<uint64_t *>(smem + C::off_LUT);
    uint32_t *STG = reinterpret_cast<uint32_t *>(smem + C::off_STG);
    uint32_t *EM = reinterpret_cast<uint32_t *>(smem + C::off_EM);
    uint32_t *MASK = reinterpret_cast<uint32_t *>(smem + C::off_MASK);
    uint32_t *MISC = reinterpret_cast<uint32_t *>(smem + C::off_MISC);
    uint16_t *RMp = reinterpret_cast<uint16_t *>(smem + C::off_RMp);

    const uint32_t tid = threadIdx.x, w = A.w, k = A.k;

    const uint32_t ctile = A.tile_base + blockIdx.x;
    // ONE 32-byte scalar load tells the workgroup everything about its tile (r02: four per-tile words, then the record's base
    // and length -- two dependent round trips before the first byte of the tile could be requested)
    const TileDesc D = A.cls_desc[ctile];
    const uint32_t rec = D.rec, tile = D.gid, ne = D.ne;
    const bool first = D.i0 == w - 1;                  // first tile of its record
    const uint32_t e_first = first ? w - 1 : A.halo;   // tile-local index of the first owned window end (halo >= w, whole runs when w > L)
    const uint32_t ph = (uint32_t)D.bfirst & 15u;
    const uint64_t word0 = D.bfirst >> 4;
    SK_STAMP(0);

    {   // stage the tile's packed words (coalesced), clear the emit bitmap, load the LUT and the warm-up table
        // Everything the tile needs from global memory is REQUESTED first and stored to LDS afterwards: one round trip.  (r02:
        // the word loop, the LUT and the table each waited for their own loads before the next were issued -- five round trips,
        // 5 000 of a wave's 27 000 cycles, r03 stamps.)
        const uint32_t nw = (ph + ne + k + 14) / 16 + 2;      // words the tile reads (<= NSTG)
        const uint32_t nq = (nw + 3) / 4;                     // in groups of four: one 16-byte load per thread
        struct __attribute__((packed, aligned(4))) Words4 { uint32_t x, y, z, w; };
        static_assert(C::NSTG % 4 == 0 && (C::off_STG % 16) == 0, "16-byte stores into the staging area");
        constexpr uint32_t QI = ((uint32_t)C::NSTG / 4 + B - 1) / B;    // groups per thread (1 for 256 threads, 3 for 64)
        constexpr uint32_t TI = 256 / B;                                  // warm-up table rows per thread
        Words4 pw[QI];
#pragma unroll
        for (uint32_t q = 0; q < QI; ++q) {
            const uint32_t g = tid + q * B;
            const uint64_t gw = word0 + 4ull * g;
            pw[q] = Words4{0, 0, 0, 0};
            if (g < nq) {
                if (gw + 4 <= A.packed_words) {
                    pw[q] = *reinterpret_cast<const Words4 *>(A.packed + gw);
                } else {                                      // the batch's last words
                    if (gw < A.packed_words) pw[q].x = A.packed[gw];
                    if (gw + 1 < A.packed_words) pw[q].y = A.packed[gw + 1];
                    if (gw + 2 < A.packed_words) pw[q].z = A.packed[gw + 2];
                }
            }
        }
        const uint64_t lutv = (tid < 40) ? A.lut[tid] : 0ull;
        ulonglong2 t4v[TI];
#pragma unroll
        for (uint32_t q = 0; q < TI; ++q) t4v[q] = reinterpret_cast<const ulonglong2 *>(A.t4)[tid + q * B];
        // (tested only here, so that the whole descriptor is ONE scalar load and the tile's loads are on their way)
        if (D.flags & 1u) return;                      // its reach crosses invalid bases: the generic kernel does it (gap list)
        for (uint32_t i = tid; i < (uint32_t)C::NEM; i += B) EM[i] = 0;
        if (tid == 0) MISC[1] = 0;
#pragma unroll
        for (uint32_t q = 0; q < QI; ++q) {
            const uint32_t g = tid + q * B;
            if (g < nq) *reinterpret_cast<uint4 *>(STG + 4 * g) = make_uint4(pw[q].x, pw[q].y, pw[q].z, pw[q].w);
        }
        if (tid < 40) LUT[tid] = lutv;
        // the 4 KiB warm-up table lives in LDS while the runs are hashed, in the space the suffix records take afterwards
        // (five dependent gathers per lane from global memory were ~4 % of the kernel's time, all of it latency)
#pragma unroll
        for (uint32_t q = 0; q < TI; ++q) reinterpret_cast<ulonglong2 *>(REC)[tid + q * B] = t4v[q];
    }
    __syncthreads();
    SK_STAMP(1);
    const uint32_t e0 = tid * L;
    const uint32_t n = (e0 < ne) ? min((uint32_t)L, ne - e0) : 0u;
    uint64_t h[L];

    if (n) {
        // ---- phase 1: ntHash over this lane's L k-mers -------------------------------------------
        const uint32_t boff = ph + tid * L;           // the lane's first base in the staged stream
        const uint32_t *wp = STG + (boff >> 4);
        const uint32_t lph = boff & 15u;              // (= ph, wave-uniform, when L is a multiple of 16)
        uint32_t icur = wp[0] >> (2u * lph), inxt = wp[1], inl = 16u - lph;
        const uint32_t *iwp = wp + 2;
        auto next_in = [&]() -> uint32_t {
            if (inl == 0) { icur = inxt; inxt = *iwp++; inl = 16; }
            const uint32_t c = icur & 3u;
            icur >>= 2;
            --inl;
            return c;
        };
        uint32_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
        auto apply = [&](uint64_t lf, uint64_t lr) {
            srol1(flo, fhi);
            flo ^= (uint32_t)lf;
            fhi ^= (uint32_t)(lf >> 32);
            rlo ^= (uint32_t)lr;
            rhi ^= (uint32_t)(lr >> 32);
            sror1(rlo, rhi);
        };
        // warm-up: the hashes of the first k-mer of the run.
        if (k <= KPAIR) {
            // (r04) k <= 32: no rotate at all.  F = XOR_i srol^(k-1-i)(S[b_i]) and R = XOR_i srol^i(S[~b_i])
            // (nthash_kmer.hpp:22-54, 104-133), so a PAIR of bases at positions (2p, 2p + 1) contributes one row of a
            // position-dependent table -- 16 rows of {F, R} per pair position, ceil(k / 2) <= 16 positions: the same 4 KiB the
            // 4-base table takes, in the same place (get_plan writes whichever the k needs).  Per pair: one bit-field, one shift,
            // one ds_read_b128, four XORs -- 66 VALU at k = 21 against ~150 for one single step + five 4-base steps with their
            // split rotates (r03 stamps: 2 944 of a wave's 22 486 cycles).  The lane's bases are brought to bit 0 of two words
            // by the wave-uniform phase; all rows of a batch of eight positions are requested before the first is used.
            const uint32_t np = (k + 1u) >> 1;                       // uniform
            const uint32_t so = 2u * lph;
            const uint32_t b0 = __builtin_amdgcn_alignbit(wp[1], wp[0], so);   // bases 0..15 of the run
            const uint32_t b1 = __builtin_amdgcn_alignbit(wp[2], wp[1], so);   // bases 16..31
            const unsigned char *T2 = reinterpret_cast<const unsigned char *>(REC);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint32_t bw = half ? b1 : b0;
                if ((uint32_t)(half * 8) < np) {                     // uniform
                    ulonglong2 te[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((uint32_t)(half * 8 + i) < np) {
                            const uint32_t off = (i == 0 ? (bw << 4) : (bw >> (4 * i - 4))) & 0xF0u;   // 16 B per row
                            te[i] = *reinterpret_cast<const ulonglong2 *>(T2 + (half * 8 + i) * 256 + off);
                        }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((uint32_t)(half * 8 + i) < np) {
                            flo ^= (uint32_t)te[i].x;
                            fhi ^= (uint32_t)(te[i].x >> 32);
                            rlo ^= (uint32_t)te[i].y;
                            rhi ^= (uint32_t)(te[i].y >> 32);
                        }
                }
            }
        } else {
            // warm-up (first k-mer of the run): k = r + 4q bases; r single-base steps through the roll LUT,
            // then q steps of 4 bases through the 256-entry table t4 (F' = srol^4(F) ^ F4[b], R' = sror^4(R) ^ R4[b]),
            // read from LDS one step ahead of use
            // All table rows of a batch are requested first (their addresses depend only on the bases) and the dependent
            // rotate / xor chain runs afterwards: one exposed LDS round trip per batch instead of one per step.
            {
                const uint32_t r = k & 3u;                      // uniform
                ulonglong2 ts[3];
    #pragma unroll
                for (int i = 0; i < 3; ++i)
                    if ((uint32_t)i < r) {
                        const uint32_t idx = 16u | next_in();
                        ts[i] = *reinterpret_cast<const ulonglong2 *>(LUT + 2 * idx);
                    }
    #pragma unroll
                for (int i = 0; i < 3; ++i)
                    if ((uint32_t)i < r) apply(ts[i].x, ts[i].y);
            }
            {
                auto next4 = [&]() -> uint32_t {             // next 4 bases as one byte (uniform control flow)
                    uint32_t b;
                    if (inl >= 4) {
                        b = icur & 0xFFu;
                        icur >>= 8;
                        inl -= 4;
                    } else {
                        b = (icur | (inxt << (2u * inl))) & 0xFFu;
                        icur = inxt >> (2u * (4u - inl));
                        inl += 12;
                        inxt = *iwp++;
                    }
                    return b;
                };
                const uint32_t q = k >> 2;
                const ulonglong2 *T4 = reinterpret_cast<const ulonglong2 *>(REC);
                constexpr int WB = 8;                           // rows in flight (32 VGPRs, free at this point of the kernel)
                for (uint32_t s4 = 0; s4 < q; s4 += WB) {
                    const uint32_t nb = min((uint32_t)WB, q - s4);   // uniform
                    ulonglong2 te[WB];
    #pragma unroll
                    for (int i = 0; i < WB; ++i)
                        if ((uint32_t)i < nb) te[i] = T4[next4()];
    #pragma unroll
                    for (int i = 0; i < WB; ++i)
                        if ((uint32_t)i < nb) {
                            srol4(flo, fhi);
                            flo ^= (uint32_t)te[i].x;
                            fhi ^= (uint32_t)(te[i].x >> 32);
                            sror4(rlo, rhi);
                            rlo ^= (uint32_t)te[i].y;
                            rhi ^= (uint32_t)(te[i].y >> 32);
                        }
                }
            }
        }
        // Rolls 1..L-1 of this lane take in-base (k + j - 1) and out-base (j - 1) of the run.  Both streams are
        // re-aligned once to the lane's first base (funnel shifts by the wave-uniform phases), so that the base
        // pair of every roll is a compile-time bit field: no per-step refill test, 4 VALU per LUT index.
        // The two 2-bit fields are interleaved once per word into 4-bit LUT rows ((out << 2) | in): rows of the even
        // bases in `ev`, of the odd bases in `od` -> one shift + one mask per roll.
        uint32_t ev[NWL], od[NWL];
        {
            const uint32_t so = 2u * lph;
            const uint32_t pk = lph + k;
            const uint32_t *wi = wp + (pk >> 4);
            const uint32_t si = 2u * (pk & 15u);
#pragma unroll
            for (int q = 0; q < NWL; ++q) {
                const uint32_t ob = __builtin_amdgcn_alignbit(wp[q + 1], wp[q], so);
                const uint32_t ib = __builtin_amdgcn_alignbit(wi[q + 1], wi[q], si);
                ev[q] = (ib & 0x33333333u) | ((ob & 0x33333333u) << 2);
                od[q] = ((ib >> 2) & 0x33333333u) | (ob & 0xCCCCCCCCu);
            }
        }
        auto lut_off = [&](int r) -> uint32_t {          // roll r (1-based): byte offset of LUT row, 16 B per row
            const int b = r - 1, q = b >> 4, sh = 4 * ((b & 15) >> 1);
            const uint32_t m = (b & 1) ? od[q] : ev[q];
            // (r03: the nibble taken to bits [7:4] by ONE sub-dword instruction -- v_and_b32_sdwa / v_lshlrev_b32_sdwa on the selected
            //  byte -- instead of shift + mask: 85.0-85.3 against 85.2 ms, no gain; not kept)
            return (sh >= 4 ? (m >> (sh - 4)) : (m << 4)) & 0xF0u;
        };
        const unsigned char *LUTb = reinterpret_cast<const unsigned char *>(LUT);
        SK_STAMP(2);   // warm-up done
        uint64_t lf, lr;
        {
            const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(LUTb + lut_off(1));
            lf = e.x;
            lr = e.y;
        }
#pragma unroll
        for (int j = 0; j < L; ++j) {
            {   // canonical() = F + R: one v_lshl_add_u64 (5.0 cycles) instead of v_add_co + v_addc_co (2 x 4.4)
                const uint64_t f64 = make64(flo, fhi), r64 = make64(rlo, rhi);
                uint64_t s64;
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(s64) : "v"(f64), "v"(r64));   // volatile: computed here, not sunk
                h[j] = s64;
            }
            if (j + 1 < L) {
                uint64_t nlf = 0, nlr = 0;
                if (j + 2 < L) {                    // LUT entry of the roll after next, issued before this roll's math
                    const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(LUTb + lut_off(j + 2));
                    nlf = e.x;
                    nlr = e.y;
                }
                apply(lf, lr);
                lf = nlf;
                lr = nlr;
                // pin the rolling state here: exactly one LUT read in flight, and the arithmetic is not sunk
                asm volatile("" : "+v"(flo), "+v"(fhi), "+v"(rlo), "+v"(rhi) : : "memory");
            }
        }
    }
    SK_STAMP(3);   // roll loop done
    __syncthreads();   // every wave is done with the warm-up table: its space now takes the suffix records
    SK_STAMP(4);
    if (SW_SK_SLEEP) __builtin_amdgcn_s_sleep(SW_SK_SLEEP);
    if (n) {
        // ---- suffix records of the run, right to left (registers only) ---------------------------
        uint64_t cur = 0;
        uint32_t mask = 0, cnt = 0;
        uint64_t *rec = REC + tid * RC;
        if (__all(n == (uint32_t)L)) {              // wave-uniform: full runs, no per-step edge predicates
            cur = h[L - 1];
            mask = 1u << (L - 1);
            rec[0] = cur;
            // the record slot is tracked as an LDS byte address (one add per record, no shift-add); slots past RC - 1
            // are not written and the count is recovered from the address
            const uint32_t ra0 = (uint32_t)(C::off_REC + (size_t)tid * RC * 8), ra_end = ra0 + RC * 8;
            uint32_t ra = ra0 + 8;
            if (SK_CLAMP && SK_SUFFIX_ASM) {
                // (r03) one statement per element: the compare writes EXEC (v_cmpx), the record is taken under it -- hash to the
                // lane's next slot, position bit, slot address advanced and clamped to the last slot -- and EXEC is restored from a
                // loop-invariant copy: seven instructions, one of them scalar, no branch (the compiler's form: nine, with an EXEC
                // round trip that waits for the compare and a branch around the body, at every element)
                using lds_ptr = __attribute__((address_space(3))) unsigned char *;
                const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;            // LDS address of the kernel's static area
                uint32_t wa = lds0 + ra;                                             // absolute LDS address of the next slot
                const uint32_t wa_last = lds0 + ra_end - 8;
                const unsigned long long ex_all = __builtin_amdgcn_read_exec();
                if (SK_SUFFIX_MIN) {
#pragma unroll
                    for (int j = L - 2; j >= 0; --j)
                        asm volatile("v_cmpx_lt_u64_e32 vcc, %[hj], %[cur]\n\t"
                                     "v_mov_b64_e32 %[cur], %[hj]\n\t"
                                     "v_or_b32_e32 %[mask], %[bit], %[mask]\n\t"
                                     "ds_write_b64 %[wa], %[hj]\n\t"
                                     "v_add_u32_e32 %[wa], 8, %[wa]\n\t"
                                     "v_min_u32_e32 %[wa], %[wa], %[last]\n\t"
                                     "s_mov_b64 exec, %[ex]"
                                     : [cur] "+v"(cur), [mask] "+v"(mask), [wa] "+v"(wa)
                                     : [hj] "v"(h[j]), [bit] "n"(1u << j), [last] "v"(wa_last), [ex] "s"(ex_all)
                                     : "vcc", "memory");
                } else {
                    // (r03) no clamp either: a lane with more than RC records writes on into the slots of the lanes behind it (the
                    // last lane: into at most (L - RC) * 8 bytes of RMh, which follows REC) -- but such a lane has cnt > rc_limit
                    // and sends the whole tile to the generic kernel below, so nothing of what it overwrote is ever read
                    static_assert(C::off_RMh == C::off_REC + (size_t)B * RC * 8 && (size_t)B * 8 >= (size_t)L * 8,
                                  "the records of the last lane may run on into RMh, and nowhere else");
#pragma unroll
                    for (int j = L - 2; j >= 0; --j)
                        asm volatile("v_cmpx_lt_u64_e32 vcc, %[hj], %[cur]\n\t"
                                     "v_mov_b64_e32 %[cur], %[hj]\n\t"
                                     "v_or_b32_e32 %[mask], %[bit], %[mask]\n\t"
                                     "ds_write_b64 %[wa], %[hj]\n\t"
                                     "v_add_u32_e32 %[wa], 8, %[wa]\n\t"
                                     "s_mov_b64 exec, %[ex]"
                                     : [cur] "+v"(cur), [mask] "+v"(mask), [wa] "+v"(wa)
                                     : [hj] "v"(h[j]), [bit] "n"(1u << j), [ex] "s"(ex_all)
                                     : "vcc", "memory");
                }
                cnt = (uint32_t)__popc(mask);
            } else if (SK_CLAMP) {
                // the slot address stops at the lane's last slot: a 13th record overwrites the 12th, and such a run (more records
                // than published, counted from the mask) sends its tile to the generic kernel anyway -- one v_min per record
                // instead of a compare and an EXEC round trip (r03: 104.6 -> 103.4 ms at 15 000 genomes)
                const uint32_t ra_last = ra_end - 8;
#pragma unroll
                for (int j = L - 2; j >= 0; --j) {
                    if (h[j] < cur) {               // strictly smaller than everything to its right
                        cur = h[j];
                        mask |= 1u << j;
                        *reinterpret_cast<uint64_t *>(smem + ra) = cur;
                        ra = min(ra + 8u, ra_last);
                    }
                }
                cnt = (uint32_t)__popc(mask);
            } else {
#pragma unroll
                for (int j = L - 2; j >= 0; --j) {
                    if (h[j] < cur) {                   // strictly smaller than everything to its right
                        cur = h[j];
                        mask |= 1u << j;
                        if (ra < ra_end) *reinterpret_cast<uint64_t *>(smem + ra) = cur;
                        ra += 8;
                    }
                }
                cnt = (ra - ra0) >> 3;
            }
        } else {
#pragma unroll
            for (int j = L - 1; j >= 0; --j) {
                const bool take = ((uint32_t)j < n) && ((uint32_t)j == n - 1 || h[j] < cur);
                if (take) {
                    cur = h[j];
                    mask |= 1u << j;
                    if (cnt < RC) rec[cnt] = cur;
                    ++cnt;
                }
            }
        }
        const uint32_t off = (uint32_t)__builtin_ctz(mask);   // the leftmost record is the run minimum
        MASK[tid] = mask;
        RMh[tid] = cur;                              // the leftmost record is the (rightmost) run minimum
        RMp[tid] = (uint16_t)(e0 + off);
        if (cnt > A.rc_limit) MISC[1] = 1;                   // more records than published: redo this tile exactly
    }
    SK_STAMP(5);   // suffix records published
    __syncthreads();
    if (MISC[1]) {                                   // uniform: hand the tile to the generic kernel (list mode)
        if (tid == 0) {
            const uint32_t slot = atomicAdd(A.ovf_count, 1u);
            A.ovf_list[slot] = ctile;
            A.tile_count[tile] = 0;
            A.tile_offset[tile] = 0;
        }
        return;
    }

    // ---- phase 2: rightmost minimum of every window ending in this lane's run -------------------
    SK_STAMP(6);
    if (n && e0 + n > w - 1) {
        const uint32_t j0 = (e0 >= w - 1) ? 0u : (w - 1 - e0);
        const uint32_t x0 = e0 + j0 - (w - 1);
        const uint32_t rxA = x0 >> LSH;
        const uint32_t bnd = (rxA + 1) << LSH;
        uint64_t mA_h = ~0ull, mB_h = ~0ull;
        uint32_t mA_e = 0, mB_e = 0, maskA = 0, maskB = 0;
        if (rxA < tid) {
            constexpr uint32_t WR = 6;   // whole runs read at once (w = 200, L = 32: five)
            if (SK_RUNMIN_UNROLLED && __all(rxA + 1 < tid && tid - rxA - 2 <= WR)) {
                // The kernel is bound by the latency of a wave's own instruction stream (r03: a sleep of 1 000 cycles per tile
                // costs 3.5 %), and the counted loop below is one LDS round trip per run: here the minima of all runs, then the
                // positions of the two winners, are requested together -- two round trips (r03 stamps: 1 144 -> ~300 cycles).
                const uint32_t nw = tid - rxA - 2;       // runs every window of the lane holds whole: tid - 1 down to rxA + 2
                uint64_t v[WR];
#pragma unroll
                for (uint32_t i = 0; i < WR; ++i) v[i] = RMh[max(tid - 1u - i, rxA + 1u)];   // (beyond nw: run rxA + 1, not looked at)
                const uint64_t hA = RMh[rxA + 1];
                maskA = MASK[rxA];
                maskB = MASK[rxA + 1];
                uint32_t br = rxA + 1;                   // run of the rightmost minimum (rxA + 1: none)
#pragma unroll
                for (uint32_t i = 0; i < WR; ++i)
                    if (i < nw && v[i] < mB_h) { mB_h = v[i]; br = tid - 1u - i; }   // right to left, strictly smaller: the rightmost wins
                const uint32_t eB = RMp[br], eA = RMp[rxA + 1];
                mB_e = nw ? eB : 0u;
                mA_h = mB_h;
                mA_e = mB_e;
                if (hA < mA_h) { mA_h = hA; mA_e = eA; }
            } else {
                for (uint32_t r = tid; r-- > rxA + 2;) {
                    const uint64_t hh = RMh[r];
                    if (hh < mB_h) { mB_h = hh; mB_e = RMp[r]; }
                }
                mA_h = mB_h;
                mA_e = mB_e;
                maskA = MASK[rxA];
                if (rxA + 1 < tid) {
                    const uint64_t hh = RMh[rxA + 1];
                    if (hh < mA_h) { mA_h = hh; mA_e = RMp[rxA + 1]; }
                    maskB = MASK[rxA + 1];
                }
            }
        }
        // Window [x, e] = left region [x, e0) (earlier runs) + own prefix [e0, e].  The left region only
        // shrinks while this lane advances, so its rightmost minimum `lc` stays valid until it leaves the
        // window; it is recomputed (suffix records of run rx + whole-run minima) only then -- about
        // 0.2 times per lane per tile -- instead of at every step.
        uint64_t pre_h = ~0ull, lc_h = ~0ull;
        uint32_t pre_e = e0, lc_e = 0;
        uint32_t prev_arg = 0xFFFFFFFFu;
        const uint32_t xb = e0 - (w - 1);   // x of step j is xb + j (mod 2^32; only used when j >= j0)
        const uint64_t *recA = REC + rxA * RC;          // suffix records of run rxA; run rxA + 1 follows
        uint32_t lc_d = 0;                               // lc_e - xb: lc leaves the window of step j when lc_d < j
        auto recompute = [&](uint32_t x, bool inB) {   // inB: x lies in run rxA + 1 (x >= bnd)
            lc_h = inB ? mB_h : mA_h;
            lc_e = inB ? mB_e : mA_e;
            const uint32_t rx = inB ? rxA + 1 : rxA;
            const uint64_t *rp = inB ? recA + RC : recA;
            const uint32_t ox = x & LM;
            const uint32_t tb = (inB ? maskB : maskA) >> ox;     // records at offsets >= ox (never 0)
            const uint32_t slot = (uint32_t)__popc(tb) - 1u;     // records to the right of the answer
            const uint64_t sh = rp[slot];                        // slot < RC: overflow tiles left above
            if (sh < lc_h) {                                     // further left: only if strictly smaller
                lc_h = sh;
                lc_e = (rx << LSH) + ox + (uint32_t)__builtin_ctz(tb);
            }
            lc_d = lc_e - xb;
        };
        auto mark = [&](bool left, uint32_t ce) {                 // rarely taken: the minimizer changed
            const uint64_t ch = left ? lc_h : pre_h;
            if (ch != ~0ull) atomicOr(&EM[ce >> 5], 1u << (ce & 31u));   // minimizer.cpp:44-45
        };
        // wave-uniform fast variant: every lane of the wave has a full run, all its windows exist and are
        // owned by this tile, and the left region is never empty -> no per-step edge predicates.  Every window
        // of a lane contains the first element of its run (w > L), so with h[0] != 2^64-1 in all lanes no window
        // minimum is the excluded value (minimizer.cpp:44-45) and that test is dropped as well.
        // Lanes of the (run-aligned) halo own no window; the last of them evaluates the window that ends just before
        // the tile's first own one (its winner is cleared below: the previous tile emits it).
        const bool owner = e0 >= e_first, last_halo = e0 + (uint32_t)L == e_first;
        const bool wave_full = __all(n == (uint32_t)L && (owner || last_halo) && h[0] != ~0ull) && w > (uint32_t)L;
        SK_STAMP(7);   // whole-run minima, masks
        if (wave_full) {
            if (last_halo) {
                recompute(xb + (uint32_t)(L - 1), xb + (uint32_t)(L - 1) >= bnd);
                const uint64_t rm_h = RMh[tid];                         // minimum of the own run, rightmost among equals
                MISC[0] = (lc_h < rm_h) ? lc_e : (uint32_t)RMp[tid];
            }
            if (owner) {
                // Winners inside the own run are collected in a register (bit j = element e0 + j, the lane's own
                // bits of the emit bitmap) and published once; a left winner is marked when it is (re)computed:
                // the own prefix minimum only falls while it stays current, so it wins some window iff it wins
                // the first one.  No per-step LDS atomic, no "did the winner change" test (marks are idempotent).
                uint32_t own = 0, pre_bit = 0;
                const unsigned long long ex_owner = __builtin_amdgcn_read_exec();   // the owner lanes: EXEC of every statement of the loop
                const uint32_t jb = ((w - 1u) & LM) ? ((w - 1u) & LM) : (uint32_t)L;   // wave-uniform (kernel argument)
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    // prefix minimum, '<=' for the newcomer (rightmost wins): lanes with pre_h >= h[j] take h[j] and bit j.
                    // Written with EXEC masking (restored inside the statement): v_cndmask issues at ~4.2 cycles per
                    // wave on gfx950, a masked v_mov / v_or at ~2.5 (scripts/micro/valu_kinds.hip).
                    // (r03) v_cmpx writes EXEC itself: one scalar instruction per masked statement (the restore from a loop-invariant
                    // copy) instead of two, and no scalar instruction that waits for the compare's result
                    unsigned long long sv;
                    if (SK_CMPX)
                        asm volatile("v_cmpx_ge_u64_e32 vcc, %[pre], %[hj]\n\t"
                                     "v_mov_b64_e32 %[pre], %[hj]\n\t"
                                     "v_mov_b32_e32 %[pb], %[bit]\n\t"
                                     "s_mov_b64 exec, %[ex]"
                                     : [pre] "+v"(pre_h), [pb] "+v"(pre_bit)
                                     : [hj] "v"(h[j]), [bit] "n"(1u << j), [ex] "s"(ex_owner)
                                     : "vcc");
                    else
                    asm volatile("v_cmp_ge_u64_e32 vcc, %[pre], %[hj]\n\t"
                                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                                 "v_mov_b64_e32 %[pre], %[hj]\n\t"
                                 "v_mov_b32_e32 %[pb], %[bit]\n\t"
                                 "s_mov_b64 exec, %[sv]"
                                 : [pre] "+v"(pre_h), [pb] "+v"(pre_bit), [sv] "=&s"(sv)
                                 : [hj] "v"(h[j]), [bit] "n"(1u << j)
                                 : "vcc", "scc");   // s_and_saveexec writes SCC
                    if (j == 0 || lc_d < (uint32_t)j) {
                        // e0 is a multiple of L, so every lane's x = e0 + j - (w - 1) enters run rxA + 1 at the same step
                        // jb: which run the lookup goes to is a scalar branch, not five per-lane selects
                        if ((uint32_t)j >= jb) recompute(xb + j, true);
                        else recompute(xb + j, false);
                        if (lc_h < pre_h) atomicOr(&EM[lc_e >> 5], 1u << (lc_e & 31u));
                    }
                    // the own prefix minimum wins this window unless the left region holds a strictly smaller hash
                    if (SK_CMPX)
                        asm volatile("v_cmpx_ge_u64_e32 vcc, %[lc], %[pre]\n\t"
                                     "v_or_b32_e32 %[own], %[own], %[pb]\n\t"
                                     "s_mov_b64 exec, %[ex]"
                                     : [own] "+v"(own)
                                     : [lc] "v"(lc_h), [pre] "v"(pre_h), [pb] "v"(pre_bit), [ex] "s"(ex_owner)
                                     : "vcc");
                    else
                    asm volatile("v_cmp_ge_u64_e32 vcc, %[lc], %[pre]\n\t"
                                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                                 "v_or_b32_e32 %[own], %[own], %[pb]\n\t"
                                 "s_mov_b64 exec, %[sv]"
                                 : [own] "+v"(own), [sv] "=&s"(sv)
                                 : [lc] "v"(lc_h), [pre] "v"(pre_h), [pb] "v"(pre_bit)
                                 : "vcc", "scc");   // s_and_saveexec writes SCC
                }
                atomicOr(&EM[(tid * L) >> 5], own << ((tid * L) & 31u));   // the lane's L bits of the emit bitmap
            }
        } else {
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const uint32_t e = e0 + j;
                if ((uint32_t)j < n && h[j] <= pre_h) { pre_h = h[j]; pre_e = e; }
                if ((uint32_t)j >= j0 && (uint32_t)j < n) {
                    const uint32_t x = xb + j;
                    if (x >= e0) {
                        lc_h = ~0ull;                                    // no left region (w == L): never wins
                    } else if ((uint32_t)j == j0 || lc_e < x) {
                        recompute(x, x >= bnd);
                    }
                    const bool left = lc_h < pre_h;
                    const uint32_t ce = left ? lc_e : pre_e;
                    if (e < e_first) {
                        if (e + 1 == e_first) MISC[0] = ce;                  // (a run-aligned halo holds more than one such window)
                    } else if (ce != prev_arg) {
                        mark(left, ce);
                    }
                    prev_arg = ce;
                }
            }
        }
    }
    SK_STAMP(8);   // window loop done
    __syncthreads();
    if (SK_TWO_BARRIERS) {
        if (tid == 0 && !first) {
            const uint32_t sarg = MISC[0];
            EM[sarg >> 5] &= ~(1u << (sarg & 31u));
        }
        __syncthreads();
    }
    SK_STAMP(9);

    // ---- phase 3: compact the set bits in position order (hashes come from registers) -----------
    uint32_t bits = 0;
    if (n) {
        bits = EM[(tid * L) >> 5] >> ((tid * L) & 31u);
        if (L < 32) bits &= (1u << (L & 31)) - 1u;
        if (n < (uint32_t)L) bits &= (1u << n) - 1u;
        if (!SK_TWO_BARRIERS && !first) {
            // the winner of the window just before the tile belongs to the previous tile: its owner lane drops it from its
            // own bits (r02: one thread cleared it in the bitmap between two barriers -- a barrier and a serial LDS round trip more)
            const uint32_t sarg = MISC[0];
            if ((sarg >> LSH) == tid) bits &= ~(1u << (sarg & LM));
        }
    }
    const uint32_t cnt = (uint32_t)__popc(bits);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t wave_bits = SK_EMIT_GUARD ? wave_or(bits) : 0u;   // positions some lane of this wave emits (scalar)
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane == 63) MISC[4 + wave] = incl;
    __syncthreads();
    SK_STAMP(10);
    uint32_t wave_off = 0, total = 0;
    for (uint32_t i = 0; i < B / 64; ++i) {
        const uint32_t v = MISC[4 + i];
        if (i < wave) wave_off += v;
        total += v;
    }
    // Output range: the tile's own slot (no atomic, no round trip); only a tile with more winners than the slot holds
    // takes a range of the shared overflow area.  (One atomic per tile on one address caps the whole launch at
    // ~80 M tiles/s on MI355X, measured: that was the sketch kernel's real bound.)
    uint64_t base = (uint64_t)tile * A.slot_cap;
    if (__builtin_amdgcn_readfirstlane(total) > A.slot_cap) {   // workgroup-uniform
        if (tid == 0) {
            const unsigned long long got = atomicAdd(A.cursor, (unsigned long long)total);
            MISC[8] = (uint32_t)got;
            MISC[9] = (uint32_t)(got >> 32);
        }
        __syncthreads();
        base = A.ovf_base + make64(MISC[8], MISC[9]);
    }
    if (tid == 0) {
        A.tile_count[tile] = total;
        A.tile_offset[tile] = base;
    }
    if (cnt && base + total <= A.cap) {
        // uniform 64-bit bases in SGPRs + a 32-bit lane offset: one shift per store instead of 64-bit address arithmetic
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        uint64_t *const sh = A.stage_hash + make64(blo, bhi);
        uint64_t *const sk = A.stage_kmer + make64(blo, bhi);
        uint32_t ob = (wave_off + (incl - cnt)) * 8u;   // byte offset inside the tile's range (< 2^32: a range holds <= NE entries)
        const uint32_t kpos = D.kpos + e0;
        // the canonical hash is staged; out_hash = extend_hashes(h) (hashing_internals.hpp:89-103: one 64-bit multiply +
        // xor-shift) is applied by k_order, which touches every tuple anyway and is HBM-bound
#define SK_EMIT(J)                                                                                                            \
    if ((J) < L && ((bits >> ((J) & 31)) & 1u)) {                                                                             \
        *reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(sh) + ob) = h[(J) < L ? (J) : 0];                     \
        *reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(sk) + ob) = make64(kpos + (uint32_t)(J), rec);        \
        ob += 8u;                                                                                                             \
    }
        if (SK_EMIT_GUARD) {
            // (r03) a scalar test in front of every position: the 53 % of positions no lane of the wave emits at cost two scalar
            // instructions instead of a lane test (v_and + v_cmp + EXEC round trip).  The wave-wide OR is six DPP steps; through
            // LDS (r03, first try) the same guard lost 0.5 %, as a scalar jump into the bodies (a while / switch over the set
            // bits) it costs six more VGPRs -- four waves per SIMD, 116.9 ms
#pragma unroll
            for (int j = 0; j < L; ++j)
                if (wave_bits & (1u << j)) {
                    asm volatile("" ::: "memory");      // (keeps the scalar test apart from the lane test)
                    SK_EMIT(j)
                }
        } else {
#pragma unroll
            for (int j = 0; j < L; ++j) { SK_EMIT(j) }
        }
#undef SK_EMIT
    }
    SK_STAMP(11);
}

// n_occ = sum of the per-tile counts (the tiles take no shared cursor any more): one atomic per workgroup of 4096 tiles.
__global__ void k_sum_counts(const uint32_t *__restrict__ tile_count, uint32_t n_tiles, unsigned long long *__restrict__ total)
{
    unsigned long long v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_tiles; i += (uint64_t)gridDim.x * blockDim.x)
        v += tile_count[i];
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(total, v);
}

// ---- the launch plan's per-tile tables, written on the device (r03) ---------------------------------------------------
// The host decides per RECORD (how many tiles of which class); these kernels expand that into one descriptor per tile.
// Before, a host loop over 750 k records / 10.2 M tiles and 160 MB of uploads took 110-120 ms per (batch, k, w) at 15 000
// genomes -- 0.6 of a device build, paid by every first build.
struct PlanTilesArgs {
    const uint32_t *cls_off;        // [R + 1] first tile of every record in this class
    const uint32_t *big_off;        // [R + 1] the same for the 256-thread class (a record's 64-thread tiles follow its big ones)
    const uint32_t *rec_tile_off;   // [R + 1] global tile numbering
    const uint32_t *rec_nvalid, *rec_seg_off, *seg_pos, *seg_idx;
    const uint64_t *rec_base;
    uint32_t n_records, n_tiles, cls, w, TW0, TW1, halo;
    TileDesc *desc;
    uint32_t *gap_list, *n_gap;
};

__device__ __forceinline__ uint32_t record_of_tile(const uint32_t *__restrict__ off, uint32_t n_records, uint32_t i)
{
    uint32_t lo = 0, hi = n_records;   // last record with off[r] <= i (records without tiles share their successor's offset)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_plan_tiles(const PlanTilesArgs P)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_tiles) return;
    const uint32_t r = record_of_tile(P.cls_off, P.n_records, i);
    const uint32_t t = i - P.cls_off[r];
    const uint32_t n_big = P.big_off[r + 1] - P.big_off[r];
    const uint32_t nv = P.rec_nvalid[r];
    const uint32_t TW = P.cls ? P.TW1 : P.TW0;
    const uint64_t I0 = (uint64_t)(P.w - 1) + (P.cls ? (uint64_t)n_big * P.TW0 + (uint64_t)t * P.TW1 : (uint64_t)t * P.TW0);
    const uint64_t I1 = min(I0 + TW, (uint64_t)nv);
    const uint64_t E0 = (I0 == P.w - 1) ? 0 : I0 - P.halo;
    // the segment that holds idx E0; the tile is a fast one if its reach [E0, I1) stays inside it
    uint32_t a = P.rec_seg_off[r], b = P.rec_seg_off[r + 1];
    const uint32_t s_end = b;
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (P.seg_idx[mid] <= E0) a = mid; else b = mid;
    }
    const uint64_t seg_end = (a + 1 < s_end) ? P.seg_idx[a + 1] : nv;
    TileDesc d;
    d.ne = (uint32_t)(I1 - E0);
    d.i0 = (uint32_t)I0;
    d.rec = r;
    d.gid = P.rec_tile_off[r] + (P.cls ? n_big : 0u) + t;
    d.flags = 0;
    const uint32_t pos0 = P.seg_pos[a] - P.seg_idx[a];   // idx g <-> pos0 + g inside the segment
    d.kpos = pos0 + (uint32_t)E0;
    d.bfirst = P.rec_base[r] + pos0 + E0;
    if (I1 > seg_end) {
        d.flags = 1;
        P.gap_list[atomicAdd(P.n_gap, 1u)] = i;
    }
    P.desc[i] = d;
}

// record of every tile of the generic class
__global__ void k_plan_gen_rec(const uint32_t *__restrict__ gen_off, uint32_t n_records, uint32_t n_tiles, uint32_t *__restrict__ tile_rec)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) tile_rec[i] = record_of_tile(gen_off, n_records, i);
}

size_t lds_bytes_for(uint32_t L)
{
    const size_t NE = (size_t)BLOCK * L;
    return NE * 8 + BLOCK * 8 + 40 * 8 + (NE / 32 + 2) * 4 + 16 * 4 + BLOCK * 2 + NE;
}

}  // namespace

Plan &get_plan(sw_batch &b, uint64_t k64, uint64_t w64, bool *cached)
{
    check_kw(k64, w64);
    // A window above SW_MAX_WINDOW does not fit the tiles (a tile holds w elements of halo), and from about half of that on
    // the halo eats the tile (window ends per 8192-element tile: 7168 at w = 1024, 4096 at w = 4096).  The minimizers of w
    // are a subset of those of any smaller window w' (the rightmost minimum of a window is the rightmost minimum of every
    // sub-window that contains it), so for w > SW_WINDOW_SPLIT the tile kernels run with w' = 1024 and order_tuples
    // (index.hip) picks the minimizers of w out of that sparse set.  Measured on 2.46 Gbp (tests/tools/lw_check.py),
    // whole build, direct / two-step: w = 1500 5.6 / 6.0 ms, 2048 7.1 / 5.9, 3000 7.3 / 5.8, 4096 10.0 / 5.7.
    // SEQWIN_AMD_WINDOW_SPLIT="T,B" (tests, A/B): windows above T <= SW_MAX_WINDOW go through base B <= T.
    uint32_t split_at = SW_WINDOW_SPLIT, split_base = 1024;
    if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_WINDOW_SPLIT")) {
        unsigned t = 0, bs = 0;
        if (sscanf(e, "%u,%u", &t, &bs) == 2 && bs >= 1 && bs <= t && t <= SW_MAX_WINDOW) {
            split_at = t;
            split_base = bs;
        }
    }
    const uint32_t k = (uint32_t)k64, w_full = (uint32_t)std::min<uint64_t>(w64, 0xFFFFFFFFull);   // (no record has 2^32 k-mers)
    const uint32_t w = w_full > split_at ? split_base : w_full;
    std::lock_guard<std::mutex> lock(b.plan_mu);
    auto key = std::make_pair(k, w_full);
    auto it = b.plans.find(key);
    if (cached) *cached = it != b.plans.end();
    if (it != b.plans.end()) return it->second;
    const auto t_begin = std::chrono::steady_clock::now();

    Plan p;
    p.k = k;
    p.w = w;
    p.w_full = w_full;
    // generic class: any record; run length odd (conflict-free 64-bit LDS accesses at stride L)
    uint32_t L = std::min<uint32_t>(w, L_MAX);
    if ((L & 1u) == 0) --L;
    p.L = L;
    p.NE = BLOCK * L;
    p.TW = p.NE - w;
    p.lds_bytes = lds_bytes_for(L);
    // fast class: single-segment records, k <= KF, power-of-two run length
    const char *force = SW_TEST_GETENV("SEQWIN_AMD_SKETCH");   // "generic" disables the fast path (debug / A-B)
    // (r04: runs of 8 and 4 for windows below 16 -- the fast kernel needs w >= L; SEQWIN_AMD_SKETCH=nosmall: the generic kernel, as before)
    const bool small_ok = !(force && !strcmp(force, "nosmall"));
    p.Lf = (k <= KF && !(force && !strcmp(force, "generic")))
               ? (w >= 32 ? 32u : (w >= 16 ? 16u : (small_ok && w >= 8 ? 8u : (small_ok && w >= 4 ? 4u : 0u)))) : 0u;
    if (p.Lf == 32 && force && !strcmp(force, "fast16")) p.Lf = 16;   // A/B: half the LDS per workgroup
    // Overflow tiles of the fast class are redone by the generic kernel, whose run length must be odd and <= w and
    // whose tile (256 * Lg elements) must hold the fast tile: Lg = Lf + 1 when w > Lf; when w == Lf the fast tiles
    // are made one run-column smaller instead (256 * (Lf - 1) elements) and Lg = Lf - 1.
    p.Lg_list = p.Lf ? (w > p.Lf ? p.Lf + 1 : p.Lf - 1) : 0;
    if (p.Lf && (uint64_t)BLOCK * std::min(p.Lf, p.Lg_list) < 2ull * w) p.Lf = 0;   // (only reachable through the fast16 A/B switch)
    // fast tiles start their halo on a run boundary (w rounded up to a multiple of L when w > L): the lanes of the halo
    // then hold no owned window at all and the first wave of a tile can run the wave-uniform window pass as well
    p.halo_f = (p.Lf && w > p.Lf) ? (w + p.Lf - 1) / p.Lf * p.Lf : w;
    p.fc[0].B = BLOCK;
    p.fc[0].TW = p.Lf ? BLOCK * std::min(p.Lf, p.Lg_list) - p.halo_f : 0;
    // 64-thread tiles for short records: only while the halo stays below a quarter of such a tile (w <= 512 at L = 32)
    p.fc[1].B = 64;
    p.fc[1].TW = (p.Lf && 4u * p.halo_f <= 64u * std::min(p.Lf, p.Lg_list) && !(force && !strcmp(force, "big")))
                     ? 64u * std::min(p.Lf, p.Lg_list) - p.halo_f : 0;
    p.mult = 1ULL ^ ((uint64_t)k * MULTISEED);

    const HostBatch &h = b.host;
    const size_t R = h.rec_len.size();
    // per RECORD: valid k-mers, segments, and how many tiles of which class; the per-tile tables follow on the device
    std::vector<uint32_t> rec_seg_off(R + 1, 0), rec_nvalid(R, 0), rec_tile_off(R + 1, 0), gen_off(R + 1, 0), off0(R + 1, 0), off1(R + 1, 0),
        seg_pos, seg_idx;
    seg_pos.reserve(R);
    seg_idx.reserve(R);
    uint64_t tiles = 0, tiles_g = 0, tiles0 = 0, tiles1 = 0;
#ifdef SW_AB
    const bool tails = force && !strcmp(force, "tails");
#else
    const bool tails = false;   // (A/B loser: -DSW_AB builds only)
#endif
    for (size_t r = 0; r < R; ++r) {
        rec_seg_off[r] = (uint32_t)seg_pos.size();
        rec_tile_off[r] = (uint32_t)tiles;
        gen_off[r] = (uint32_t)tiles_g;
        off0[r] = (uint32_t)tiles0;
        off1[r] = (uint32_t)tiles1;
        uint64_t nv = 0;
        for (uint32_t q = h.rec_run_off[r]; q < h.rec_run_off[r + 1]; ++q) {
            if (h.run_len[q] < k) continue;
            seg_pos.push_back(h.run_pos[q]);
            seg_idx.push_back((uint32_t)nv);
            nv += h.run_len[q] - k + 1;
        }
        rec_nvalid[r] = (uint32_t)nv;
        p.n_valid += nv;
        // minimize_sequence's guard (minimizer.cpp:56-58) is implied: n_valid <= len-k+1
        if (nv >= w) {
            const uint64_t windows = nv - w + 1;
            p.n_windows += windows;
            // Fast classes: tiles that lie in ONE valid segment (idx <-> pos is affine there).  A tile costs its workgroup's
            // time whatever its fill, and a 64-thread tile costs about a third of a 256-thread one (measured,
            // tests/tools/fragment_timing.py: 3.6-4.1 ns against 11.0-12.5 ns per tile).  A record is cut, whichever is
            // cheaper in tile count x cost, into (a) 256-thread tiles or (c) 64-thread tiles only.  (b) -- full 256-thread
            // tiles + 64-thread tiles for the tail -- is implemented (every tile carries its own descriptor) but measured
            // SLOWER on the default workload (96 kbp contigs, 12 + 1 tiles instead of 13: 3.85 against 3.72 ms): a nearly empty
            // 256-thread tile leaves the VALU to its neighbours and costs well under a full one, so the model above does not
            // hold for tails; it stays behind SEQWIN_AMD_SKETCH=tails for experiments.
            bool fast = false;
            uint64_t n_big = 0, n_small = 0;
            if (p.Lf) {
                const uint64_t TW0 = p.fc[0].TW, TW1 = p.fc[1].TW;
                const uint64_t nb = windows / TW0, rem = windows % TW0;
                const uint64_t cost_a = (nb + (rem ? 1 : 0)) * 100;
                const uint64_t cost_b = (TW1 && tails) ? nb * 100 + (rem + TW1 - 1) / TW1 * 32 : ~0ull;
                const uint64_t cost_c = TW1 ? (windows + TW1 - 1) / TW1 * 32 : ~0ull;
                n_big = nb + (rem ? 1 : 0);                                                                  // (a)
                if (cost_b < cost_a && cost_b <= cost_c) { n_big = nb; n_small = (rem + TW1 - 1) / TW1; }   // (b)
                else if (cost_c < cost_a) { n_big = 0; n_small = (windows + TW1 - 1) / TW1; }               // (c)
                // a record with invalid bases is cut into the same tiles; those whose reach [E0, I1) crosses a gap are
                // listed for the generic kernel's list mode (the fast kernels skip them) -- unless most of its tiles
                // would be, then the whole record goes to the generic class.  Only such records are walked tile by tile here.
                const uint32_t s0 = rec_seg_off[r], s1 = (uint32_t)seg_pos.size();
                if (s1 - s0 <= 1) {
                    fast = true;
                } else {
                    uint32_t sgm = s0;
                    uint64_t gap_free = 0;
                    for (uint64_t t = 0; t < n_big + n_small; ++t) {
                        const uint64_t I0 = (w - 1) + (t < n_big ? t * TW0 : n_big * TW0 + (t - n_big) * TW1);
                        const uint64_t I1 = std::min<uint64_t>(I0 + (t < n_big ? TW0 : TW1), nv);
                        const uint64_t E0 = (I0 == w - 1) ? 0 : I0 - p.halo_f;
                        while (sgm + 1 < s1 && seg_idx[sgm + 1] <= E0) ++sgm;          // segment holding idx E0
                        const uint64_t seg_end = (sgm + 1 < s1) ? seg_idx[sgm + 1] : nv;
                        if (I1 <= seg_end) ++gap_free;
                    }
                    fast = gap_free * 2 >= n_big + n_small;
                }
            }
            if (fast) {
                tiles0 += n_big;
                tiles1 += n_small;
                tiles += n_big + n_small;
            } else {
                const uint64_t nt = (windows + p.TW - 1) / p.TW;
                tiles_g += nt;
                tiles += nt;
            }
        }
        if (tiles > 0x7FFFFFFFull) raise(SW_ERR_RUNTIME, "batch too large: more than 2^31-1 tiles");
    }
    rec_seg_off[R] = (uint32_t)seg_pos.size();
    rec_tile_off[R] = (uint32_t)tiles;
    gen_off[R] = (uint32_t)tiles_g;
    off0[R] = (uint32_t)tiles0;
    off1[R] = (uint32_t)tiles1;
    p.n_tiles = (uint32_t)tiles;
    p.n_tiles_gen = (uint32_t)tiles_g;
    {   // a tile's own stage slot: 1.5 x the expected 2 / (w + 1) minimizers per window end, + 16
        const uint64_t tw = std::max<uint64_t>(p.TW, p.fc[0].TW);
        p.slot_cap = (uint32_t)std::min<uint64_t>(tw + w, (3 * tw / (w + 1) + 16 + 7) / 8 * 8);
        if (const char *e = SW_TEST_GETENV("SEQWIN_AMD_SLOT_CAP")) p.slot_cap = (uint32_t)std::max(1, atoi(e));   // test hook
    }

    auto up32 = [](DevArray<uint32_t> &d, const std::vector<uint32_t> &v) {
        d.alloc(v.size());
        if (!v.empty()) SW_HIP(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    };
    up32(p.rec_seg_off, rec_seg_off);
    up32(p.rec_nvalid, rec_nvalid);
    up32(p.rec_tile_off, rec_tile_off);
    up32(p.gen_tile_off, gen_off);
    up32(p.seg_pos, seg_pos);
    up32(p.seg_idx, seg_idx);
    up32(p.fc[0].rec_off, off0);
    up32(p.fc[1].rec_off, off1);
    {   // the per-tile tables, on the device
        DevArray<uint32_t> n_gap(2);
        SW_HIP(hipMemset(n_gap.p, 0, 8));
        const uint32_t nt[2] = {(uint32_t)tiles0, (uint32_t)tiles1};
        for (int c = 0; c < 2; ++c) {
            p.fc[c].n_tiles = nt[c];
            p.fc[c].desc.alloc(nt[c]);
            p.fc[c].gap_list.alloc(nt[c]);
            if (!nt[c]) continue;
            PlanTilesArgs a;
            a.cls_off = p.fc[c].rec_off.p;
            a.big_off = p.fc[0].rec_off.p;
            a.rec_tile_off = p.rec_tile_off.p;
            a.rec_nvalid = p.rec_nvalid.p;
            a.rec_seg_off = p.rec_seg_off.p;
            a.seg_pos = p.seg_pos.p;
            a.seg_idx = p.seg_idx.p;
            a.rec_base = b.d_rec_base.p;
            a.n_records = (uint32_t)R;
            a.n_tiles = nt[c];
            a.cls = (uint32_t)c;
            a.w = w;
            a.TW0 = p.fc[0].TW;
            a.TW1 = p.fc[1].TW;
            a.halo = p.halo_f;
            a.desc = p.fc[c].desc.p;
            a.gap_list = p.fc[c].gap_list.p;
            a.n_gap = n_gap.p + c;
            hipLaunchKernelGGL(k_plan_tiles, dim3((nt[c] + 255) / 256), dim3(256), 0, 0, a);
            SW_HIP(hipGetLastError());
        }
        p.gen_tile_rec.alloc(p.n_tiles_gen);
        if (p.n_tiles_gen) {
            hipLaunchKernelGGL(k_plan_gen_rec, dim3((p.n_tiles_gen + 255) / 256), dim3(256), 0, 0, p.gen_tile_off.p, (uint32_t)R,
                               p.n_tiles_gen, p.gen_tile_rec.p);
            SW_HIP(hipGetLastError());
        }
        uint32_t ng[2] = {0, 0};
        SW_HIP(hipMemcpy(ng, n_gap.p, 8, hipMemcpyDeviceToHost));   // (synchronises: the tables are complete)
        p.fc[0].n_gap = ng[0];
        p.fc[1].n_gap = ng[1];
    }

    uint64_t lut[40];
    const uint64_t S[4] = {SEED_A, SEED_C, SEED_G, SEED_T};
    uint64_t Sk[4];
    for (int c = 0; c < 4; ++c) Sk[c] = host_srol(S[c], k);  // srol_table(c, k), hashing_internals.hpp:347-352
    for (int out = 0; out < 5; ++out)
        for (int in = 0; in < 4; ++in) {
            const int idx = in | (out << 2);
            lut[2 * idx] = S[in] ^ (out < 4 ? Sk[out] : 0);              // forward: + S[in] + srol^k(S[out])
            lut[2 * idx + 1] = Sk[3 - in] ^ (out < 4 ? S[3 - out] : 0);  // reverse, before the sror
        }

    p.lut.alloc(40);
    SW_HIP(hipMemcpy(p.lut.p, lut, sizeof lut, hipMemcpyHostToDevice));
    if (k <= KPAIR) {
        // pair table of the rotate-free warm-up: row (p, b0 | b1 << 2) = {srol^(k-1-2p)(S[b0]) ^ srol^(k-2-2p)(S[b1]),
        // srol^(2p)(S[~b0]) ^ srol^(2p+1)(S[~b1])}; at odd k the last position holds one base (b1 ignored)
        std::vector<uint64_t> t2(512, 0);
        const uint32_t np = ((uint32_t)k + 1u) / 2u;
        for (uint32_t pp = 0; pp < np; ++pp)
            for (int b = 0; b < 16; ++b) {
                const int c0 = b & 3, c1 = b >> 2;
                const uint32_t i0 = 2 * pp, i1 = 2 * pp + 1;
                uint64_t f = host_srol(S[c0], (unsigned)(k - 1 - i0)), r = host_srol(S[3 - c0], i0);
                if (i1 < k) {
                    f ^= host_srol(S[c1], (unsigned)(k - 1 - i1));
                    r ^= host_srol(S[3 - c1], i1);
                }
                t2[2 * (pp * 16 + b)] = f;
                t2[2 * (pp * 16 + b) + 1] = r;
            }
        p.t4.alloc(512);
        SW_HIP(hipMemcpy(p.t4.p, t2.data(), 512 * 8, hipMemcpyHostToDevice));
    } else {   // 4-base warm-up table; byte b = c0 | c1 << 2 | c2 << 4 | c3 << 6 with c0 the earliest base
        std::vector<uint64_t> t4(512);
        auto host_sror = [&](uint64_t x, unsigned d) { return host_srol(x, 1023u - (d % 1023u)); };
        for (int b = 0; b < 256; ++b) {
            uint64_t f = 0, r = 0;
            for (int i = 0; i < 4; ++i) {
                const int c = (b >> (2 * i)) & 3;
                f = host_srol1(f) ^ S[c];              // F' = srol(F) ^ S[c]
                r = host_sror(r ^ Sk[3 - c], 1);       // R' = sror(R ^ srol^k(S[~c]))
            }
            t4[2 * b] = f;
            t4[2 * b + 1] = r;
        }
        p.t4.alloc(512);
        SW_HIP(hipMemcpy(p.t4.p, t4.data(), 512 * 8, hipMemcpyHostToDevice));
    }
    p.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return b.plans.emplace(key, std::move(p)).first->second;
}

void run_sketch(const sw_batch &b, const Plan &plan, hipStream_t stream, SketchOut &out, float *sketch_ms)
{
    {   // > 64 KiB of dynamic LDS must be opted into, once per device
        static std::mutex &mu = *new std::mutex;                 // leaked on purpose (see api.hip: pool())
        static std::map<int, bool> &done = *new std::map<int, bool>;
        int dev = 0;
        SW_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (!done[dev]) {
            SW_HIP(hipFuncSetAttribute((const void *)sketch_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes_for(L_MAX)));
            done[dev] = true;
        }
    }
    // the 64-thread tile class runs beside the 256-thread one on a stream of its own (forked from / joined into `stream`):
    // back to back in one stream the second kernel would start only when the first has drained
    hipStream_t side = nullptr;
    if (plan.fc[0].n_tiles && plan.fc[1].n_tiles) {
        static std::mutex &mu2 = *new std::mutex;                 // leaked on purpose (see api.hip: pool())
        static std::map<std::pair<int, hipStream_t>, hipStream_t> &sides = *new std::map<std::pair<int, hipStream_t>, hipStream_t>;
        int dev = 0;
        SW_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu2);
        hipStream_t &sref = sides[std::make_pair(dev, stream)];    // one per (device, caller's stream): callers do not share it
        if (!sref) SW_HIP(hipStreamCreateWithFlags(&sref, hipStreamNonBlocking));
        side = sref;
    }
    out.n_occ = 0;
    out.launches = 0;
    if (sketch_ms) *sketch_ms = 0.f;
    out.tile_count.alloc(plan.n_tiles);
    out.tile_offset.alloc(plan.n_tiles);
    if (plan.n_tiles == 0) return;

    DevArray<unsigned long long> cursor(2);   // [0] entries taken from the overflow area, [1] sum of the tile counts
    // per fast class: the list of tiles for the generic kernel's list mode (gap tiles of the plan, then the tiles the fast
    // kernel hands over) and its length
    DevArray<uint32_t> ovf_count(2), ovf_list0(plan.fc[0].n_tiles), ovf_list1(plan.fc[1].n_tiles);
    uint32_t *const ovf_list[2] = {ovf_list0.p, ovf_list1.p};
    // every tile owns a slot of slot_cap entries (1.5 x the expected density 2/(w+1) per window end); tiles with more
    // winners share an overflow area behind the slots, which is grown to the exact size and the pass re-run if it is too small
    const uint64_t slots = (uint64_t)plan.n_tiles * plan.slot_cap;
    uint64_t ovf_cap = std::max<uint64_t>(4096, slots / 64);
    Event ev0, ev1, evj(false);
#ifdef SW_SK_STAMPS
    DevArray<unsigned long long> stamps;
    const size_t n_stamp_waves = ((size_t)plan.fc[0].n_tiles / STAMP_EVERY + 1) * (BLOCK / 64);
    if (getenv("SEQWIN_AMD_STAMPS")) {
        stamps.alloc(n_stamp_waves * STAMP_SLOTS);
        SW_HIP(hipMemsetAsync(stamps.p, 0, stamps.bytes(), stream));
    }
#endif
    for (;;) {
        const uint64_t cap = slots + ovf_cap;
        out.stage_hash.alloc(cap);
        out.stage_kmer.alloc(cap);
        SW_HIP(hipMemsetAsync(cursor.p, 0, 2 * sizeof(unsigned long long), stream));
        SketchArgs a;
        a.packed = b.d_packed.p;
        a.rec_base = b.d_rec_base.p;
        a.rec_seg_off = plan.rec_seg_off.p;
        a.rec_nvalid = plan.rec_nvalid.p;
        a.rec_tile_off = plan.rec_tile_off.p;
        a.seg_pos = plan.seg_pos.p;
        a.seg_idx = plan.seg_idx.p;
        a.lut = plan.lut.p;
        a.t4 = plan.t4.p;
        a.n_records = (uint32_t)b.n_records;
        a.k = plan.k;
        a.w = plan.w;
        a.packed_words = b.packed_words;
        a.mult = plan.mult;
        a.stage_hash = out.stage_hash.p;
        a.stage_kmer = out.stage_kmer.p;
        a.cursor = cursor.p;
        a.cap = cap;
        a.ovf_base = slots;
        a.slot_cap = plan.slot_cap;
        a.tile_count = out.tile_count.p;
        a.tile_offset = out.tile_offset.p;
        a.list = nullptr;
        a.stamps = nullptr;
        a.cls_tile_rec = nullptr;
        a.cls_desc = nullptr;
        a.cls_tile_off = nullptr;
        a.ovf_count = nullptr;
        a.ovf_list = nullptr;
        a.halo = plan.halo_f;
        {
            const char *rc = SW_TEST_GETENV("SEQWIN_AMD_RC");   // test hook: force the overflow (list-mode) path
            a.rc_limit = rc ? std::min<uint32_t>(RC, (uint32_t)atoi(rc)) : RC;
        }
        // the lists start with the plan's gap tiles; the fast kernels append
        for (int c = 0; c < 2; ++c) {
            SW_HIP(hipMemsetD32Async((hipDeviceptr_t)(ovf_count.p + c), (int)plan.fc[c].n_gap, 1, stream));
            if (plan.fc[c].n_gap)
                SW_HIP(hipMemcpyAsync(ovf_list[c], plan.fc[c].gap_list.p, (size_t)plan.fc[c].n_gap * 4, hipMemcpyDeviceToDevice,
                                      stream));
        }
        SW_HIP(hipEventRecord(ev0, stream));
        if (side) {
            alloc_fork(side);
            SW_HIP(hipStreamWaitEvent(side, ev0, 0));
        }
        for (int c = 1; c >= 0; --c) {
            const Plan::FastClass &fc = plan.fc[c];
            if (!fc.n_tiles) continue;
            hipStream_t cs = (c == 1 && side) ? side : stream;
            a.cls_desc = fc.desc.p;
            a.ovf_count = ovf_count.p + c;
            a.ovf_list = ovf_list[c];
            a.L = plan.Lf;
            a.TW = fc.TW;
            a.n_tiles = fc.n_tiles;
            for (uint32_t tb = 0; tb < fc.n_tiles; tb += MAX_TILES_PER_LAUNCH) {
                const uint32_t nt = std::min(fc.n_tiles - tb, MAX_TILES_PER_LAUNCH);
                a.tile_base = tb;
#ifdef SW_SK_STAMPS
                a.stamps = (c == 0 && tb == 0) ? stamps.p : nullptr;
#endif
                if (plan.Lf == 32 && c == 0) hipLaunchKernelGGL((sketch_fast_kernel<32, BLOCK>), dim3(nt), dim3(BLOCK), 0, cs, a);
                else if (plan.Lf == 32) hipLaunchKernelGGL((sketch_fast_kernel<32, 64>), dim3(nt), dim3(64), 0, cs, a);
                else if (plan.Lf == 16 && c == 0) hipLaunchKernelGGL((sketch_fast_kernel<16, BLOCK>), dim3(nt), dim3(BLOCK), 0, cs, a);
                else if (plan.Lf == 16) hipLaunchKernelGGL((sketch_fast_kernel<16, 64>), dim3(nt), dim3(64), 0, cs, a);
                else if (plan.Lf == 8 && c == 0) hipLaunchKernelGGL((sketch_fast_kernel<8, BLOCK>), dim3(nt), dim3(BLOCK), 0, cs, a);
                else if (plan.Lf == 8) hipLaunchKernelGGL((sketch_fast_kernel<8, 64>), dim3(nt), dim3(64), 0, cs, a);
                else if (c == 0) hipLaunchKernelGGL((sketch_fast_kernel<4, BLOCK>), dim3(nt), dim3(BLOCK), 0, cs, a);
                else hipLaunchKernelGGL((sketch_fast_kernel<4, 64>), dim3(nt), dim3(64), 0, cs, a);
                SW_HIP(hipGetLastError());
            }
            if (c == 1 && side) SW_HIP(hipEventRecord(evj, side));
        }
        if (side) {
            SW_HIP(hipStreamWaitEvent(stream, evj, 0));   // join: everything below is ordered after both classes
            alloc_join(side);
        }
        if (plan.n_tiles_gen) {
            a.cls_tile_off = plan.gen_tile_off.p;
            a.cls_tile_rec = plan.gen_tile_rec.p;
            a.cls_desc = nullptr;         // own class: tile t of its record, (w - 1) + t * TW
            a.L = plan.L;
            a.TW = plan.TW;
            a.n_tiles = plan.n_tiles_gen;
            for (uint32_t tb = 0; tb < plan.n_tiles_gen; tb += MAX_TILES_PER_LAUNCH) {
                const uint32_t nt = std::min(plan.n_tiles_gen - tb, MAX_TILES_PER_LAUNCH);
                a.tile_base = tb;
                hipLaunchKernelGGL(sketch_generic_kernel, dim3(nt), dim3(BLOCK), plan.lds_bytes, stream, a);
                SW_HIP(hipGetLastError());
            }
        }
        // the listed tiles are done, exactly, by the generic kernel in list mode (same window ranges as the fast tiles of
        // their class).  The list length is only known on the device: a first batch of LIST_GRID workgroups per class is
        // enqueued blind (surplus workgroups exit at once), the rest -- rare -- after the counts have come back.
        constexpr uint32_t LIST_GRID = 2048;
        // (r06: the gap tiles of a class are known to the plan -- ragged assemblies with scaffold gaps list thousands of them --, so the
        //  blind launch covers them + LIST_GRID overflow tiles: no read-back and second launch for what the host already knows)
        uint32_t list_grid[2] = {LIST_GRID, LIST_GRID};
        SketchArgs al[2] = {a, a};
        for (int c = 0; c < 2; ++c) {
            const Plan::FastClass &fc = plan.fc[c];
            if (!fc.n_tiles) continue;
            list_grid[c] = (uint32_t)std::min<uint64_t>((uint64_t)fc.n_gap + LIST_GRID, MAX_TILES_PER_LAUNCH);
            al[c].cls_desc = fc.desc.p;
            al[c].ovf_count = ovf_count.p + c;
            al[c].L = plan.Lg_list;
            al[c].TW = fc.TW;
            al[c].list = ovf_list[c];
            al[c].tile_base = 0;
            al[c].n_tiles = list_grid[c];
            hipLaunchKernelGGL(sketch_generic_kernel, dim3(std::min(list_grid[c], fc.n_tiles)), dim3(BLOCK),
                               lds_bytes_for(plan.Lg_list), stream, al[c]);
            SW_HIP(hipGetLastError());
        }
        auto sum_counts = [&]() {
            const uint32_t blocks = (uint32_t)std::min<uint64_t>(1024, ((uint64_t)plan.n_tiles + 4095) / 4096);
            hipLaunchKernelGGL(k_sum_counts, dim3(blocks), dim3(256), 0, stream, out.tile_count.p, plan.n_tiles, cursor.p + 1);
            SW_HIP(hipGetLastError());
        };
        sum_counts();
        SW_HIP(hipEventRecord(ev1, stream));
        unsigned long long total[2] = {0, 0};   // overflow entries taken, n_occ
        uint32_t n_ovf[2] = {0, 0};
        SW_HIP(hipMemcpyAsync(total, cursor.p, sizeof total, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipMemcpyAsync(n_ovf, ovf_count.p, sizeof n_ovf, hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));
        float ms = 0.f;
        SW_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        if (sketch_ms) *sketch_ms += ms;
        ++out.launches;
        out.n_ovf_tiles += (uint64_t)n_ovf[0] + n_ovf[1];
        if (n_ovf[0] > list_grid[0] || n_ovf[1] > list_grid[1]) {
            SW_HIP(hipEventRecord(ev0, stream));
            for (int c = 0; c < 2; ++c)
                for (uint32_t tb = list_grid[c]; tb < n_ovf[c]; tb += MAX_TILES_PER_LAUNCH) {
                    al[c].tile_base = tb;
                    hipLaunchKernelGGL(sketch_generic_kernel, dim3(std::min(n_ovf[c] - tb, MAX_TILES_PER_LAUNCH)), dim3(BLOCK),
                                       lds_bytes_for(plan.Lg_list), stream, al[c]);
                    SW_HIP(hipGetLastError());
                }
            SW_HIP(hipMemsetAsync(cursor.p + 1, 0, sizeof(unsigned long long), stream));
            sum_counts();
            SW_HIP(hipEventRecord(ev1, stream));
            SW_HIP(hipMemcpyAsync(total, cursor.p, sizeof total, hipMemcpyDeviceToHost, stream));
            SW_HIP(hipStreamSynchronize(stream));
            SW_HIP(hipEventElapsedTime(&ms, ev0, ev1));
            if (sketch_ms) *sketch_ms += ms;
        }
        if (total[0] <= ovf_cap) {
            out.n_occ = total[1];
#ifdef SW_SK_STAMPS
            if (stamps.p) {   // mean shader clocks between the stamps of a wave, over the sampled tiles
                std::vector<unsigned long long> hs(n_stamp_waves * STAMP_SLOTS);
                SW_HIP(hipMemcpy(hs.data(), stamps.p, stamps.bytes(), hipMemcpyDeviceToHost));
                double acc[STAMP_SLOTS] = {0}, tile_life = 0;
                size_t nw = 0, nt = 0;
                for (size_t t = 0; t + (BLOCK / 64) <= n_stamp_waves; t += BLOCK / 64) {
                    unsigned long long t_first = ~0ull, t_last = 0;
                    bool ok = true;
                    for (size_t wv = 0; wv < BLOCK / 64; ++wv) {
                        const unsigned long long *st = &hs[(t + wv) * STAMP_SLOTS];
                        if (!st[0] || !st[11]) { ok = false; continue; }
                        for (int i = 1; i <= 11; ++i) acc[i] += st[i] && st[i - 1] ? (double)(st[i] - st[i - 1]) : 0.0;
                        ++nw;
                        t_first = std::min(t_first, st[0]);
                        t_last = std::max(t_last, st[11]);
                    }
                    if (ok) { tile_life += (double)(t_last - t_first); ++nt; }
                }
                fprintf(stderr, "[stamps] %zu waves of %zu tiles; mean clocks per phase:", nw, nt);
                static const char *nm[] = {"", "stage+barrier", "warmup", "roll", "barrier", "suffix", "barrier+ovf", "runmins", "windows",
                                           "barrier x2", "scan+barrier", "emit"};
                double sum = 0;
                for (int i = 1; i <= 11; ++i) { fprintf(stderr, " %s=%.0f", nm[i], nw ? acc[i] / nw : 0.0); sum += nw ? acc[i] / nw : 0.0; }
                fprintf(stderr, " | wave total %.0f, tile first-to-last %.0f\n", sum, nt ? tile_life / nt : 0.0);
            }
#endif
            break;
        }
        ovf_cap = total[0];  // exact size is now known
    }
}

}  // namespace sw
