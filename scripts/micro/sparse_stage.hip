// sparse_stage.hip -- stand-alone prototype of the SPARSE stage of the "threshold form" of the window minima (NOTES.md r05,
// VERDICT r5 item 4): measure it instead of leaving it an analysis.
//
// Threshold form: keep only the k-mers whose hash lies below T = 2^64 / D (D = 16 or 32); element j (hash h, position p) is the
// rightmost minimum of some window of w consecutive k-mers (btllib minimizer.cpp:14-49: `<=`, the rightmost wins) iff
//     b - a > w,   a = position of the nearest STRICTLY smaller kept element to its left  (none within w: a = p - w - 1 ... -inf)
//                  b = position of the nearest smaller-OR-EQUAL kept element to its right (none within w: +inf)
// provided every window holds a kept element (two neighbouring kept elements more than w apart: the tile goes to the fallback).
// The DENSE part (roll, compare with T, compaction of ~NE/D kept elements per tile into LDS) is not prototyped here -- its cost is
// known (roll 16 + ~6 VALU per element); what was only estimated is the part that works on the kept elements in LOCKSTEP:
// a wave pays the longest neighbour search of its 64 lanes.  This program times exactly that part, per tile of NE = 8192 elements
// as in sketch_fast_kernel<32, 256> (one 256-thread workgroup, the kept elements of the tile + halo in LDS), in two forms:
//   scan      every kept element walks left / right over the kept list until it finds a better one or leaves the window
//   blocked   minima of groups of four kept elements first (one LDS word per group), then inside the group that stopped the walk
// and checks the winners against a brute-force sliding-window minimum over the dense hashes of a few tiles.
// Build: hipcc --offload-arch=gfx950 -O3 sparse_stage.hip -o sparse_stage        Run: ./sparse_stage [tiles] [w] [D]
// Instruction counts: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -- ./sparse_stage
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NE = 8192;     // elements (k-mers) per tile, halo included
constexpr int CAP = 1024;    // kept elements a tile may hold (D = 16: 512 +- 22; D = 8 would need more)
constexpr int SLOT = 144;    // winners a tile's output slot holds (as the real kernel at w = 200)

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__host__ __device__ inline uint64_t dense_hash(uint64_t tile, uint32_t e) { return mix64(tile * 0x9E3779B97F4A7C15ULL + e * 2 + 1); }

// setup (not timed): what the dense part would leave in LDS -- the tile's kept elements in position order
__global__ void k_setup(uint64_t *kh, uint32_t *kp, uint32_t *kn, uint32_t *far, uint64_t T, int w)
{
    const uint64_t tile = blockIdx.x;
    if (threadIdx.x) return;
    uint32_t n = 0, last = 0, gap = 0;
    for (uint32_t e = 0; e < NE; ++e) {
        const uint64_t h = dense_hash(tile, e);
        if (h < T && n < CAP) {
            if (n && e - last > (uint32_t)w) gap = 1;
            kh[tile * CAP + n] = h;
            kp[tile * CAP + n] = e;
            last = e;
            ++n;
        }
    }
    kn[tile] = n;
    far[tile] = gap;   // a window without a kept element: the real kernel would hand the tile to the fallback
}

// one workgroup = one tile.  Winners among the window ENDS owned by the tile: a winner is emitted by the tile that owns the window
// end max(p, a + w) -- here simply every winner whose position is >= w (the halo's are the previous tile's).
template <bool BLOCKED>
__global__ __launch_bounds__(256) void k_sparse(const uint64_t *__restrict__ kh, const uint32_t *__restrict__ kp, const uint32_t *__restrict__ kn,
                                               uint64_t *__restrict__ out_h, uint32_t *__restrict__ out_p, uint32_t *__restrict__ out_n, int w)
{
    __shared__ uint64_t H[CAP + 8];
    __shared__ uint32_t P[CAP + 8];
    __shared__ uint64_t GM[CAP / 4 + 2];
    __shared__ uint32_t wave_cnt[4];
    const uint64_t tile = blockIdx.x;
    const int n = (int)kn[tile], t = threadIdx.x;
    for (int j = t; j < n; j += 256) {
        H[j] = kh[tile * CAP + j];
        P[j] = kp[tile * CAP + j];
    }
    __syncthreads();
    if (BLOCKED) {
        for (int g = t; g * 4 < n; g += 256) {
            uint64_t m = ~0ull;
            for (int i = g * 4; i < g * 4 + 4 && i < n; ++i) m = H[i] < m ? H[i] : m;
            GM[g] = m;
        }
        __syncthreads();
    }
    uint32_t total_before = 0;
    for (int base = 0; base < n; base += 256) {   // (D = 16: two rounds and a short third)
        const int j = base + t;
        bool win = false;
        uint64_t h = 0;
        uint32_t p = 0;
        if (j < n) {
            h = H[j];
            p = P[j];
            // a: nearest strictly smaller to the left, at most w - 1 positions away (further left it cannot share a window with p)
            int64_t a = (int64_t)p - w;          // "none in reach": any window that holds p starts after it
            if (!BLOCKED) {
                for (int i = j - 1; i >= 0 && p - P[i] < (uint32_t)w; --i)
                    if (H[i] < h) { a = P[i]; break; }
            } else {
                int i = j - 1;
                for (; i >= 0 && (i & 3) != 3 && p - P[i] < (uint32_t)w; --i)      // the rest of the own group
                    if (H[i] < h) { a = P[i]; i = -2; break; }
                while (i >= 3 && p - P[i] < (uint32_t)w) {                          // whole groups by their minima
                    if (GM[i >> 2] < h) {
                        for (int q = i; q > i - 4; --q)
                            if (H[q] < h) { if (p - P[q] < (uint32_t)w) a = P[q]; break; }
                        i = -2;
                        break;
                    }
                    i -= 4;
                }
                if (i >= 0)
                    for (; i >= 0 && p - P[i] < (uint32_t)w; --i)
                        if (H[i] < h) { a = P[i]; break; }
            }
            // b: nearest smaller-or-equal to the right, inside (p, a + w]: beyond a + w no window holds both a's successor and b
            const int64_t reach = a + w;         // b > reach <=> winner
            win = true;
            if (!BLOCKED) {
                for (int i = j + 1; i < n && (int64_t)P[i] <= reach; ++i)
                    if (H[i] <= h) { win = false; break; }
            } else {
                int i = j + 1;
                for (; i < n && (i & 3) != 0 && (int64_t)P[i] <= reach; ++i)
                    if (H[i] <= h) { win = false; break; }
                while (win && i + 3 < n && (int64_t)P[i] <= reach) {
                    if (GM[i >> 2] <= h) {
                        for (int q = i; q < i + 4; ++q)
                            if (H[q] <= h) { if ((int64_t)P[q] <= reach) win = false; break; }
                        if (!win) break;
                        // (the group's minimum lies beyond the reach: nothing further right is in reach either)
                        i = n;
                        break;
                    }
                    i += 4;
                }
                if (win)
                    for (; i < n && (int64_t)P[i] <= reach; ++i)
                        if (H[i] <= h) { win = false; break; }
            }
            win = win && p >= (uint32_t)w;       // (halo elements belong to the previous tile)
        }
        // emit in position order: wave ballots, counts across the four waves through LDS
        const uint64_t bal = __ballot(win);
        const uint32_t below = __builtin_popcountll(bal & ((1ull << (t & 63)) - 1));
        if ((t & 63) == 0) wave_cnt[t >> 6] = __builtin_popcountll(bal);
        __syncthreads();
        uint32_t off = total_before;
        for (int q = 0; q < (t >> 6); ++q) off += wave_cnt[q];
        const uint32_t round_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (win && off + below < SLOT) {
            out_h[tile * SLOT + off + below] = h;
            out_p[tile * SLOT + off + below] = p;
        }
        total_before += round_total;
        __syncthreads();
    }
    if (t == 0) out_n[tile] = total_before;
}

// brute force over the dense hashes: the rightmost minimum of every window [x, x + w - 1] inside the tile, x >= 1 (windows that end
// at e >= w, like the tile-owned ends above), marked; one thread per window
__global__ void k_brute(uint8_t *mark, int w)
{
    const uint64_t tile = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x + w - 1 >= NE) return;
    uint64_t best = ~0ull;
    int arg = -1;
    for (int e = x; e < x + w; ++e) {
        const uint64_t h = dense_hash(tile, e);
        if (h <= best) { best = h; arg = e; }
    }
    if (arg >= w) mark[tile * NE + arg] = 1;
}

int main(int argc, char **argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 200000, w = argc > 2 ? atoi(argv[2]) : 200, D = argc > 3 ? atoi(argv[3]) : 16;
    const uint64_t T = ~0ull / (uint64_t)D;
    uint64_t *kh, *out_h;
    uint32_t *kp, *kn, *far, *out_p, *out_n;
    CK(hipMalloc(&kh, (size_t)tiles * CAP * 8));
    CK(hipMalloc(&kp, (size_t)tiles * CAP * 4));
    CK(hipMalloc(&kn, (size_t)tiles * 4));
    CK(hipMalloc(&far, (size_t)tiles * 4));
    CK(hipMalloc(&out_h, (size_t)tiles * SLOT * 8));
    CK(hipMalloc(&out_p, (size_t)tiles * SLOT * 4));
    CK(hipMalloc(&out_n, (size_t)tiles * 4));
    hipLaunchKernelGGL(k_setup, dim3(tiles), dim3(64), 0, 0, kh, kp, kn, far, T, w);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> hn(tiles), hfar(tiles);
    CK(hipMemcpy(hn.data(), kn, (size_t)tiles * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hfar.data(), far, (size_t)tiles * 4, hipMemcpyDeviceToHost));
    double kept = 0, fallback = 0;
    for (int i = 0; i < tiles; ++i) { kept += hn[i]; fallback += hfar[i]; }
    printf("%d tiles of %d elements, w = %d, T = 2^64 / %d: %.1f kept elements per tile, %.2f %% of the tiles hold a gap > w between kept elements (fallback)\n",
           tiles, NE, w, D, kept / tiles, 100.0 * fallback / tiles);
    // correctness on the first 64 tiles without such a gap
    {
        const int nt = 64;
        uint8_t *mark;
        CK(hipMalloc(&mark, (size_t)nt * NE));
        CK(hipMemset(mark, 0, (size_t)nt * NE));
        hipLaunchKernelGGL(k_brute, dim3((NE + 255) / 256, nt), dim3(256), 0, 0, mark, w);
        std::vector<uint8_t> hm((size_t)nt * NE);
        CK(hipMemcpy(hm.data(), mark, hm.size(), hipMemcpyDeviceToHost));
        for (int form = 0; form < 2; ++form) {
            if (form) hipLaunchKernelGGL(k_sparse<true>, dim3(nt), dim3(256), 0, 0, kh, kp, kn, out_h, out_p, out_n, w);
            else hipLaunchKernelGGL(k_sparse<false>, dim3(nt), dim3(256), 0, 0, kh, kp, kn, out_h, out_p, out_n, w);
            CK(hipDeviceSynchronize());
            std::vector<uint32_t> on(nt), op((size_t)nt * SLOT);
            CK(hipMemcpy(on.data(), out_n, nt * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(op.data(), out_p, (size_t)nt * SLOT * 4, hipMemcpyDeviceToHost));
            long bad = 0, checked = 0, winners = 0;
            for (int ti = 0; ti < nt; ++ti) {
                if (hfar[ti]) continue;
                ++checked;
                std::vector<uint8_t> got(NE, 0);
                for (uint32_t q = 0; q < on[ti] && q < (uint32_t)SLOT; ++q) got[op[(size_t)ti * SLOT + q]] = 1;
                for (int e = 0; e < NE; ++e) {
                    // (the last w - 1 positions can win windows that end in the NEXT tile: brute force only sees windows inside this one)
                    if (e + w - 1 < NE && got[e] != hm[(size_t)ti * NE + e]) ++bad;
                    winners += hm[(size_t)ti * NE + e];
                }
            }
            printf("%s form: %ld tiles checked against the brute-force window minima, %ld winners, %ld differences\n", form ? "blocked" : "scan", checked, winners, bad);
            if (bad) return 1;
        }
        CK(hipFree(mark));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int form = 0; form < 2; ++form) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (form) hipLaunchKernelGGL(k_sparse<true>, dim3(tiles), dim3(256), 0, 0, kh, kp, kn, out_h, out_p, out_n, w);
            else hipLaunchKernelGGL(k_sparse<false>, dim3(tiles), dim3(256), 0, 0, kh, kp, kn, out_h, out_p, out_n, w);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        // 10.2 M tiles per 75 Gbp launch of sketch_fast_kernel<32, 256> (80.4 ms: 7.9 ns per tile for ALL its phases)
        printf("%s form: %.3f ms for %d tiles = %.2f ns per tile -> %.1f ms for the 10.2 M tiles of the 15 000-genome launch\n", form ? "blocked" : "scan", best, tiles,
               best * 1e6 / tiles, best / tiles * 10.2e6);
    }
    return 0;
}
