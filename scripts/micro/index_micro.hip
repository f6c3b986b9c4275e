// micro-benchmarks behind the index-stage design: rocPRIM onesweep pass cost by element size
// and digit width, random gather / scatter, and an open-addressing u64 table (atomicCAS insert, lookup).
//   hipcc -O3 --offload-arch=gfx950 index_micro.hip -o index_micro && ./index_micro [n_million] [distinct_million]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
struct V12 { uint32_t a, b, c; };
struct V16 { uint32_t a, b, c, d; };
struct V2 { uint16_t a; };

template <class K, class V> __global__ void fill(K *k, V *v, size_t n, uint64_t distinct, unsigned bits)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t id = mix64(i * 0x9E3779B97F4A7C15ull + 7) % distinct;
    uint64_t h = mix64(id + 0x1234567);
    if (bits < 64) h &= (1ull << bits) - 1ull;
    k[i] = (K)h;
    if (v) memset(&v[i], 0, sizeof(V)), *reinterpret_cast<uint16_t *>(&v[i]) = (uint16_t)i;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { hipEventCreate(&a); hipEventCreate(&b); }
    void start() { hipEventRecord(a); }
    float stop() { hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
};

template <class Config, class K, class V> void sort_pairs(const char *name, size_t n, unsigned end_bit, uint64_t distinct)
{
    K *k0, *k1; V *v0, *v1;
    CK(hipMalloc(&k0, n * sizeof(K))); CK(hipMalloc(&k1, n * sizeof(K))); CK(hipMalloc(&v0, n * sizeof(V))); CK(hipMalloc(&v1, n * sizeof(V)));
    size_t tmp = 0;
    rocprim::double_buffer<K> dk(k0, k1); rocprim::double_buffer<V> dv(v0, v1);
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, dk, dv, n, 0, end_bit, 0)));
    void *t; CK(hipMalloc(&t, tmp));
    Timer T; float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        fill<K, V><<<(n + 255) / 256, 256>>>(k0, v0, n, distinct, end_bit);
        rocprim::double_buffer<K> a(k0, k1); rocprim::double_buffer<V> b(v0, v1);
        T.start();
        CK((rocprim::radix_sort_pairs<Config>(t, tmp, a, b, n, 0, end_bit, 0)));
        float ms = T.stop(); if (ms < best) best = ms;
    }
    printf("pairs %-26s key %zu B val %2zu B bits %2u n=%zu: %8.3f ms  (%.2f ns/elem)\n", name, sizeof(K), sizeof(V), end_bit, n, best, best * 1e6 / n);
    fflush(stdout);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
}
template <class Config, class K> void sort_keys(const char *name, size_t n, unsigned end_bit, uint64_t distinct)
{
    K *k0, *k1;
    CK(hipMalloc(&k0, n * sizeof(K))); CK(hipMalloc(&k1, n * sizeof(K)));
    size_t tmp = 0;
    rocprim::double_buffer<K> dk(k0, k1);
    CK((rocprim::radix_sort_keys<Config>(nullptr, tmp, dk, n, 0, end_bit, 0)));
    void *t; CK(hipMalloc(&t, tmp));
    Timer T; float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        fill<K, V2><<<(n + 255) / 256, 256>>>(k0, nullptr, n, distinct, end_bit);
        rocprim::double_buffer<K> a(k0, k1);
        T.start();
        CK((rocprim::radix_sort_keys<Config>(t, tmp, a, n, 0, end_bit, 0)));
        float ms = T.stop(); if (ms < best) best = ms;
    }
    printf("keys  %-26s key %zu B          bits %2u n=%zu: %8.3f ms  (%.2f ns/elem)\n", name, sizeof(K), end_bit, n, best, best * 1e6 / n);
    fflush(stdout);
    hipFree(k0); hipFree(k1); hipFree(t);
}

using namespace rocprim;
template <unsigned BS, unsigned IPT, unsigned BITS>
using Cfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<BS, IPT>, BITS>>;

// ---- random gather / scatter -----------------------------------------------------------------------
__global__ void k_perm(uint32_t *perm, size_t n)   // a bijection of [0, n): multiplicative step modulo n is not one in general; use sort instead
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)i;
}
__global__ void k_scatter4(const uint32_t *__restrict__ perm, uint32_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm[i]] = (uint32_t)i;
}
__global__ void k_gather8(const uint32_t *__restrict__ perm, const uint64_t *__restrict__ in, uint64_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}
__global__ void k_gather4(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

// ---- open-addressing table of u64 keys (linear probing, 64-bit CAS) -----------------------------------
constexpr uint64_t EMPTY = ~0ull;
__global__ void k_insert(const uint64_t *__restrict__ keys, size_t n, unsigned long long *__restrict__ tab, uint64_t mask,
                         uint32_t *__restrict__ slot_of, uint32_t *__restrict__ count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint64_t s = mix64(key) & mask;
    for (;;) {
        unsigned long long cur = tab[s];
        if (cur == key) break;
        if (cur == EMPTY) {
            cur = atomicCAS(&tab[s], EMPTY, (unsigned long long)key);
            if (cur == EMPTY || cur == key) break;
        }
        s = (s + 1) & mask;
    }
    slot_of[i] = (uint32_t)s;
    if (count) atomicAdd(&count[s], 1u);
}
__global__ void k_lookup(const uint64_t *__restrict__ keys, size_t n, const unsigned long long *__restrict__ tab, uint64_t mask,
                         const uint32_t *__restrict__ val, uint32_t *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint64_t s = mix64(key) & mask;
    while (tab[s] != key) s = (s + 1) & mask;
    out[i] = val[s];
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atof(argv[1]) : 200) * 1000000;
    const uint64_t distinct = (uint64_t)((argc > 2 ? atof(argv[2]) : 21.3) * 1000000);
    const char *only = argc > 3 ? argv[3] : "";
    auto want = [&](const char *s) { return !only[0] || strstr(only, s); };
    printf("n = %zu elements, %llu distinct keys\n", n, (unsigned long long)distinct);
    if (want("sort")) {
        sort_pairs<default_config, uint32_t, uint64_t>("default", n, 32, distinct);
        sort_pairs<default_config, uint32_t, V12>("default", n, 32, distinct);
        sort_pairs<default_config, uint32_t, V16>("default", n, 32, distinct);
        sort_pairs<default_config, uint32_t, uint32_t>("default", n, 30, distinct);
        sort_pairs<default_config, uint32_t, uint32_t>("default (16 bits)", n, 16, distinct);
        sort_pairs<default_config, uint64_t, uint32_t>("default", n, 54, distinct);
        sort_pairs<default_config, uint64_t, V2>("default", n, 54, distinct);
        sort_keys<default_config, uint64_t>("default", n, 54, distinct);
        sort_keys<default_config, uint64_t>("default", n, 64, distinct);
    }
#ifdef WIDE
    if (want("wide")) {
        sort_pairs<Cfg<256, 12, 8>, uint32_t, uint64_t>("256x12 r8", n, 32, distinct);
        sort_pairs<Cfg<256, 16, 8>, uint32_t, uint64_t>("256x16 r8", n, 32, distinct);
        sort_pairs<Cfg<256, 12, 9>, uint32_t, uint64_t>("256x12 r9", n, 32, distinct);
        sort_pairs<Cfg<256, 16, 11>, uint32_t, uint64_t>("256x16 r11", n, 32, distinct);
        sort_pairs<Cfg<512, 16, 11>, uint32_t, uint64_t>("512x16 r11", n, 32, distinct);
        sort_pairs<Cfg<256, 12, 9>, uint64_t, uint32_t>("256x12 r9", n, 54, distinct);
        sort_pairs<Cfg<256, 16, 11>, uint64_t, uint32_t>("256x16 r11", n, 54, distinct);
        sort_keys<Cfg<256, 16, 9>, uint64_t>("256x16 r9", n, 54, distinct);
        sort_keys<Cfg<256, 16, 11>, uint64_t>("256x16 r11", n, 54, distinct);
    }
#endif
    if (want("rand")) {
        uint32_t *perm, *pk0, *pk1, *pv1, *out4;
        uint64_t *in8, *out8;
        CK(hipMalloc(&perm, n * 4)); CK(hipMalloc(&pk0, n * 4)); CK(hipMalloc(&pk1, n * 4)); CK(hipMalloc(&pv1, n * 4));
        CK(hipMalloc(&out4, n * 4)); CK(hipMalloc(&in8, n * 8)); CK(hipMalloc(&out8, n * 8));
        // a random permutation: sort iota by random 32-bit keys
        fill<uint32_t, V2><<<(n + 255) / 256, 256>>>(pk0, nullptr, n, ~0ull, 32);
        k_perm<<<(n + 255) / 256, 256>>>(perm, n);
        {
            size_t tmp = 0;
            rocprim::double_buffer<uint32_t> dk(pk0, pk1), dv(perm, pv1);
            CK(rocprim::radix_sort_pairs(nullptr, tmp, dk, dv, n, 0, 32, 0));
            void *t; CK(hipMalloc(&t, tmp));
            CK(rocprim::radix_sort_pairs(t, tmp, dk, dv, n, 0, 32, 0));
            CK(hipDeviceSynchronize());
            if (dv.current() != perm) CK(hipMemcpy(perm, dv.current(), n * 4, hipMemcpyDeviceToDevice));
            hipFree(t);
        }
        CK(hipMemset(in8, 1, n * 8));
        Timer T;
        for (int it = 0; it < 2; ++it) {
            T.start(); k_scatter4<<<(n + 255) / 256, 256>>>(perm, out4, n); float a = T.stop();
            T.start(); k_gather8<<<(n + 255) / 256, 256>>>(perm, in8, out8, n); float b = T.stop();
            T.start(); k_gather4<<<(n + 255) / 256, 256>>>(perm, pk0, out4, n); float c = T.stop();
            printf("random over n=%zu: scatter 4 B %.3f ms, gather 8 B %.3f ms, gather 4 B %.3f ms\n", n, a, b, c);
        }
        hipFree(perm); hipFree(pk0); hipFree(pk1); hipFree(pv1); hipFree(out4); hipFree(in8); hipFree(out8);
    }
    if (want("table")) {
        uint64_t *keys; uint32_t *slot_of, *count, *out;
        CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&slot_of, n * 4)); CK(hipMalloc(&out, n * 4));
        fill<uint64_t, V2><<<(n + 255) / 256, 256>>>(keys, nullptr, n, distinct, 64);
        for (unsigned lg = 0; lg < 2; ++lg) {
            uint64_t cap = 1; while (cap < distinct * (lg ? 4 : 2)) cap <<= 1;
            unsigned long long *tab;
            CK(hipMalloc(&tab, cap * 8)); CK(hipMalloc(&count, cap * 4));
            Timer T;
            for (int it = 0; it < 2; ++it) {
                CK(hipMemset(tab, 0xFF, cap * 8)); CK(hipMemset(count, 0, cap * 4));
                T.start(); k_insert<<<(n + 255) / 256, 256>>>(keys, n, tab, cap - 1, slot_of, nullptr); float a = T.stop();
                CK(hipMemset(tab, 0xFF, cap * 8));
                T.start(); k_insert<<<(n + 255) / 256, 256>>>(keys, n, tab, cap - 1, slot_of, count); float b = T.stop();
                T.start(); k_lookup<<<(n + 255) / 256, 256>>>(keys, n, tab, cap - 1, count, out); float c = T.stop();
                T.start(); k_gather4<<<(n + 255) / 256, 256>>>(slot_of, count, out, n); float d = T.stop();
                printf("table 2^%d slots (load %.2f): insert %.3f ms, insert+count %.3f ms, lookup %.3f ms, value gather by slot %.3f ms\n",
                       (int)__builtin_ctzll(cap), (double)distinct / cap, a, b, c, d);
            }
            hipFree(tab); hipFree(count);
        }
        hipFree(keys); hipFree(slot_of); hipFree(out);
    }
    return 0;
}
