// Does rocprim::radix_sort_pairs honour begin_bit > 0 on this stack (ROCm 7.2, gfx950)?  Sort u64 keys on bits [32, 64)
// with a 12-byte value, check on the device that the result is a stable sort by the top half and a permutation; time it
// against the split form (u32 key + 16-byte value) the library uses.   ./sort_beginbit [n_million]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__host__ __device__ inline uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31; return x; }
struct V12 { uint32_t pos, rec, idx; };
struct alignas(16) V16 { uint32_t low, pos, rec, idx; };
__global__ void fill64(uint64_t *k, V12 *v, size_t n, uint64_t distinct)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    k[i] = mix64(mix64(i * 0x9E3779B97F4A7C15ull + 7) % distinct + 0x1234567);
    v[i] = V12{(uint32_t)(i * 3), (uint32_t)(i >> 7), (uint32_t)i};
}
__global__ void fill32(uint32_t *k, V16 *v, size_t n, uint64_t distinct)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = mix64(mix64(i * 0x9E3779B97F4A7C15ull + 7) % distinct + 0x1234567);
    k[i] = (uint32_t)(h >> 32);
    v[i] = V16{(uint32_t)h, (uint32_t)(i * 3), (uint32_t)(i >> 7), (uint32_t)i};
}
// violations: [0] top halves descending, [1] equal top halves with descending original index (instability), [2] value does not
// belong to key, then [3] sum of idx, [4] xor of mix(idx)
__global__ void check(const uint64_t *k, const V12 *v, size_t n, uint64_t distinct, unsigned long long *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t expect = mix64(mix64((uint64_t)v[i].idx * 0x9E3779B97F4A7C15ull + 7) % distinct + 0x1234567);
    if (expect != k[i] || v[i].pos != v[i].idx * 3u) atomicAdd(&out[2], 1ull);
    if (i) {
        const uint32_t a = (uint32_t)(k[i - 1] >> 32), b = (uint32_t)(k[i] >> 32);
        if (a > b) atomicAdd(&out[0], 1ull);
        if (a == b && v[i - 1].idx > v[i].idx) atomicAdd(&out[1], 1ull);
    }
    atomicAdd(&out[3], (unsigned long long)v[i].idx);
    atomicXor(&out[4], (unsigned long long)mix64(v[i].idx));
}
int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atof(argv[1]) : 100) * 1000000;
    const uint64_t distinct = n / 9 + 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {
        uint64_t *k0, *k1; V12 *v0, *v1; unsigned long long *out;
        CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 12)); CK(hipMalloc(&v1, n * 12)); CK(hipMalloc(&out, 40));
        size_t tmp = 0;
        rocprim::double_buffer<uint64_t> dk(k0, k1); rocprim::double_buffer<V12> dv(v0, v1);
        CK(rocprim::radix_sort_pairs(nullptr, tmp, dk, dv, n, 32, 64, 0));
        void *t; CK(hipMalloc(&t, tmp));
        for (int it = 0; it < 3; ++it) {
            fill64<<<(n + 255) / 256, 256>>>(k0, v0, n, distinct);
            rocprim::double_buffer<uint64_t> a(k0, k1); rocprim::double_buffer<V12> b(v0, v1);
            hipEventRecord(e0);
            CK(rocprim::radix_sort_pairs(t, tmp, a, b, n, 32, 64, 0));
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            CK(hipMemset(out, 0, 40));
            check<<<(n + 255) / 256, 256>>>(a.current(), b.current(), n, distinct, out);
            unsigned long long h[5]; CK(hipMemcpy(h, out, 40, hipMemcpyDeviceToHost));
            unsigned long long xs = 0; for (size_t i = 0; i < n; ++i) xs ^= mix64((uint32_t)i);
            printf("u64 key bits [32,64) + 12 B value, n=%zu: %.3f ms; descending %llu, unstable %llu, mismatched %llu, idx sum %s, idx xor %s\n", n, ms,
                   h[0], h[1], h[2], h[3] == (unsigned long long)n * (n - 1) / 2 ? "ok" : "WRONG", h[4] == xs ? "ok" : "WRONG");
        }
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t); hipFree(out);
    }
    {
        uint32_t *k0, *k1; V16 *v0, *v1;
        CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 16)); CK(hipMalloc(&v1, n * 16));
        size_t tmp = 0;
        rocprim::double_buffer<uint32_t> dk(k0, k1); rocprim::double_buffer<V16> dv(v0, v1);
        CK(rocprim::radix_sort_pairs(nullptr, tmp, dk, dv, n, 0, 32, 0));
        void *t; CK(hipMalloc(&t, tmp));
        for (int it = 0; it < 3; ++it) {
            fill32<<<(n + 255) / 256, 256>>>(k0, v0, n, distinct);
            rocprim::double_buffer<uint32_t> a(k0, k1); rocprim::double_buffer<V16> b(v0, v1);
            hipEventRecord(e0);
            CK(rocprim::radix_sort_pairs(t, tmp, a, b, n, 0, 32, 0));
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("u32 key bits [0,32) + 16 B value, n=%zu: %.3f ms\n", n, ms);
        }
    }
    return 0;
}
