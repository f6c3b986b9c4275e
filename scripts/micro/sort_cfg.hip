// micro-benchmark: rocPRIM radix_sort_pairs<u64,u32> with different onesweep configurations
#include <cstring>
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void fill(unsigned long long* k, unsigned* v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32; k[i] = x; v[i] = (unsigned)i; }
}
template <class Config> int run(const char* name, size_t n, unsigned end_bit) {
    unsigned long long *k0, *k1; unsigned *v0, *v1;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    size_t tmp = 0;
    rocprim::double_buffer<unsigned long long> dk(k0, k1); rocprim::double_buffer<unsigned> dv(v0, v1);
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, dk, dv, n, 0, end_bit, 0)));
    void* t; CK(hipMalloc(&t, tmp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        fill<<<(n + 255) / 256, 256>>>(k0, v0, n);
        rocprim::double_buffer<unsigned long long> a(k0, k1); rocprim::double_buffer<unsigned> b(v0, v1);
        hipEventRecord(e0);
        CK((rocprim::radix_sort_pairs<Config>(t, tmp, a, b, n, 0, end_bit, 0)));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-34s n=%zu bits=%u  %.3f ms\n", name, n, end_bit, best);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
    return 0;
}
using namespace rocprim;
template <unsigned BS, unsigned IPT, unsigned BITS>
using Cfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<BS, IPT>, BITS>>;
int main() {
    for (size_t n : {(size_t)24415401, (size_t)200000000}) {
        run<default_config>("default", n, 64);
        run<Cfg<256, 8, 8>>("256x8 r8", n, 64);
        run<Cfg<256, 12, 8>>("256x12 r8", n, 64);
        run<Cfg<256, 16, 8>>("256x16 r8", n, 64);
        run<Cfg<256, 20, 8>>("256x20 r8", n, 64);
        run<Cfg<128, 16, 8>>("128x16 r8", n, 64);
        run<Cfg<256, 12, 7>>("256x12 r7", n, 64);
        run<Cfg<256, 16, 6>>("256x16 r6", n, 64);
        run<default_config>("default 44 bits", n, 44);
    }
    return 0;
}
