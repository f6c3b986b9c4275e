// host check of the half-word formulas against the reference definitions (hashing_internals.hpp:29-35, 69-74)
#include <cstdint>
#include <cstdio>
#include <random>
static uint32_t alignbit(uint32_t a, uint32_t b, unsigned s) { return (uint32_t)(((((uint64_t)a) << 32) | b) >> s); }
static uint32_t bfi(uint32_t x, uint32_t m, uint32_t y) { return (x & m) | (y & ~m); }
static uint64_t srol(uint64_t x) { uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32); return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m; }
static uint64_t sror(uint64_t x) { uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32); return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m; }
int main() {
    std::mt19937_64 g(1); long bad = 0;
    for (int i = 0; i < 2000000; ++i) {
        uint64_t x = g(); if (i < 64) x = 1ULL << i; if (i >= 64 && i < 128) x = ~(1ULL << (i - 64));
        uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        { // srol1
            uint32_t t = alignbit(hi, lo, 31), u = hi >> 30, nhi = bfi(u, 2, t), l2 = lo + lo, nlo = bfi(hi, 1, l2);
            if ((((uint64_t)nhi << 32) | nlo) != srol(x)) ++bad; }
        { // sror1
            uint32_t nlo = alignbit(hi, lo, 1), t1 = hi >> 1, t2 = alignbit(t1, hi, 1), nhi = bfi(lo, 1, t2);
            if ((((uint64_t)nhi << 32) | nlo) != sror(x)) ++bad; }
        { // srol4
            uint32_t xx = alignbit(hi, lo, 1), nlo = alignbit(lo, xx, 28), r4 = alignbit(hi, hi, 28), low5 = alignbit(hi >> 28, lo << 3, 31), nhi = bfi(low5, 31, r4);
            if ((((uint64_t)nhi << 32) | nlo) != srol(srol(srol(srol(x))))) ++bad; }
        { // sror4
            uint32_t l2 = lo + lo, y = bfi(hi, 1, l2), nlo = alignbit(y, lo, 4), n = bfi(l2, 16, hi), nhi = alignbit(hi >> 1, n, 4);
            if ((((uint64_t)nhi << 32) | nlo) != sror(sror(sror(sror(x))))) ++bad; }
    }
    printf("bad = %ld\n", bad); return bad != 0;
}
