// Host->device copy paths on the target box: pageable, registered, pinned; cost of pinning.  hipcc -O2 h2d.hip -o h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main()
{
    const size_t N = 320u << 20;   // 320 MiB
    void *d;
    hipMalloc(&d, N);
    char *pg = (char *)malloc(N);
    memset(pg, 1, N);
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = clk::now();
        hipMemcpy(d, pg, N, hipMemcpyHostToDevice);
        auto t1 = clk::now();
        printf("pageable hipMemcpy        : %.1f ms  %.1f GB/s\n", ms(t0, t1), N / ms(t0, t1) / 1e6);
    }
    {
        auto t0 = clk::now();
        hipHostRegister(pg, N, hipHostRegisterDefault);
        auto t1 = clk::now();
        hipMemcpy(d, pg, N, hipMemcpyHostToDevice);
        auto t2 = clk::now();
        hipHostUnregister(pg);
        auto t3 = clk::now();
        printf("hipHostRegister           : %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms\n", ms(t0, t1), ms(t1, t2),
               N / ms(t1, t2) / 1e6, ms(t2, t3));
    }
    {
        auto t0 = clk::now();
        char *pin;
        hipHostMalloc((void **)&pin, N, hipHostMallocDefault);
        auto t1 = clk::now();
        memset(pin, 2, N);
        auto t2 = clk::now();
        hipMemcpy(d, pin, N, hipMemcpyHostToDevice);
        auto t3 = clk::now();
        hipMemcpy(d, pin, N, hipMemcpyHostToDevice);
        auto t4 = clk::now();
        hipHostFree(pin);
        auto t5 = clk::now();
        printf("hipHostMalloc             : %.1f ms, first touch %.1f ms, copy %.1f / %.1f ms (%.1f GB/s), free %.1f ms\n", ms(t0, t1),
               ms(t1, t2), ms(t2, t3), ms(t3, t4), N / ms(t3, t4) / 1e6, ms(t4, t5));
    }
    {   // staged: T threads memcpy 8 MiB pieces into a pinned ring, async copies
        const size_t P = 8u << 20;
        const int SLOTS = 8;
        char *ring;
        hipHostMalloc((void **)&ring, P * SLOTS, hipHostMallocDefault);
        memset(ring, 0, P * SLOTS);
        hipStream_t st;
        hipStreamCreate(&st);
        hipEvent_t ev[SLOTS];
        for (auto &e : ev) hipEventCreate(&e);
        for (int T : {1, 2, 4}) {
            auto t0 = clk::now();
            const size_t pieces = N / P;
            for (size_t p0 = 0; p0 < pieces; p0 += SLOTS) {
                for (int s = 0; s < SLOTS && p0 + s < pieces; ++s) hipEventSynchronize(ev[s]);
                std::vector<std::thread> th;
                const int cnt = (int)std::min<size_t>(SLOTS, pieces - p0);
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        for (int s = t; s < cnt; s += T) memcpy(ring + s * P, pg + (p0 + s) * P, P);
                    });
                for (auto &x : th) x.join();
                for (int s = 0; s < cnt; ++s) {
                    hipMemcpyAsync((char *)d + (p0 + s) * P, ring + s * P, P, hipMemcpyHostToDevice, st);
                    hipEventRecord(ev[s], st);
                }
            }
            hipStreamSynchronize(st);
            auto t1 = clk::now();
            printf("staged ring, %d copy thread(s): %.1f ms  %.1f GB/s\n", T, ms(t0, t1), N / ms(t0, t1) / 1e6);
        }
    }
    return 0;
}
