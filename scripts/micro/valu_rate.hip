// valu_rate.hip -- integer VALU issue rate on gfx950 at a given occupancy (waves per SIMD), for the instruction kinds
// the sketch kernel is made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int KIND> __global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    uint64_t q0 = a0, q1 = a1 | 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) {          // 8 independent 32-bit xor/add chains (VOP2)
                a0 = (a0 ^ a1) + 1; a1 = (a1 ^ a2) + 2; a2 = (a2 ^ a3) + 3; a3 = (a3 ^ a4) + 4;
                a4 = (a4 ^ a5) + 5; a5 = (a5 ^ a6) + 6; a6 = (a6 ^ a7) + 7; a7 = (a7 ^ a0) + 8;
            } else if (KIND == 1) {   // one dependent chain (VOP2)
                a0 = (a0 ^ a1) + 1; a0 = (a0 ^ a2) + 2; a0 = (a0 ^ a3) + 3; a0 = (a0 ^ a4) + 4;
                a0 = (a0 ^ a5) + 5; a0 = (a0 ^ a6) + 6; a0 = (a0 ^ a7) + 7; a0 = (a0 ^ a1) + 8;
            } else if (KIND == 2) {   // VOP3: alignbit + bitop3-like and_or, dependent pairs
                a0 = __builtin_amdgcn_alignbit(a0, a1, 31); a1 = (a1 & a2) | a3; a2 = __builtin_amdgcn_alignbit(a2, a3, 1); a3 = (a3 & a4) | a5;
                a4 = __builtin_amdgcn_alignbit(a4, a5, 31); a5 = (a5 & a6) | a7; a6 = __builtin_amdgcn_alignbit(a6, a7, 1); a7 = (a7 & a0) | a1;
            } else {                  // 64-bit compare + 2 cndmask (prefix-minimum step)
                const uint64_t h = ((uint64_t)a1 << 32) | a2;
                const bool keep = q0 < h; q0 = keep ? q0 : h; a1 += a3; a2 ^= a1;
                const uint64_t g = ((uint64_t)a4 << 32) | a5;
                const bool kp = q1 < g; q1 = kp ? q1 : g; a4 += a6; a5 ^= a4;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)q0 ^ (uint32_t)(q1 >> 32);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
    printf("%s CUs=%d clock=%.2f GHz\n", p.gcnArchName, cus, ghz);
    uint32_t *out; hipMalloc(&out, (size_t)cus * 32 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char *names[4] = {"8 indep xor+add (VOP2)", "1 dependent chain (VOP2)", "alignbit + and_or (VOP3)", "cmp_u64 + 2 cndmask + 2 alu"};
    // instructions per unrolled step: KIND0: 16, KIND1: 16, KIND2: 8 (and_or fuses), KIND3: ~2*(1 cmp + 2 cnd + 2) = 10
    const double ipu[4] = {16, 16, 8, 10};
    for (int kind = 0; kind < 4; ++kind)
        for (int wg_per_cu : {1, 2, 3, 4, 5, 8}) {   // 256-thread workgroups = 1 wave per SIMD each
            const int grid = cus * wg_per_cu;
            auto launch = [&](int it) {
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, it, 1u); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, it, 1u); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, it, 1u); break;
                default: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, it, 1u); break;
                }
            };
            launch(10); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_wave = (double)iters * 16 * ipu[kind];
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf("%-30s waves/SIMD=%d  %.3f ms  cycles per wave-instruction per SIMD = %.2f  (%.1f T lane-ops/s)\n", names[kind],
                   wg_per_cu, ms, cyc / (insts_per_wave * wg_per_cu), insts_per_wave * 64 * 4 * wg_per_cu * cus / (ms * 1e-3) / 1e12);
        }
    return 0;
}
