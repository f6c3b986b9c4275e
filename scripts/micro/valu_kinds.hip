// valu_kinds.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU instruction kinds the sketch kernel
// is made of, on gfx950, at 5 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_kinds.hip -o valu_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)

#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#define KERNEL(NAME, STMT)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed)                    \
    {                                                                                                        \
        uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11,   \
                 a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19, b = a0 * 9 + 5, c = a0 * 11 + 7;                   \
        uint64_t q = ((uint64_t)a1 << 32) | a2, r = ((uint64_t)a3 << 32) | a4;                                \
        for (int i = 0; i < iters; ++i) {                                                                    \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { REP8(STMT) }                                     \
        }                                                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b ^ c ^ (uint32_t)q ^ (uint32_t)r; \
    }

#define S_XOR(x) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x) : "v"(b));
#define S_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(b));
#define S_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_BITOP3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36" : "+v"(x) : "v"(b), "v"(c));
#define S_BITOP3C(x) asm volatile("v_bitop3_b32 %0, %0, 1, %1 bitop3:0x48" : "+v"(x) : "v"(b));
#define S_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define S_CND32(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : );
#define S_CND64(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(b) : );
#define S_CMP64(x) asm volatile("v_cmp_lt_u64_e32 vcc, %0, %1" : : "v"(q), "v"(r) : "vcc");
#define S_CMP32(x) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define S_CMP64S(x) asm volatile("v_cmp_lt_u64_e64 s[20:21], %0, %1" : : "v"(q), "v"(r) : "s20", "s21");
#define S_SHL(x) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(x));
#define S_ANDLIT(x) asm volatile("v_and_b32_e32 %0, 0x7ffffffe, %0" : "+v"(x));
#define S_ANDINL(x) asm volatile("v_and_b32_e32 %0, -3, %0" : "+v"(x));
#define S_ADDCO(x) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1\n v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define S_MOV64(x) asm volatile("v_mov_b64_e32 %0, %1" : "=v"(q) : "v"(r));
#define S_BFE(x) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(x));
#define S_LSHR(x) asm volatile("v_lshrrev_b32_e32 %0, 30, %0" : "+v"(x));
#define S_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_BCNT(x) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(x));
#define S_ADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(r));
#define S_ADDU32(x) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_MOV32(x) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(b));
#define S_MINU32(x) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_XOR3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
#define S_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_CMPCND(x) asm volatile("v_cmp_lt_u64_e32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(x) : "v"(q), "v"(r), "v"(b) : "vcc");

KERNEL(k_xor, S_XOR) KERNEL(k_align, S_ALIGN) KERNEL(k_andor, S_ANDOR) KERNEL(k_bitop3, S_BITOP3) KERNEL(k_bitop3c, S_BITOP3C)
KERNEL(k_or3, S_OR3) KERNEL(k_lshlor, S_LSHLOR) KERNEL(k_cnd32, S_CND32) KERNEL(k_cnd64, S_CND64) KERNEL(k_cmp64, S_CMP64)
KERNEL(k_cmp32, S_CMP32) KERNEL(k_cmp64s, S_CMP64S) KERNEL(k_shl, S_SHL) KERNEL(k_andlit, S_ANDLIT) KERNEL(k_andinl, S_ANDINL)
KERNEL(k_addco, S_ADDCO) KERNEL(k_mov64, S_MOV64) KERNEL(k_bfe, S_BFE) KERNEL(k_lshr, S_LSHR) KERNEL(k_add3, S_ADD3)
KERNEL(k_bcnt, S_BCNT) KERNEL(k_cmpcnd, S_CMPCND) KERNEL(k_add64, S_ADD64) KERNEL(k_addu32, S_ADDU32) KERNEL(k_mov32, S_MOV32) KERNEL(k_minu32, S_MINU32) KERNEL(k_xor3, S_XOR3) KERNEL(k_perm, S_PERM)

typedef void (*kern_t)(uint32_t *, int, uint32_t);
int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct { const char *name; kern_t f; double per_stmt; } ks[] = {
        {"v_xor_b32_e32 (VOP2)", k_xor, 1}, {"v_alignbit_b32 v,v,v,31", k_align, 1}, {"v_and_or_b32 (3 VGPR)", k_andor, 1},
        {"v_bitop3_b32 (3 VGPR)", k_bitop3, 1}, {"v_bitop3_b32 v,v,1,v", k_bitop3c, 1}, {"v_or3_b32 (3 VGPR)", k_or3, 1},
        {"v_lshl_or_b32 v,v,1,v", k_lshlor, 1}, {"v_cndmask_b32_e32 vcc", k_cnd32, 1}, {"v_cndmask_b32_e64 sgpr", k_cnd64, 1},
        {"v_cmp_lt_u64_e32", k_cmp64, 1}, {"v_cmp_lt_u32_e32", k_cmp32, 1}, {"v_cmp_lt_u64_e64 sgpr", k_cmp64s, 1},
        {"v_lshlrev_b32_e32 const", k_shl, 1}, {"v_and_b32 literal", k_andlit, 1}, {"v_and_b32 inline const", k_andinl, 1},
        {"v_add_co + v_addc pair", k_addco, 2}, {"v_mov_b64", k_mov64, 1}, {"v_bfe_u32 v,v,4,4", k_bfe, 1},
        {"v_lshrrev_b32_e32 const", k_lshr, 1}, {"v_add3_u32 (3 VGPR)", k_add3, 1}, {"v_bcnt_u32_b32", k_bcnt, 1},
        {"cmp_u64 + s_nop 1 + cndmask (triple)", k_cmpcnd, 1},
        {"v_lshl_add_u64 (64-bit add)", k_add64, 1}, {"v_add_u32_e32", k_addu32, 1}, {"v_mov_b32_e32", k_mov32, 1},
        {"v_min_u32_e32", k_minu32, 1}, {"v_bitop3_b32 xor3 (3 VGPR)", k_xor3, 1}, {"v_perm_b32 (3 VGPR)", k_perm, 1}};
    const int iters = 20000, wg_per_cu = 5;
    printf("%s CUs=%d clock=%.2f GHz, %d waves per SIMD, cycles per wave-instruction per SIMD:\n", p.gcnArchName, cus, ghz, wg_per_cu);
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.f, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.f, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)iters * 64 * k.per_stmt * wg_per_cu;   // wave-instructions per SIMD
        printf("  %-40s %6.2f\n", k.name, ms * 1e-3 * ghz * 1e9 / n);
    }
    return 0;
}
