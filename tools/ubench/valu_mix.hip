// Micro-benchmark: issue cost on gfx950 of the VALU instruction kinds the nonbonded kernel's
// pair body is made of (inline asm, 8 independent registers, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)

template <int KIND>
__global__ void __launch_bounds__(64) k(float *out, int iters, unsigned long long *clk)
{
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f + threadIdx.x * 1e-9f;
    asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21");
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
#define I_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define I_FMAC(x) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(b));
#define I_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define I_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
#define I_CMP(x) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define I_CMPS(x) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" : : "v"(x), "v"(b) : "s20", "s21");
#define I_DPP(x) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define I_BFE(x) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(x));
#define I_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_FMAS(x) asm volatile("v_fma_f32 %0, %0, %1, s8" : "+v"(x) : "v"(b));
#define I_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define I_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define I_FMAAK(x) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800001" : "+v"(x) : "v"(b));
#define I_CND64(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(b));
#define I_CNDZ(x) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x));
#define I_CMPCND(x) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(x) : "v"(b) : "vcc");
#define I_FMA3(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_FMAC2(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_MUL2(x) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x) : "v"(b), "v"(c));
#define I_FMAN(x) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
            if (KIND == 18) { REP8(I_CND64) }
            if (KIND == 19) { REP8(I_CNDZ) }
            if (KIND == 20) { REP8(I_CMPCND) }
            if (KIND == 21) { REP8(I_FMA3) }
            if (KIND == 22) { REP8(I_FMAC2) }
            if (KIND == 23) { REP8(I_MUL2) }
            if (KIND == 24) { REP8(I_FMAN) }
            if (KIND == 0) { REP8(I_FMA) }
            if (KIND == 1) { REP8(I_FMAC) }
            if (KIND == 2) { REP8(I_MUL) }
            if (KIND == 3) { REP8(I_SUB) }
            if (KIND == 4) { REP8(I_ADD) }
            if (KIND == 5) { REP8(I_MOV) }
            if (KIND == 6) { REP8(I_AND) }
            if (KIND == 7) { REP8(I_CND) }
            if (KIND == 8) { REP8(I_CMP) }
            if (KIND == 9) { REP8(I_CMPS) }
            if (KIND == 10) { REP8(I_DPP) }
            if (KIND == 11) { REP8(I_BFE) }
            if (KIND == 12) { REP8(I_ADDU) }
            if (KIND == 13) { REP8(I_FMAS) }
            if (KIND == 14) { REP8(I_LSHL) }
            if (KIND == 15) { REP8(I_MAX) }
            if (KIND == 16) { REP8(I_RCP) }
            if (KIND == 17) { REP8(I_FMAAK) }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char *name)
{
    float *out; unsigned long long *clk, h[2];
    hipMalloc(&out, 64 * 65536 * sizeof(float)); hipMalloc(&clk, 16);
    const int iters = 2000;
    for (int wps : {1, 4})
    {
        const int nblk = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<nblk, 64>>>(out, 10, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<nblk, 64>>>(out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ninstr = (double)iters * 64 * wps;
        const double ghz = (double)h[0] / h[1] * 0.1;
        printf("%-14s waves/SIMD %d: %.2f cycles per wave-instr per SIMD at measured %.2f GHz; one wave: %.2f cycles/instr\n",
               name, wps, ms * 1e-3 * ghz * 1e9 / ninstr, ghz, (double)h[0] / (iters * 64.0));
    }
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("v_fma_f32"); run<1>("v_fmac_f32"); run<2>("v_mul_f32"); run<3>("v_sub_f32"); run<4>("v_add_f32");
    run<5>("v_mov_b32"); run<6>("v_and_b32"); run<7>("v_cndmask"); run<8>("v_cmp vcc"); run<9>("v_cmp sgpr");
    run<10>("v_add_dpp"); run<11>("v_bfe_i32"); run<12>("v_add_u32"); run<13>("v_fma sgpr"); run<14>("v_lshlrev");
    run<15>("v_max_f32"); run<18>("cndmask e64 sgpr"); run<19>("cndmask 0,x,vcc"); run<20>("cmp+cndmask (2)"); run<21>("fma 3 vgpr"); run<22>("fmac 2 src"); run<23>("mul 2 src"); run<24>("fma neg acc"); run<16>("v_rcp_f32"); run<17>("v_fmaak_f32");
    return 0;
}
