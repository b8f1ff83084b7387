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
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0p = {a0, a1}, a1p = {a1, a2}, a2p = {a2, a3}, a3p = {a3, a4}, a4p = {a4, a5}, a5p = {a5, a6}, a6p = {a6, a7}, a7p = {a7, a0}, bp = {b, c};
    asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21");
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
    asm volatile("s_mov_b32 s22, 0\n\ts_mov_b32 s23, 0" ::: "s22", "s23");
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
#define I_BFI(x) asm volatile("v_bfi_b32 %0, %1, 0, %0" : "+v"(x) : "v"(b));
#define I_BFIS(x) asm volatile("v_bfi_b32 %0, %1, 0, s8" : "+v"(x) : "v"(b));
#define I_FMA1(x) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x) : "v"(b));
#define I_MULS(x) asm volatile("v_mul_f32 %0, s8, %0" : "+v"(x));
#define I_ANDC(x) asm volatile("v_and_b32 %0, 63, %0" : "+v"(x));
#define I_FRACT(x) asm volatile("v_fract_f32 %0, %0" : "+v"(x));
#define I_CVTU(x) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));
#define I_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(x) : "v"(b));
#define I_SUBS(x) asm volatile("v_sub_f32 %0, s8, %0" : "+v"(x));
#define I_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MULC(x) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x));
#define I_BFEU(x) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x##p) : "v"(bp));
#define I_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x##p) : "v"(bp));
#define I_ASHR(x) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x));
#define I_CNDV(x) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(x) : "v"(b));
#define I_MOVC(x) asm volatile("v_mov_b32 %0, 0" : "=v"(x));
#define I_FMAMK(x) asm volatile("v_fmamk_f32 %0, %0, 0x3f800001, %1" : "+v"(x) : "v"(b));
#define I_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define I_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_SUBREV(x) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(x) : "v"(b));
#define I_DPPMOV(x) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define I_DPPHM(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(x));
            if (KIND == 50) { I_RCP(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) I_FMA3(a4) I_FMA3(a5) I_FMA3(a6) I_FMA3(a7) }
            if (KIND == 51) { I_RCP(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) I_RCP(a4) I_FMA3(a5) I_FMA3(a6) I_FMA3(a7) }
            if (KIND == 52) { I_CMPCND(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) I_CMPCND(a4) I_FMA3(a5) I_FMA3(a6) I_FMA3(a7) }
            if (KIND == 53) { I_DPP(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) I_DPP(a4) I_FMA3(a5) I_FMA3(a6) I_FMA3(a7) }
            if (KIND == 54) { I_FMAS(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) I_FMAS(a4) I_FMA3(a5) I_FMA3(a6) I_FMA3(a7) }
#define I_SALU asm volatile("s_add_u32 s22, s22, 1" ::: "s22", "scc");
#define I_BRNT asm volatile("s_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1f\n\t1:" ::: "scc");
#define I_BRT(x) asm volatile("s_cmp_lg_u32 s23, 0\n\ts_cbranch_scc0 1f\n\tv_mov_b32 %0, 0\n\t1:" : "+v"(x) :: "scc");
            if (KIND == 60) { I_FMA3(a0) I_SALU I_FMA3(a1) I_SALU I_FMA3(a2) I_SALU I_FMA3(a3) I_SALU I_FMA3(a4) I_SALU I_FMA3(a5) I_SALU I_FMA3(a6) I_SALU I_FMA3(a7) I_SALU }
            if (KIND == 61) { I_FMA3(a0) I_SALU I_SALU I_FMA3(a1) I_SALU I_SALU I_FMA3(a2) I_SALU I_SALU I_FMA3(a3) I_SALU I_SALU I_FMA3(a4) I_SALU I_SALU I_FMA3(a5) I_SALU I_SALU I_FMA3(a6) I_SALU I_SALU I_FMA3(a7) I_SALU I_SALU }
            if (KIND == 62) { I_FMA3(a0) I_FMA3(a1) I_BRNT I_FMA3(a2) I_FMA3(a3) I_BRNT I_FMA3(a4) I_FMA3(a5) I_BRNT I_FMA3(a6) I_FMA3(a7) I_BRNT }
            if (KIND == 63) { I_FMA3(a0) I_FMA3(a1) I_BRT(a0) I_FMA3(a2) I_FMA3(a3) I_BRT(a2) I_FMA3(a4) I_FMA3(a5) I_BRT(a4) I_FMA3(a6) I_FMA3(a7) I_BRT(a6) }
            // dependent chains: 8 instructions on 1 / 2 / 4 registers (what a lone wave's pair body looks like)
            if (KIND == 70) { I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) }
            if (KIND == 71) { I_FMA3(a0) I_FMA3(a1) I_FMA3(a0) I_FMA3(a1) I_FMA3(a0) I_FMA3(a1) I_FMA3(a0) I_FMA3(a1) }
            if (KIND == 72) { I_FMA3(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) I_FMA3(a0) I_FMA3(a1) I_FMA3(a2) I_FMA3(a3) }
            if (KIND == 73) { I_RCP(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) I_RCP(a0) I_FMA3(a0) I_FMA3(a0) I_FMA3(a0) }
            if (KIND == 74) { I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU I_FMA3(a0) I_SALU }
            if (KIND == 25) { REP8(I_BFI) }
            if (KIND == 26) { REP8(I_BFIS) }
            if (KIND == 27) { REP8(I_FMA1) }
            if (KIND == 28) { REP8(I_MULS) }
            if (KIND == 29) { REP8(I_ANDC) }
            if (KIND == 30) { REP8(I_FRACT) }
            if (KIND == 31) { REP8(I_CVTU) }
            if (KIND == 32) { REP8(I_LSHLADD) }
            if (KIND == 33) { REP8(I_SUBS) }
            if (KIND == 34) { REP8(I_MED3) }
            if (KIND == 35) { REP8(I_MIN) }
            if (KIND == 36) { REP8(I_MULC) }
            if (KIND == 37) { REP8(I_BFEU) }
            if (KIND == 38) { REP8(I_PKFMA) }
            if (KIND == 39) { REP8(I_PKMUL) }
            if (KIND == 40) { REP8(I_ASHR) }
            if (KIND == 41) { REP8(I_CNDV) }
            if (KIND == 42) { REP8(I_MOVC) }
            if (KIND == 43) { REP8(I_FMAMK) }
            if (KIND == 44) { REP8(I_EXP) }
            if (KIND == 45) { REP8(I_XOR) }
            if (KIND == 46) { REP8(I_SUBREV) }
            if (KIND == 47) { REP8(I_DPPMOV) }
            if (KIND == 48) { REP8(I_DPPHM) }
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
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a0p.x + a1p.y + a2p.x + a3p.y + a4p.x + a5p.y + a6p.x + a7p.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char *name)
{
    float *out; unsigned long long *clk, h[2];
    hipMalloc(&out, 64 * 65536 * sizeof(float)); hipMalloc(&clk, 16);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 5})
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

int main(int argc, char **)
{
    if (argc > 2) { run<21>("8 independent fma"); run<70>("8 fma, one chain"); run<71>("8 fma, two chains"); run<72>("8 fma, four chains"); run<73>("2 rcp + 6 fma, one chain"); run<74>("8 fma one chain + 8 salu"); run<60>("8 fma indep + 8 salu"); return 0; }
    if (argc > 1) { run<60>("8 fma + 8 salu"); run<61>("8 fma + 16 salu"); run<62>("8 fma + 4 (cmp+branch not taken)"); run<63>("8 fma + 4 (cmp + taken branch over a mov)"); run<21>("fma 3 vgpr"); run<50>("1 rcp + 7 fma"); run<51>("2 rcp + 6 fma"); run<52>("2 (cmp+cnd) + 6 fma"); run<53>("2 dpp + 6 fma"); run<54>("2 fma-sgpr + 6 fma"); return 0; }
    run<0>("v_fma_f32"); run<1>("v_fmac_f32"); run<2>("v_mul_f32"); run<3>("v_sub_f32"); run<4>("v_add_f32");
    run<5>("v_mov_b32"); run<6>("v_and_b32"); run<7>("v_cndmask"); run<8>("v_cmp vcc"); run<9>("v_cmp sgpr");
    run<10>("v_add_dpp"); run<11>("v_bfe_i32"); run<12>("v_add_u32"); run<13>("v_fma sgpr"); run<14>("v_lshlrev");
    run<15>("v_max_f32"); run<18>("cndmask e64 sgpr"); run<19>("cndmask 0,x,vcc"); run<20>("cmp+cndmask (2)"); run<21>("fma 3 vgpr"); run<22>("fmac 2 src"); run<23>("mul 2 src"); run<24>("fma neg acc"); run<16>("v_rcp_f32"); run<17>("v_fmaak_f32");
    run<50>("1 rcp + 7 fma"); run<51>("2 rcp + 6 fma"); run<52>("2 (cmp+cnd) + 6 fma"); run<53>("2 dpp + 6 fma"); run<54>("2 fma-sgpr + 6 fma");
    run<25>("v_bfi vgpr"); run<26>("v_bfi sgpr"); run<27>("fma inline 1.0"); run<28>("v_mul sgpr"); run<29>("v_and inline"); run<30>("v_fract");
    run<31>("v_cvt_u32_f32"); run<32>("v_lshl_add_u32"); run<33>("v_sub sgpr"); run<34>("v_med3"); run<35>("v_min"); run<36>("v_mul inline 0.5");
    run<37>("v_bfe_u32 vgprs"); run<38>("v_pk_fma_f32"); run<39>("v_pk_mul_f32"); run<40>("v_ashrrev"); run<41>("v_cndmask v,v,vcc"); run<42>("v_mov 0");
    run<43>("v_fmamk"); run<44>("v_exp_f32"); run<45>("v_xor"); run<46>("v_subrev"); run<47>("v_mov_dpp"); run<48>("v_add_dpp half_mirror");
    return 0;
}
