// Micro-benchmark: does a wave64 VALU instruction on gfx950 get cheaper when part of EXEC is off?  (The nonbonded kernel
// runs its pair body under the cut-off's EXEC mask, lane = 8 j + i: if the hardware skipped a pass whose lanes are all
// off, the order of the lanes would matter.)  8 independent v_fma_f32 with three VGPR sources per unrolled step, EXEC set
// by inline asm for the timed loop only and restored before the result is stored.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#define I_FMA3(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));

__global__ void __launch_bounds__(64) k(float *out, int iters, unsigned long long *clk, unsigned mlo, unsigned mhi)
{
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f + threadIdx.x * 1e-9f;
    const unsigned lo = __builtin_amdgcn_readfirstlane(mlo), hi = __builtin_amdgcn_readfirstlane(mhi);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %1" :: "s"(lo), "s"(hi) : "s20", "s21");
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++) { REP8(I_FMA3) }
    }
    asm volatile("s_mov_b64 exec, s[20:21]" ::: "s20", "s21");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// as the pair body does it: 8 instructions on all lanes, then 16 under the mask, again and again
__global__ void __launch_bounds__(64) k_alt(float *out, int iters, unsigned long long *clk, unsigned mlo, unsigned mhi)
{
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f + threadIdx.x * 1e-9f;
    const unsigned lo = __builtin_amdgcn_readfirstlane(mlo), hi = __builtin_amdgcn_readfirstlane(mhi);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21");
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            REP8(I_FMA3)
            asm volatile("s_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %1" :: "s"(lo), "s"(hi));
            REP8(I_FMA3) REP8(I_FMA3)
            asm volatile("s_mov_b64 exec, s[20:21]" ::: "s20", "s21");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char **)
{
    const bool alt = argc > 1;
    float *out; unsigned long long *clk, h[2];
    hipMalloc(&out, 64 * 65536 * sizeof(float)); hipMalloc(&clk, 16);
    const int iters = 2000;
    struct { const char *name; unsigned lo, hi; } masks[] = {
        {"all 64 lanes", 0xffffffffu, 0xffffffffu}, {"lanes 0-31", 0xffffffffu, 0u}, {"lanes 32-63", 0u, 0xffffffffu},
        {"lanes 0-47", 0xffffffffu, 0xffffu}, {"lanes 0-23", 0xffffffu, 0u}, {"lanes 0-16", 0x1ffffu, 0u},
        {"lanes 0-15", 0xffffu, 0u}, {"lanes 16-31", 0xffff0000u, 0u}, {"lanes 0-15 and 32-47", 0xffffu, 0xffffu},
        {"lanes 0-7", 0xffu, 0u}, {"lanes 0-3", 0xfu, 0u}, {"lane 0", 1u, 0u}, {"lanes 0 and 32", 1u, 1u}, {"lanes 0 and 16", 0x10001u, 0u},
        {"lane 0 of each 16", 0x10001u, 0x10001u}, {"lane 0 of each 8", 0x01010101u, 0x01010101u},
        {"every other lane", 0x55555555u, 0x55555555u}, {"lanes 0-7 of each 16", 0x00ff00ffu, 0x00ff00ffu},
        {"lanes 5, 13, 42", 0x2020u, 0x400u}, {"all 64 lanes again", 0xffffffffu, 0xffffffffu}};
    for (auto &m : masks)
    {
        for (int wps : {1, 5})
        {
            const int nblk = 256 * 4 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (alt) { k_alt<<<nblk, 64>>>(out, 10, clk, m.lo, m.hi); } else { k<<<nblk, 64>>>(out, 10, clk, m.lo, m.hi); }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (alt) { k_alt<<<nblk, 64>>>(out, iters * 64 / 96, clk, m.lo, m.hi); } else { k<<<nblk, 64>>>(out, iters, clk, m.lo, m.hi); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double ninstr = (double)(alt ? (iters * 64 / 96) * 96 : iters * 64) * wps;
            const double ghz = (double)h[0] / h[1] * 0.1;
            printf("%-22s waves/SIMD %d: %.2f cycles per wave-instr per SIMD at measured %.2f GHz; one wave: %.2f cycles/instr\n",
                   m.name, wps, ms * 1e-3 * ghz * 1e9 / ninstr, ghz, (double)h[0] / (alt ? (iters * 64 / 96) * 96.0 : iters * 64.0));
        }
    }
    hipFree(out); hipFree(clk);
    return 0;
}
