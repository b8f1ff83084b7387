// Micro-benchmark: achievable VALU issue rate on gfx950 for plain v_fma_f32, v_mul_f32,
// v_rsq_f32 and v_pk_fma_f32 as a function of waves per SIMD.  Prints cycles per
// wave-instruction per SIMD derived from wall time and the nominal 2.4 GHz clock, plus the
// in-kernel clock ratio (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) k(float *out, int iters, unsigned long long *clk)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 1e-7f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (KIND == 0) { a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c); a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c); }
            if (KIND == 1) { a0 *= b; a1 *= b; a2 *= b; a3 *= b; a4 *= b; a5 *= b; a6 *= b; a7 *= b; }
            if (KIND == 2) { a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3); a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7); }
            if (KIND == 4) { a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); }
            if (KIND == 5) { a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); }
            if (KIND == 3)
            {
                typedef float v2 __attribute__((ext_vector_type(2)));
                v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
                p0 = __builtin_elementwise_fma(p0, bb, cc); p1 = __builtin_elementwise_fma(p1, bb, cc);
                p2 = __builtin_elementwise_fma(p2, bb, cc); p3 = __builtin_elementwise_fma(p3, bb, cc);
                a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char *name, int per_instr)
{
    float *out; unsigned long long *clk, h[2];
    hipMalloc(&out, 64 * 65536 * sizeof(float)); hipMalloc(&clk, 16);
    const int iters = 4000;
    for (int wps : {1, 2, 3, 4, 8})
    {
        const int nblk = 256 * 4 * wps;   // one-wave blocks; wps waves per SIMD if spread evenly
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<nblk, 64>>>(out, 10, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<nblk, 64>>>(out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ninstr = (double)iters * 64 / per_instr * wps;   // wave-instructions per SIMD
        const double ghz = (double)h[0] / h[1] * 0.1;
        printf("%-10s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instr per SIMD at measured %.2f GHz (%.2f at 2.4)\n",
               name, wps, ms, ms * 1e-3 * ghz * 1e9 / ninstr, ghz, ms * 1e-3 * 2.4e9 / ninstr);
    }
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("v_fma_f32", 1);
    run<1>("v_mul_f32", 1);
    run<2>("v_rsq_f32", 1);
    run<3>("v_pk_fma", 2);
    run<4>("fma dep1", 1);
    run<5>("fma dep2", 1);
    return 0;
}
