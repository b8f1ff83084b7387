// What does a scalar branch cost a wave on gfx950?  One loop body = 8 x { s_bitcmp1; s_cbranch; 4 x v_fma (one of them
// behind the branch target's label) }, with the branches all TAKEN (jumping over one v_fma each) or all NOT taken; the
// difference per branch is its cost beyond the two scalar instructions.  1, 2 and 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 branch_cost.hip -o branch_cost       usage: ./branch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

#define UNIT(n)                                   \
    "s_bitcmp1_b32 %[m], " #n "\n"                \
    "s_cbranch_scc1 .Lskip" #n "_%=\n"            \
    "v_fma_f32 %[a], %[a], %[b], %[b]\n"          \
    ".Lskip" #n "_%=:\n"                          \
    "v_fma_f32 %[b], %[a], %[b], %[a]\n"          \
    "v_fma_f32 %[c], %[a], %[c], %[b]\n"          \
    "v_fma_f32 %[d], %[a], %[d], %[c]\n"

__global__ void k(float *out, unsigned long long *cyc, int iters, unsigned mask)
{
    float a = threadIdx.x * 1e-3f, b = 1.0f, c = 0.5f, d = 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
    {
        __asm__ volatile(UNIT(0) UNIT(1) UNIT(2) UNIT(3) UNIT(4) UNIT(5) UNIT(6) UNIT(7)
                         : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d)
                         : [m] "s"(mask)
                         : "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) { cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

int main()
{
    const int iters = 20000;
    float *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, 256 * 1024 * sizeof(float)));
    CHECK(hipMalloc(&d_cyc, 4096 * sizeof(unsigned long long)));
    for (int wps : {1, 2, 4})
    {
        double res[2];
        for (int taken = 0; taken < 2; taken++)
        {
            const int threads = 64 * 4 * wps;      // one workgroup per CU: wps waves on each of the 4 SIMDs
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d_out, d_cyc, iters, taken ? 0xffu : 0u);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[4096];
            CHECK(hipMemcpy(h, d_cyc, 256 * 4 * wps * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double s = 0;
            for (int i = 0; i < 256 * 4 * wps; i++) { s += (double)h[i]; }
            res[taken] = s / (256 * 4 * wps) / iters / 8;     // cycles per unit per wave
        }
        printf("%d waves per SIMD: cycles per unit {bitcmp, cbranch, 3-4 v_fma} per wave: not taken (4 fma) %.1f, taken (3 fma) %.1f  -> a taken branch costs %.1f cycles more than a not-taken one plus a v_fma\n",
               wps, res[0], res[1], res[1] - res[0]);
    }
    return 0;
}
