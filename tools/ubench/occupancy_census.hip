// How many one-wave (64-thread) workgroups does a CU of MI355X hold at once?  A grid far larger than the chip runs a
// kernel that counts itself in and out of a per-CU counter (CU identity from HW_REG_HW_ID / HW_REG_XCC_ID) and records
// the largest count it saw; the kernel's register and LDS needs are template / launch parameters, so the limit that
// binds (wave slots, VGPRs, LDS, or a workgroup limit) can be told apart.  Also run with 128- and 256-thread groups.
// build: hipcc --offload-arch=gfx950 -O2 occupancy_census.hip -o occupancy_census      usage: ./occupancy_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NV>
__global__ void census(int *count, int *peak, float *sink, int spin)
{
    extern __shared__ float lds[];
    const unsigned hw  = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    const unsigned cu  = (xcc & 0xf) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xf);   // xcc, se, cu (sh is 0 on this part)
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) { v[k] = threadIdx.x + k; }
    if (threadIdx.x % 64 == 0)
    {
        const int now = atomicAdd(&count[cu], 1) + 1;
        atomicMax(&peak[cu], now);
    }
    for (int i = 0; i < spin; i++)
    {
#pragma unroll
        for (int k = 0; k < NV; k++) { v[k] = fmaf(v[k], 1.000001f, 0.5f); }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < NV; k++) { s += v[k]; }
    if (s == 12345.f) { sink[0] = s + lds[threadIdx.x]; }
    if (threadIdx.x % 64 == 0) { atomicAdd(&count[cu], -1); }
}

template <int NV>
static void run(int threads, size_t lds, int *d_count, int *d_peak, float *d_sink)
{
    CHECK(hipMemset(d_count, 0, 1024 * sizeof(int)));
    CHECK(hipMemset(d_peak, 0, 1024 * sizeof(int)));
    hipLaunchKernelGGL(census<NV>, dim3(256 * 64), dim3(threads), lds, 0, d_count, d_peak, d_sink, 3000);
    CHECK(hipDeviceSynchronize());
    std::vector<int> h(1024);
    CHECK(hipMemcpy(h.data(), d_peak, 1024 * sizeof(int), hipMemcpyDeviceToHost));
    int mx = 0, used = 0;
    long sum = 0;
    for (int v : h) { if (v > 0) { used++; sum += v; mx = std::max(mx, v); } }
    int api = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census<NV>, threads, lds));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(census<NV>)));
    printf("threads %4d  regs %3d  lds %6zu B: peak waves per CU max %2d mean %.1f over %d CUs (occupancy API: %d groups = %d waves)\n", threads,
           fa.numRegs, lds, mx, used ? (double)sum / used : 0.0, used, api, api * threads / 64);
}

int main()
{
    int *d_count, *d_peak;
    float *d_sink;
    CHECK(hipMalloc(&d_count, 1024 * sizeof(int)));
    CHECK(hipMalloc(&d_peak, 1024 * sizeof(int)));
    CHECK(hipMalloc(&d_sink, 64));
    for (int threads : {64, 128, 256})
    {
        run<8>(threads, 0, d_count, d_peak, d_sink);
        run<8>(threads, 4096 * threads / 64, d_count, d_peak, d_sink);
        run<8>(threads, 6144 * threads / 64, d_count, d_peak, d_sink);
        run<56>(threads, 0, d_count, d_peak, d_sink);
        run<56>(threads, 5632 * threads / 64, d_count, d_peak, d_sink);
        run<72>(threads, 0, d_count, d_peak, d_sink);
        run<100>(threads, 0, d_count, d_peak, d_sink);
    }
    return 0;
}
