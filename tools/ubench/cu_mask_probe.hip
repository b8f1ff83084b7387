// Which CUs does a CU-masked stream run on?  Launches a grid that fills the chip on a stream made with
// hipExtStreamCreateWithCUMask and lets every workgroup report where it ran (XCC_ID and HW_ID registers), then
// prints, per mask, the set of (xcc, se, cu) that did NOT appear compared with the unmasked stream.
// build: hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe      usage: ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where(unsigned *out, int spin)
{
    // HW_REG_HW_ID (4): all 32 bits; HW_REG_XCC_ID (20): bits 3:0
    const unsigned hw  = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    float          v   = threadIdx.x;
    for (int i = 0; i < spin; i++) { v = v * 1.000001f + 0.5f; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (v == 12345.f ? 1 : 0); }
}

static std::set<unsigned> run(hipStream_t st, unsigned *d_out, int nwg)
{
    std::vector<unsigned> h(2 * nwg);
    hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, st, d_out, 20000);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::set<unsigned> s;
    for (int i = 0; i < nwg; i++)
    {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        s.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    }
    return s;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount, nwords = (ncu + 31) / 32, nwg = ncu * 64;
    unsigned *d_out;
    CHECK(hipMalloc(&d_out, 2 * nwg * sizeof(unsigned)));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const std::set<unsigned> all = run(st, d_out, nwg);
    printf("%d CUs reported by the device, %zu distinct (xcc, se, sh, cu) seen unmasked\n", ncu, all.size());
    const int tests[][8] = {{0, -1}, {1, -1}, {7, -1}, {8, -1}, {31, -1}, {32, -1}, {33, -1}, {64, -1}, {255, -1}, {0, 1, 2, 3, 4, 5, 6, 7}};
    for (const auto &t : tests)
    {
        std::vector<uint32_t> mask(nwords, 0xffffffffu);
        printf("mask without bits");
        for (int k = 0; k < 8 && t[k] >= 0; k++) { mask[t[k] / 32] &= ~(1u << (t[k] % 32)); printf(" %d", t[k]); }
        hipStream_t ms;
        CHECK(hipExtStreamCreateWithCUMask(&ms, nwords, mask.data()));
        const std::set<unsigned> got = run(ms, d_out, nwg);
        printf(": %zu seen; missing (xcc se sh cu):", got.size());
        for (unsigned v : all) { if (!got.count(v)) { printf(" (%u %u %u %u)", v >> 12, (v >> 8) & 0xf, (v >> 4) & 0xf, v & 0xf); } }
        printf("\n");
        CHECK(hipStreamDestroy(ms));
    }
    return 0;
}
