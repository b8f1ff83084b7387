// Device detection, selection and memory tests: the counterparts of gmxlib/gpu_utils/gpu_utils.cu
// behind plain C types (declared in include/nbnxn_hip.h).  gmx_gpu_info_t embeds cudaDeviceProp
// and gmx_gpu_opt_t owns heap arrays, so neither crosses this ABI: the caller holds an array of
// nbx_gpu_dev_info_t (what cuda_dev_info_t is to the reference) and a plain list of indices.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

#include "nbnxn_hip.h"

namespace {

// the reference runs a dummy kernel and checks that it ran (do_sanity_checks, gpu_utils.cu:108-190)
__global__ void k_sanity(int *out, int v) { out[threadIdx.x] = v + (int)threadIdx.x; }

int sanity_check(int dev_id, hipDeviceProp_t *prop)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { return -1; }
    int id = dev_id;
    if (dev_id == -1) { if (hipGetDevice(&id) != hipSuccess) { return -1; } }   // the device already in use
    if (id < 0 || id > ndev - 1) { return -1; }
    if (hipGetDeviceProperties(prop, id) != hipSuccess) { return -1; }
    if (prop->major < 1) { return -1; }
    if (dev_id != -1 && hipSetDevice(id) != hipSuccess) { return -1; }
    // code objects of another architecture make the launch fail: such a device is reported as
    // incompatible by the caller, not as broken, so the dummy kernel only runs where it can
    if (strncmp(prop->gcnArchName, "gfx950", 6) != 0) { return 0; }
    int *d = nullptr, h[64];
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) { return -1; }
    hipLaunchKernelGGL(k_sanity, dim3(1), dim3(64), 0, 0, d, 1000);
    const bool ok = hipGetLastError() == hipSuccess && hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) { return -1; }
    for (int i = 0; i < 64; i++) { if (h[i] != 1000 + i) { return -1; } }
    return 0;
}

// is_gmx_supported_gpu (gpu_utils.cu:618-637) asks for compute capability >= 2.0; this library
// carries gfx950 code objects only
bool is_supported(const hipDeviceProp_t &p) { return strncmp(p.gcnArchName, "gfx950", 6) == 0; }

// ---- memory tests: the reference links memtestG80 (moving inversions, modulo-20, LCG logic tests);
// here the same families of tests as plain kernels over one buffer of 32-bit words ----
__device__ __forceinline__ unsigned lcg(unsigned v) { return v * 1664525u + 1013904223u; }

__global__ void k_mt_write(unsigned *p, size_t n, unsigned seed, int mode)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        unsigned v;
        if (mode == 0) { v = seed; }                                             // moving inversions: constant pattern
        else if (mode == 1) { v = (i % 20 == seed % 20) ? 0xffffffffu : 0u; }    // modulo-20
        else { v = lcg((unsigned)i ^ seed); }                                    // address-dependent pseudo-random
        p[i] = v;
    }
}

__global__ void k_mt_check_invert(unsigned *p, size_t n, unsigned seed, int mode, unsigned *errors)
{
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        unsigned v;
        if (mode == 0) { v = seed; }
        else if (mode == 1) { v = (i % 20 == seed % 20) ? 0xffffffffu : 0u; }
        else { v = lcg((unsigned)i ^ seed); }
        const unsigned got = p[i];
        bad += (got != v);
        p[i] = ~v;
    }
    if (bad) { atomicAdd(errors, bad); }
}

__global__ void k_mt_check_inverted(const unsigned *p, size_t n, unsigned seed, int mode, unsigned *errors)
{
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        unsigned v;
        if (mode == 0) { v = seed; }
        else if (mode == 1) { v = (i % 20 == seed % 20) ? 0xffffffffu : 0u; }
        else { v = lcg((unsigned)i ^ seed); }
        bad += (p[i] != ~v);
    }
    if (bad) { atomicAdd(errors, bad); }
}

// returns the number of mismatching words (0 = pass), -1 if the test could not run
long long run_memtest(size_t megs, int iterations, int time_limit_s)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { return -1; }
    size_t bytes = megs << 20;
    if (bytes > free_b - free_b / 10) { bytes = free_b - free_b / 10; }
    const size_t n = bytes / sizeof(unsigned);
    if (n == 0) { return -1; }
    unsigned *buf = nullptr, *d_err = nullptr;
    if (hipMalloc((void **)&buf, n * sizeof(unsigned)) != hipSuccess) { return -1; }
    if (hipMalloc((void **)&d_err, sizeof(unsigned)) != hipSuccess) { (void)hipFree(buf); return -1; }
    (void)hipMemset(d_err, 0, sizeof(unsigned));
    const auto t0 = std::chrono::steady_clock::now();
    const dim3 grid(4096), block(256);
    bool       fail = false;
    for (int it = 0; (time_limit_s > 0 || it < iterations) && !fail; it++)
    {
        for (int mode = 0; mode < 3 && !fail; mode++)
        {
            const int nseed = (mode == 1) ? 20 : 2;
            for (int s = 0; s < nseed && !fail; s++)
            {
                const unsigned seed = (mode == 0) ? (s ? 0xaaaaaaaau : 0x55555555u) : (mode == 1 ? (unsigned)s : 0x9e3779b9u * (unsigned)(it * 2 + s + 1));
                hipLaunchKernelGGL(k_mt_write, grid, block, 0, 0, buf, n, seed, mode);
                hipLaunchKernelGGL(k_mt_check_invert, grid, block, 0, 0, buf, n, seed, mode, d_err);
                hipLaunchKernelGGL(k_mt_check_inverted, grid, block, 0, 0, (const unsigned *)buf, n, seed, mode, d_err);
                fail = hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess;
            }
        }
        if (time_limit_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= time_limit_s) { break; }
    }
    unsigned h_err = 0;
    if (!fail) { fail = hipMemcpy(&h_err, d_err, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess; }
    (void)hipFree(buf); (void)hipFree(d_err);
    return fail ? -1 : (long long)h_err;
}

int memtest_entry(int dev_id, size_t megs, int iterations, int time_limit_s, bool whole_memory)
{
    hipDeviceProp_t prop;
    if (sanity_check(dev_id, &prop) != 0 || !is_supported(prop)) { return -1; }
    if (whole_memory) { megs = prop.totalGlobalMem >> 20; }
    const long long r = run_memtest(megs, iterations, time_limit_s);
    return r < 0 ? -1 : (r > 0x7fffffff ? 0x7fffffff : (int)r);
}

} // namespace

extern "C" {

size_t nbnxn_hip_sizeof_dev_info(void) { return sizeof(nbx_gpu_dev_info_t); }

int nbnxn_hip_detect_gpus_info(nbx_gpu_dev_info_t *info, int info_alloc, int *ndev, int *ndev_compatible, char *err_str, int err_len)
{
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (ndev) { *ndev = 0; }
    if (ndev_compatible) { *ndev_compatible = 0; }
    if (e != hipSuccess)
    {
        if (err_str && err_len > 0) { snprintf(err_str, err_len, "%s", hipGetErrorString(e)); }
        return 1;
    }
    if (ndev) { *ndev = n; }
    int prev = -1;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    int ncompat = 0;
    for (int i = 0; i < n; i++)
    {
        hipDeviceProp_t p;
        memset(&p, 0, sizeof(p));
        const int sane = sanity_check(i, &p);
        const int stat = sane != 0 ? NBX_GPU_INSANE : (is_supported(p) ? NBX_GPU_COMPATIBLE : NBX_GPU_INCOMPATIBLE);
        if (stat == NBX_GPU_COMPATIBLE) { ncompat++; }
        if (info && i < info_alloc)
        {
            memset(&info[i], 0, sizeof(info[i]));
            info[i].id = i; info[i].stat = stat;
            snprintf(info[i].name, sizeof(info[i].name), "%s", p.name);
            snprintf(info[i].arch, sizeof(info[i].arch), "%s", p.gcnArchName);
            info[i].compute_units = p.multiProcessorCount;
            info[i].clock_mhz     = p.clockRate / 1000;
            info[i].total_mem     = p.totalGlobalMem;
        }
    }
    if (have_prev) { (void)hipSetDevice(prev); }
    if (ndev_compatible) { *ndev_compatible = ncompat; }
    return 0;
}

int nbnxn_hip_pick_compatible_gpus(const nbx_gpu_dev_info_t *info, int ndev, int *dev_use, int dev_use_alloc, int *ndev_use)
{
    int n = 0;
    for (int i = 0; i < ndev; i++)
    {
        if (info[i].stat == NBX_GPU_COMPATIBLE)
        {
            if (dev_use && n < dev_use_alloc) { dev_use[n] = i; }
            n++;
        }
    }
    if (ndev_use) { *ndev_use = n; }
    return (dev_use && n > dev_use_alloc) ? 1 : 0;
}

int nbnxn_hip_check_selected_gpus(int *checkres, const nbx_gpu_dev_info_t *info, int ndev, const int *dev_use, int ndev_use)
{
    int all_ok = 1;
    for (int i = 0; i < ndev_use; i++)
    {
        const int id  = dev_use[i];
        const int res = (id < 0 || id >= ndev) ? NBX_GPU_NONEXISTENT : info[id].stat;
        if (checkres) { checkres[i] = res; }
        all_ok = all_ok && (res == NBX_GPU_COMPATIBLE);
    }
    return all_ok;
}

int nbnxn_hip_get_gpu_device_id(const nbx_gpu_dev_info_t *info, int ndev, const int *dev_use, int ndev_use, int index)
{
    if (!info || !dev_use || index < 0 || index >= ndev_use || dev_use[index] < 0 || dev_use[index] >= ndev) { return -1; }
    return info[dev_use[index]].id;
}

int nbnxn_hip_do_quick_memtest(int dev_id) { return memtest_entry(dev_id, 250, 1, 0, false); }
int nbnxn_hip_do_full_memtest(int dev_id) { return memtest_entry(dev_id, 0, 2, 0, true); }
int nbnxn_hip_do_timed_memtest(int dev_id, int time_limit) { return memtest_entry(dev_id, 0, 1, time_limit > 0 ? time_limit : 1, true); }

} // extern "C"
