// Device side of the two per-step atom-data moves the reference does on the host
// (SURVEY 8f-1): nbnxn_atomdata_copy_x_to_nbat_x (nbnxn_atomdata.c:1141, inner loops :256-295)
// and nbnxn_atomdata_add_nbat_f_to_f (:1600, inner loop :1329-1336).  Both are index
// permutations over ~30 bytes per atom: HBM/L2-bound, one thread per grid slot / atom,
// coalesced on the grid-ordered side, gathered on the caller-ordered side.
#include <hip/hip_runtime.h>

#include "nbnxn_kernel.h"

namespace {

// xq[s].xyz = x[a[s]] for the slots that hold an atom; fillers and the charges stay as they
// were uploaded with the atom data (the host fixed them at search time).
__global__ void __launch_bounds__(256) nbnxn_gather_x(const int *__restrict__ a, const float *__restrict__ x,
                                                      float4 *__restrict__ xq, int begin, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) { return; }
    const int s  = begin + i;
    const int at = a[s];
    if (at < 0) { return; }
    float4 v = xq[s];
    v.x = x[3 * at + 0];
    v.y = x[3 * at + 1];
    v.z = x[3 * at + 2];
    xq[s] = v;
}

// ADD: f[at] += fnb[cell[at]]   (f on the device)      !ADD: fa[i] = fnb[cell[a0 + i]] (staging for a host f)
template <bool ADD>
__global__ void __launch_bounds__(256) nbnxn_scatter_f(const int *__restrict__ cell, const float *__restrict__ fnb,
                                                       float *__restrict__ f, int a0, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) { return; }
    const int    c  = cell[a0 + i];
    const float  fx = fnb[3 * c + 0], fy = fnb[3 * c + 1], fz = fnb[3 * c + 2];
    float       *o  = f + (size_t)3 * (ADD ? a0 + i : i);
    if (ADD) { o[0] += fx; o[1] += fy; o[2] += fz; }
    else { o[0] = fx; o[1] = fy; o[2] = fz; }
}

} // namespace

hipError_t nbx_launch_gather_x(const int *a, const float *x, float4 *xq, int begin, int n, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    hipLaunchKernelGGL(nbnxn_gather_x, dim3((n + 255) / 256), dim3(256), 0, stream, a, x, xq, begin, n);
    return hipGetLastError();
}

hipError_t nbx_launch_scatter_f(const int *cell, const float *fnb, float *f, int a0, int n, bool add, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    if (add) { hipLaunchKernelGGL(nbnxn_scatter_f<true>, dim3((n + 255) / 256), dim3(256), 0, stream, cell, fnb, f, a0, n); }
    else { hipLaunchKernelGGL(nbnxn_scatter_f<false>, dim3((n + 255) / 256), dim3(256), 0, stream, cell, fnb, f, a0, n); }
    return hipGetLastError();
}
