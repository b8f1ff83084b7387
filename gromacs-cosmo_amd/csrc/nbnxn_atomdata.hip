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

// Halo moves of a domain-decomposed step.  The reference packs and unpacks these
// buffers on the CPU (dd_move_x, mdlib/domdec.c:651-758; dd_move_f, :761-875); here the
// coordinates and forces never leave the device, so the same moves are four small kernels over
// the grid-ordered arrays, one thread per halo atom.
//   pack_x:   buf[i] = {xq[slot[i]].xyz + shift, q}     (the owner's side of dd_move_x)
//   unpack_x: xq[slot[i]] = buf[i]                      (the importer's side)
//   pack_f:   buf[i] = f[slot[i]]                       (the importer's side of dd_move_f)
//   unpack_f: f[slot[i]] += buf[i], and with WRAP the sum of buf goes to one shift-force entry
//             (what dd_move_f does for a pulse that crosses a periodic boundary, :837-851)
// One-wave workgroups: they run beside force kernels that hold every wave slot, and a single wave fits
// into whichever slot frees up, where a 256-thread workgroup would wait for four free slots on one CU.
__global__ void __launch_bounds__(64) nbnxn_halo_pack_x(const int *__restrict__ slot, int n, const float4 *__restrict__ xq,
                                                         float sx, float sy, float sz, float4 *__restrict__ buf)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) { return; }
    float4 v = xq[slot[i]];
    v.x += sx; v.y += sy; v.z += sz;
    buf[i] = v;
}

__global__ void __launch_bounds__(64) nbnxn_halo_unpack_x(const int *__restrict__ slot, int n, const float4 *__restrict__ buf,
                                                           float4 *__restrict__ xq)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) { return; }
    const int s = slot[i];
    if (s < 0) { return; }          // padding row of a multi-message layout
    xq[s] = buf[i];
}

__global__ void __launch_bounds__(64) nbnxn_halo_pack_f(const int *__restrict__ slot, int n, const float *__restrict__ f,
                                                         float *__restrict__ buf)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) { return; }
    const int s = slot[i];
    if (s < 0) { return; }          // padding row of a multi-message layout
    buf[3 * i + 0] = f[3 * s + 0];
    buf[3 * i + 1] = f[3 * s + 1];
    buf[3 * i + 2] = f[3 * s + 2];
}

// The force kernels may still be adding to f on the other stream: atomic adds (one address per
// thread, no contention).  WRAP: wave sum, one float atomic per wave and component into one of the 64 accumulation
// slots (fshift_entry points into slot 0; the waves spread over the slots, see nbnxn_halo_unpack_f_msgs).
template <bool WRAP>
__global__ void __launch_bounds__(64) nbnxn_halo_unpack_f(const int *__restrict__ slot, int n, const float *__restrict__ buf,
                                                           float *__restrict__ f, float *__restrict__ fshift_entry)
{
    const int i  = blockIdx.x * 64 + threadIdx.x;
    float     fx = 0, fy = 0, fz = 0;
    if (i < n)
    {
        const int s = slot[i];
        fx = buf[3 * i + 0]; fy = buf[3 * i + 1]; fz = buf[3 * i + 2];
        atomicAdd(f + 3 * s + 0, fx);
        atomicAdd(f + 3 * s + 1, fy);
        atomicAdd(f + 3 * s + 2, fz);
    }
    if (WRAP)
    {
        for (int o = 32; o > 0; o >>= 1)
        {
            fx += __shfl_down(fx, o);
            fy += __shfl_down(fy, o);
            fz += __shfl_down(fz, o);
        }
        if (threadIdx.x == 0)
        {
            float *entry = fshift_entry + (size_t)(blockIdx.x % NBX_OUT_SLOTS) * NBX_OUT_STRIDE;
            atomicAdd(entry + 0, fx);
            atomicAdd(entry + 1, fy);
            atomicAdd(entry + 2, fz);
        }
    }
}

// Several messages in one launch (nbnxn_hip_dd_set_messages): the rows of message k start at a multiple of 64, so a wave
// belongs to one message (wave_msg[wave]) and its shift / shift-force entry are wave-uniform; rows that pad a message
// to the next multiple of 64 have slot -1.  msg_shift[k] = {shift x, y, z, shift index as int bits (-1 = none)}.
__global__ void __launch_bounds__(64) nbnxn_halo_pack_x_msgs(const int *__restrict__ slot, int n, const float4 *__restrict__ xq,
                                                              const int *__restrict__ wave_msg, const float4 *__restrict__ msg_shift,
                                                              float4 *__restrict__ buf)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) { return; }
    const int s = slot[i];
    if (s < 0) { return; }
    const float4 sh = msg_shift[wave_msg[blockIdx.x]];
    float4       v  = xq[s];
    v.x += sh.x; v.y += sh.y; v.z += sh.z;
    buf[i] = v;
}

__global__ void __launch_bounds__(64) nbnxn_halo_unpack_f_msgs(const int *__restrict__ slot, int n, const float *__restrict__ buf,
                                                                float *__restrict__ f, const int *__restrict__ wave_msg,
                                                                const float4 *__restrict__ msg_shift, float *__restrict__ out)
{
    const int i  = blockIdx.x * 64 + threadIdx.x;
    float     fx = 0, fy = 0, fz = 0;
    const int s  = i < n ? slot[i] : -1;
    if (s >= 0)
    {
        fx = buf[3 * i + 0]; fy = buf[3 * i + 1]; fz = buf[3 * i + 2];
        atomicAdd(f + 3 * s + 0, fx);
        atomicAdd(f + 3 * s + 1, fy);
        atomicAdd(f + 3 * s + 2, fz);
    }
    const int is = __float_as_int(msg_shift[wave_msg[blockIdx.x]].w);   // wave-uniform
    if (is >= 0)
    {
        for (int o = 32; o > 0; o >>= 1)
        {
            fx += __shfl_down(fx, o);
            fy += __shfl_down(fy, o);
            fz += __shfl_down(fz, o);
        }
        if (threadIdx.x == 0)
        {
            // one of the 64 accumulation slots {e_lj, e_el, fshift[45][3]}, as the force kernel's waves do: the waves of a
            // message all add to the same shift force, and atomics on ONE address queue up behind each other (500 waves on
            // 21 addresses took 12 us; the fold sums the slots anyway)
            float *entry = out + (size_t)(blockIdx.x % NBX_OUT_SLOTS) * NBX_OUT_STRIDE + 2 + 3 * is;
            atomicAdd(entry + 0, fx);
            atomicAdd(entry + 1, fy);
            atomicAdd(entry + 2, fz);
        }
    }
}

} // namespace

hipError_t nbx_launch_halo_pack_x(const int *slot, int n, const float4 *xq, const float *shift, float4 *buf, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    hipLaunchKernelGGL(nbnxn_halo_pack_x, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, xq, shift[0], shift[1], shift[2], buf);
    return hipGetLastError();
}

hipError_t nbx_launch_halo_unpack_x(const int *slot, int n, const float4 *buf, float4 *xq, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    hipLaunchKernelGGL(nbnxn_halo_unpack_x, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, buf, xq);
    return hipGetLastError();
}

hipError_t nbx_launch_halo_pack_f(const int *slot, int n, const float *f, float *buf, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    hipLaunchKernelGGL(nbnxn_halo_pack_f, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, f, buf);
    return hipGetLastError();
}

hipError_t nbx_launch_halo_unpack_f(const int *slot, int n, const float *buf, float *f, float *fshift_entry, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    if (fshift_entry) { hipLaunchKernelGGL(nbnxn_halo_unpack_f<true>, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, buf, f, fshift_entry); }
    else { hipLaunchKernelGGL(nbnxn_halo_unpack_f<false>, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, buf, f, fshift_entry); }
    return hipGetLastError();
}

hipError_t nbx_launch_halo_pack_x_msgs(const int *slot, int n, const float4 *xq, const int *wave_msg, const float4 *msg_shift, float4 *buf,
                                       hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    hipLaunchKernelGGL(nbnxn_halo_pack_x_msgs, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, xq, wave_msg, msg_shift, buf);
    return hipGetLastError();
}

hipError_t nbx_launch_halo_unpack_f_msgs(const int *slot, int n, const float *buf, float *f, const int *wave_msg, const float4 *msg_shift,
                                         float *out, hipStream_t stream)
{
    if (n <= 0) { return hipSuccess; }
    hipLaunchKernelGGL(nbnxn_halo_unpack_f_msgs, dim3((n + 63) / 64), dim3(64), 0, stream, slot, n, buf, f, wave_msg, msg_shift, out);
    return hipGetLastError();
}

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
