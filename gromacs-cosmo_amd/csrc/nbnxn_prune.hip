// Prune-only kernel: dynamic ("dual") pair-list pruning.  GROMACS 5.0.4 prunes a list once, inside
// the first force kernel that runs on it (nbnxn_cuda_kernel.cuh:332-340, 503); later versions
// added a list built with a larger outer radius that is re-pruned to a smaller inner radius every
// few steps by a kernel of its own, optionally a slice of the list per step ("rolling").  SURVEY
// 8f-2 names it with the device search, so it is here: the outer interaction masks are kept in an
// array beside the list, the inner ones are what the force kernels read from nbnxn_cj4_t.
//
// One wave64 per list entry, lane = 8 * jatom + iatom as in the force kernel; the distance is
// computed from the raw coordinates with the expression of the reference's pruning rule, so the
// result equals oracle/nbnxn_oracle.c:oracle_prune_list bit for bit.
#include <hip/hip_runtime.h>

#include "nbnxn_kernel.h"

namespace {

template <bool FRESH>
__global__ void __launch_bounds__(64)
nbnxn_prune_kernel(const float4 *__restrict__ xq, const float *__restrict__ shift_vec, const nbx_sci_t *__restrict__ sci,
                   nbx_cj4_t *__restrict__ cj4, unsigned *__restrict__ imask_outer, int sci_begin, float rl_outer2,
                   float rl_inner2)
{
    const int       lane  = threadIdx.x;
    const int       tidxi = lane & 7, tidxj = lane >> 3;
    const nbx_sci_t e     = sci[sci_begin + blockIdx.x];
    __shared__ float4 s_xi[64];
    {
        float4 x = xq[e.sci * 64 + lane];
        x.x += shift_vec[3 * e.shift + 0];
        x.y += shift_vec[3 * e.shift + 1];
        x.z += shift_vec[3 * e.shift + 2];
        s_xi[lane] = x;
    }
    __syncthreads();
    float4 xi[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { xi[i] = s_xi[i * 8 + tidxi]; }

    // software pipeline as in the force kernel: the next group's mask and j-cluster indices are requested
    // before this group's arithmetic, and a group's four j-atoms (one per lane and cluster) all at once
    auto load_hdr = [&](int j4, unsigned &in, int (&cj)[4]) {
        // a list pruned per 32-lane half by the reference's own kernels carries two different masks
        // (nbnxn_cuda_kernel.cuh:503): their union is what a wave64 walks, here as in the force kernel
        in = FRESH ? (cj4[j4].imei[0].imask | cj4[j4].imei[1].imask) : imask_outer[j4];
#pragma unroll
        for (int jm = 0; jm < 4; jm++) { cj[jm] = ((in >> (8 * jm)) & 0xffu) ? cj4[j4].cj[jm] : 0; }   // unused slots hold anything
    };
    unsigned in = 0, in_next = 0;
    int      cj[4] = {0, 0, 0, 0}, cj_next[4] = {0, 0, 0, 0};
    if (e.cj4_ind_start < e.cj4_ind_end) { load_hdr(e.cj4_ind_start, in, cj); }
    for (int j4 = e.cj4_ind_start; j4 < e.cj4_ind_end; j4++)
    {
        float4 xj[4];
#pragma unroll
        for (int jm = 0; jm < 4; jm++) { xj[jm] = xq[cj[jm] * 8 + tidxj]; }
        if (j4 + 1 < e.cj4_ind_end) { load_hdr(j4 + 1, in_next, cj_next); }
        unsigned outer = in, inner = in;
#pragma unroll
        for (int jm = 0; jm < 4; jm++)
        {
            if (!((in >> (8 * jm)) & 0xffu)) { continue; }
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const unsigned bit = 1u << (jm * 8 + i);
                if (in & bit)
                {
                    const float dx = xi[i].x - xj[jm].x, dy = xi[i].y - xj[jm].y, dz = xi[i].z - xj[jm].z;
                    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (FRESH) { if (!__any(r2 < rl_outer2)) { outer &= ~bit; } }
                    if (!__any(r2 < rl_inner2)) { inner &= ~bit; }
                }
            }
        }
        if (lane == 0)
        {
            if (FRESH) { imask_outer[j4] = outer; }
            cj4[j4].imei[0].imask = inner & outer;
            cj4[j4].imei[1].imask = inner & outer;
        }
        in = in_next;
#pragma unroll
        for (int jm = 0; jm < 4; jm++) { cj[jm] = cj_next[jm]; }
    }
}

} // namespace

hipError_t nbx_launch_prune(const float4 *xq, const float *shift_vec, const nbx_sci_t *sci, nbx_cj4_t *cj4, unsigned *imask_outer,
                            int sci_begin, int nsci, bool fresh, float rlist_outer, float rlist_inner, hipStream_t stream)
{
    if (nsci <= 0) { return hipSuccess; }
    if (fresh)
    {
        hipLaunchKernelGGL(nbnxn_prune_kernel<true>, dim3(nsci), dim3(64), 0, stream, xq, shift_vec, sci, cj4, imask_outer, sci_begin,
                           rlist_outer * rlist_outer, rlist_inner * rlist_inner);
    }
    else
    {
        hipLaunchKernelGGL(nbnxn_prune_kernel<false>, dim3(nsci), dim3(64), 0, stream, xq, shift_vec, sci, cj4, imask_outer, sci_begin,
                           rlist_outer * rlist_outer, rlist_inner * rlist_inner);
    }
    return hipGetLastError();
}
