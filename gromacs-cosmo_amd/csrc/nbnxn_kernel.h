// Internal interface between the C-ABI layer (nbnxn_hip_api.hip) and the kernels
// (nbnxn_kernel.hip).  Not part of the public ABI.
#ifndef NBNXN_KERNEL_H
#define NBNXN_KERNEL_H

#include <hip/hip_runtime.h>

#include "nbnxn_hip.h"

// LJ tables up to this many atom types are staged in LDS (2 KB); larger ones are read
// through L1/L2.
#define NBX_KERNEL_MAX_LDS_TYPES 16
// floats per per-entry debug record (diagnostic NBX_STAMPS build only)
#define NBX_PARTIAL_STRIDE 8
// energy / shift-force accumulation slots: each is {e_lj, e_el, fshift[45][3]} padded to 144 floats
#define NBX_OUT_SLOTS 64
#define NBX_OUT_STRIDE 144

// Everything a launch needs, passed by value as the kernel argument (the reference passes
// cu_atomdata_t / cu_nbparam_t / cu_plist_t by value: nbnxn_cuda.cu:384).
struct nbx_kernel_args_t {
    // atom data (cu_atomdata_t, nbnxn_cuda_types.h:114-139)
    const float4 *xq;
    float        *f;
    const int    *type;
    const float  *shift_vec;
    int           ntype;
    // parameters (cu_nbparam_t, :142-185)
    const float2 *nbfp;
    float         epsfac, inv_epsfac;
    float         rc2, rlist2;
    float         two_k_rf, c_rf;
    float         beta, beta2, beta3, sh_ewald;
    float         disp_cpot, rep_cpot;
    // pair list (cu_plist_t, :188-204)
    const nbx_sci_t  *sci;
    nbx_cj4_t        *cj4;
    const nbx_excl_t *excl;
    // per-entry debug records [nsci][NBX_PARTIAL_STRIDE] (diagnostic build)
    float *partial;
    // accumulation slots [NBX_OUT_SLOTS][NBX_OUT_STRIDE]
    float *out;
};

hipError_t nbx_launch_nonbonded(const nbx_kernel_args_t &a, int nsci, int el_type, bool ener, bool prune,
                                int variant, hipStream_t stream);

#endif
