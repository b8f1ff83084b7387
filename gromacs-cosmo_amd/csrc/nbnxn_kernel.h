// Internal interface between the C-ABI layer (nbnxn_hip_api.hip) and the kernels
// (nbnxn_kernel.hip).  Not part of the public ABI.
#ifndef NBNXN_KERNEL_H
#define NBNXN_KERNEL_H

#include <hip/hip_runtime.h>

#include "nbnxn_hip.h"

// LJ tables up to this many atom types are staged in LDS (2 KB); larger ones are read
// through L1/L2.
#define NBX_KERNEL_MAX_LDS_TYPES 16
// energy / shift-force accumulation slots: each is {e_lj, e_el, fshift[45][3]} padded to 144 floats
#define NBX_OUT_SLOTS 64
#define NBX_OUT_STRIDE 144

// Kernel-ready constants, derived on the host in double (nbnxn_hip_api.hip: fill_kernel_consts).
// For analytical Ewald the kernel works in coordinates scaled by beta, so that
// r'^2 = (beta r)^2 is directly the argument of the correction fits and no pair ever
// multiplies by beta: with u = 1/r' the pair force over r' is
//   qq beta^2 (m u^3 + g(r'^2)) + m u^8 (c12 beta^13 u^6 - c6 beta^7)      (m: exclusion bit)
// and the energies come out multiplied by beta.  For reaction field all scales are 1.
struct nbx_kernel_k_t {
    float xscale;      // beta | 1: multiplies coordinates
    float qscale;      // epsfac beta^2 | epsfac: multiplies the i-charges
    float c6scale;     // beta^7 | 1
    float c12scale;    // beta^13 | 1
    float escale;      // 1/beta | 1: multiplies the accumulated energies at the end
    float rc2, rlist2; // scaled cut-off and list radius, squared
    float rvdw2;       // scaled LJ cut-off squared (twin-range flavour)
    float r2inc;       // NBX_AVOID_SING_R2_INC in scaled units (beta^2 x 1e-12 | 1e-12): added to r^2 of excluded pairs
    // LJ force / potential switch (forcerec.c:1942-1975), in the kernel's scaled units: with
    // r' = beta r the constants of a term c*rsw^n become c / beta^(n + its prefactor power),
    // see fill_kernel_consts in nbnxn_hip_api.hip
    float vdw_rsw;     // scaled rvdw_switch
    float fsw_d2, fsw_d3, fsw_r2, fsw_r3;   // force switch: dispersion c2, c3; repulsion c2, c3
    float psw_c3, psw_c4, psw_c5;           // potential switch
    float sh;          // sh_ewald / beta
    float rep12;       // rep_cpot / 12, scaled by beta^-12 for Ewald
    float ndisp6;      // -disp_cpot / 6, scaled by beta^-6 for Ewald
    float self;        // coefficient of q_i'^2 (scaled charges) in the scaled self energy
    float two_k_rf;    // reaction field: 2 k_rf
    float c_rf;        // reaction field: c_rf
    // LJ-PME grid part (nbnxn_cuda_kernel_utils.cuh:186-302), scaled units: with cl = ewaldcoeff_lj / beta
    float lje2;        // cl^2
    float lje2_nlog2e; // -cl^2 log2(e): exp(-cl^2 r'^2) = exp2(r'^2 * lje2_nlog2e)
    float lje6_6;      // cl^6 / 6
    float lje_sh6;     // sh_lj_ewald / beta^6 / 6
    // tabulated Ewald (nbnxn_cuda_kernel_utils.cuh:304-317): table index = r' * tab_scale
    float tab_scale;   // coulomb_tab_scale / beta
    float tab_max;     // table size - 2, as float: clamp of the interpolation index
};

// Everything a launch needs (the reference passes cu_atomdata_t / cu_nbparam_t / cu_plist_t
// by value: nbnxn_cuda.cu:384).
struct nbx_kernel_args_t {
    // atom data (cu_atomdata_t, nbnxn_cuda_types.h:114-139)
    const float4 *xq;
    float        *f;
    const int    *type;
    const float  *shift_vec;
    int           ntype;
    // parameters (cu_nbparam_t, :142-185)
    const float2  *nbfp;
    const float   *c6grid;      // LJ-PME: grid C6 per type pair [ntype][ntype] (6*C6 convention), else null
    const float   *coulomb_tab; // tabulated Ewald: force table / beta^2, else null
    nbx_kernel_k_t k;
    // pair list (cu_plist_t, :188-204)
    const nbx_sci_t  *sci;
    nbx_cj4_t        *cj4;
    const nbx_excl_t *excl;
    // accumulation slots [NBX_OUT_SLOTS][NBX_OUT_STRIDE]
    float *out;
    // optional: zero_n4 float4s at `zero` are cleared by this launch (the idle output buffer)
    float4 *zero;
    int     zero_n4;
};

// el_type: NBX_EL_*; vdw_mod: 0 plain cut-off (with or without potential shift), 1 force switch,
// 2 potential switch, 3 LJ-PME (either combination rule: the rule only shapes the c6grid table)
hipError_t nbx_launch_nonbonded(const nbx_kernel_args_t &a, int nsci, int el_type, bool ener, bool prune, bool vdw_cutoff,
                                int vdw_mod, hipStream_t stream);

// one translation unit per electrostatics flavour (nbnxn_kernel.hip compiled with -DNBX_UNIT_EL=n)
hipError_t nbx_launch_nonbonded_rf(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, bool vdw_cutoff, int vdw_mod, hipStream_t stream);
hipError_t nbx_launch_nonbonded_ewald_ana(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, bool vdw_cutoff, int vdw_mod, hipStream_t stream);
hipError_t nbx_launch_nonbonded_ewald_tab(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, bool vdw_cutoff, int vdw_mod, hipStream_t stream);

// nbnxn_prune.hip: prune-only pass over sci[sci_begin, sci_begin + nsci); fresh = the list has not been
// pruned yet (outer masks are produced from the list's own masks with rlist_outer)
hipError_t nbx_launch_prune(const float4 *xq, const float *shift_vec, const nbx_sci_t *sci, nbx_cj4_t *cj4, unsigned *imask_outer,
                            int sci_begin, int nsci, bool fresh, float rlist_outer, float rlist_inner, hipStream_t stream);

// nbnxn_atomdata.hip
hipError_t nbx_launch_gather_x(const int *a, const float *x, float4 *xq, int begin, int n, hipStream_t stream);
hipError_t nbx_launch_scatter_f(const int *cell, const float *fnb, float *f, int a0, int n, bool add, hipStream_t stream);
// DD halo moves on the grid-ordered arrays (nbnxn_atomdata.hip)
hipError_t nbx_launch_halo_pack_x(const int *slot, int n, const float4 *xq, const float *shift, float4 *buf, hipStream_t stream);
hipError_t nbx_launch_halo_unpack_x(const int *slot, int n, const float4 *buf, float4 *xq, hipStream_t stream);
hipError_t nbx_launch_halo_pack_f(const int *slot, int n, const float *f, float *buf, hipStream_t stream);
hipError_t nbx_launch_halo_unpack_f(const int *slot, int n, const float *buf, float *f, float *fshift_entry, hipStream_t stream);
// several messages in one launch: rows of a message start at a multiple of 64, padding rows have slot -1
hipError_t nbx_launch_halo_pack_x_msgs(const int *slot, int n, const float4 *xq, const int *wave_msg, const float4 *msg_shift, float4 *buf,
                                       hipStream_t stream);
hipError_t nbx_launch_halo_unpack_f_msgs(const int *slot, int n, const float *buf, float *f, const int *wave_msg, const float4 *msg_shift,
                                         float *out, hipStream_t stream);

#endif
