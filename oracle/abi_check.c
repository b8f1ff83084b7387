/*
 * abi_check.c -- proves that the structs of include/nbnxn_hip.h are layout-identical to
 * the reference's (so the reference host code can pass its own objects across the C ABI).
 * TEST INFRASTRUCTURE: compiled against the reference headers by oracle/Makefile where
 * /root/reference exists; prints one line per struct and exits non-zero on a mismatch.
 */
#include <stddef.h>
#include <stdio.h>

#include "typedefs.h"
#include "types/nbnxn_pairlist.h"
#include "types/interaction_const.h"
#include "types/nb_verlet.h"
#include "types/nbnxn_cuda_types_ext.h"
#include "types/force_flags.h"
#include "types/ishift.h"
#include "nbnxn_consts.h"

#include "nbnxn_hip.h"

static int bad = 0;

#define SZ(A, B) do { if (sizeof(A) != sizeof(B)) { printf("SIZE MISMATCH %s %zu vs %s %zu\n", #A, sizeof(A), #B, sizeof(B)); bad++; } \
                      else { printf("ok  sizeof(%s) == sizeof(%s) == %zu\n", #A, #B, sizeof(A)); } } while (0)
#define OFF(A, B, F) do { if (offsetof(A, F) != offsetof(B, F)) { printf("OFFSET MISMATCH %s.%s %zu vs %zu\n", #A, #F, offsetof(A, F), offsetof(B, F)); bad++; } } while (0)
#define VAL(A, B) do { if ((A) != (B)) { printf("VALUE MISMATCH %s=%d %s=%d\n", #A, (int)(A), #B, (int)(B)); bad++; } } while (0)

int main(void)
{
    SZ(nbnxn_sci_t, nbx_sci_t);
    OFF(nbnxn_sci_t, nbx_sci_t, sci); OFF(nbnxn_sci_t, nbx_sci_t, shift);
    OFF(nbnxn_sci_t, nbx_sci_t, cj4_ind_start); OFF(nbnxn_sci_t, nbx_sci_t, cj4_ind_end);
    SZ(nbnxn_cj4_t, nbx_cj4_t);
    OFF(nbnxn_cj4_t, nbx_cj4_t, cj); OFF(nbnxn_cj4_t, nbx_cj4_t, imei);
    SZ(nbnxn_im_ei_t, nbx_im_ei_t);
    OFF(nbnxn_im_ei_t, nbx_im_ei_t, imask); OFF(nbnxn_im_ei_t, nbx_im_ei_t, excl_ind);
    SZ(nbnxn_excl_t, nbx_excl_t);
    SZ(nbnxn_ci_t, nbx_ci_t);
    SZ(nbnxn_cj_t, nbx_cj_t);

    SZ(nbnxn_pairlist_t, nbx_pairlist_t);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, bSimple); OFF(nbnxn_pairlist_t, nbx_pairlist_t, na_ci);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, na_sc);   OFF(nbnxn_pairlist_t, nbx_pairlist_t, rlist);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, nci);     OFF(nbnxn_pairlist_t, nbx_pairlist_t, ci);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, nsci);    OFF(nbnxn_pairlist_t, nbx_pairlist_t, sci);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, ncj);     OFF(nbnxn_pairlist_t, nbx_pairlist_t, cj);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, ncj4);    OFF(nbnxn_pairlist_t, nbx_pairlist_t, cj4);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, nexcl);   OFF(nbnxn_pairlist_t, nbx_pairlist_t, excl);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, nci_tot); OFF(nbnxn_pairlist_t, nbx_pairlist_t, work);
    OFF(nbnxn_pairlist_t, nbx_pairlist_t, cp1);

    SZ(nbnxn_pairlist_set_t, nbx_pairlist_set_t);
    OFF(nbnxn_pairlist_set_t, nbx_pairlist_set_t, nbl); OFF(nbnxn_pairlist_set_t, nbx_pairlist_set_t, natpair_q);
    OFF(nbnxn_pairlist_set_t, nbx_pairlist_set_t, nbl_fep);

    SZ(nbnxn_atomdata_output_t, nbx_atomdata_output_t);
    OFF(nbnxn_atomdata_output_t, nbx_atomdata_output_t, f);  OFF(nbnxn_atomdata_output_t, nbx_atomdata_output_t, fshift);
    OFF(nbnxn_atomdata_output_t, nbx_atomdata_output_t, Vvdw); OFF(nbnxn_atomdata_output_t, nbx_atomdata_output_t, Vc);

    SZ(nbnxn_atomdata_t, nbx_atomdata_t);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, ntype);       OFF(nbnxn_atomdata_t, nbx_atomdata_t, nbfp);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, comb_rule);   OFF(nbnxn_atomdata_t, nbx_atomdata_t, nbfp_comb);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, natoms);      OFF(nbnxn_atomdata_t, nbx_atomdata_t, natoms_local);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, type);        OFF(nbnxn_atomdata_t, nbx_atomdata_t, XFormat);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, FFormat);     OFF(nbnxn_atomdata_t, nbx_atomdata_t, q);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, nenergrp);    OFF(nbnxn_atomdata_t, nbx_atomdata_t, bDynamicBox);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, shift_vec);   OFF(nbnxn_atomdata_t, nbx_atomdata_t, xstride);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, fstride);     OFF(nbnxn_atomdata_t, nbx_atomdata_t, x);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, nout);        OFF(nbnxn_atomdata_t, nbx_atomdata_t, out);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, nalloc);      OFF(nbnxn_atomdata_t, nbx_atomdata_t, buffer_flags);
    OFF(nbnxn_atomdata_t, nbx_atomdata_t, syncStep);

    SZ(interaction_const_t, nbx_interaction_const_t);
    OFF(interaction_const_t, nbx_interaction_const_t, vdwtype);      OFF(interaction_const_t, nbx_interaction_const_t, vdw_modifier);
    OFF(interaction_const_t, nbx_interaction_const_t, rvdw);         OFF(interaction_const_t, nbx_interaction_const_t, dispersion_shift);
    OFF(interaction_const_t, nbx_interaction_const_t, repulsion_shift); OFF(interaction_const_t, nbx_interaction_const_t, vdw_switch);
    OFF(interaction_const_t, nbx_interaction_const_t, sh_invrc6);    OFF(interaction_const_t, nbx_interaction_const_t, eeltype);
    OFF(interaction_const_t, nbx_interaction_const_t, rcoulomb);     OFF(interaction_const_t, nbx_interaction_const_t, rlist);
    OFF(interaction_const_t, nbx_interaction_const_t, ewaldcoeff_q); OFF(interaction_const_t, nbx_interaction_const_t, sh_ewald);
    OFF(interaction_const_t, nbx_interaction_const_t, epsfac);       OFF(interaction_const_t, nbx_interaction_const_t, k_rf);
    OFF(interaction_const_t, nbx_interaction_const_t, c_rf);         OFF(interaction_const_t, nbx_interaction_const_t, tabq_scale);
    OFF(interaction_const_t, nbx_interaction_const_t, tabq_coul_F);  OFF(interaction_const_t, nbx_interaction_const_t, tabq_vdw_FDV0);

    SZ(nonbonded_verlet_group_t, nbx_verlet_group_t);
    OFF(nonbonded_verlet_group_t, nbx_verlet_group_t, nbat); OFF(nonbonded_verlet_group_t, nbx_verlet_group_t, ewald_excl);

    SZ(wallclock_gpu_t, nbx_wallclock_gpu_t);
    OFF(wallclock_gpu_t, nbx_wallclock_gpu_t, nb_h2d_t); OFF(wallclock_gpu_t, nbx_wallclock_gpu_t, pl_h2d_c);

    VAL(NBX_CLUSTER_SIZE, NBNXN_GPU_CLUSTER_SIZE);
    VAL(NBX_NCL_PER_SUPERCL, NBNXN_GPU_NCLUSTER_PER_SUPERCLUSTER);
    VAL(NBX_JGROUP_SIZE, NBNXN_GPU_JGROUP_SIZE);
    VAL(NBX_SHIFTS, SHIFTS);
    VAL(NBX_CENTRAL, CENTRAL);
    VAL(NBX_FORCE_VIRIAL, GMX_FORCE_VIRIAL);
    VAL(NBX_FORCE_ENERGY, GMX_FORCE_ENERGY);
    VAL(NBX_LOCAL, eintLocal); VAL(NBX_NONLOCAL, eintNonlocal);
    VAL(NBX_EEL_CUT, eelCUT); VAL(NBX_EEL_RF, eelRF); VAL(NBX_EEL_GRF, eelGRF);
    VAL(NBX_EEL_PME, eelPME); VAL(NBX_EEL_EWALD, eelEWALD);
    VAL(NBX_VDW_CUT, evdwCUT);
    VAL(NBX_MOD_POTSHIFT, eintmodPOTSHIFT); VAL(NBX_MOD_NONE, eintmodNONE);
    if (NBX_AVOID_SING_R2_INC != (float)NBNXN_AVOID_SING_R2_INC) { printf("VALUE MISMATCH r2 inc\n"); bad++; }

    printf(bad ? "ABI CHECK FAILED: %d mismatches\n" : "ABI check passed (%d mismatches)\n", bad);
    return bad != 0;
}
