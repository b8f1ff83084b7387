/*
 * binding_check.c -- compiles the reference-side binding (include/nbnxn_hip_gmx.h) the way a maintainer would:
 * the reference's own headers give the prototypes of every GPU entry point mdrun calls, the binding header gives
 * their definitions, and use_every_name() calls each one once with the reference's own argument types.
 * TEST INFRASTRUCTURE: built by oracle/Makefile only where /root/reference exists, with
 * -Wall -Wextra -Werror=incompatible-pointer-types -Werror=implicit-function-declaration -Werror=int-conversion;
 * the object is linked against libnbnxn_hip.so into oracle/_ref/binding_check so that every nbnxn_hip_* name the
 * binding uses is proven to be exported (the program only prints the device count when run without a GPU).
 */
#define GMX_GPU
#include <stdio.h>

#include "typedefs.h"
#include "types/nb_verlet.h"
#include "types/hw_info.h"
#include "types/interaction_const.h"
#include "types/nbnxn_pairlist.h"
#include "types/force_flags.h"
#include "nbnxn_cuda_data_mgmt.h"
#include "nbnxn_cuda/nbnxn_cuda.h"
#include "gpu_utils.h"
#include "pmalloc_cuda.h"
#include "types/enums.h"
#include "types/ishift.h"

#include "nbnxn_hip_gmx.h"

/* every mapped name, called as forcerec.c / sim_util.c / runner.c / md.c / pme_loadbal.c / gmx_detect_hardware.c
 * call it (call sites in INTEGRATION.md section 1); never executed */
void use_every_name(FILE *fplog, gmx_hw_info_t *hwinfo, gmx_hw_opt_t *hw_opt, nonbonded_verlet_t *nbv,
                    const interaction_const_t *ic, real *e_lj, real *e_el, rvec *fshift)
{
    nbnxn_cuda_ptr_t         *p_nb = &nbv->cu_nbv;
    nbnxn_cuda_ptr_t          nb;
    nonbonded_verlet_group_t *grp  = &nbv->grp[eintLocal];
    const nbnxn_atomdata_t   *nbat = grp->nbat;
    char                      str[STRLEN];
    int                       checkres[8], id;
    void                     *h = NULL;
    wallclock_gpu_t          *t;
    gmx_bool                  b;

    (void)detect_cuda_gpus(&hwinfo->gpu_info, str);
    pick_compatible_gpus(&hwinfo->gpu_info, &hw_opt->gpu_opt);
    b  = check_selected_cuda_gpus(checkres, &hwinfo->gpu_info, &hw_opt->gpu_opt);
    id = get_gpu_device_id(&hwinfo->gpu_info, &hw_opt->gpu_opt, 0);
    get_gpu_device_info_string(str, &hwinfo->gpu_info, 0);
    (void)sizeof_cuda_dev_info();
    b  = init_gpu(0, str, &hwinfo->gpu_info, &hw_opt->gpu_opt);
    id = get_current_gpu_device_id();
    (void)do_quick_memtest(id); (void)do_full_memtest(id); (void)do_timed_memtest(id, 1);
    pmalloc(&h, 64); pfree(h); pmalloc_wc(&h, 64); pfree(h);

    nbnxn_cuda_init(fplog, p_nb, &hwinfo->gpu_info, &hw_opt->gpu_opt, 0, TRUE);
    nb = *p_nb;
    nbnxn_cuda_init_const(nb, ic, nbv->grp);
    nbnxn_cuda_init_atomdata(nb, nbat);
    nbnxn_cuda_upload_shiftvec(nb, nbat);
    nbnxn_cuda_init_pairlist(nb, grp->nbl_lists.nbl[0], eintLocal);
    nbnxn_cuda_pme_loadbal_update_param(nb, ic);
    nbnxn_cuda_clear_outputs(nb, GMX_FORCE_VIRIAL);
    nbnxn_cuda_launch_kernel(nb, nbat, GMX_FORCE_VIRIAL, eintLocal);
    nbnxn_cuda_launch_cpyback(nb, nbat, GMX_FORCE_VIRIAL, eatLocal);
    nbnxn_cuda_wait_gpu(nb, nbat, GMX_FORCE_VIRIAL, eatLocal, e_lj, e_el, fshift);
    t = nbnxn_cuda_get_timings(nb);
    nbnxn_cuda_reset_timings(nb);
    (void)nbnxn_cuda_min_ci_balanced(nb);
    b = nbnxn_cuda_is_kernel_ewald_analytical(nb);
    nbnxn_cuda_free(nb);
    b = free_gpu(str);
    free_gpu_info(&hwinfo->gpu_info);
    (void)b; (void)t; (void)id;
}

/* ---- run mode: one real step through the reference's own names and struct types ------------------------------------
 * binding_check <in> <out>: reads a small system (written by tests/util.py write_step_file), builds the REFERENCE's
 * objects around it -- gmx_gpu_info_t / gmx_gpu_opt_t, nonbonded_verlet_t, nbnxn_atomdata_t with x and f from pmalloc
 * as forcerec.c:2247-2262 arranges, nbnxn_pairlist_t, interaction_const_t -- and makes the calls mdrun makes, in
 * mdrun's order (gmx_detect_hardware.c:585-772, forcerec.c:1825, 2108, 2201; sim_util.c:1007-1034, 622, 1147, 1385,
 * 1486; runner.c:1058-1077), every one of them by its reference name, resolved by include/nbnxn_hip_gmx.h.
 * Step 1 with energies and shift forces, step 2 forces only; results go to <out> for the GPU test to compare with
 * the oracle (tests/test_gpu_parity.py::test_reference_named_calls_run_a_step). */
#include <string.h>
#include <stdlib.h>
#include <stdint.h>

typedef struct {
    int32_t magic, natoms, ntype, nsci, ncj4, nexcl, nci_tot, eeltype, vdwtype, vdw_modifier, pad[6];
    float   rcoulomb, rvdw, rlist, epsfac, k_rf, c_rf, ewaldcoeff_q, sh_ewald, disp_cpot, rep_cpot, epsilon_r, epsilon_rf, fpad[4];
} step_header_t;

static int read_all(FILE *fp, void *p, size_t n) { return n == 0 || fread(p, 1, n, fp) == n; }

static int run_step(const char *in, const char *out)
{
    step_header_t       h;
    gmx_hw_info_t       hwinfo;
    gmx_hw_opt_t        hw_opt;
    nonbonded_verlet_t  nbv;
    nbnxn_atomdata_t    nbat;
    nbnxn_atomdata_output_t nbat_out;
    nbnxn_pairlist_t    nbl;
    nbnxn_pairlist_t   *nblp[1];
    interaction_const_t ic;
    char                str[STRLEN] = "";
    int                 checkres[8];
    rvec                fshift[SHIFTS];
    real                e_lj = 0, e_el = 0;
    float              *f2;
    FILE               *fp = fopen(in, "rb");
    int32_t             info[8];
    void               *p;
    wallclock_gpu_t    *t;

    if (!fp || !read_all(fp, &h, sizeof(h)) || h.magic != 0x4e42584e) { fprintf(stderr, "bad input file %s\n", in); return 2; }
    memset(&hwinfo, 0, sizeof(hwinfo)); memset(&hw_opt, 0, sizeof(hw_opt)); memset(&nbv, 0, sizeof(nbv));
    memset(&nbat, 0, sizeof(nbat)); memset(&nbat_out, 0, sizeof(nbat_out)); memset(&nbl, 0, sizeof(nbl)); memset(&ic, 0, sizeof(ic));

    /* gmx_detect_hardware.c: detection, selection, check; forcerec.c:1825: init_gpu */
    if (detect_cuda_gpus(&hwinfo.gpu_info, str) != 0 || hwinfo.gpu_info.ncuda_dev_compatible < 1) { fprintf(stderr, "no compatible GPU: %s\n", str); return 3; }
    pick_compatible_gpus(&hwinfo.gpu_info, &hw_opt.gpu_opt);
    if (hw_opt.gpu_opt.ncuda_dev_use < 1 || !check_selected_cuda_gpus(checkres, &hwinfo.gpu_info, &hw_opt.gpu_opt)) { fprintf(stderr, "GPU check failed\n"); return 3; }
    if (!init_gpu(0, str, &hwinfo.gpu_info, &hw_opt.gpu_opt)) { fprintf(stderr, "init_gpu: %s\n", str); return 3; }
    get_gpu_device_info_string(str, &hwinfo.gpu_info, 0);
    printf("device %d: %s\n", get_gpu_device_id(&hwinfo.gpu_info, &hw_opt.gpu_opt, 0), str);

    /* forcerec.c:2201 */
    nbv.bUseGPU = TRUE; nbv.ngrp = 1;
    nbnxn_cuda_init(stdout, &nbv.cu_nbv, &hwinfo.gpu_info, &hw_opt.gpu_opt, 0, FALSE);
    nbv.min_ci_balanced = nbnxn_cuda_min_ci_balanced(nbv.cu_nbv);

    /* the atom data as nbnxn_atomdata_init / nbnxn_atomdata_set leave it for the GPU (nbatXYZQ / nbatXYZ), x and f pinned */
    nbat.alloc = pmalloc; nbat.free = pfree;
    nbat.ntype = h.ntype; nbat.natoms = h.natoms; nbat.natoms_local = h.natoms;
    nbat.comb_rule = ljcrNONE; nbat.XFormat = nbatXYZQ; nbat.FFormat = nbatXYZ; nbat.na_c = 8; nbat.nenergrp = 1;
    nbat.xstride = 4; nbat.fstride = 3; nbat.bDynamicBox = FALSE;
    nbat.nbfp = malloc((size_t)h.ntype * h.ntype * 2 * sizeof(real));
    nbat.type = malloc((size_t)h.natoms * sizeof(int));
    nbat.shift_vec = malloc(SHIFTS * sizeof(rvec));
    nbat.alloc(&p, (size_t)h.natoms * 4 * sizeof(real)); nbat.x = p;
    nbat.alloc(&p, (size_t)h.natoms * 3 * sizeof(real)); nbat_out.f = p;
    nbat.nout = 1; nbat.out = &nbat_out;
    f2 = malloc((size_t)h.natoms * 3 * sizeof(float));
    nbl.bSimple = FALSE; nbl.na_ci = 8; nbl.na_cj = 8; nbl.na_sc = 64; nbl.rlist = h.rlist;
    nbl.nsci = h.nsci; nbl.ncj4 = h.ncj4; nbl.nexcl = h.nexcl; nbl.nci_tot = h.nci_tot;
    nbl.sci = malloc((size_t)(h.nsci + 1) * sizeof(nbnxn_sci_t));
    nbl.cj4 = malloc((size_t)(h.ncj4 + 1) * sizeof(nbnxn_cj4_t));
    nbl.excl = malloc((size_t)(h.nexcl + 1) * sizeof(nbnxn_excl_t));
    if (!nbat.x || !nbat_out.f || !read_all(fp, nbat.x, (size_t)h.natoms * 16) || !read_all(fp, nbat.type, (size_t)h.natoms * 4)
        || !read_all(fp, nbat.nbfp, (size_t)h.ntype * h.ntype * 8) || !read_all(fp, nbat.shift_vec, SHIFTS * 12)
        || !read_all(fp, nbl.sci, (size_t)h.nsci * sizeof(nbnxn_sci_t)) || !read_all(fp, nbl.cj4, (size_t)h.ncj4 * sizeof(nbnxn_cj4_t))
        || !read_all(fp, nbl.excl, (size_t)h.nexcl * sizeof(nbnxn_excl_t))) { fprintf(stderr, "short input file\n"); return 2; }
    fclose(fp);
    nblp[0] = &nbl;
    nbv.grp[0].nbat = &nbat; nbv.grp[0].kernel_type = nbnxnk8x8x8_CUDA;
    nbv.grp[0].nbl_lists.nnbl = 1; nbv.grp[0].nbl_lists.nbl = nblp; nbv.grp[0].nbl_lists.bSimple = FALSE;

    ic.vdwtype = h.vdwtype; ic.vdw_modifier = h.vdw_modifier; ic.rvdw = h.rvdw;
    ic.dispersion_shift.cpot = h.disp_cpot; ic.repulsion_shift.cpot = h.rep_cpot; ic.sh_invrc6 = -h.disp_cpot;
    ic.eeltype = h.eeltype; ic.coulomb_modifier = eintmodPOTSHIFT; ic.rcoulomb = h.rcoulomb; ic.rlist = h.rlist; ic.rlistlong = h.rlist;
    ic.ewaldcoeff_q = h.ewaldcoeff_q; ic.sh_ewald = h.sh_ewald; ic.epsilon_r = h.epsilon_r; ic.epsfac = h.epsfac;
    ic.epsilon_rf = h.epsilon_rf; ic.k_rf = h.k_rf; ic.c_rf = h.c_rf;

    /* forcerec.c:2108; sim_util.c:1007, 1012, 1034 (a search step) */
    nbnxn_cuda_init_const(nbv.cu_nbv, &ic, nbv.grp);
    nbnxn_cuda_init_atomdata(nbv.cu_nbv, nbv.grp[0].nbat);
    nbnxn_cuda_upload_shiftvec(nbv.cu_nbv, nbv.grp[0].nbat);
    nbnxn_cuda_init_pairlist(nbv.cu_nbv, nbv.grp[0].nbl_lists.nbl[0], eintLocal);

    /* step 1, energies and virial: sim_util.c:622 (do_nb_verlet), 1147, 1385, 1486 */
    memset(fshift, 0, sizeof(fshift));
    nbnxn_cuda_launch_kernel(nbv.cu_nbv, nbv.grp[0].nbat, GMX_FORCE_VIRIAL | GMX_FORCE_ENERGY, eintLocal);
    nbnxn_cuda_launch_cpyback(nbv.cu_nbv, nbv.grp[0].nbat, GMX_FORCE_VIRIAL | GMX_FORCE_ENERGY, eatLocal);
    nbnxn_cuda_wait_gpu(nbv.cu_nbv, nbv.grp[0].nbat, GMX_FORCE_VIRIAL | GMX_FORCE_ENERGY, eatLocal, &e_lj, &e_el, fshift);
    nbnxn_cuda_clear_outputs(nbv.cu_nbv, GMX_FORCE_VIRIAL | GMX_FORCE_ENERGY);
    memcpy(f2, nbat_out.f, (size_t)h.natoms * 3 * sizeof(float));
    /* step 2, forces only */
    nbnxn_cuda_launch_kernel(nbv.cu_nbv, nbv.grp[0].nbat, GMX_FORCE_FORCES, eintLocal);
    nbnxn_cuda_launch_cpyback(nbv.cu_nbv, nbv.grp[0].nbat, GMX_FORCE_FORCES, eatLocal);
    nbnxn_cuda_wait_gpu(nbv.cu_nbv, nbv.grp[0].nbat, GMX_FORCE_FORCES, eatLocal, &e_lj, &e_el, fshift);
    nbnxn_cuda_clear_outputs(nbv.cu_nbv, GMX_FORCE_FORCES);

    t = nbnxn_cuda_get_timings(nbv.cu_nbv);
    info[0] = nbv.min_ci_balanced;
    info[1] = nbnxn_cuda_is_kernel_ewald_analytical(nbv.cu_nbv);
    info[2] = t ? t->nb_c : -1;
    info[3] = t ? t->ktime[1][1].c : -1;     /* the first launch on a new list prunes and had energies */
    info[4] = t ? t->ktime[0][0].c : -1;     /* the second: neither */
    info[5] = get_current_gpu_device_id();
    info[6] = hwinfo.gpu_info.ncuda_dev; info[7] = (int)sizeof_cuda_dev_info();
    nbnxn_cuda_reset_timings(nbv.cu_nbv);

    fp = fopen(out, "wb");
    if (!fp) { return 2; }
    fwrite(info, sizeof(info), 1, fp);
    fwrite(&e_lj, sizeof(real), 1, fp); fwrite(&e_el, sizeof(real), 1, fp);
    fwrite(fshift, sizeof(fshift), 1, fp);
    fwrite(f2, sizeof(float), (size_t)h.natoms * 3, fp);            /* forces of step 1 */
    fwrite(nbat_out.f, sizeof(float), (size_t)h.natoms * 3, fp);    /* forces of step 2 */
    fclose(fp);

    /* runner.c:1058-1077 */
    nbnxn_cuda_free(nbv.cu_nbv);
    nbat.free(nbat.x); nbat.free(nbat_out.f);
    if (!free_gpu(str)) { fprintf(stderr, "free_gpu: %s\n", str); return 3; }
    free_gpu_info(&hwinfo.gpu_info);
    free(hw_opt.gpu_opt.cuda_dev_use);
    free(nbat.nbfp); free(nbat.type); free(nbat.shift_vec); free(nbl.sci); free(nbl.cj4); free(nbl.excl); free(f2);
    printf("step done: e_lj %.6e e_el %.6e\n", e_lj, e_el);
    return 0;
}

int main(int argc, char **argv)
{
    /* without arguments, the one thing that is safe anywhere: count devices through the reference's own detection call */
    gmx_gpu_info_t info;
    char           err[STRLEN] = "";
    int            rc;
    if (argc == 3) { return run_step(argv[1], argv[2]); }
    nbnxn_hip_set_abort_on_error(0);
    rc = detect_cuda_gpus(&info, err);
    printf("binding ok: detect_cuda_gpus -> %d, %d device(s), %d compatible%s%s\n", rc, info.ncuda_dev, info.ncuda_dev_compatible,
           err[0] ? ": " : "", err);
    free_gpu_info(&info);
    return 0;
}
