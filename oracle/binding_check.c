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

int main(void)
{
    /* the one thing that is safe anywhere: count devices through the reference's own detection call */
    gmx_gpu_info_t info;
    char           err[STRLEN] = "";
    int            rc;
    nbnxn_hip_set_abort_on_error(0);
    rc = detect_cuda_gpus(&info, err);
    printf("binding ok: detect_cuda_gpus -> %d, %d device(s), %d compatible%s%s\n", rc, info.ncuda_dev, info.ncuda_dev_compatible,
           err[0] ? ": " : "", err);
    free_gpu_info(&info);
    return 0;
}
