/*
 * nbnxn_hip_gmx.h -- the reference-side binding: DEFINES the reference's own GPU entry points in terms of
 * libnbnxn_hip.so (include/nbnxn_hip.h), so that no call site and no header of GROMACS 5.0.4 changes.
 *
 * How a maintainer uses it: one new translation unit in libgromacs, src/gromacs/mdlib/nbnxn_hip/nbnxn_hip_gmx.c,
 * that consists of
 *
 *     #define GMX_GPU                                  (config.h has it already in a GPU build)
 *     #include "typedefs.h"
 *     #include "types/nb_verlet.h"                     nonbonded_verlet_group_t
 *     #include "types/hw_info.h"                       gmx_gpu_info_t, gmx_gpu_opt_t
 *     #include "nbnxn_cuda_data_mgmt.h"                the reference's own prototypes ...
 *     #include "nbnxn_cuda/nbnxn_cuda.h"
 *     #include "gpu_utils.h"
 *     #include "pmalloc_cuda.h"
 *     #include "nbnxn_hip_gmx.h"                       ... and their definitions
 *
 * and takes the place of the CUDA targets nbnxn_cuda, gpu_utils and cuda_tools.  Because
 * the reference's prototypes are in scope, the compiler checks every definition below against them: a wrong argument
 * type is a hard error.  oracle/binding_check.c is that translation unit plus one call of every name with the
 * reference's argument types; oracle/Makefile compiles it against /root/reference with
 * -Wall -Werror=incompatible-pointer-types (tests/test_abi.py runs it).
 *
 * The struct arguments are passed through by pointer cast: the nbx_* mirrors are layout-identical
 * (oracle/abi_check.c proves sizes and offsets against the reference's headers).
 * cuda_dev_info_t (gmxlib/cuda_tools/cudautils.cuh:116-122), opaque to everything outside the CUDA module
 * (types/nbnxn_cuda_types_ext.h:47), is defined here as nbx_gpu_dev_info_t.
 */
#ifndef NBNXN_HIP_GMX_H
#define NBNXN_HIP_GMX_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbnxn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

struct cuda_dev_info { nbx_gpu_dev_info_t d; };

#define NBX_GMX_NB(p) ((nbnxn_hip_ptr_t)(p))

/* ---- legacyheaders/nbnxn_cuda_data_mgmt.h:59-139 ---- */

void nbnxn_cuda_init(FILE *fplog, nbnxn_cuda_ptr_t *p_cu_nb, const gmx_gpu_info_t *gpu_info, const gmx_gpu_opt_t *gpu_opt,
                     int my_gpu_index, gmx_bool bLocalAndNonlocal)
{
    /* the device id nbnxn_cuda_init derives itself (nbnxn_cuda_data_mgmt.cu:566) */
    nbnxn_hip_init(fplog, (nbnxn_hip_ptr_t *)p_cu_nb, get_gpu_device_id(gpu_info, gpu_opt, my_gpu_index), bLocalAndNonlocal);
    /* mdrun clears the outputs before every step (nbnxn_cuda_clear_outputs, sim_util.c:1547): with this switch the local
     * launch of step n zeroes the buffer of step n + 1 and that call only flips buffers -- no fill between two steps */
    if (*p_cu_nb && getenv("GMX_HIP_NB_NO_CLEAR_IN_KERNEL") == NULL) { nbnxn_hip_set_clear_in_kernel((nbnxn_hip_ptr_t)*p_cu_nb, 1); }
}

void nbnxn_cuda_init_const(nbnxn_cuda_ptr_t cu_nb, const interaction_const_t *ic, const nonbonded_verlet_group_t *nbv_group)
{
    nbnxn_hip_init_const(NBX_GMX_NB(cu_nb), (const nbx_interaction_const_t *)ic, (const nbx_verlet_group_t *)nbv_group);
}

void nbnxn_cuda_init_pairlist(nbnxn_cuda_ptr_t cu_nb, const nbnxn_pairlist_t *h_nblist, int iloc)
{
    nbnxn_hip_init_pairlist(NBX_GMX_NB(cu_nb), (const nbx_pairlist_t *)h_nblist, iloc);
}

void nbnxn_cuda_init_atomdata(nbnxn_cuda_ptr_t cu_nb, const nbnxn_atomdata_t *atomdata)
{
    nbnxn_hip_init_atomdata(NBX_GMX_NB(cu_nb), (const nbx_atomdata_t *)atomdata);
}

void nbnxn_cuda_pme_loadbal_update_param(nbnxn_cuda_ptr_t cu_nb, const interaction_const_t *ic)
{
    nbnxn_hip_pme_loadbal_update_param(NBX_GMX_NB(cu_nb), (const nbx_interaction_const_t *)ic);
}

void nbnxn_cuda_upload_shiftvec(nbnxn_cuda_ptr_t cu_nb, const nbnxn_atomdata_t *nbatom)
{
    nbnxn_hip_upload_shiftvec(NBX_GMX_NB(cu_nb), (const nbx_atomdata_t *)nbatom);
}

void nbnxn_cuda_clear_outputs(nbnxn_cuda_ptr_t cu_nb, int flags) { nbnxn_hip_clear_outputs(NBX_GMX_NB(cu_nb), flags); }

void nbnxn_cuda_free(nbnxn_cuda_ptr_t cu_nb) { nbnxn_hip_free(NBX_GMX_NB(cu_nb)); }

wallclock_gpu_t *nbnxn_cuda_get_timings(nbnxn_cuda_ptr_t cu_nb)
{
    /* nbx_wallclock_gpu_t mirrors struct wallclock_gpu (types/nbnxn_cuda_types_ext.h:62-73; oracle/abi_check.c) */
    return (wallclock_gpu_t *)nbnxn_hip_get_timings(NBX_GMX_NB(cu_nb));
}

void nbnxn_cuda_reset_timings(nbnxn_cuda_ptr_t cu_nb) { nbnxn_hip_reset_timings(NBX_GMX_NB(cu_nb)); }

int nbnxn_cuda_min_ci_balanced(nbnxn_cuda_ptr_t cu_nb) { return nbnxn_hip_min_ci_balanced(NBX_GMX_NB(cu_nb)); }

gmx_bool nbnxn_cuda_is_kernel_ewald_analytical(const nbnxn_cuda_ptr_t cu_nb)
{
    return nbnxn_hip_is_kernel_ewald_analytical(NBX_GMX_NB(cu_nb));
}

/* ---- mdlib/nbnxn_cuda/nbnxn_cuda.h:64-86 ---- */

void nbnxn_cuda_launch_kernel(nbnxn_cuda_ptr_t cu_nb, const nbnxn_atomdata_t *nbdata, int flags, int iloc)
{
    nbnxn_hip_launch_kernel(NBX_GMX_NB(cu_nb), (const nbx_atomdata_t *)nbdata, flags, iloc);
}

void nbnxn_cuda_launch_cpyback(nbnxn_cuda_ptr_t cu_nb, const nbnxn_atomdata_t *nbatom, int flags, int aloc)
{
    nbnxn_hip_launch_cpyback(NBX_GMX_NB(cu_nb), (const nbx_atomdata_t *)nbatom, flags, aloc);
}

void nbnxn_cuda_wait_gpu(nbnxn_cuda_ptr_t cu_nb, const nbnxn_atomdata_t *nbatom, int flags, int aloc,
                         real *e_lj, real *e_el, rvec *fshift)
{
    /* real is float in the mixed-precision build the GPU path requires; rvec * is float (*)[3] */
    nbnxn_hip_wait_gpu(NBX_GMX_NB(cu_nb), (const nbx_atomdata_t *)nbatom, flags, aloc, e_lj, e_el, fshift);
}

/* ---- legacyheaders/pmalloc_cuda.h:57-63 ---- */

void pmalloc(void **h_ptr, size_t nbytes) { nbnxn_hip_pmalloc(h_ptr, nbytes); }
void pmalloc_wc(void **h_ptr, size_t nbytes) { nbnxn_hip_pmalloc(h_ptr, nbytes); }   /* no caller in 5.0.4 */
void pfree(void *h_ptr) { nbnxn_hip_pfree(h_ptr); }

/* ---- legacyheaders/gpu_utils.h:63-107 ---- */

int do_quick_memtest(int dev_id) { return nbnxn_hip_do_quick_memtest(dev_id); }
int do_full_memtest(int dev_id) { return nbnxn_hip_do_full_memtest(dev_id); }
int do_timed_memtest(int dev_id, int time_limit) { return nbnxn_hip_do_timed_memtest(dev_id, time_limit); }

int detect_cuda_gpus(gmx_gpu_info_t *gpu_info, char *err_str)
{
    int ndev = 0, ncompat = 0, rc;
    gpu_info->ncuda_dev = 0;
    gpu_info->ncuda_dev_compatible = 0;
    gpu_info->cuda_dev = NULL;
    if (nbnxn_hip_detect_gpus(&ndev, err_str, STRLEN) != 0) { return -1; }
    if (ndev > 0)
    {
        /* freed by free_gpu_info below; gmx_detect_hardware.c:610-623 broadcasts ndev * sizeof_cuda_dev_info() bytes */
        gpu_info->cuda_dev = (struct cuda_dev_info *)calloc((size_t)ndev, sizeof(struct cuda_dev_info));
        if (!gpu_info->cuda_dev) { strncpy(err_str, "out of memory for the device table", STRLEN - 1); return -1; }
    }
    rc = nbnxn_hip_detect_gpus_info(gpu_info->cuda_dev ? &gpu_info->cuda_dev[0].d : NULL, ndev, &ndev, &ncompat, err_str, STRLEN);
    gpu_info->ncuda_dev = ndev;
    gpu_info->ncuda_dev_compatible = ncompat;
    return rc == 0 ? 0 : -1;
}

void pick_compatible_gpus(const gmx_gpu_info_t *gpu_info, gmx_gpu_opt_t *gpu_opt)
{
    /* released by the caller with sfree (= free), as the array the reference makes with snew */
    gpu_opt->cuda_dev_use = (int *)calloc((size_t)(gpu_info->ncuda_dev > 0 ? gpu_info->ncuda_dev : 1), sizeof(int));
    gpu_opt->ncuda_dev_use = 0;
    nbnxn_hip_pick_compatible_gpus(gpu_info->cuda_dev ? &gpu_info->cuda_dev[0].d : NULL, gpu_info->ncuda_dev,
                                   gpu_opt->cuda_dev_use, gpu_info->ncuda_dev, &gpu_opt->ncuda_dev_use);
}

gmx_bool check_selected_cuda_gpus(int *checkres, const gmx_gpu_info_t *gpu_info, gmx_gpu_opt_t *gpu_opt)
{
    return nbnxn_hip_check_selected_gpus(checkres, gpu_info->cuda_dev ? &gpu_info->cuda_dev[0].d : NULL, gpu_info->ncuda_dev,
                                         gpu_opt->cuda_dev_use, gpu_opt->ncuda_dev_use);
}

void free_gpu_info(const gmx_gpu_info_t *gpu_info)
{
    if (gpu_info) { free(gpu_info->cuda_dev); }
}

int get_gpu_device_id(const gmx_gpu_info_t *gpu_info, const gmx_gpu_opt_t *gpu_opt, int idx)
{
    return nbnxn_hip_get_gpu_device_id(gpu_info->cuda_dev ? &gpu_info->cuda_dev[0].d : NULL, gpu_info->ncuda_dev,
                                       gpu_opt->cuda_dev_use, gpu_opt->ncuda_dev_use, idx);
}

gmx_bool init_gpu(int mygpu, char *result_str, const gmx_gpu_info_t *gpu_info, const gmx_gpu_opt_t *gpu_opt)
{
    return nbnxn_hip_init_gpu(get_gpu_device_id(gpu_info, gpu_opt, mygpu), result_str, STRLEN) == 0;
}

gmx_bool free_gpu(char *result_str) { return nbnxn_hip_free_gpu(result_str, STRLEN) == 0; }

int get_current_gpu_device_id(void) { return nbnxn_hip_get_current_gpu_device_id(); }

void get_gpu_device_info_string(char *s, const gmx_gpu_info_t *gpu_info, int index)
{
    /* index counts detected devices (gmx_detect_hardware.c:100-103), whose id is their index */
    if (index < 0 || index >= gpu_info->ncuda_dev) { s[0] = 0; return; }
    nbnxn_hip_get_gpu_device_info_string(s, STRLEN, gpu_info->cuda_dev[index].d.id);
}

size_t sizeof_cuda_dev_info(void) { return nbnxn_hip_sizeof_dev_info(); }

#undef NBX_GMX_NB

#ifdef __cplusplus
}
#endif

#endif /* NBNXN_HIP_GMX_H */
