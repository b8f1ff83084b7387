/*
 * nbnxn_hip.h -- C ABI of the MI355X (gfx950) Verlet cluster-pair nonbonded module.
 *
 * This is the drop-in boundary: a plain C interface (pointers, ints, floats) that the
 * GROMACS 5.0.4 host code binds instead of its nbnxn_cuda_* module.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference tree,
 * src/gromacs/...).  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions kept from the reference: real == float, gmx_bool == int, rvec == float[3],
 * flags are GMX_FORCE_* bits (legacyheaders/types/force_flags.h:50-72), iloc/aloc are
 * eintLocal/eintNonlocal and eatLocal/eatNonlocal (legacyheaders/types/nb_verlet.h:68-82).
 *
 * Error convention: the reference returns void and aborts through gmx_fatal
 * (gmxlib/cuda_tools/cudautils.cuh:65-90).  Here every call returns 0 on success and a
 * non-zero code on failure; by default a failure also prints the message and abort()s
 * (gmx_fatal behaviour).  nbnxn_hip_set_abort_on_error(0) turns the abort off so that a
 * host language runtime (the Python mirror, tests) can raise instead.
 */
#ifndef NBNXN_HIP_H
#define NBNXN_HIP_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * Constants of the list format (mdlib/nbnxn_consts.h:50-83, legacyheaders/types/ishift.h)
 * ---------------------------------------------------------------------------------- */
#define NBX_CLUSTER_SIZE         8   /* atoms per i/j cluster (NBNXN_GPU_CLUSTER_SIZE)   */
#define NBX_NCL_PER_SUPERCL      8   /* clusters per super-cluster (2x2x2)               */
#define NBX_JGROUP_SIZE          4   /* j-clusters per cj4 entry                         */
#define NBX_SHIFTS              45   /* SHIFTS                                           */
#define NBX_MAX_ZONES           14   /* grid zones of one domain: home + 13 halo zones (half shell in 3-D); the reference's eighth shell has 8 (domdec.c:433-449) */
#define NBX_DD_MAX_MESSAGES     13   /* halo messages of one domain per step and direction (one per halo zone)                        */
#define NBX_CENTRAL             22   /* CENTRAL                                          */
#define NBX_AVOID_SING_R2_INC   1.0e-12f

/* GMX_FORCE_* bits that this module looks at (types/force_flags.h:66-68) */
#define NBX_FORCE_VIRIAL        (1<<8)
#define NBX_FORCE_ENERGY        (1<<9)

/* interaction / atom locality (types/nb_verlet.h:68-82) */
enum { NBX_LOCAL = 0, NBX_NONLOCAL = 1 };

/* eeltype / vdwtype / modifier values this module accepts (types/enums.h:92-122) */
enum { NBX_EEL_CUT = 0, NBX_EEL_RF = 1, NBX_EEL_GRF = 2, NBX_EEL_PME = 3, NBX_EEL_EWALD = 4 };
enum { NBX_VDW_CUT = 0, NBX_VDW_PME = 5 };
enum { NBX_LJCR_GEOM = 0, NBX_LJCR_LB = 1, NBX_LJCR_NONE = 2 };   /* ljcr*, types/nbnxn_pairlist.h; = eljpme* */
enum { NBX_MOD_POTSHIFT_VERLET = 0, NBX_MOD_POTSHIFT = 1, NBX_MOD_NONE = 2, NBX_MOD_POTSWITCH = 3,
       NBX_MOD_EXACTCUTOFF = 4, NBX_MOD_FORCESWITCH = 5 };   /* enum eintmod, types/enums.h:92-93 */

/* electrostatics flavour of the device kernel (nbnxn_cuda/nbnxn_cuda_types.h:80-82) */
enum { NBX_EL_RF = 0, NBX_EL_EWALD_ANA = 1, NBX_EL_EWALD_TAB = 2, NBX_EL_NR };

/* ------------------------------------------------------------------------------------
 * ABI structs.  Each is layout-identical to the named reference struct so that the
 * reference host code can pass its own objects; oracle/abi_check.c static-asserts
 * size and field offsets against the reference headers when /root/reference exists.
 * ---------------------------------------------------------------------------------- */

/* nbnxn_sci_t, types/nbnxn_pairlist.h:102-107 */
typedef struct {
    int sci;
    int shift;
    int cj4_ind_start;
    int cj4_ind_end;
} nbx_sci_t;

/* nbnxn_im_ei_t / nbnxn_cj4_t, types/nbnxn_pairlist.h:109-117.  imei[w] belongs to lanes
 * 32w..32w+31 of the wave (the reference's "warp" w). */
typedef struct {
    unsigned int imask;
    int          excl_ind;
} nbx_im_ei_t;

typedef struct {
    int         cj[4];
    nbx_im_ei_t imei[2];
} nbx_cj4_t;

/* nbnxn_excl_t, types/nbnxn_pairlist.h:119-123 */
typedef struct {
    unsigned int pair[32];
} nbx_excl_t;

/* nbnxn_ci_t / nbnxn_cj_t (simple 4x4 list, only carried, never read here) :75-99 */
typedef struct { int cj; unsigned int excl; } nbx_cj_t;
typedef struct { int ci; int shift; int cj_ind_start; int cj_ind_end; } nbx_ci_t;

typedef struct { int dummy[16]; } nbx_cache_protect_t;   /* gmx_cache_protect_t :50-52 */

typedef void nbx_alloc_t(void **ptr, size_t nbytes);     /* nbnxn_alloc_t :61 */
typedef void nbx_free_t(void *ptr);                      /* nbnxn_free_t  :66 */

/* nbnxn_pairlist_t, types/nbnxn_pairlist.h:125-160 */
typedef struct {
    nbx_cache_protect_t cp0;
    nbx_alloc_t        *alloc;
    nbx_free_t         *free;
    int                 bSimple;
    int                 na_ci;
    int                 na_cj;
    int                 na_sc;
    float               rlist;
    int                 nci;
    nbx_ci_t           *ci;
    int                 ci_nalloc;
    int                 nsci;
    nbx_sci_t          *sci;
    int                 sci_nalloc;
    int                 ncj;
    nbx_cj_t           *cj;
    int                 cj_nalloc;
    int                 ncj4;
    nbx_cj4_t          *cj4;
    int                 cj4_nalloc;
    int                 nexcl;
    nbx_excl_t         *excl;
    int                 excl_nalloc;
    int                 nci_tot;
    void               *work;
    nbx_cache_protect_t cp1;
} nbx_pairlist_t;

/* nbnxn_pairlist_set_t, types/nbnxn_pairlist.h:162-172 */
typedef struct {
    int              nnbl;
    nbx_pairlist_t **nbl;
    int              bCombined;
    int              bSimple;
    int              natpair_ljq;
    int              natpair_lj;
    int              natpair_q;
    void           **nbl_fep;
} nbx_pairlist_set_t;

/* nbnxn_atomdata_output_t, types/nbnxn_pairlist.h:178-187 */
typedef struct {
    float *f;
    float *fshift;
    int    nV;
    float *Vvdw;
    float *Vc;
    int    nVS;
    float *VSvdw;
    float *VSc;
} nbx_atomdata_output_t;

/* nbnxn_buffer_flags_t :212-216 */
typedef struct {
    int           nflag;
    unsigned int *flag;
    int           flag_nalloc;
} nbx_buffer_flags_t;

/* nbnxn_atomdata_t, types/nbnxn_pairlist.h:226-271.  For the GPU path the reference sets
 * XFormat = nbatXYZQ (x is float4 xyzq, xstride 4), FFormat = nbatXYZ (fstride 3), nout 1
 * (mdlib/nbnxn_atomdata.c:734-749). */
typedef struct {
    nbx_alloc_t           *alloc;
    nbx_free_t            *free;
    int                    ntype;
    float                 *nbfp;
    int                    comb_rule;
    float                 *nbfp_comb;
    float                 *nbfp_s4;
    int                    natoms;
    int                    natoms_local;
    int                   *type;
    float                 *lj_comb;
    int                    XFormat;
    int                    FFormat;
    float                 *q;
    int                    na_c;
    int                    nenergrp;
    int                    neg_2log;
    int                   *energrp;
    int                    bDynamicBox;
    float                (*shift_vec)[3];
    int                    xstride;
    int                    fstride;
    float                 *x;
    float                 *simd_4xn_diagonal_j_minus_i;
    float                 *simd_2xnn_diagonal_j_minus_i;
    unsigned int          *simd_exclusion_filter1;
    unsigned int          *simd_exclusion_filter2;
    float                 *simd_interaction_array;
    int                    nout;
    nbx_atomdata_output_t *out;
    int                    nalloc;
    int                    bUseBufferFlags;
    nbx_buffer_flags_t     buffer_flags;
    int                    bUseTreeReduce;
    void                  *syncStep;
} nbx_atomdata_t;

/* shift_consts_t / switch_consts_t / interaction_const_t, types/interaction_const.h:49-130 */
typedef struct { float c2, c3, cpot; } nbx_shift_consts_t;
typedef struct { float c3, c4, c5; } nbx_switch_consts_t;

typedef struct {
    int                 vdwtype;
    int                 vdw_modifier;
    float               rvdw;
    float               rvdw_switch;
    nbx_shift_consts_t  dispersion_shift;
    nbx_shift_consts_t  repulsion_shift;
    nbx_switch_consts_t vdw_switch;
    float               sh_invrc6;
    int                 eeltype;
    int                 coulomb_modifier;
    float               rcoulomb;
    float               rlist;
    float               rlistlong;
    float               ewaldcoeff_q;
    float               ewaldcoeff_lj;
    int                 ljpme_comb_rule;
    float               sh_ewald;
    float               sh_lj_ewald;
    float               epsilon_r;
    float               epsfac;
    float               epsilon_rf;
    float               k_rf;
    float               c_rf;
    float               tabq_scale;
    int                 tabq_size;
    float              *tabq_coul_F;
    float              *tabq_coul_V;
    float              *tabq_coul_FDV0;
    float              *tabq_vdw_F;
    float              *tabq_vdw_V;
    float              *tabq_vdw_FDV0;
} nbx_interaction_const_t;

/* nonbonded_verlet_group_t, types/nb_verlet.h:91-96 */
typedef struct {
    nbx_pairlist_set_t nbl_lists;
    nbx_atomdata_t    *nbat;
    int                kernel_type;
    int                ewald_excl;
} nbx_verlet_group_t;

/* nbnxn_cuda_ktime / wallclock_gpu, types/nbnxn_cuda_types_ext.h:55-71 */
typedef struct { double t; int c; } nbx_ktime_t;
typedef struct {
    nbx_ktime_t ktime[2][2];   /* [prune][energy], milliseconds and call count */
    double      nb_h2d_t;
    double      nb_d2h_t;
    int         nb_c;
    double      pl_h2d_t;
    int         pl_h2d_c;
} nbx_wallclock_gpu_t;

/* opaque module handle; replaces nbnxn_cuda_ptr_t (types/nbnxn_cuda_types_ext.h:45) */
typedef struct nbnxn_hip *nbnxn_hip_ptr_t;

/* ------------------------------------------------------------------------------------
 * Error handling
 * ---------------------------------------------------------------------------------- */
void        nbnxn_hip_set_abort_on_error(int abort_on_error);
const char *nbnxn_hip_last_error(void);

/* ------------------------------------------------------------------------------------
 * Device selection; replaces detect_cuda_gpus / init_gpu / free_gpu / get_current_gpu_device_id /
 * get_gpu_device_info_string (legacyheaders/gpu_utils.h:72-104).
 * ---------------------------------------------------------------------------------- */
int nbnxn_hip_detect_gpus(int *ngpu, char *err_str, int err_len);
int nbnxn_hip_init_gpu(int device_id, char *result_str, int result_len);
int nbnxn_hip_free_gpu(char *result_str, int result_len);
int nbnxn_hip_get_current_gpu_device_id(void);
int nbnxn_hip_get_gpu_device_info_string(char *s, int len, int device_id);

/* The rest of gpu_utils.h behind plain types.  nbx_gpu_dev_info_t stands for cuda_dev_info_t
 * (gmxlib/cuda_tools/cudautils.cuh:116-122; it embeds cudaDeviceProp there), an array of them for
 * gmx_gpu_info_t.cuda_dev, a list of indices into it for gmx_gpu_opt_t.cuda_dev_use; both arrays
 * belong to the caller, so free_gpu_info (gpu_utils.h:84) has nothing to free here. */
enum { NBX_GPU_COMPATIBLE = 0, NBX_GPU_NONEXISTENT = 1, NBX_GPU_INCOMPATIBLE = 2, NBX_GPU_INSANE = 3 };   /* egpu*, types/hw_info.h:57 */
typedef struct {
    int    id;              /* device id */
    int    stat;            /* NBX_GPU_*: result of the sanity check (a dummy kernel) and of the architecture check */
    char   name[256];
    char   arch[64];        /* gcnArchName; compatible = gfx950, the only code objects this library carries */
    int    compute_units;
    int    clock_mhz;
    size_t total_mem;
} nbx_gpu_dev_info_t;
/* sizeof_cuda_dev_info (gpu_utils.h:107): for broadcasting the array between ranks */
size_t nbnxn_hip_sizeof_dev_info(void);
/* detect_cuda_gpus (gpu_utils.h:72) with the per-device checks of is_gmx_supported_gpu_id: fills
 * up to info_alloc entries, counts all devices and the compatible ones; non-zero on an API failure */
int nbnxn_hip_detect_gpus_info(nbx_gpu_dev_info_t *info, int info_alloc, int *ndev, int *ndev_compatible, char *err_str, int err_len);
/* pick_compatible_gpus (gpu_utils.h:75): indices of the compatible devices, in increasing order */
int nbnxn_hip_pick_compatible_gpus(const nbx_gpu_dev_info_t *info, int ndev, int *dev_use, int dev_use_alloc, int *ndev_use);
/* check_selected_cuda_gpus (gpu_utils.h:79): checkres[i] = NBX_GPU_* of dev_use[i]; returns 1 iff all are compatible */
int nbnxn_hip_check_selected_gpus(int *checkres, const nbx_gpu_dev_info_t *info, int ndev, const int *dev_use, int ndev_use);
/* get_gpu_device_id (gpu_utils.h:99): device id of the index-th selected device, -1 if out of range */
int nbnxn_hip_get_gpu_device_id(const nbx_gpu_dev_info_t *info, int ndev, const int *dev_use, int ndev_use, int index);
/* do_quick_memtest / do_full_memtest / do_timed_memtest (gpu_utils.h:63-69): device memory tests
 * (moving inversions, modulo-20, address-dependent pseudo-random words; the reference links the
 * memtestG80 library for the same families).  dev_id -1 = the device in use.  Quick: 250 MiB, one
 * pass; full: 90 % of the free memory, two passes; timed: the same, repeated for time_limit
 * seconds.  Return 0 = no error found, > 0 = mismatching words, -1 = the test could not run. */
int nbnxn_hip_do_quick_memtest(int dev_id);
int nbnxn_hip_do_full_memtest(int dev_id);
int nbnxn_hip_do_timed_memtest(int dev_id, int time_limit);

/* pinned host memory; replaces pmalloc / pfree (legacyheaders/pmalloc_cuda.h:57,63),
 * the nbnxn_alloc_t / nbnxn_free_t hooks handed to nbnxn_init_pairlist_set and
 * nbnxn_atomdata_init (mdlib/forcerec.c:2247-2262). */
void nbnxn_hip_pmalloc(void **h_ptr, size_t nbytes);
void nbnxn_hip_pfree(void *h_ptr);

/* ------------------------------------------------------------------------------------
 * Data management; one-to-one with legacyheaders/nbnxn_cuda_data_mgmt.h:59-139
 * ---------------------------------------------------------------------------------- */

/* nbnxn_cuda_init (nbnxn_cuda_data_mgmt.h:59): creates the handle on device device_id with
 * one stream (local) or two (local + high-priority non-local).  The reference's
 * gpu_info/gpu_opt/my_gpu_index triple only selects the device id, passed here directly. */
int nbnxn_hip_init(FILE *fplog, nbnxn_hip_ptr_t *p_nb, int device_id, int bLocalAndNonlocal);

/* nbnxn_cuda_init_const (:69): cut-offs, RF/Ewald constants, LJ table; picks the kernel
 * flavour; clears the energy and shift-force outputs. */
int nbnxn_hip_init_const(nbnxn_hip_ptr_t nb, const nbx_interaction_const_t *ic,
                         const nbx_verlet_group_t *nbv_group);

/* nbnxn_cuda_init_pairlist (:75): uploads sci/cj4/excl of a GPU-format list and arms the
 * pruning kernel flavour for the next launch of that locality. */
int nbnxn_hip_init_pairlist(nbnxn_hip_ptr_t nb, const nbx_pairlist_t *h_nblist, int iloc);

/* nbnxn_cuda_init_atomdata (:81): (re)allocates xq/f/type for natoms, uploads types. */
int nbnxn_hip_init_atomdata(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat);

/* nbnxn_cuda_pme_loadbal_update_param (:86) */
int nbnxn_hip_pme_loadbal_update_param(nbnxn_hip_ptr_t nb, const nbx_interaction_const_t *ic);

/* nbnxn_cuda_upload_shiftvec (:91): uploads the 45 shift vectors if the box is dynamic or
 * they were never uploaded. */
int nbnxn_hip_upload_shiftvec(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat);

/* nbnxn_cuda_clear_outputs (:96): zeroes f; zeroes fshift/energies if flags has VIRIAL. */
int nbnxn_hip_clear_outputs(nbnxn_hip_ptr_t nb, int flags);

/* nbnxn_cuda_free (:101) */
int nbnxn_hip_free(nbnxn_hip_ptr_t nb);

/* nbnxn_cuda_get_timings / _reset_timings (:105,116); NULL when timing is off. */
nbx_wallclock_gpu_t *nbnxn_hip_get_timings(nbnxn_hip_ptr_t nb);
int                  nbnxn_hip_reset_timings(nbnxn_hip_ptr_t nb);

/* nbnxn_cuda_min_ci_balanced (:121): the pair-list split target (number of sci entries the
 * list builder should aim for); tuned for 256 CUs x 4 SIMDs instead of 40 x #SM. */
int nbnxn_hip_min_ci_balanced(nbnxn_hip_ptr_t nb);

/* nbnxn_cuda_is_kernel_ewald_analytical (:132) */
int nbnxn_hip_is_kernel_ewald_analytical(const nbnxn_hip_ptr_t nb);

/* ------------------------------------------------------------------------------------
 * Per-step calls; one-to-one with mdlib/nbnxn_cuda/nbnxn_cuda.h:64-86
 * ---------------------------------------------------------------------------------- */

/* nbnxn_cuda_launch_kernel (:64): async H2D of nbat->x for the locality, then the kernel. */
int nbnxn_hip_launch_kernel(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat, int flags, int iloc);

/* nbnxn_cuda_launch_cpyback (:72): async D2H of f (into nbat->out[0].f), and of
 * fshift/energies on the local stream when flags has VIRIAL. */
int nbnxn_hip_launch_cpyback(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat, int flags, int aloc);

/* nbnxn_cuda_wait_gpu (:80): blocks until the locality's stream is idle; ADDS e_lj, e_el
 * and fshift[45] into the caller's accumulators (local only); disarms pruning. */
int nbnxn_hip_wait_gpu(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat, int flags, int aloc,
                       float *e_lj, float *e_el, float (*fshift)[3]);

/* ------------------------------------------------------------------------------------
 * Flat variants (plain pointers and sizes) of the struct-taking calls above, for hosts
 * that do not hold the GROMACS structs (the Python mirror, tests, bench).  Same semantics.
 * ---------------------------------------------------------------------------------- */

/* = nbnxn_hip_init_const.  nbfp is [ntype*ntype*2] (6*C6, 12*C12 pairs,
 * mdlib/nbnxn_atomdata.c:611-617); el_type is NBX_EL_RF, NBX_EL_EWALD_ANA or NBX_EL_EWALD_TAB
 * (the reference's tabulated-force flavour, eelCuEWALD_TAB: nbnxn_cuda_types.h:80-82). */
int nbnxn_hip_init_const_flat(nbnxn_hip_ptr_t nb, int el_type,
                              float rcoulomb, float rvdw, float rlist,
                              float epsfac, float k_rf, float c_rf,
                              float ewaldcoeff_q, float sh_ewald,
                              float disp_cpot, float rep_cpot,
                              int ntype, const float *nbfp);

/* = nbnxn_hip_init_pairlist */
/* LJ modifier of interaction_const_t for hosts that use init_const_flat (the struct-taking
 * nbnxn_hip_init_const reads it from ic): vdw_modifier is enum eintmod (NBX_MOD_*), the
 * constants are ic->rvdw_switch, dispersion_shift.c2/c3, repulsion_shift.c2/c3, vdw_switch.c3/c4/c5
 * (forcerec.c:1942-1975).  The cpot values travel with init_const_flat.  Call after init_const_flat. */
int nbnxn_hip_set_vdw_modifier_flat(nbnxn_hip_ptr_t nb, int vdw_modifier, float rvdw_switch,
                                    float disp_c2, float disp_c3, float rep_c2, float rep_c3,
                                    float sw_c3, float sw_c4, float sw_c5);

/* LJ-PME (ic->vdwtype == evdwPME) for hosts that use init_const_flat; the struct-taking
 * nbnxn_hip_init_const reads the same from ic and nbat (nbnxn_cuda_data_mgmt.cu:281-282, 322-334,
 * 367-378).  comb_rule is NBX_LJCR_GEOM or NBX_LJCR_LB (ic->ljpme_comb_rule), nbfp_comb the 2*ntype
 * per-type combination parameters of nbat->nbfp_comb (nbnxn_atomdata.c:397-429).  Replaces any LJ
 * modifier set before: the reference has no switched LJ-PME flavour.  Call after init_const_flat. */
int nbnxn_hip_set_ljpme_flat(nbnxn_hip_ptr_t nb, int comb_rule, float ewaldcoeff_lj, float sh_lj_ewald,
                             const float *nbfp_comb);

int nbnxn_hip_init_pairlist_flat(nbnxn_hip_ptr_t nb, int iloc, float rlist,
                                 int nsci, const nbx_sci_t *sci,
                                 int ncj4, const nbx_cj4_t *cj4,
                                 int nexcl, const nbx_excl_t *excl, int nci_tot);

/* = nbnxn_hip_init_atomdata */
int nbnxn_hip_init_atomdata_flat(nbnxn_hip_ptr_t nb, int natoms, int natoms_local,
                                 int ntype, const int *type);

/* = nbnxn_hip_upload_shiftvec (always uploads) */
int nbnxn_hip_upload_shiftvec_flat(nbnxn_hip_ptr_t nb, const float *shift_vec /* [45*3] */);

/* = nbnxn_hip_launch_kernel: xq is the host float4 array [natoms*4]. */
int nbnxn_hip_launch_kernel_flat(nbnxn_hip_ptr_t nb, const float *xq, int flags, int iloc);

/* = nbnxn_hip_launch_cpyback: f is the host array [natoms*3]. */
int nbnxn_hip_launch_cpyback_flat(nbnxn_hip_ptr_t nb, float *f, int flags, int aloc);

/* = nbnxn_hip_wait_gpu */
int nbnxn_hip_wait_gpu_flat(nbnxn_hip_ptr_t nb, int flags, int aloc,
                            float *e_lj, float *e_el, float *fshift /* [45*3] */);

/* ------------------------------------------------------------------------------------
 * Resident-coordinate extension: x and f stay in HBM between steps, so the
 * per-step PCIe round trip of launch_kernel/launch_cpyback disappears.  Used by bench.py
 * (the timed region starts with inputs in HBM) and by the multi-GPU halo exchange, which
 * hands these device pointers to RCCL send/recv.
 * ---------------------------------------------------------------------------------- */

/* device addresses of the xq (float4), f (float[3]) and atom-type arrays */
int nbnxn_hip_get_device_buffers(nbnxn_hip_ptr_t nb, void **d_xq, void **d_f, void **d_type);

/* device address and shape of the energy / shift-force accumulation slots: float
 * [nslots][stride], each slot {e_lj, e_el, fshift[45][3], padding}; the kernels add into
 * slot (entry index mod nslots), read_outputs / wait_gpu sum the slots in double.  For
 * resident callers that reduce them on the device (the multi-GPU all-reduce). */
int nbnxn_hip_get_output_slots(nbnxn_hip_ptr_t nb, void **d_out, int *nslots, int *stride);

/* ------------------------------------------------------------------------------------
 * Pair search on the device.  The two calls stand where the
 * reference's host code has nbnxn_put_on_grid (nbnxn_search.c:1777, called at sim_util.c:967)
 * + nbnxn_atomdata_set (nbnxn_atomdata.c:1094) + nbnxn_cuda_init_atomdata
 * (nbnxn_cuda_data_mgmt.h:81), and nbnxn_make_pairlist (nbnxn_search.c:5583, sim_util.c:1021)
 * + nbnxn_cuda_init_pairlist (nbnxn_cuda_data_mgmt.h:75): atoms and list never visit the host.
 * ---------------------------------------------------------------------------------- */

/* Grids natoms atoms into columns / super-cells / clusters exactly as the host-side builder
 * (include/nbnxn_host.h) does.  box[9] is the GROMACS box matrix as nbnxn_put_on_grid takes it
 * (rows are box vectors; rectangular or triclinic within check_box's limits, gmxlib/pbc.c:121-156);
 * the atoms lie in the brick [0, box[d][d]) that put_atoms_in_box leaves them in.  The call also fills the module's grid-ordered
 * coordinates+charges (with fillers), types (dummy type = ntype - 1 of init_const), atom-order
 * arrays (as nbnxn_hip_set_atom_order would) and shift vectors.  x [natoms][3] may be a host or a
 * device array; type and q are host arrays in caller order; pbc_shift_dim NULL = periodic in
 * x, y and z; atom_density <= 0 = natoms / volume.  type = q = NULL reuses the arrays of the last
 * call (the topology is constant between searches; natoms must agree).  Invalidates the lists.
 * Call after init_const. */
int nbnxn_hip_put_on_grid(nbnxn_hip_ptr_t nb, const float box[9], const int pbc_shift_dim[3], int natoms, const float *x,
                          int x_on_device, const int *type, const float *q, float atom_density);

/* The same for a domain with halo zones, as nbnxn_put_on_grid is called once per zone
 * (sim_util.c:967 for the home zone, :978 nbnxn_put_on_grid_nonlocal for the others):
 * zone 0 = home, zones 1.. = the halo zones in the decomposition's order, each with its corners,
 * its atom range [a0, a0 + n) of x / type / q (home atoms first, then the communicated ones, as in
 * the reference's state arrays) and its atom density (<= 0: n / zone volume).  The home zone's
 * padded atoms come first in the grid-ordered arrays (they are the "local" range of
 * nbnxn_hip_launch_kernel / _cpyback). */
typedef struct {
    float corner0[3], corner1[3];
    int   a0, n;
    float atom_density;
} nbx_grid_zone_t;
int nbnxn_hip_put_on_grid_zones(nbnxn_hip_ptr_t nb, const float box[9], const int pbc_shift_dim[3], int nzones,
                                const nbx_grid_zone_t *zones, int natoms, const float *x, int x_on_device, const int *type,
                                const float *q);

/* Builds the local list for the atoms gridded by nbnxn_hip_put_on_grid and installs it as
 * nbnxn_hip_init_pairlist would (its first kernel prunes it).  excl_index [natoms+1] / excl_a:
 * topology exclusions in caller atom numbering (t_blocka index / a); NULL = those of the last call
 * that passed them (none if there was no such call: use nbnxn_hip_forget_exclusions to drop them);
 * min_ci_balanced as for nbnxn_make_pairlist (nbnxn_hip_min_ci_balanced()).  iloc NBX_LOCAL: the
 * home zone against itself; NBX_NONLOCAL: the home zone against the halo zones given to
 * nbnxn_hip_put_on_grid_zones (nbnxn_search.c:5652-5699). */
int nbnxn_hip_make_pairlist(nbnxn_hip_ptr_t nb, int iloc, float rlist, const int *excl_index, const int *excl_a, int min_ci_balanced);
/* The same with the zone pairs of the reference's eighth shell: zone_pairs [3 * nzp] = {i-zone, first j-zone,
 * end j-zone}, the dd_zp1 / dd_zp2 / dd_zp3 tables (domdec.c:433-449) that nbnxn_make_pairlist walks in its i-zone
 * loop (nbnxn_search.c:5652-5699).  The local list is {0, 0, 1}; the non-local list pairs up to four i-zones
 * (imported ones included) with halo zones other than their own.  Same lists as nbnxn_host_make_pairlist_zones
 * (include/nbnxn_host.h).  nzp = 0 is nbnxn_hip_make_pairlist. */
int nbnxn_hip_make_pairlist_zones(nbnxn_hip_ptr_t nb, int iloc, int nzp, const int *zone_pairs, float rlist, const int *excl_index,
                                  const int *excl_a, int min_ci_balanced);
int nbnxn_hip_forget_exclusions(nbnxn_hip_ptr_t nb);
/* Progressive balancing of the lists nbnxn_hip_make_pairlist splits (default on), the device builder's
 * side of nbnxn_host_set_progressive_balance (include/nbnxn_host.h): same rule, same lists. */
int nbnxn_hip_set_progressive_balance(nbnxn_hip_ptr_t nb, int on);

/* Verification aids: the device grid (dims = {ncx, ncy of the home zone, super-cells and padded atoms of all zones};
 * a [padded], cell [natoms], bb [clusters][6] = lo xyz, hi xyz, bbcz [super-cells][2],
 * nsubc [super-cells], xq [padded][4], type [padded]; any pointer may be NULL) and the list
 * installed for iloc (counts = {nsci, ncj4, nexcl, nci_tot}; call once with NULL arrays for the sizes). */
int nbnxn_hip_download_grid(nbnxn_hip_ptr_t nb, int *dims, int *a, int *cell, float *bb, float *bbcz, int *nsubc, float *xq, int *type);
int nbnxn_hip_download_pairlist(nbnxn_hip_ptr_t nb, int iloc, int *counts, nbx_sci_t *sci, nbx_cj4_t *cj4, nbx_excl_t *excl);

/* Verification aid for the tabulated Ewald flavour: size and scale of the force table this
 * module generated (init_ewald_coulomb_force_table, nbnxn_cuda_data_mgmt.cu:120-183:
 * 1536 points, scale = (size - 2) / rcoulomb) and, when tab != NULL, its entries in the
 * reference's units. */
int nbnxn_hip_download_coulomb_tab(nbnxn_hip_ptr_t nb, float *tab, int tab_alloc, int *tab_size, float *tab_scale);

/* the HIP stream (hipStream_t) of a locality, as an opaque pointer */
void *nbnxn_hip_get_stream(nbnxn_hip_ptr_t nb, int iloc);

/* Dynamic pruning: re-prunes the list of iloc, built with the (outer) radius it was created with, to
 * rlist_inner <= rlist using the coordinates now on the device; the force kernels then run on the
 * inner list until the next call.  No counterpart in 5.0.4, whose lists are pruned once by their
 * first force kernel (nbnxn_cuda_kernel.cuh:332-340); it is the nbnxn_gpu_launch_kernel_pruneonly of
 * later GROMACS versions, named in SURVEY 8f-2.  The first call on a new list prunes all of it (and
 * takes the place of the pruning force-kernel flavour); later calls prune one of num_parts slices
 * of the entries in turn ("rolling": 1 = the whole list every call).  The caller chooses how many
 * steps an inner list stays valid, as it does for the outer one. */
int nbnxn_hip_launch_kernel_pruneonly(nbnxn_hip_ptr_t nb, int iloc, float rlist_inner, int num_parts);

/* kernel only, no H2D: coordinates already resident. */
int nbnxn_hip_launch_kernel_resident(nbnxn_hip_ptr_t nb, int flags, int iloc);

/* device-side energy / shift-force outputs without the host accumulate of wait_gpu:
 * copies e_lj, e_el, fshift[135] to host (synchronises the local stream). */
int nbnxn_hip_read_outputs(nbnxn_hip_ptr_t nb, float *e_lj, float *e_el, float *fshift);

/* bit-exact list verification (north_star: "uploaded once per list lifetime and verified
 * bit-exact"): reads sci/cj4/excl back from HBM and compares with the host list; cj4.cj
 * slots whose imask byte is 0 are padding and are ignored (SURVEY 8a trap 1).
 * Returns 0 when identical, otherwise the number of differing words. */
int nbnxn_hip_verify_pairlist_flat(nbnxn_hip_ptr_t nb, int iloc,
                                   int nsci, const nbx_sci_t *sci,
                                   int ncj4, const nbx_cj4_t *cj4,
                                   int nexcl, const nbx_excl_t *excl);

/* ------------------------------------------------------------------------------------
 * Device-side atom-data moves.  The reference permutes coordinates into grid
 * order and forces back into caller order on the host every step
 * (nbnxn_atomdata_copy_x_to_nbat_x, mdlib/nbnxn_atomdata.c:1141; nbnxn_atomdata_add_nbat_f_to_f,
 * :1600).  Here the two index arrays of the search (nbs->a and nbs->cell,
 * nbnxn_search.c:1588-1660) are uploaded once per search and both permutations run on the GPU,
 * so a caller moves 12 B per atom each way instead of the padded 16 B + 12 B, or nothing at
 * all when its x and f live in HBM.
 * ---------------------------------------------------------------------------------- */

/* a[natoms padded, as given to init_atomdata]: grid slot -> caller atom, -1 for fillers;
 * cell[natoms_real]: caller atom -> grid slot.  Call after init_atomdata of every search. */
int nbnxn_hip_set_atom_order(nbnxn_hip_ptr_t nb, int natoms_real, const int *a, const int *cell);

/* uploads a whole host xq range (charges and filler positions change only with the search) */
int nbnxn_hip_upload_xq_flat(nbnxn_hip_ptr_t nb, const float *xq, int iloc);

/* nbnxn_atomdata_copy_x_to_nbat_x: x [natoms_real*3] in caller order (host memory, or device
 * memory when x_on_device) -> the xyz of the atom slots of locality iloc; async on its stream. */
int nbnxn_hip_copy_x_to_xq(nbnxn_hip_ptr_t nb, const float *x, int x_on_device, int iloc);

/* nbnxn_atomdata_add_nbat_f_to_f: f[at] += fnb[cell[at]] for at in [a0, a0+n).  Device f: async
 * on the stream of aloc.  Host f: returns when f is updated. */
int nbnxn_hip_add_f(nbnxn_hip_ptr_t nb, float *f, int f_on_device, int a0, int n, int aloc);

/* ------------------------------------------------------------------------------------
 * Halo moves of a domain-decomposed step on the device.  The reference moves halo
 * coordinates and forces through host buffers and MPI: dd_move_x (mdlib/domdec.c:651-758) copies
 * the owner's atoms, plus the box shift of a pulse that crosses a periodic boundary (:685-706),
 * into a send buffer; dd_move_f (:761-875) adds the forces that come back to the owner's atoms and,
 * for such a pulse, to the shift force of that boundary as well (:837-851).  Here x and f live in
 * HBM in grid order, so the buffers do too and a communication library (RCCL) takes them as they
 * are.  d_slots are grid slots (device memory, n of them); d_buf is device memory; iloc names the
 * stream the move is put on (one-stream handles: their only stream).
 * ---------------------------------------------------------------------------------- */

/* owner's side of dd_move_x: d_buf[i] = {xq[d_slots[i]].xyz + shift, q}; shift NULL = none */
int nbnxn_hip_halo_pack_x(nbnxn_hip_ptr_t nb, const int *d_slots, int n, const float *shift, float *d_buf, int iloc);
/* importer's side: xq[d_slots[i]] = d_buf[i]  (float4 rows) */
int nbnxn_hip_halo_unpack_x(nbnxn_hip_ptr_t nb, const int *d_slots, int n, const float *d_buf, int iloc);
/* importer's side of dd_move_f: d_buf[i] = f[d_slots[i]]  (3 floats per row) */
int nbnxn_hip_halo_pack_f(nbnxn_hip_ptr_t nb, const int *d_slots, int n, float *d_buf, int iloc);
/* owner's side: f[d_slots[i]] += d_buf[i] (atomic, the force kernels may still be running);
 * shift_index >= 0: the sum of d_buf is also added to that entry of the shift forces of the
 * output in use (one of its accumulation slots), shift_index -1: not. */
int nbnxn_hip_halo_unpack_f(nbnxn_hip_ptr_t nb, const int *d_slots, int n, const float *d_buf, int shift_index, int iloc);
/* everything put on the stream of iloc_waits from here on runs after what the stream of
 * iloc_done holds now (two pre-made events, no allocation); a no-op on one-stream handles */
int nbnxn_hip_stream_wait(nbnxn_hip_ptr_t nb, int iloc_waits, int iloc_done);
/* Two-stream handles: how many CUs the local stream leaves free (a CU mask: one CU per shader engine by default,
 * GMX_HIP_NB_RESERVE_CUS overrides, 0 = none), so that the halo exchange's kernels -- RCCL's above all -- can run
 * beside a local kernel that would otherwise hold every wave slot.  No counterpart in the reference. */
int nbnxn_hip_reserved_cus(nbnxn_hip_ptr_t nb);

/* ------------------------------------------------------------------------------------
 * Domain decomposition over RCCL: the halo exchange of one decomposed dimension,
 * issued by this library on its own streams.  These calls stand where the reference has
 * dd_move_x (sim_util.c:1114; mdlib/domdec.c:651-758) and dd_move_f (sim_util.c:1428;
 * domdec.c:761-875), whose MPI_Sendrecv of host buffers (dd_sendrecv_rvec,
 * domdec_network.c:114) becomes pack kernel -> one grouped ncclSend / ncclRecv between devices ->
 * unpack kernel on the non-local stream; energies and shift forces are summed over ranks by one
 * ncclAllReduce of the accumulation slots (the reference sums them on the host in global_stat).
 * One pulse along one dimension (nbnxn_hip_dd_set_pulse, the slab decomposition of gromacs-cosmo_amd/dd.py), or up to
 * NBX_DD_MAX_MESSAGES messages with direct peers (nbnxn_hip_dd_set_messages: a decomposition in two or three dimensions
 * whose halo zones each come straight from the rank that owns them, no forwarding through intermediate ranks as in
 * the staged pulses of domdec.c:667-758 -- on a node whose GPUs are all linked to each other every neighbour is
 * one hop away).  RCCL is bound at run time: the copy the process already holds (librccl.so.1), else the system's.
 * ---------------------------------------------------------------------------------- */
#define NBX_DD_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: one rank makes the id, the host's launcher hands it to the others (mdrun:
 * MPI_Bcast over cr->mpi_comm_mysim; the Python mirror: torch.distributed.broadcast) */
int nbnxn_hip_dd_get_unique_id(char *id128);
/* ncclCommInitRank on the handle's device: collective over the nranks ranks of the id */
int nbnxn_hip_dd_init(nbnxn_hip_ptr_t nb, const char *id128, int rank, int nranks);
/* how many ranks the module's communicator holds, as RCCL itself reports it (ncclCommCount): what a benchmark line
 * quotes to show that the halo messages went between that many devices.  No counterpart in the reference (its halo is
 * host MPI, domdec.c:651-875). */
int nbnxn_hip_dd_comm_count(nbnxn_hip_ptr_t nb, int *nranks);
/* Per search: coordinates of the grid slots send_slots go to rank send_to (plus shift, the box
 * vector of a pulse across a periodic boundary, NULL = none), coordinates for the slots recv_slots
 * arrive from rank recv_from; forces travel the reverse way, and those that arrive add to the shift
 * force shift_index as well (-1: not).  Host arrays; copied. */
int nbnxn_hip_dd_set_pulse(nbnxn_hip_ptr_t nb, int send_to, int recv_from, const int *send_slots, int n_send,
                           const int *recv_slots, int n_recv, const float *shift, int shift_index);
/* The general form: nmsg messages (1..NBX_DD_MAX_MESSAGES) per step and direction.  Message k carries the coordinates
 * of n_send grid slots to rank send_to, each plus shift (the box vectors of the periodic boundaries the message
 * crosses; zeros = none), and fills n_recv grid slots with what rank recv_from sends as ITS message k; forces travel
 * the reverse way, and those that arrive for message k add to the shift force shift_index as well (-1: not), the rule
 * of dd_move_f for pulses across a periodic boundary (domdec.c:837-851).  send_slots / recv_slots hold the slots of
 * all messages back to back, in message order.  Several messages may go to the same rank (a dimension cut in two has
 * the same neighbour on both sides): every rank must then list them in the same order, RCCL matches the sends and
 * receives between two ranks in the order they were issued.  One pack kernel, one grouped RCCL call and one unpack
 * kernel per step and direction, whatever nmsg is.  Host arrays; copied. */
typedef struct {
    int   send_to, recv_from;
    int   n_send, n_recv;
    int   shift_index;
    float shift[3];
} nbx_dd_message_t;
int nbnxn_hip_dd_set_messages(nbnxn_hip_ptr_t nb, int nmsg, const nbx_dd_message_t *msg, const int *send_slots, const int *recv_slots);
/* The reference's staged pulses (dd_move_x, domdec.c:667-758; dd_move_f, :784-870): the messages of
 * nbnxn_hip_dd_set_messages split into nphase phases, phase p = messages [phase_end[p - 1], phase_end[p]).  The
 * coordinate exchange runs the phases in order (pack, grouped send / receive, unpack, each behind the one before on
 * the non-local stream), so a later phase may send slots that an earlier phase filled -- the pulse of a later
 * dimension forwards the zones the earlier dimensions brought; the force exchange runs them in reverse order, so
 * what comes back for a forwarded atom is added to its slot before that slot's own pulse returns it.  Send slots of
 * one phase must be distinct.  Default after nbnxn_hip_dd_set_messages: one phase. */
int nbnxn_hip_dd_set_phases(nbnxn_hip_ptr_t nb, int nphase, const int *phase_end);
/* dd_move_x / dd_move_f for that pulse (or those messages), asynchronous on the non-local stream; every rank calls them.
 * Ordering: the pack kernel of dd_move_x (and nbnxn_hip_halo_pack_x) waits for the last write of home coordinates
 * that this library issued on the LOCAL stream (the H2D of nbnxn_hip_launch_kernel / nbnxn_hip_upload_xq_flat, the
 * gather of nbnxn_hip_copy_x_to_xq) -- for that write, not for a local kernel launched behind it, beside which the
 * exchange is meant to run.  A caller that writes the coordinate array by other means (its own kernels through
 * nbnxn_hip_get_device_buffers) orders that itself (nbnxn_hip_stream_wait).  The next step's coordinate write must
 * in turn come after this step's pack: nbnxn_hip_stream_wait(nb, NBX_LOCAL, NBX_NONLOCAL) at the end of the step,
 * which the force reduction needs anyway, gives that. */
int nbnxn_hip_dd_move_x(nbnxn_hip_ptr_t nb);
int nbnxn_hip_dd_move_f(nbnxn_hip_ptr_t nb);
/* in-place ncclAllReduce(sum) of the output slots in use, on the stream of iloc (order it after both
 * kernels with nbnxn_hip_stream_wait) */
int nbnxn_hip_dd_sum_outputs(nbnxn_hip_ptr_t nb, int iloc);
/* One decomposed step with resident coordinates from ONE call (the host's time to issue a step limits a short step):
 * [dd_move_x, if the exchange is not to run beside the local kernel] -> [dd_sum_outputs of the LAST step's slots, if
 * sum_last_outputs] -> clear_outputs -> local kernel -> [dd_move_x beside it] -> non-local kernel -> dd_move_f ->
 * nbnxn_hip_stream_wait(LOCAL, NONLOCAL).  ev_before_local / ev_after_local: hipEvent_t handles recorded around the
 * local kernel on the local stream (NULL: none).  The sequence of gromacs-cosmo_amd/dd.py::SlabRun._step_rccl_direct;
 * stands where sim_util.c:1012-1212 interleaves launches with dd_move_x / dd_move_f. */
int nbnxn_hip_dd_step(nbnxn_hip_ptr_t nb, int flags, int sum_last_outputs, int exchange_beside_local, void *ev_before_local,
                      void *ev_after_local);
/* destroys the communicator and the halo buffers (nbnxn_hip_free does it too) */
int nbnxn_hip_dd_finish(nbnxn_hip_ptr_t nb);

/* reads the (possibly pruned) cj4 array back from HBM; for tests of the prune flavour. */
int nbnxn_hip_download_cj4(nbnxn_hip_ptr_t nb, int iloc, int ncj4, nbx_cj4_t *cj4);

/* last kernel duration of a locality in milliseconds from the module's HIP events
 * (the wallclock_gpu_t source); <0 when no kernel was timed yet. */
float nbnxn_hip_last_kernel_ms(nbnxn_hip_ptr_t nb, int iloc);

/* arms (1) or disarms (0) the pruning flavour for the next launch of a locality;
 * the reference arms it in init_pairlist and disarms it in wait_gpu. */
int nbnxn_hip_set_prune(nbnxn_hip_ptr_t nb, int iloc, int do_prune);

/* In-kernel clear (off by default).  When on, the module keeps two output buffers {energy
 * slots, f}; every LOCAL launch also zeroes the one that is not in use and
 * nbnxn_hip_clear_outputs just switches to it instead of dispatching a fill.  For resident
 * callers that step every ~100 us, where the fill and its dispatch gap are ~7 % of the step.
 * The addresses returned by get_device_buffers / get_output_slots change with every
 * clear_outputs while this is on: query them after it. */
int nbnxn_hip_set_clear_in_kernel(nbnxn_hip_ptr_t nb, int on);

/* Copy kernels (on by default; NBNXN_HIP_COPY_KERNELS=0 in the environment starts handles with them off).  The per-step
 * moves of the reference's calling convention -- xq in at nbnxn_hip_launch_kernel (nbnxn_cuda.cu:347-348), f and the
 * energy / shift-force partial sums out at nbnxn_hip_launch_cpyback (:492-522) -- are done by a kernel that reads /
 * writes the caller's arrays directly when they are pinned and mapped into the device's address space (nbnxn_hip_pmalloc,
 * hipHostMalloc, hipHostRegister) and 16-byte aligned; the forces then leave in ONE launch that also sums the partial-sum
 * slots (in double, in slot order: the bits the host-side sum gives) so that the host reads 137 numbers.  Pageable,
 * unaligned or device-resident arrays go through hipMemcpyAsync as before.  Same bytes either way. */
int nbnxn_hip_set_copy_kernels(nbnxn_hip_ptr_t nb, int on);

/* switches the module's HIP-event timing (wallclock_gpu_t accounting) on or off; the default
 * follows the reference's bDoTime (nbnxn_cuda_data_mgmt.cu:714-715): on with one stream unless
 * GMX_DISABLE_HIP_TIMING is set, off with two.  Each timed launch carries two event records
 * (~7 us of device time on MI355X), so resident callers that launch every ~100 us turn it off. */
int nbnxn_hip_set_timing(nbnxn_hip_ptr_t nb, int on);

/* ---- the entry point that allocates nothing ----
 * No counterpart in the reference.  It is for callers that cannot call hipMalloc: code under HIP graph capture,
 * tight serving loops, harnesses that forbid allocation.  A variant of nbnxn_hip_launch_kernel_resident: the same
 * kernel over the pair list, atom types and parameters installed in the handle (by init_const / init_atomdata /
 * init_pairlist or the device search, which do allocate and come before), on the handle's stream of `iloc`, but
 *   - coordinates are read from the CALLER's device array d_xq [natoms][4] {x, y, z, q} in grid order,
 *   - forces are ADDED into the caller's device array d_f [natoms][3] (the caller clears it),
 *   - with NBX_FORCE_VIRIAL in flags, {e_lj, e_el, fshift[45][3]} (137 floats) are ASSIGNED to the caller's device
 *     array d_e_fshift; without it that array is not touched and may be NULL,
 * and the kernel's scratch buffer -- the 64 x 144-float partial-sum slots the waves add into, which every other
 * launch takes from the handle's allocation -- is a file-scope device array of the library: 73,728 bytes per GPU (a half per locality),
 * held until the process ends, whatever the system size.  The call issues one fill of that array, the kernel and a
 * 137-thread fold kernel; it calls no hipMalloc / hipFree / allocating routine and returns without synchronising
 * (order later work after it on nbnxn_hip_get_stream(nb, iloc), or synchronise that stream).
 * Limits, refused with an error: natoms must equal the installed atom count and be <= NBX_NOALLOC_MAX_ATOMS, the
 * installed list may have at most NBX_NOALLOC_MAX_NSCI entries (32-bit indexing inside the kernels; the scratch
 * array itself does not grow with the system).  Every call on one GPU shares that GPU's array, which keeps the last
 * call's contents (the call clears its half first): the local and the non-local call of one two-stream handle may
 * overlap (each locality has its own half of the array), calls of TWO handles that overlap on one GPU race on the
 * energies; forces are unaffected.  A fresh list is pruned by this
 * launch as by any other (the handle's list masks are rewritten in place). */
#define NBX_NOALLOC_MAX_ATOMS (1 << 28)
#define NBX_NOALLOC_MAX_NSCI  (1 << 30)
int nbnxn_hip_launch_kernel_noalloc(nbnxn_hip_ptr_t nb, int flags, int iloc, int natoms, const float *d_xq, float *d_f,
                                    float *d_e_fshift);

/* kernel tuning knob for benchmarking: 0 = default variant */
int nbnxn_hip_set_kernel_variant(nbnxn_hip_ptr_t nb, int variant);

#ifdef __cplusplus
}
#endif

#endif /* NBNXN_HIP_H */
