/*
 * ref_harness.c -- drives the REFERENCE's own plain-C kernels on flat arrays.
 * TEST INFRASTRUCTURE ONLY (see oracle/nbnxn_oracle.c header); built into
 * oracle/_ref/libnbnxn_ref.so by oracle/Makefile, only where /root/reference exists.
 *
 * What is reference code in that library: the translation units
 *   mdlib/nbnxn_kernels/nbnxn_kernel_ref.c      (nbnxn_kernel_ref, the contract oracle)
 *   mdlib/nbnxn_kernels/nbnxn_kernel_gpu_ref.c  (nbnxn_kernel_gpu_ref, same-list oracle)
 *   mdlib/nbnxn_kernels/nbnxn_kernel_common.c   (clear_f, reduce_energies_over_lists)
 *   math/utilities.c                            (gmx_erff)
 * compiled unmodified from where they lie, against the reference's own headers.
 *
 * What is NOT reference code, and why: those units call four host services that live in
 * units needing cmake-generated headers (gmx_header_config_gen.h), which this image
 * cannot produce: gmx_fatal / _gmx_error (error exit), the `debug` FILE*, and
 * gmx_omp_nthreads_get (thread count).  This file supplies them as the host program
 * (abort with the message; debug off; the thread count the caller asked for).  None of
 * them takes part in the arithmetic.  The pair search (nbnxn_search.c) and the Ewald table
 * filler (tables.c) are unbuildable for the same reason: lists come from the repo's own
 * builder (identical inputs for both sides), and fill_ewald_table() below restates
 * table_spline3_fill_ewald_lr (tables.c:165-311) / ewald_spline3_table_scale (:347-369).
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "typedefs.h"
#include "types/nbnxn_pairlist.h"
#include "types/interaction_const.h"
#include "types/force_flags.h"
#include "types/nb_verlet.h"
#include "nbnxn_kernels/nbnxn_kernel_ref.h"
#include "nbnxn_kernels/nbnxn_kernel_gpu_ref.h"

/* ---- host services the kernel units expect from the program they are linked into ---- */
FILE *debug = NULL;
static int g_nthreads = 1;

void gmx_fatal(int fatal_errno, const char *file, int line, const char *fmt, ...)
{
    va_list ap;
    (void)fatal_errno;
    fprintf(stderr, "reference gmx_fatal at %s:%d: ", file, line);
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    abort();
}

void _gmx_error(const char *key, const char *msg, const char *file, int line)
{
    fprintf(stderr, "reference gmx_error(%s) at %s:%d: %s\n", key, file, line, msg);
    abort();
}

int gmx_omp_nthreads_get(int mod)
{
    (void)mod;
    return g_nthreads;
}

/* ---- Ewald correction table, restating tables.c:165-311 (quadratic spline, force = average
 * of the two adjacent interval derivatives) and the spacing rule of tables.c:317-369 ---- */
static double v_q_ewald_lr(double beta, double r)
{
    return r == 0 ? beta * 2 / sqrt(M_PI) : erf(beta * r) / r;
}

static void fill_ewald_table(float *tab_f, float *tab_v, float *tab_fdv0, int ntab, double dx, double beta)
{
    int    i;
    double dc = 0;
    for (i = ntab - 1; i >= 0; i--)
    {
        const double x0 = i * dx;
        const double v0 = v_q_ewald_lr(beta, x0);
        double       v1, a2dx;
        tab_v[i] = (float)v0;
        if (i == 0) { continue; }
        v1   = v_q_ewald_lr(beta, (i - 1) * dx);
        a2dx = (v0 + v1 - 2 * v_q_ewald_lr(beta, x0 - 0.5 * dx)) / (0.25 * dx);
        dc   = (v0 - v1) / dx + 0.5 * a2dx;
        if (i == ntab - 1) { tab_f[i] = (float)(-dc); }
        else { tab_f[i] += (float)(-0.5 * dc); }
        a2dx = (dc * dx + v1 - v0) * 2 / dx;
        dc   = dc - a2dx;
        tab_f[i - 1] = (float)(-0.5 * dc);
    }
    tab_f[0] *= 2;
    for (i = 0; i < ntab - 1; i++)
    {
        tab_fdv0[4 * i]     = tab_f[i];
        tab_fdv0[4 * i + 1] = tab_f[i + 1] - tab_f[i];
        tab_fdv0[4 * i + 2] = tab_v[i];
        tab_fdv0[4 * i + 3] = 0;
    }
    tab_fdv0[4 * (ntab - 1)]     = tab_f[ntab - 1];
    tab_fdv0[4 * (ntab - 1) + 1] = -tab_f[ntab - 1];
    tab_fdv0[4 * (ntab - 1) + 2] = tab_v[ntab - 1];
    tab_fdv0[4 * (ntab - 1) + 3] = 0;
}

static double ewald_table_scale(double beta, double rc)
{
    const double d3 = 1.0522;                     /* max |(erf(x)/x)'''|, tables.c:355 */
    const double etol = 0.1 * erfc(beta * rc);
    const double sc_deriv = sqrt(d3 / (6 * 4 * 5.96046448E-08 * beta)) * beta;
    const double ftol = etol > 5.96046448E-08 ? etol : 5.96046448E-08;
    const double sc_func = pow(d3 / (6 * 12 * sqrt(3) * ftol), 1.0 / 3.0) * beta;
    return sc_deriv > sc_func ? sc_deriv : sc_func;
}

typedef struct {
    int   el_type;      /* 0 RF, 1 Ewald */
    float rcoulomb, rvdw, rlist;
    float epsfac, k_rf, c_rf;
    float ewaldcoeff_q, sh_ewald;
    float disp_cpot, rep_cpot;
    int   vdw_modifier;   /* enum eintmod */
    float rvdw_switch;
    float disp_c2, disp_c3, rep_c2, rep_c3;
    float sw_c3, sw_c4, sw_c5;
    int   ljpme;          /* 0 off, 1 LJ-PME geometric, 2 LJ-PME Lorentz-Berthelot */
    float ewaldcoeff_lj, sh_lj_ewald;
} ref_params_t;

static void make_ic(const ref_params_t *p, interaction_const_t *ic)
{
    memset(ic, 0, sizeof(*ic));
    ic->vdwtype               = p->ljpme ? evdwPME : evdwCUT;
    ic->ljpme_comb_rule       = p->ljpme == 2 ? ljcrLB : ljcrGEOM;
    ic->ewaldcoeff_lj         = p->ewaldcoeff_lj;
    ic->sh_lj_ewald           = p->sh_lj_ewald;
    ic->vdw_modifier          = p->vdw_modifier;
    ic->rvdw                  = p->rvdw;
    ic->rvdw_switch           = p->rvdw_switch;
    ic->dispersion_shift.c2   = p->disp_c2;
    ic->dispersion_shift.c3   = p->disp_c3;
    ic->repulsion_shift.c2    = p->rep_c2;
    ic->repulsion_shift.c3    = p->rep_c3;
    ic->vdw_switch.c3         = p->sw_c3;
    ic->vdw_switch.c4         = p->sw_c4;
    ic->vdw_switch.c5         = p->sw_c5;
    ic->dispersion_shift.cpot = p->disp_cpot;
    ic->repulsion_shift.cpot  = p->rep_cpot;
    ic->sh_invrc6             = -p->disp_cpot;
    ic->eeltype               = p->el_type == 0 ? eelRF : eelPME;
    ic->coulomb_modifier      = eintmodPOTSHIFT;
    ic->rcoulomb              = p->rcoulomb;
    ic->rlist                 = p->rlist;
    ic->rlistlong             = p->rlist;
    ic->ewaldcoeff_q          = p->ewaldcoeff_q;
    ic->sh_ewald              = p->sh_ewald;
    ic->epsilon_r             = 1;
    ic->epsfac                = p->epsfac;
    ic->k_rf                  = p->k_rf;
    ic->c_rf                  = p->c_rf;
    if (p->el_type != 0)
    {
        /* init_ewald_f_table, forcerec.c:1865-1905, bUsesSimpleTables branch */
        const float maxr = p->rlist > p->rcoulomb ? p->rlist : p->rcoulomb;
        ic->tabq_scale = (float)ewald_table_scale(p->ewaldcoeff_q, p->rcoulomb);
        ic->tabq_size  = (int)(maxr * ic->tabq_scale) + 2;
        ic->tabq_coul_F    = calloc(ic->tabq_size, sizeof(float));
        ic->tabq_coul_V    = calloc(ic->tabq_size, sizeof(float));
        ic->tabq_coul_FDV0 = calloc((size_t)ic->tabq_size * 4, sizeof(float));
        fill_ewald_table(ic->tabq_coul_F, ic->tabq_coul_V, ic->tabq_coul_FDV0,
                         ic->tabq_size, 1.0 / ic->tabq_scale, p->ewaldcoeff_q);
    }
}

static void free_ic(interaction_const_t *ic)
{
    free(ic->tabq_coul_F);
    free(ic->tabq_coul_V);
    free(ic->tabq_coul_FDV0);
}

/* The Coulomb table make_ic() hands to the reference kernels (the harness's restatement of tables.c, see the
 * header): size and scale are returned, FDV0 [4 * size] is copied when fdv0 is non-NULL with room for it.  The tests
 * check every point of it against the closed forms (tests/test_oracle_golden.py), so that an error in the
 * restatement cannot hide inside a force tolerance. */
int ref_ewald_table(const ref_params_t *p, float *scale, float *fdv0, int fdv0_alloc)
{
    interaction_const_t ic;
    int                 n;
    if (p->el_type == 0) { return 0; }
    make_ic(p, &ic);
    n = ic.tabq_size;
    if (scale) { *scale = ic.tabq_scale; }
    if (fdv0 && fdv0_alloc >= 4 * n) { memcpy(fdv0, ic.tabq_coul_FDV0, (size_t)n * 4 * sizeof(float)); }
    free_ic(&ic);
    return n;
}

/* nbnxn_kernel_gpu_ref on the GPU-format list.  f [natoms*3] is cleared and filled. */
int ref_run_kernel_gpu_ref(int natoms, const float *xq, const int *type, int ntype,
                           const float *nbfp, const float *shift_vec,
                           int nsci, const nbnxn_sci_t *sci, int ncj4, const nbnxn_cj4_t *cj4,
                           int nexcl, const nbnxn_excl_t *excl,
                           const ref_params_t *p, int calc_ener,
                           float *f, float *fshift, float *Vc, float *Vvdw)
{
    nbnxn_pairlist_t    nbl;
    nbnxn_atomdata_t    nbat;
    interaction_const_t ic;
    int                 flags = GMX_FORCE_FORCES | GMX_FORCE_VIRIAL | (calc_ener ? GMX_FORCE_ENERGY : 0);

    memset(&nbl, 0, sizeof(nbl));
    nbl.bSimple = FALSE;
    nbl.na_ci = 8; nbl.na_cj = 8; nbl.na_sc = 64;
    nbl.rlist = p->rlist;
    nbl.nsci = nsci;   nbl.sci = (nbnxn_sci_t *)sci;
    nbl.ncj4 = ncj4;   nbl.cj4 = (nbnxn_cj4_t *)cj4;
    nbl.nexcl = nexcl; nbl.excl = (nbnxn_excl_t *)excl;

    memset(&nbat, 0, sizeof(nbat));
    nbat.ntype = ntype;  nbat.nbfp = (real *)nbfp;
    nbat.natoms = natoms; nbat.natoms_local = natoms;
    nbat.type = (int *)type;
    nbat.XFormat = nbatXYZQ; nbat.FFormat = nbatXYZ;
    nbat.na_c = 8; nbat.nenergrp = 1;
    nbat.xstride = 4; nbat.fstride = 3;
    nbat.x = (real *)xq;
    nbat.shift_vec = (rvec *)shift_vec;

    make_ic(p, &ic);
    memset(fshift, 0, SHIFTS * 3 * sizeof(float));
    *Vc = 0; *Vvdw = 0;
    nbnxn_kernel_gpu_ref(&nbl, &nbat, &ic, (rvec *)shift_vec, flags, enbvClearFYes, f, fshift, Vc, Vvdw);
    free_ic(&ic);
    return 0;
}

/* ---- GPU list -> simple 4x4 list on the same atom order (each 8-cluster = two 4-clusters).
 * Covers exactly the same atom pairs with the same interaction bits. ---- */
typedef struct { int ci, shift, cj; unsigned excl; } pair44_t;

static int cmp_pair44(const void *a, const void *b)
{
    const pair44_t *x = a, *y = b;
    int             fx, fy, dx, dy;
    if (x->ci != y->ci) { return x->ci < y->ci ? -1 : 1; }
    if (x->shift != y->shift) { return x->shift < y->shift ? -1 : 1; }
    /* within an i-entry: entries with exclusions first (nbnxn_pairlist.h:68-74), the
     * diagonal first of all (kernel_ref_outer.h:305), then ascending cj */
    fx = x->excl == 0xffffu; fy = y->excl == 0xffffu;
    if (fx != fy) { return fx - fy; }
    dx = !(x->shift == CENTRAL && x->cj == x->ci); dy = !(y->shift == CENTRAL && y->cj == y->ci);
    if (dx != dy) { return dx - dy; }
    return x->cj < y->cj ? -1 : (x->cj > y->cj);
}

typedef struct {
    int         nci, ncj;
    nbnxn_ci_t *ci;
    nbnxn_cj_t *cj;
} simple_list_t;

static void make_simple_list(int nsci, const nbnxn_sci_t *sci, const nbnxn_cj4_t *cj4,
                             const nbnxn_excl_t *excl, simple_list_t *sl)
{
    size_t    n = 0, cap = 1 << 20, k;
    pair44_t *pr = malloc(cap * sizeof(*pr));
    int       s, j4, jm, im, hi, hj, i, j;
    for (s = 0; s < nsci; s++)
    {
        for (j4 = sci[s].cj4_ind_start; j4 < sci[s].cj4_ind_end; j4++)
        {
            for (jm = 0; jm < 4; jm++)
            {
                for (im = 0; im < 8; im++)
                {
                    const int ci8 = sci[s].sci * 8 + im, cj8 = cj4[j4].cj[jm];
                    if (!((cj4[j4].imei[0].imask >> (jm * 8 + im)) & 1u)) { continue; }
                    for (hi = 0; hi < 2; hi++)
                    {
                        for (hj = 0; hj < 2; hj++)
                        {
                            unsigned m = 0;
                            if (sci[s].shift == CENTRAL && ci8 == cj8 && hj < hi) { continue; }
                            for (i = 0; i < 4; i++)
                            {
                                for (j = 0; j < 4; j++)
                                {
                                    const int ic = hi * 4 + i, jc = hj * 4 + j;
                                    const unsigned w = excl[cj4[j4].imei[jc >> 2].excl_ind].pair[(jc & 3) * 8 + ic];
                                    m |= ((w >> (jm * 8 + im)) & 1u) << (i * 4 + j);
                                }
                            }
                            if (n == cap) { cap *= 2; pr = realloc(pr, cap * sizeof(*pr)); }
                            pr[n].ci = ci8 * 2 + hi; pr[n].shift = sci[s].shift; pr[n].cj = cj8 * 2 + hj; pr[n].excl = m;
                            n++;
                        }
                    }
                }
            }
        }
    }
    qsort(pr, n, sizeof(*pr), cmp_pair44);
    sl->cj  = malloc((n + 1) * sizeof(*sl->cj));
    sl->ci  = malloc((n + 1) * sizeof(*sl->ci));
    sl->nci = 0; sl->ncj = (int)n;
    for (k = 0; k < n; k++)
    {
        sl->cj[k].cj = pr[k].cj; sl->cj[k].excl = pr[k].excl;
        if (k == 0 || pr[k].ci != pr[k - 1].ci || pr[k].shift != pr[k - 1].shift)
        {
            if (sl->nci > 0) { sl->ci[sl->nci - 1].cj_ind_end = (int)k; }
            sl->ci[sl->nci].ci = pr[k].ci;
            /* full LJ + Coulomb for every i-cluster: NBNXN_CI_DO_LJ(0) | NBNXN_CI_DO_COUL(0) */
            sl->ci[sl->nci].shift = pr[k].shift | NBNXN_CI_DO_LJ(0) | NBNXN_CI_DO_COUL(0);
            sl->ci[sl->nci].cj_ind_start = (int)k;
            sl->nci++;
        }
    }
    if (sl->nci > 0) { sl->ci[sl->nci - 1].cj_ind_end = (int)n; }
    free(pr);
}

/* persistent context so that the kernel can be timed without the set-up */
typedef struct {
    int                  natoms, nlist;
    simple_list_t        sl;
    nbnxn_pairlist_t    *nbl;
    nbnxn_pairlist_t   **nblp;
    nbnxn_pairlist_set_t set;
    nbnxn_atomdata_t     nbat;
    interaction_const_t  ic;
    float               *x3, *q, *shift_vec;
    int                 *type;
} ref_ctx_t;

/* Prepares nbnxn_kernel_ref (plain-C 4x4) for the atoms/list given in GPU layout.
 * nlist = number of per-thread lists (nnbl); the i-entries are split in contiguous blocks
 * with about equal j counts. */
ref_ctx_t *ref_kernel_ref_setup(int natoms, const float *xq, const int *type, int ntype,
                                const float *nbfp, const float *shift_vec,
                                int nsci, const nbnxn_sci_t *sci, const nbnxn_cj4_t *cj4,
                                const nbnxn_excl_t *excl, const ref_params_t *p, int nlist)
{
    ref_ctx_t *c = calloc(1, sizeof(*c));
    int        a, l, i0;
    c->natoms = natoms; c->nlist = nlist;
    make_simple_list(nsci, sci, cj4, excl, &c->sl);
    c->x3 = malloc((size_t)natoms * 3 * sizeof(float));
    c->q  = malloc((size_t)natoms * sizeof(float));
    c->type = malloc((size_t)natoms * sizeof(int));
    c->shift_vec = malloc(SHIFTS * 3 * sizeof(float));
    memcpy(c->shift_vec, shift_vec, SHIFTS * 3 * sizeof(float));
    memcpy(c->type, type, (size_t)natoms * sizeof(int));
    for (a = 0; a < natoms; a++)
    {
        c->x3[3 * a] = xq[4 * a]; c->x3[3 * a + 1] = xq[4 * a + 1]; c->x3[3 * a + 2] = xq[4 * a + 2];
        c->q[a] = xq[4 * a + 3];
    }
    c->nbl  = calloc(nlist, sizeof(*c->nbl));
    c->nblp = calloc(nlist, sizeof(*c->nblp));
    i0 = 0;
    for (l = 0; l < nlist; l++)
    {
        const long long target = (long long)c->sl.ncj * (l + 1) / nlist;
        int             i1 = i0;
        while (i1 < c->sl.nci && (l == nlist - 1 || c->sl.ci[i1].cj_ind_end <= target)) { i1++; }
        c->nbl[l].bSimple = TRUE;
        c->nbl[l].na_ci = 4; c->nbl[l].na_cj = 4; c->nbl[l].na_sc = 4;
        c->nbl[l].rlist = p->rlist;
        c->nbl[l].nci = i1 - i0; c->nbl[l].ci = c->sl.ci + i0;
        c->nbl[l].ncj = c->sl.ncj; c->nbl[l].cj = c->sl.cj;   /* ci entries index the shared cj array */
        c->nblp[l] = &c->nbl[l];
        i0 = i1;
    }
    c->set.nnbl = nlist; c->set.nbl = c->nblp; c->set.bCombined = FALSE; c->set.bSimple = TRUE;

    memset(&c->nbat, 0, sizeof(c->nbat));
    c->nbat.ntype = ntype; c->nbat.nbfp = malloc((size_t)ntype * ntype * 2 * sizeof(float));
    memcpy(c->nbat.nbfp, nbfp, (size_t)ntype * ntype * 2 * sizeof(float));
    /* nbat->nbfp_comb as nbnxn_atomdata.c:397-429 fills it (that unit is unbuildable here, see header) */
    c->nbat.comb_rule = p->ljpme == 1 ? ljcrGEOM : (p->ljpme == 2 ? ljcrLB : ljcrNONE);
    c->nbat.nbfp_comb = calloc((size_t)ntype * 2, sizeof(float));
    for (a = 0; a < ntype && p->ljpme; a++)
    {
        const double c6 = nbfp[(a * ntype + a) * 2], c12 = nbfp[(a * ntype + a) * 2 + 1];
        if (p->ljpme == 1)
        {
            c->nbat.nbfp_comb[a * 2] = sqrt(c6); c->nbat.nbfp_comb[a * 2 + 1] = sqrt(c12);
        }
        else if (c6 > 0 && c12 > 0)
        {
            c->nbat.nbfp_comb[a * 2] = 0.5 * pow(c12 / c6, 1.0 / 6.0); c->nbat.nbfp_comb[a * 2 + 1] = sqrt(c6 * c6 / c12);
        }
    }
    c->nbat.natoms = natoms; c->nbat.natoms_local = natoms;
    c->nbat.type = c->type;
    c->nbat.XFormat = nbatXYZ; c->nbat.FFormat = nbatXYZ;
    c->nbat.q = c->q;
    c->nbat.na_c = 4; c->nbat.nenergrp = 1;
    c->nbat.xstride = 3; c->nbat.fstride = 3;
    c->nbat.x = c->x3;
    c->nbat.shift_vec = (rvec *)c->shift_vec;
    c->nbat.nout = nlist;
    c->nbat.out = calloc(nlist, sizeof(*c->nbat.out));
    for (l = 0; l < nlist; l++)
    {
        c->nbat.out[l].f = calloc((size_t)natoms * 3, sizeof(float));
        c->nbat.out[l].fshift = calloc(SHIFTS * 3, sizeof(float));
        c->nbat.out[l].nV = 1;
        c->nbat.out[l].Vvdw = calloc(1, sizeof(float));
        c->nbat.out[l].Vc = calloc(1, sizeof(float));
    }
    make_ic(p, &c->ic);
    return c;
}

int ref_kernel_ref_nci(const ref_ctx_t *c) { return c->sl.nci; }
int ref_kernel_ref_ncj(const ref_ctx_t *c) { return c->sl.ncj; }

/* One call of nbnxn_kernel_ref (mdlib/nbnxn_kernels/nbnxn_kernel_ref.c:169) with nthreads
 * OpenMP threads over the nlist lists.  Outputs may be NULL (timing runs). */
int ref_kernel_ref_run(ref_ctx_t *c, int calc_ener, int nthreads,
                       float *f, float *fshift, float *Vc, float *Vvdw)
{
    int   flags = GMX_FORCE_FORCES | GMX_FORCE_VIRIAL | (calc_ener ? GMX_FORCE_ENERGY : 0);
    float fsh[SHIFTS * 3], vc = 0, vv = 0;
    int   l, k;
    size_t a;
    g_nthreads = nthreads;
    memset(fsh, 0, sizeof(fsh));
    nbnxn_kernel_ref(&c->set, &c->nbat, &c->ic, (rvec *)c->shift_vec, flags, enbvClearFYes, fsh, &vc, &vv);
    if (f)
    {
        memset(f, 0, (size_t)c->natoms * 3 * sizeof(float));
        for (l = 0; l < c->nlist; l++)
        {
            for (a = 0; a < (size_t)c->natoms * 3; a++) { f[a] += c->nbat.out[l].f[a]; }
        }
    }
    if (fshift)
    {
        /* with one list the kernel writes the caller's array, otherwise out[l].fshift
         * (kernel_ref.c:257-268) */
        for (k = 0; k < SHIFTS * 3; k++) { fshift[k] = fsh[k]; }
        if (c->nlist > 1)
        {
            for (l = 0; l < c->nlist; l++)
            {
                for (k = 0; k < SHIFTS * 3; k++) { fshift[k] += c->nbat.out[l].fshift[k]; }
            }
        }
    }
    if (Vc) { *Vc = vc; }
    if (Vvdw) { *Vvdw = vv; }
    return 0;
}

void ref_kernel_ref_free(ref_ctx_t *c)
{
    int l;
    for (l = 0; l < c->nlist; l++)
    {
        free(c->nbat.out[l].f); free(c->nbat.out[l].fshift); free(c->nbat.out[l].Vvdw); free(c->nbat.out[l].Vc);
    }
    free(c->nbat.out); free(c->nbat.nbfp); free(c->nbat.nbfp_comb);
    free(c->sl.ci); free(c->sl.cj);
    free(c->nbl); free(c->nblp);
    free(c->x3); free(c->q); free(c->type); free(c->shift_vec);
    free_ic(&c->ic);
    free(c);
}
