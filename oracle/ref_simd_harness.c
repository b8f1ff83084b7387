/*
 * ref_simd_harness.c -- drives the REFERENCE's SIMD 4xN OpenMP kernel (AVX2_256: 4x8) on flat arrays.
 * TEST INFRASTRUCTURE ONLY, and the TIMING leg only: this is the CPU yardstick bench.py quotes as
 * cpu_baseline ("the reference's SIMD 4xN OpenMP CPU kernel timed in the same run").  It pins nothing: parity is
 * pinned by the plain-C kernels of oracle/_ref/libnbnxn_ref.so (ref_harness.c); a test only checks that what is
 * timed here computes the same forces (tests/test_oracle_golden.py).
 *
 * Built into oracle/_ref/libnbnxn_ref_simd.so by oracle/Makefile, only where /root/reference exists.
 * Reference code in that library, compiled unmodified from where it lies:
 *   mdlib/nbnxn_kernels/simd_4xn/nbnxn_kernel_simd_4xn.c and the 95 generated nbnxn_kernel_*_4xn.c units beside it
 *   mdlib/nbnxn_kernels/nbnxn_kernel_common.c
 * The SIMD selection comes from the compiler command line (-mavx2 -mfma -DGMX_SIMD_X86_AVX2_256 -DGMX_TARGET_X86
 * -Dgmx_simdcall= : the definitions cmake writes into config.h for this CPU, src/config.h.cmakein:93,114,129); no
 * header is stood in for.
 * Not reference code: the four host services of ref_harness.c (error exit, debug off, thread count) and the
 * fillers below for what nbnxn_atomdata.c / nbnxn_search.c produce (both unbuildable here: they need the generated
 * gmx_header_config_gen.h): the X8 atom layout (nbnxn_atomdata.c:734-738, :834-852), the exclusion filters
 * (:441-492), nbfp_s4 (:374-389), the combination-rule detection (:578-697, geometric or none) and the simple 4x8
 * list, converted from the GPU list so that both sides see the same atom pairs.
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
#include "nbnxn_kernels/simd_4xn/nbnxn_kernel_simd_4xn.h"

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

/* the subset of ref_harness.c's parameter block this leg needs (same layout: the Python side passes one struct) */
typedef struct {
    int   el_type;      /* 0 RF, 1 Ewald */
    float rcoulomb, rvdw, rlist;
    float epsfac, k_rf, c_rf;
    float ewaldcoeff_q, sh_ewald;
    float disp_cpot, rep_cpot;
    int   vdw_modifier;
    float rvdw_switch;
    float disp_c2, disp_c3, rep_c2, rep_c3;
    float sw_c3, sw_c4, sw_c5;
    int   ljpme;
    float ewaldcoeff_lj, sh_lj_ewald;
} ref_params_t;

typedef struct { int ci, shift, cj; unsigned excl; } pair48_t;

static int cmp_pair48(const void *a, const void *b)
{
    const pair48_t *x = a, *y = b;
    int             fx, fy, dx, dy;
    if (x->ci != y->ci) { return x->ci < y->ci ? -1 : 1; }
    if (x->shift != y->shift) { return x->shift < y->shift ? -1 : 1; }
    /* entries with exclusion bits first (the kernel's masked loop runs while excl != all ones,
     * nbnxn_kernel_simd_4xn_outer.h:628), the diagonal first of all (self term test, :464-470), then ascending cj */
    fx = x->excl == 0xffffffffu; fy = y->excl == 0xffffffffu;
    if (fx != fy) { return fx - fy; }
    dx = !(x->shift == CENTRAL && x->cj == (x->ci >> 1)); dy = !(y->shift == CENTRAL && y->cj == (y->ci >> 1));
    if (dx != dy) { return dx - dy; }
    return x->cj < y->cj ? -1 : (x->cj > y->cj);
}

typedef struct {
    int                  natoms, nlist, nci, ncj;
    nbnxn_ci_t          *ci;
    nbnxn_cj_t          *cj;
    nbnxn_pairlist_t    *nbl;
    nbnxn_pairlist_t   **nblp;
    nbnxn_pairlist_set_t set;
    nbnxn_atomdata_t     nbat;
    interaction_const_t  ic;
    float               *shift_vec;
} simd_ctx_t;

static void *amalloc(size_t n)
{
    void *p = NULL;
    if (posix_memalign(&p, 128, n ? n : 128) != 0) { abort(); }
    memset(p, 0, n);
    return p;
}

/* GPU list -> simple 4x8 list: every 8-atom i-cluster is two 4-atom i-clusters, j-clusters stay 8 atoms;
 * exclusion bit of (i, j) is bit i * 8 + j (nbnxn_search.c: simple lists, na_cj_2log = 3) */
static void make_list48(simd_ctx_t *c, int nsci, const nbnxn_sci_t *sci, const nbnxn_cj4_t *cj4, const nbnxn_excl_t *excl)
{
    size_t    n = 0, cap = 1 << 20, k;
    pair48_t *pr = malloc(cap * sizeof(*pr));
    int       s, j4, jm, im, hi, i, j;
    for (s = 0; s < nsci; s++)
    {
        for (j4 = sci[s].cj4_ind_start; j4 < sci[s].cj4_ind_end; j4++)
        {
            for (jm = 0; jm < 4; jm++)
            {
                for (im = 0; im < 8; im++)
                {
                    const int ci8 = sci[s].sci * 8 + im, cj8 = cj4[j4].cj[jm];
                    if (!(((cj4[j4].imei[0].imask | cj4[j4].imei[1].imask) >> (jm * 8 + im)) & 1u)) { continue; }
                    for (hi = 0; hi < 2; hi++)
                    {
                        unsigned m = 0;
                        for (i = 0; i < 4; i++)
                        {
                            for (j = 0; j < 8; j++)
                            {
                                const int      ic = hi * 4 + i;
                                const unsigned w  = excl[cj4[j4].imei[j >> 2].excl_ind].pair[(j & 3) * 8 + ic];
                                unsigned       b  = (w >> (jm * 8 + im)) & 1u;
                                /* the j <= i half of a diagonal pair is excluded in the list as well
                                 * (set_self_and_newton_excls, nbnxn_search.c:2716-2757) */
                                if (sci[s].shift == CENTRAL && ci8 == cj8 && j <= ic) { b = 0; }
                                m |= b << (i * 8 + j);
                            }
                        }
                        if (n == cap) { cap *= 2; pr = realloc(pr, cap * sizeof(*pr)); }
                        pr[n].ci = ci8 * 2 + hi; pr[n].shift = sci[s].shift; pr[n].cj = cj8; pr[n].excl = m;
                        n++;
                    }
                }
            }
        }
    }
    qsort(pr, n, sizeof(*pr), cmp_pair48);
    c->cj  = amalloc((n + 1) * sizeof(*c->cj));
    c->ci  = amalloc((n + 1) * sizeof(*c->ci));
    c->nci = 0; c->ncj = (int)n;
    for (k = 0; k < n; k++)
    {
        c->cj[k].cj = pr[k].cj; c->cj[k].excl = pr[k].excl;
        if (k == 0 || pr[k].ci != pr[k - 1].ci || pr[k].shift != pr[k - 1].shift)
        {
            if (c->nci > 0) { c->ci[c->nci - 1].cj_ind_end = (int)k; }
            c->ci[c->nci].ci = pr[k].ci;
            c->ci[c->nci].shift = pr[k].shift | NBNXN_CI_DO_LJ(0) | NBNXN_CI_DO_COUL(0);
            c->ci[c->nci].cj_ind_start = (int)k;
            c->nci++;
        }
    }
    if (c->nci > 0) { c->ci[c->nci - 1].cj_ind_end = (int)n; }
    free(pr);
}

static int within_tol(double a, double b, double tol) { return fabs(a - b) <= tol * 0.5 * (fabs(a) + fabs(b)); }

simd_ctx_t *ref_simd_setup(int natoms, const float *xq, const int *type, int ntype, const float *nbfp, const float *shift_vec,
                           int nsci, const nbnxn_sci_t *sci, const nbnxn_cj4_t *cj4, const nbnxn_excl_t *excl,
                           const ref_params_t *p, int nlist)
{
    simd_ctx_t       *c = calloc(1, sizeof(*c));
    nbnxn_atomdata_t *nbat = &c->nbat;
    interaction_const_t *ic = &c->ic;
    int               a, l, i0, i, j, geom = 1;
    if (natoms % 8 != 0 || p->ljpme || nlist < 1 || nlist > NBNXN_BUFFERFLAG_MAX_THREADS) { free(c); return NULL; }
    c->natoms = natoms; c->nlist = nlist;
    make_list48(c, nsci, sci, cj4, excl);
    c->shift_vec = amalloc(SHIFTS * 3 * sizeof(float));
    memcpy(c->shift_vec, shift_vec, SHIFTS * 3 * sizeof(float));

    c->nbl  = calloc(nlist, sizeof(*c->nbl));
    c->nblp = calloc(nlist, sizeof(*c->nblp));
    i0 = 0;
    for (l = 0; l < nlist; l++)
    {
        const long long target = (long long)c->ncj * (l + 1) / nlist;
        int             i1 = i0;
        while (i1 < c->nci && (l == nlist - 1 || c->ci[i1].cj_ind_end <= target)) { i1++; }
        c->nbl[l].bSimple = TRUE;
        c->nbl[l].na_ci = 4; c->nbl[l].na_cj = 8; c->nbl[l].na_sc = 4;
        c->nbl[l].rlist = p->rlist;
        c->nbl[l].nci = i1 - i0; c->nbl[l].ci = c->ci + i0;
        c->nbl[l].ncj = c->ncj; c->nbl[l].cj = c->cj;
        c->nblp[l] = &c->nbl[l];
        i0 = i1;
    }
    c->set.nnbl = nlist; c->set.nbl = c->nblp; c->set.bCombined = FALSE; c->set.bSimple = TRUE;

    memset(nbat, 0, sizeof(*nbat));
    nbat->ntype = ntype;
    nbat->nbfp  = amalloc((size_t)ntype * ntype * 2 * sizeof(float));
    memcpy(nbat->nbfp, nbfp, (size_t)ntype * ntype * 2 * sizeof(float));
    nbat->nbfp_s4 = amalloc((size_t)ntype * ntype * 4 * sizeof(float));
    for (i = 0; i < ntype * ntype; i++) { nbat->nbfp_s4[4 * i] = nbfp[2 * i]; nbat->nbfp_s4[4 * i + 1] = nbfp[2 * i + 1]; }
    /* combination-rule detection (nbnxn_atomdata.c:611-660): geometric, or none */
    for (i = 0; i < ntype; i++)
    {
        for (j = 0; j < ntype; j++)
        {
            const double c6 = nbfp[(i * ntype + j) * 2], c12 = nbfp[(i * ntype + j) * 2 + 1];
            geom = geom && within_tol(c6 * c6, (double)nbfp[(i * ntype + i) * 2] * nbfp[(j * ntype + j) * 2], 1e-5)
                && within_tol(c12 * c12, (double)nbfp[(i * ntype + i) * 2 + 1] * nbfp[(j * ntype + j) * 2 + 1], 1e-5);
        }
    }
    nbat->comb_rule = geom ? ljcrGEOM : ljcrNONE;
    nbat->nbfp_comb = amalloc((size_t)ntype * 2 * sizeof(float));
    for (i = 0; i < ntype && geom; i++)
    {
        nbat->nbfp_comb[2 * i] = sqrt(nbfp[(i * ntype + i) * 2]); nbat->nbfp_comb[2 * i + 1] = sqrt(nbfp[(i * ntype + i) * 2 + 1]);
    }
    nbat->natoms = natoms; nbat->natoms_local = natoms;
    nbat->type = amalloc((size_t)natoms * sizeof(int));
    memcpy(nbat->type, type, (size_t)natoms * sizeof(int));
    nbat->q = amalloc((size_t)natoms * sizeof(float));
    nbat->x = amalloc((size_t)natoms * 3 * sizeof(float));
    nbat->lj_comb = amalloc((size_t)natoms * 2 * sizeof(float));
    for (a = 0; a < natoms; a++)
    {
        const int b = (a >> 3) * 24 + (a & 7);
        nbat->x[b] = xq[4 * a]; nbat->x[b + 8] = xq[4 * a + 1]; nbat->x[b + 16] = xq[4 * a + 2];
        nbat->q[a] = xq[4 * a + 3];
        nbat->lj_comb[(a >> 3) * 16 + (a & 7)]     = nbat->nbfp_comb[type[a] * 2];
        nbat->lj_comb[(a >> 3) * 16 + 8 + (a & 7)] = nbat->nbfp_comb[type[a] * 2 + 1];
    }
    nbat->XFormat = nbatX8; nbat->FFormat = nbatX8;
    nbat->na_c = 4; nbat->nenergrp = 1;
    nbat->xstride = 3; nbat->fstride = 3;
    nbat->shift_vec = (rvec *)c->shift_vec;
    nbat->simd_4xn_diagonal_j_minus_i = amalloc(8 * sizeof(float));
    for (j = 0; j < 8; j++) { nbat->simd_4xn_diagonal_j_minus_i[j] = j - 0.5f; }
    nbat->simd_exclusion_filter1 = amalloc(32 * sizeof(unsigned));
    nbat->simd_exclusion_filter2 = amalloc(64 * sizeof(unsigned));
    for (j = 0; j < 32; j++)
    {
        nbat->simd_exclusion_filter1[j] = 1u << j;
        nbat->simd_exclusion_filter2[2 * j] = 1u << j; nbat->simd_exclusion_filter2[2 * j + 1] = 1u << j;
    }
    nbat->nout = nlist;
    nbat->out  = calloc(nlist, sizeof(*nbat->out));
    for (l = 0; l < nlist; l++)
    {
        nbat->out[l].f      = amalloc((size_t)natoms * 3 * sizeof(float));
        nbat->out[l].fshift = amalloc(SHIFTS * 3 * sizeof(float));
        nbat->out[l].nV     = 1;
        nbat->out[l].Vvdw   = amalloc(sizeof(float));
        nbat->out[l].Vc     = amalloc(sizeof(float));
    }

    memset(ic, 0, sizeof(*ic));
    ic->vdwtype = evdwCUT; ic->vdw_modifier = p->vdw_modifier;
    ic->rvdw = p->rvdw; ic->rvdw_switch = p->rvdw_switch;
    ic->dispersion_shift.c2 = p->disp_c2; ic->dispersion_shift.c3 = p->disp_c3; ic->dispersion_shift.cpot = p->disp_cpot;
    ic->repulsion_shift.c2 = p->rep_c2; ic->repulsion_shift.c3 = p->rep_c3; ic->repulsion_shift.cpot = p->rep_cpot;
    ic->vdw_switch.c3 = p->sw_c3; ic->vdw_switch.c4 = p->sw_c4; ic->vdw_switch.c5 = p->sw_c5;
    ic->sh_invrc6 = -p->disp_cpot;
    ic->eeltype = p->el_type == 0 ? eelRF : eelPME;
    ic->coulomb_modifier = eintmodPOTSHIFT;
    ic->rcoulomb = p->rcoulomb; ic->rlist = p->rlist; ic->rlistlong = p->rlist;
    ic->ewaldcoeff_q = p->ewaldcoeff_q; ic->sh_ewald = p->sh_ewald;
    ic->epsilon_r = 1; ic->epsfac = p->epsfac; ic->k_rf = p->k_rf; ic->c_rf = p->c_rf;
    return c;
}

int ref_simd_nci(const simd_ctx_t *c) { return c->nci; }
int ref_simd_ncj(const simd_ctx_t *c) { return c->ncj; }
int ref_simd_comb_rule(const simd_ctx_t *c) { return c->nbat.comb_rule; }

/* One call of nbnxn_kernel_simd_4xn (simd_4xn/nbnxn_kernel_simd_4xn.c:254) with nthreads OpenMP threads over the
 * nlist lists, analytical Ewald as forcerec.c:1656-1660 chooses for a SIMD width of 8.  Outputs may be NULL. */
int ref_simd_run(simd_ctx_t *c, int calc_ener, int nthreads, float *f, float *fshift, float *Vc, float *Vvdw)
{
    int   flags = GMX_FORCE_FORCES | GMX_FORCE_VIRIAL | (calc_ener ? GMX_FORCE_ENERGY : 0);
    float fsh[SHIFTS * 3], vc = 0, vv = 0;
    int   l, k, a;
    g_nthreads = nthreads;
    memset(fsh, 0, sizeof(fsh));
    nbnxn_kernel_simd_4xn(&c->set, &c->nbat, &c->ic, ewaldexclAnalytical, (rvec *)c->shift_vec, flags, enbvClearFYes, fsh, &vc, &vv);
    if (f)
    {
        memset(f, 0, (size_t)c->natoms * 3 * sizeof(float));
        for (l = 0; l < c->nlist; l++)
        {
            const float *fo = c->nbat.out[l].f;
            for (a = 0; a < c->natoms; a++)
            {
                const int b = (a >> 3) * 24 + (a & 7);
                f[3 * a] += fo[b]; f[3 * a + 1] += fo[b + 8]; f[3 * a + 2] += fo[b + 16];
            }
        }
    }
    if (fshift)
    {
        for (k = 0; k < SHIFTS * 3; k++) { fshift[k] = fsh[k]; }
        if (c->nlist > 1)
        {
            for (l = 0; l < c->nlist; l++) { for (k = 0; k < SHIFTS * 3; k++) { fshift[k] += c->nbat.out[l].fshift[k]; } }
        }
    }
    if (Vc) { *Vc = vc; }
    if (Vvdw) { *Vvdw = vv; }
    return 0;
}

void ref_simd_free(simd_ctx_t *c)
{
    int l;
    nbnxn_atomdata_t *nbat = &c->nbat;
    for (l = 0; l < c->nlist; l++) { free(nbat->out[l].f); free(nbat->out[l].fshift); free(nbat->out[l].Vvdw); free(nbat->out[l].Vc); }
    free(nbat->out); free(nbat->nbfp); free(nbat->nbfp_s4); free(nbat->nbfp_comb); free(nbat->type); free(nbat->q); free(nbat->x);
    free(nbat->lj_comb); free(nbat->simd_4xn_diagonal_j_minus_i); free(nbat->simd_exclusion_filter1); free(nbat->simd_exclusion_filter2);
    free(c->ci); free(c->cj); free(c->nbl); free(c->nblp); free(c->shift_vec);
    free(c);
}
