/*
 * nbnxn_oracle.c -- CPU restatement of the reference's Verlet cluster-pair nonbonded
 * kernel.  TEST INFRASTRUCTURE ONLY: nothing in the product path (gromacs-cosmo_amd/,
 * include/) may call, link or load this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * What it restates (paths under /root/reference/src/gromacs/mdlib/):
 *   - the loop structure and list semantics of nbnxn_kernels/nbnxn_kernel_gpu_ref.c:54-376
 *     (super-cluster i-entries, cj4 groups, imask bits, per-half exclusion words,
 *     diagonal rule ja <= ia, r^2 += (1-int_bit)*1e-12, shift forces, self terms);
 *   - the pair arithmetic of nbnxn_kernels/nbnxn_kernel_ref_inner.h:117-372 for
 *     reaction-field + LJ with potential shift, and for Ewald real space in ANALYTICAL
 *     form (erf/exp from libm) instead of the reference's spline table
 *     (nbnxn_kernel_ref_inner.h:304-330; the table is accurate to float epsilon by
 *     construction, tables.c:324-326, so both agree at the 1e-6 level);
 *   - the CUDA kernel's conventions where the C kernels differ
 *     (nbnxn_cuda/nbnxn_cuda_kernel.cuh:222-256 self terms, :346-347 cut-off + diagonal,
 *     :461-471 energies).
 * All pair arithmetic and accumulation is in double from the float inputs, so it is the
 * arbiter when the reference's own float kernels disagree at the 1e-5 level (SURVEY 7 iv-b).
 *
 * Pinning: the reference ships no golden vectors for this path.  The oracle
 * is pinned against the reference ITSELF run in the build container: oracle/_ref is the
 * reference's nbnxn_kernel_ref.c + nbnxn_kernel_gpu_ref.c compiled from /root/reference
 * (oracle/Makefile), tests/golden/ holds its outputs, tests/test_oracle_golden.py checks
 * this file against them.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "nbnxn_hip.h"

#define CL  NBX_CLUSTER_SIZE
#define NSC NBX_NCL_PER_SUPERCL

typedef struct {
    int    el_type;        /* NBX_EL_RF or NBX_EL_EWALD_ANA */
    double rcoulomb, rvdw;
    double epsfac;
    double k_rf, c_rf;
    double beta, sh_ewald;
    double disp_cpot, rep_cpot;
    /* LJ modifier: enum eintmod (types/enums.h:92-93): 1 potential shift, 3 potential switch,
     * 5 force switch; constants of forcerec.c:1942-1975 */
    int    vdw_modifier;
    double rvdw_switch;
    double disp_c2, disp_c3, rep_c2, rep_c3;
    double sw_c3, sw_c4, sw_c5;
    /* LJ-PME (ic->vdwtype == evdwPME): ljpme 0 off, 1 geometric, 2 Lorentz-Berthelot grid;
     * nbfp_comb as nbat->nbfp_comb (nbnxn_atomdata.c:397-429) */
    int          ljpme;
    double       ewaldcoeff_lj, sh_lj_ewald;
    const float *nbfp_comb;
    /* tabulated Ewald force (el_type NBX_EL_EWALD_TAB): the GPU flavour's table
     * (nbnxn_cuda_data_mgmt.cu:120-183), linear interpolation (nbnxn_cuda_kernel_utils.cuh:304-317) */
    const float *coul_tab;
    int          tab_size;
    double       tab_scale;
} oracle_params_t;

/* grid C6 of a type pair (nbnxn_kernel_ref_inner.h:211-226) */
static double c6grid_of(const oracle_params_t *p, int ti, int tj)
{
    const float *ljc = p->nbfp_comb;
    if (p->ljpme == 1) { return (double)ljc[ti * 2] * ljc[tj * 2]; }
    {
        const double sigma = (double)ljc[ti * 2] + ljc[tj * 2], eps = (double)ljc[ti * 2 + 1] * ljc[tj * 2 + 1];
        const double s2 = sigma * sigma;
        return eps * s2 * s2 * s2;
    }
}

/* Ewald long-range force table as table_spline3_fill_ewald_lr makes it (tables.c:165-311),
 * force column only, with the float roundings of the reference's real-typed table. */
void oracle_ewald_force_table(int ntab, double dx, double beta, float *tab_f)
{
    int    i;
    double dc;
    for (i = ntab - 1; i >= 1; i--)
    {
        const double x0 = i * dx;
        const double v0 = erf(beta * x0) / x0;
        const double x1 = (i - 1) * dx;
        const double v1 = i == 1 ? beta * 2 / sqrt(M_PI) : erf(beta * x1) / x1;
        const double vm = erf(beta * (x0 - 0.5 * dx)) / (x0 - 0.5 * dx);
        double       a2dx = (v0 + v1 - 2 * vm) / (0.25 * dx);          /* :247 */
        dc = (v0 - v1) / dx + 0.5 * a2dx;                             /* :252 */
        if (i == ntab - 1) { tab_f[i] = (float)(-dc); }               /* :255-264 */
        else { tab_f[i] = (float)(tab_f[i] + -0.5 * dc); }
        a2dx = (dc * dx + v1 - v0) * 2 / dx;                          /* :272-277 */
        dc   = dc - a2dx;
        tab_f[i - 1] = (float)(-0.5 * dc);                            /* :289 */
    }
    tab_f[0] *= 2;                                                    /* :292 */
}

/* Pair term shared by the list kernel and the brute-force sum.
 * dx = xi - xj (i already shifted).  int_bit = 1: interacting pair, 0: excluded pair that
 * still gets the reaction-field / Ewald-mesh correction (the reference's EXCL_FORCES).
 * Returns the scalar F/r; adds the pair energies when vc/vv are non-NULL. */
static double pair_term(const oracle_params_t *p, double rsq, int int_bit,
                        double qq /* epsfac*qi*qj */, double c6, double c12, double c6grid,
                        double *vc, double *vv)
{
    double rinv, rinvsq, rinvsix, fscal, r;

    rsq   += (1.0 - int_bit) * (double)NBX_AVOID_SING_R2_INC;   /* kernel_gpu_ref.c:257 */
    rinv   = 1.0 / sqrt(rsq);
    rinvsq = rinv * rinv;
    r      = rsq * rinv;

    if (p->el_type == NBX_EL_RF)
    {
        /* kernel_gpu_ref.c:264-273, kernel_ref_inner.h:296-299 */
        fscal = qq * (int_bit * rinv * rinvsq - 2.0 * p->k_rf);
        if (vc) { *vc += qq * (int_bit * rinv + p->k_rf * rsq - p->c_rf); }
    }
    else if (p->el_type == NBX_EL_EWALD_TAB)
    {
        /* nbnxn_cuda_kernel.cuh:451-458 (force from the table), :468-471 (analytical energy) */
        const double norm = p->tab_scale * r;
        const int    idx  = (int)norm;
        const double fr2  = norm - idx, fr1 = 1.0 - fr2;
        const double br   = p->beta * r;
        if (idx + 1 >= p->tab_size) { abort(); }
        fscal = qq * (int_bit * rinvsq - (fr1 * p->coul_tab[idx] + fr2 * p->coul_tab[idx + 1])) * rinv;
        if (vc) { *vc += qq * ((int_bit - erf(br)) * rinv - int_bit * p->sh_ewald); }
    }
    else
    {
        /* analytical form of kernel_gpu_ref.c:276-289:
         * F/r = qq*(int_bit/r^3 - [erf(br)/r - 2b/sqrt(pi) exp(-b^2 r^2)]/r^2) */
        const double br = p->beta * r;
        const double fexcl_r = (erf(br) * rinv - 2.0 * p->beta * M_2_SQRTPI * 0.5 * exp(-br * br)) * rinvsq;
        fscal = qq * (int_bit * rinv * rinvsq - fexcl_r);
        if (vc) { *vc += qq * ((int_bit - erf(br)) * rinv - int_bit * p->sh_ewald); }
    }

    if (rsq < p->rvdw * p->rvdw)
    {
        /* kernel_ref_inner.h:157-205 (kernel_gpu_ref.c:291-312 for the plain cut-off) */
        double frLJ, VLJ, rsw = 0;
        rinvsix = int_bit * rinvsq * rinvsq * rinvsq;
        const double frLJ6  = c6 * rinvsix;
        const double frLJ12 = c12 * rinvsix * rinvsix;
        frLJ = frLJ12 - frLJ6;                       /* force times r */
        VLJ  = (frLJ12 + c12 * p->rep_cpot) / 12.0 - (frLJ6 + c6 * p->disp_cpot) / 6.0;
        if (p->vdw_modifier == 5 || p->vdw_modifier == 3)
        {
            rsw = r - p->rvdw_switch;
            rsw = rsw >= 0.0 ? rsw : 0.0;
        }
        if (p->vdw_modifier == 5)
        {
            /* force switch :176-185; like the reference it is NOT masked by int_bit */
            frLJ += -c6 * (p->disp_c2 + p->disp_c3 * rsw) * rsw * rsw * r
                + c12 * (p->rep_c2 + p->rep_c3 * rsw) * rsw * rsw * r;
            VLJ  += -c6 * (-p->disp_c2 / 3 - p->disp_c3 / 4 * rsw) * rsw * rsw * rsw
                + c12 * (-p->rep_c2 / 3 - p->rep_c3 / 4 * rsw) * rsw * rsw * rsw;
        }
        VLJ *= int_bit;                               /* :187-193 */
        if (p->vdw_modifier == 3)
        {
            /* potential switch :195-205 */
            const double sw  = 1.0 + (p->sw_c3 + (p->sw_c4 + p->sw_c5 * rsw) * rsw) * rsw * rsw * rsw;
            const double dsw = (3 * p->sw_c3 + (4 * p->sw_c4 + 5 * p->sw_c5 * rsw) * rsw) * rsw * rsw;
            frLJ = frLJ * sw - r * VLJ * dsw;
            VLJ *= sw;
        }
        if (p->ljpme)
        {
            /* kernel_ref_inner.h:207-251: mesh part of the dispersion, not masked by int_bit */
            const double rinvsix_nm = rinvsq * rinvsq * rinvsq;
            const double lje2 = p->ewaldcoeff_lj * p->ewaldcoeff_lj;
            const double cr2 = lje2 * rsq, expmcr2 = exp(-cr2), poly = 1 + cr2 + 0.5 * cr2 * cr2;
            if (cr2 < 1e-2)
            {
                /* 1 - exp(-x)(1 + x + x^2/2) = sum_k (-1)^k x^(k+3) / (2 k! (k+3)): the reference's expression
                 * cancels to nothing as r -> 0 (excluded atoms on one point, r^2 = 1e-12) and r^-6 then
                 * multiplies the rounding noise; the arbiter takes the series there */
                const double x = cr2;
                const double one_m = x * x * x * (1.0 / 6 + x * (-1.0 / 8 + x * (1.0 / 20 + x * (-1.0 / 72 + x * (1.0 / 336 - x / 1920)))));
                frLJ += c6grid * (rinvsix_nm * one_m - expmcr2 * lje2 * lje2 * lje2 / 6.0);
                VLJ  += c6grid / 6 * (rinvsix_nm * one_m + p->sh_lj_ewald * int_bit);
            }
            else
            {
                frLJ += c6grid * (rinvsix_nm - expmcr2 * (rinvsix_nm * poly + lje2 * lje2 * lje2 / 6.0));
                VLJ  += c6grid / 6 * (rinvsix_nm * (1 - expmcr2 * poly) + p->sh_lj_ewald * int_bit);
            }
        }
        fscal += frLJ * rinvsq;
        if (vv) { *vv += VLJ; }
    }
    return fscal;
}

/* Cluster-pair kernel over the GPU list (sci / cj4 / excl).
 * f is ADDED to (caller clears); f[natoms*3], fshift[45*3] (CENTRAL included, as the C
 * reference kernels do: kernel_ref_outer.h:379-398), Vc/Vvdw scalars are added to.
 * calc_ener != 0 also accumulates the energies including the self terms.
 * pair_count (may be NULL) receives: [0] pair slots of set imask bits, [1] slots of
 * cluster pairs with an atom pair inside rlist, [2] pairs inside the cut-off that are not
 * fillers (the three counts of kernel_gpu_ref.c:362-375). */
int oracle_kernel_gpu_list(int natoms, const float *xq, const int *type, int ntype,
                           const float *nbfp, const float *shift_vec,
                           int nsci, const nbx_sci_t *sci, const nbx_cj4_t *cj4,
                           const nbx_excl_t *excl, double rlist,
                           const oracle_params_t *p, int calc_ener,
                           double *f, double *fshift, double *Vc, double *Vvdw,
                           long long *pair_count)
{
    const double rc2 = p->rcoulomb * p->rcoulomb;
    const double rl2 = rlist * rlist;
    long long    n_slots = 0, n_pruned = 0, n_in = 0;
    int          n, j4, jm, im, ic, jc;

    for (n = 0; n < nsci; n++)
    {
        const nbx_sci_t *e   = &sci[n];
        const int        s   = e->sci;
        const double     shx = shift_vec[3 * e->shift], shy = shift_vec[3 * e->shift + 1], shz = shift_vec[3 * e->shift + 2];
        double           vctot = 0, vvtot = 0;

        if (e->cj4_ind_end > e->cj4_ind_start && calc_ener &&
            e->shift == NBX_CENTRAL && cj4[e->cj4_ind_start].cj[0] == s * NSC)
        {
            /* self term, once per diagonal super-cluster (kernel_gpu_ref.c:156-181) */
            double q2 = 0;
            for (ic = 0; ic < NSC * CL; ic++)
            {
                const double qi = xq[(size_t)(s * NSC * CL + ic) * 4 + 3];
                q2 += qi * qi;
            }
            if (p->el_type == NBX_EL_RF)
            {
                vctot += -p->epsfac * 0.5 * p->c_rf * q2;
            }
            else
            {
                vctot += -p->epsfac * p->beta * (0.5 * M_2_SQRTPI) * q2;
            }
            if (p->ljpme)
            {
                /* LJ-PME self term (kernel_ref_outer.h:318-321, nbnxn_cuda_kernel.cuh:232-246) */
                const double l2 = p->ewaldcoeff_lj * p->ewaldcoeff_lj;
                for (ic = 0; ic < NSC * CL; ic++)
                {
                    const int t = type[s * NSC * CL + ic];
                    vvtot += 0.5 * nbfp[(size_t)t * (ntype + 1) * 2] / 6 * (l2 * l2 * l2 / 6.0);
                }
            }
        }

        for (j4 = e->cj4_ind_start; j4 < e->cj4_ind_end; j4++)
        {
            const nbx_cj4_t *g = &cj4[j4];
            for (jm = 0; jm < NBX_JGROUP_SIZE; jm++)
            {
                for (im = 0; im < NSC; im++)
                {
                    int within_rlist = 0;
                    /* nbnxn_kernel_gpu_ref.c:194-197 reads imei[0] alone (host-built lists carry the same mask twice,
                     * nbnxn_search.c:3108-3111); a list pruned per 32-lane half on a GPU (nbnxn_cuda_kernel.cuh:503) carries
                     * two, and a cluster pair is alive while either half keeps it: the union */
                    if (!(((g->imei[0].imask | g->imei[1].imask) >> (jm * NSC + im)) & 1u)) { continue; }
                    {
                        const int cj = g->cj[jm];
                        const int ci = s * NSC + im;
                        if (cj < 0 || (cj + 1) * CL > natoms || (ci + 1) * CL > natoms) { return -1; }
                        for (ic = 0; ic < CL; ic++)
                        {
                            const int    ia = ci * CL + ic;
                            const double ix = shx + xq[(size_t)ia * 4], iy = shy + xq[(size_t)ia * 4 + 1], iz = shz + xq[(size_t)ia * 4 + 2];
                            const double iq = p->epsfac * xq[(size_t)ia * 4 + 3];
                            double       fix = 0, fiy = 0, fiz = 0;
                            for (jc = 0; jc < CL; jc++)
                            {
                                const int ja = cj * CL + jc;
                                int       int_bit;
                                double    dx, dy, dz, rsq, fscal, c6, c12;
                                if (e->shift == NBX_CENTRAL && ci == cj && ja <= ia) { continue; }
                                /* half w = jc>>2 of the wave, word (jc&3)*8+ic (kernel_gpu_ref.c:231) */
                                int_bit = (excl[g->imei[jc >> 2].excl_ind].pair[(jc & 3) * CL + ic] >> (jm * NSC + im)) & 1u;
                                dx  = ix - xq[(size_t)ja * 4];
                                dy  = iy - xq[(size_t)ja * 4 + 1];
                                dz  = iz - xq[(size_t)ja * 4 + 2];
                                rsq = dx * dx + dy * dy + dz * dz;
                                if (rsq < rl2) { within_rlist = 1; }
                                if (rsq >= rc2) { continue; }
                                if (type[ia] != ntype - 1 && type[ja] != ntype - 1) { n_in++; }
                                c6    = nbfp[(type[ia] * ntype + type[ja]) * 2];
                                c12   = nbfp[(type[ia] * ntype + type[ja]) * 2 + 1];
                                fscal = pair_term(p, rsq, int_bit, iq * xq[(size_t)ja * 4 + 3], c6, c12,
                                                  p->ljpme ? c6grid_of(p, type[ia], type[ja]) : 0.0,
                                                  calc_ener ? &vctot : NULL, calc_ener ? &vvtot : NULL);
                                fix += fscal * dx; fiy += fscal * dy; fiz += fscal * dz;
                                f[(size_t)ja * 3]     -= fscal * dx;
                                f[(size_t)ja * 3 + 1] -= fscal * dy;
                                f[(size_t)ja * 3 + 2] -= fscal * dz;
                            }
                            f[(size_t)ia * 3]     += fix;
                            f[(size_t)ia * 3 + 1] += fiy;
                            f[(size_t)ia * 3 + 2] += fiz;
                            fshift[3 * e->shift]     += fix;
                            fshift[3 * e->shift + 1] += fiy;
                            fshift[3 * e->shift + 2] += fiz;
                        }
                    }
                    n_slots += CL * CL;
                    if (within_rlist) { n_pruned += CL * CL; }
                }
            }
        }
        if (calc_ener) { *Vc += vctot; *Vvdw += vvtot; }
    }
    if (pair_count) { pair_count[0] = n_slots; pair_count[1] = n_pruned; pair_count[2] = n_in; }
    return 0;
}

/* The pruning rule of the GPU kernels (nbnxn_cuda_kernel.cuh:332-340): a cluster pair
 * keeps its imask bit iff at least one of its 64 atom pairs has r^2 < rlist^2
 * (float arithmetic, same expression order as the kernel).  Writes both imei halves. */
int oracle_prune_list(const float *xq, const float *shift_vec, int nsci, const nbx_sci_t *sci,
                      nbx_cj4_t *cj4, float rlist)
{
    const float rl2 = rlist * rlist;
    int         n, j4, jm, im, ic, jc;
    for (n = 0; n < nsci; n++)
    {
        const nbx_sci_t *e = &sci[n];
        const float     *sh = shift_vec + 3 * e->shift;
        for (j4 = e->cj4_ind_start; j4 < e->cj4_ind_end; j4++)
        {
            unsigned imask = cj4[j4].imei[0].imask | cj4[j4].imei[1].imask;   /* the union, as the wave64 kernels walk it */
            for (jm = 0; jm < NBX_JGROUP_SIZE; jm++)
            {
                for (im = 0; im < NSC; im++)
                {
                    int any = 0;
                    if (!((imask >> (jm * NSC + im)) & 1u)) { continue; }
                    for (ic = 0; ic < CL && !any; ic++)
                    {
                        const float *xi = xq + (size_t)((e->sci * NSC + im) * CL + ic) * 4;
                        const float  ix = xi[0] + sh[0], iy = xi[1] + sh[1], iz = xi[2] + sh[2];
                        for (jc = 0; jc < CL; jc++)
                        {
                            const float *xj = xq + (size_t)(cj4[j4].cj[jm] * CL + jc) * 4;
                            const float  dx = ix - xj[0], dy = iy - xj[1], dz = iz - xj[2];
                            /* same evaluation order as the device code: fma chain */
                            const float  r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                            if (r2 < rl2) { any = 1; break; }
                        }
                    }
                    if (!any) { imask &= ~(1u << (jm * NSC + im)); }
                }
            }
            cj4[j4].imei[0].imask = imask;
            cj4[j4].imei[1].imask = imask;
        }
    }
    return 0;
}

/* Brute force over all atom pairs with the minimum-image convention:
 * the list-independent ground truth for small systems (validates list completeness and
 * exclusion masks, not only the kernel).  Atoms in caller order; excl as CSR; excluded
 * pairs inside the cut-off get the int_bit = 0 term, self pairs the self term.
 * box9: the GROMACS box matrix (rows are box vectors, lower triangular).  The nearest image is
 * searched over k box[XX] + l box[YY] + m box[ZZ] with |k| <= 2, |l|, |m| <= 1 -- the shift
 * vectors the reference's search can use (calc_shifts, gmxlib/pbc.c:1326-1349) -- which holds the
 * nearest image of any two atoms in the brick [0, box[d][d]) for a box within check_box's skew
 * limits (pbc.c:143-146). */
static int brute_force_core(int natoms, const float *x, const float *q, const int *type, int ntype,
                            const float *nbfp, const double box9[9],
                            const int *excl_index, const int *excl_a,
                            const oracle_params_t *p, int calc_ener,
                            double *f, double *vir, double *Vc, double *Vvdw)
{
    const double rc2 = p->rcoulomb * p->rcoulomb;
    int          i, j, k, d, e;
    double       q2 = 0;
    const int    tric = (box9[3] != 0 || box9[6] != 0 || box9[7] != 0);
    float        bf[9];
    for (d = 0; d < 9; d++) { bf[d] = (float)box9[d]; }
    for (i = 0; i < natoms; i++)
    {
        q2 += (double)q[i] * q[i];
        for (j = i + 1; j < natoms; j++)
        {
            double dx[3], rsq = 0, fscal, c6, c12;
            int    int_bit = 1;
            for (d = 0; d < 3; d++) { dx[d] = (double)x[3 * i + d] - x[3 * j + d]; }
            if (!tric)
            {
                for (d = 0; d < 3; d++)
                {
                    dx[d] -= box9[4 * d] * floor(dx[d] / box9[4 * d] + 0.5);
                    rsq   += dx[d] * dx[d];
                }
            }
            else
            {
                double best[3] = {0, 0, 0};
                int    tx, ty, tz;
                rsq = 1e300;
                for (tz = -1; tz <= 1; tz++)
                {
                    for (ty = -1; ty <= 1; ty++)
                    {
                        for (tx = -2; tx <= 2; tx++)
                        {
                            /* the shift vector as the path has it: summed in float (calc_shifts) */
                            const float  sx = tx * bf[0] + ty * bf[3] + tz * bf[6];
                            const float  sy = ty * bf[4] + tz * bf[7];
                            const float  sz = tz * bf[8];
                            const double vx = dx[0] + sx, vy = dx[1] + sy, vz = dx[2] + sz;
                            const double r2 = vx * vx + vy * vy + vz * vz;
                            if (r2 < rsq) { rsq = r2; best[0] = vx; best[1] = vy; best[2] = vz; }
                        }
                    }
                }
                for (d = 0; d < 3; d++) { dx[d] = best[d]; }
            }
            if (rsq >= rc2) { continue; }
            if (excl_index)
            {
                for (k = excl_index[i]; k < excl_index[i + 1]; k++)
                {
                    if (excl_a[k] == j) { int_bit = 0; break; }
                }
            }
            c6    = nbfp[(type[i] * ntype + type[j]) * 2];
            c12   = nbfp[(type[i] * ntype + type[j]) * 2 + 1];
            fscal = pair_term(p, rsq, int_bit, p->epsfac * q[i] * q[j], c6, c12,
                              p->ljpme ? c6grid_of(p, type[i], type[j]) : 0.0,
                              calc_ener ? Vc : NULL, calc_ener ? Vvdw : NULL);
            for (d = 0; d < 3; d++)
            {
                f[3 * i + d] += fscal * dx[d];
                f[3 * j + d] -= fscal * dx[d];
                if (vir)
                {
                    for (e = 0; e < 3; e++) { vir[3 * d + e] += -0.5 * dx[d] * fscal * dx[e]; }
                }
            }
        }
    }
    if (calc_ener)
    {
        if (p->el_type == NBX_EL_RF) { *Vc += -p->epsfac * 0.5 * p->c_rf * q2; }
        else { *Vc += -p->epsfac * p->beta * (0.5 * M_2_SQRTPI) * q2; }
        if (p->ljpme)
        {
            const double l2 = p->ewaldcoeff_lj * p->ewaldcoeff_lj;
            for (i = 0; i < natoms; i++) { *Vvdw += 0.5 * nbfp[(size_t)type[i] * (ntype + 1) * 2] / 6 * (l2 * l2 * l2 / 6.0); }
        }
    }
    return 0;
}

int oracle_brute_force(int natoms, const float *x, const float *q, const int *type, int ntype,
                       const float *nbfp, const float *box,
                       const int *excl_index, const int *excl_a,
                       const oracle_params_t *p, int calc_ener,
                       double *f, double *vir /* 9, -0.5 sum r_ij (x) f_ij */, double *Vc, double *Vvdw)
{
    const double b9[9] = {box[0], 0, 0, 0, box[1], 0, 0, 0, box[2]};
    return brute_force_core(natoms, x, q, type, ntype, nbfp, b9, excl_index, excl_a, p, calc_ener, f, vir, Vc, Vvdw);
}

int oracle_brute_force_tric(int natoms, const float *x, const float *q, const int *type, int ntype,
                            const float *nbfp, const float *box9,
                            const int *excl_index, const int *excl_a,
                            const oracle_params_t *p, int calc_ener,
                            double *f, double *vir, double *Vc, double *Vvdw)
{
    double b9[9];
    int    d;
    for (d = 0; d < 9; d++) { b9[d] = box9[d]; }
    return brute_force_core(natoms, x, q, type, ntype, nbfp, b9, excl_index, excl_a, p, calc_ener, f, vir, Vc, Vvdw);
}

/* Virial from shift forces and per-atom forces, as calc_vir does for the Verlet path
 * (mdlib/calcvir.c:55-93, sim_util.c:289): vir = -0.5 * (sum_s shift_vec[s] (x) fshift[s]
 * + sum_a x[a] (x) f[a]), accumulated in double. */
void oracle_virial(int natoms, const float *xq, const double *f, const float *shift_vec,
                   const double *fshift, double *vir)
{
    int a, s, d, e;
    for (d = 0; d < 9; d++) { vir[d] = 0; }
    for (a = 0; a < natoms; a++)
    {
        for (d = 0; d < 3; d++)
        {
            for (e = 0; e < 3; e++) { vir[3 * d + e] += -0.5 * (double)xq[(size_t)a * 4 + d] * f[(size_t)a * 3 + e]; }
        }
    }
    for (s = 0; s < NBX_SHIFTS; s++)
    {
        for (d = 0; d < 3; d++)
        {
            for (e = 0; e < 3; e++) { vir[3 * d + e] += -0.5 * (double)shift_vec[3 * s + d] * fshift[3 * s + e]; }
        }
    }
}

/* Scalar single-thread float port of the same kernel, for the bench's cpu_baseline "port"
 * leg when oracle/_ref is not available: identical loop, float arithmetic, F only. */
int oracle_kernel_gpu_list_f32(int natoms, const float *xq, const int *type, int ntype,
                               const float *nbfp, const float *shift_vec,
                               int nsci, const nbx_sci_t *sci, const nbx_cj4_t *cj4,
                               const nbx_excl_t *excl, const oracle_params_t *p, float *f)
{
    const float rc2 = (float)(p->rcoulomb * p->rcoulomb);
    const float k2  = (float)(2 * p->k_rf), beta = (float)p->beta, epsfac = (float)p->epsfac;
    int         n, j4, jm, im, ic, jc;
    (void)natoms;
    for (n = 0; n < nsci; n++)
    {
        const nbx_sci_t *e = &sci[n];
        const float     *sh = shift_vec + 3 * e->shift;
        for (j4 = e->cj4_ind_start; j4 < e->cj4_ind_end; j4++)
        {
            const nbx_cj4_t *g = &cj4[j4];
            for (jm = 0; jm < NBX_JGROUP_SIZE; jm++)
            {
                for (im = 0; im < NSC; im++)
                {
                    const int cj = g->cj[jm], ci = e->sci * NSC + im;
                    /* nbnxn_kernel_gpu_ref.c:194-197 reads imei[0] alone (host-built lists carry the same mask twice,
                     * nbnxn_search.c:3108-3111); a list pruned per 32-lane half on a GPU (nbnxn_cuda_kernel.cuh:503) carries
                     * two, and a cluster pair is alive while either half keeps it: the union */
                    if (!(((g->imei[0].imask | g->imei[1].imask) >> (jm * NSC + im)) & 1u)) { continue; }
                    for (ic = 0; ic < CL; ic++)
                    {
                        const int   ia = ci * CL + ic;
                        const float ix = sh[0] + xq[ia * 4], iy = sh[1] + xq[ia * 4 + 1], iz = sh[2] + xq[ia * 4 + 2];
                        const float iq = epsfac * xq[ia * 4 + 3];
                        float       fix = 0, fiy = 0, fiz = 0;
                        for (jc = 0; jc < CL; jc++)
                        {
                            const int ja = cj * CL + jc;
                            float     dx, dy, dz, rsq, rinv, rinvsq, rinvsix, fscal, qq, ib;
                            if (e->shift == NBX_CENTRAL && ci == cj && ja <= ia) { continue; }
                            ib  = (float)((excl[g->imei[jc >> 2].excl_ind].pair[(jc & 3) * CL + ic] >> (jm * NSC + im)) & 1u);
                            dx  = ix - xq[ja * 4]; dy = iy - xq[ja * 4 + 1]; dz = iz - xq[ja * 4 + 2];
                            rsq = dx * dx + dy * dy + dz * dz;
                            if (rsq >= rc2) { continue; }
                            rsq   += (1.0f - ib) * NBX_AVOID_SING_R2_INC;
                            rinv   = 1.0f / sqrtf(rsq);
                            rinvsq = rinv * rinv;
                            qq     = iq * xq[ja * 4 + 3];
                            if (p->el_type == NBX_EL_RF)
                            {
                                fscal = qq * (ib * rinv * rinvsq - k2);
                            }
                            else
                            {
                                const float br = beta * rsq * rinv;
                                fscal = qq * (ib * rinv * rinvsq - (erff(br) * rinv - beta * 1.1283791671f * expf(-br * br)) * rinvsq);
                            }
                            rinvsix = ib * rinvsq * rinvsq * rinvsq;
                            fscal  += (nbfp[(type[ia] * ntype + type[ja]) * 2 + 1] * rinvsix * rinvsix
                                       - nbfp[(type[ia] * ntype + type[ja]) * 2] * rinvsix) * rinvsq;
                            fix += fscal * dx; fiy += fscal * dy; fiz += fscal * dz;
                            f[ja * 3] -= fscal * dx; f[ja * 3 + 1] -= fscal * dy; f[ja * 3 + 2] -= fscal * dz;
                        }
                        f[ia * 3] += fix; f[ia * 3 + 1] += fiy; f[ia * 3 + 2] += fiz;
                    }
                }
            }
        }
    }
    return 0;
}
