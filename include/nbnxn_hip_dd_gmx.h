/*
 * nbnxn_hip_dd_gmx.h -- the reference-side binding of the halo exchange: turns the pulses the reference's domain
 * decomposition has set up (setup_dd_communication, mdlib/domdec.c:8085-8560; used by dd_move_x, domdec.c:651-758, and
 * dd_move_f, :761-875) into the messages and phases of libnbnxn_hip.so (nbnxn_hip_dd_set_messages / _set_phases,
 * include/nbnxn_hip.h), so that
 *
 *     dd_move_x(cr->dd, box, x)            sim_util.c:1114    becomes    nbnxn_hip_dd_move_x(nbv->cu_nbv)
 *     dd_move_f(cr->dd, f, fr->fshift)     sim_util.c:1428    becomes    nbnxn_hip_dd_move_f(nbv->cu_nbv)
 *
 * for the coordinates and forces that stay in HBM, with the pulses, their order, their neighbours, their periodic
 * shifts and their shift-force bookkeeping exactly the reference's.
 *
 * What the translation needs from the reference, and where a maintainer gets it:
 *   gmx_domdec_t            public (legacyheaders/types/commrec.h:134-226): ndim, dim[], ci[], neighbor[][], cgindex,
 *                           nat_home, bScrewPBC
 *   the pulses              PRIVATE to domdec.c: gmx_domdec_comm_t (domdec.c:226-380) holds cd[d].np, cd[d].bInPlace
 *                           and cd[d].ind[p] (gmx_domdec_ind_t, :100-114).  One accessor inside domdec.c copies those
 *                           fields into nbx_dd_gmx_pulse_t below; it is given at the end of this file under
 *                           NBNXN_HIP_DD_GMX_IN_DOMDEC, to be included by domdec.c itself.
 *   atom -> grid slot       nbnxn_get_atomorder(nbv->nbs, &a, &na) (mdlib/nbnxn_search.h:105) gives slot -> atom for
 *                           home and halo atoms once nbnxn_put_on_grid_nonlocal (sim_util.c:978) has run; it is inverted here.
 *
 * Include it below typedefs.h (gmx_domdec_t, matrix, ivec) and types/ishift.h (IVEC2IS).
 * oracle/binding_check_dd.c is the translation unit that includes this header after the reference's own headers;
 * oracle/Makefile compiles it with -Wall -Wextra -Werror=incompatible-pointer-types against /root/reference, and
 * tests/test_abi.py builds it and runs its host-only self-test (a two-dimensional decomposition with forwarding,
 * a pulse that is not in place, a periodic boundary).  The accessor block cannot be compiled here (domdec.c needs
 * the cmake-generated config.h); it touches nothing but the fields named above.
 */
#ifndef NBNXN_HIP_DD_GMX_H
#define NBNXN_HIP_DD_GMX_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbnxn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one pulse of one decomposed dimension, as dd_move_x walks them (for d: for p:), copied out of gmx_domdec_comm_t */
typedef struct {
    int        dim_index;   /* d, the index into dd->dim[] / dd->neighbor[]                                      */
    int        nzone;       /* zones that send in this pulse: 1, 2, 4 for d = 0, 1, 2 (domdec.c:667, 757)        */
    int        nsend_cg;    /* ind->nsend[nzone]:     charge groups sent                                         */
    int        nsend_at;    /* ind->nsend[nzone + 1]: atoms sent                                                 */
    int        nrecv_at;    /* ind->nrecv[nzone + 1]: atoms received                                             */
    const int *index;       /* ind->index: the charge groups sent, nsend_cg of them                              */
    int        in_place;    /* cd->bInPlace: received atoms are appended (domdec.c:731-734), else scattered ...  */
    const int *cell2at0;    /* ... to the atom ranges ind->cell2at0[z] .. ind->cell2at1[z], z < nzone (:743-753) */
    const int *cell2at1;
} nbx_dd_gmx_pulse_t;

/*
 * The translation, host only.  npulse pulses in dd_move_x's order -> msg[npulse], phase_end[npulse] (every pulse a
 * phase of its own: a later pulse forwards what an earlier one brought, domdec.c:685-694 reads x[] beyond nat_home),
 * send_slots / recv_slots (malloc'ed here, the caller frees; lengths *n_send_tot / *n_recv_tot).
 * a[na]: grid slot -> atom of nbnxn_get_atomorder (-1 in filler slots).  with_fshift: whether the step's dd_move_f
 * gets a shift-force array (sim_util.c:1428 passes fr->fshift; the twin-range call at :1434 passes NULL).
 * Returns 0, or -1 with a message in err (at least 256 bytes).
 */
static int nbnxn_hip_dd_gmx_translate(const gmx_domdec_t *dd, matrix box, gmx_bool with_fshift, int npulse, const nbx_dd_gmx_pulse_t *pulse,
                                      const int *a, int na, nbx_dd_message_t *msg, int *phase_end, int **send_slots, int *n_send_tot,
                                      int **recv_slots, int *n_recv_tot, char *err)
{
    int  k, i, j, z, ns = 0, nr = 0, nat = 0, nat_tot, s, *cell = NULL, *ss = NULL, *rs = NULL;
    ivec vis;

    *send_slots = NULL; *recv_slots = NULL; *n_send_tot = 0; *n_recv_tot = 0;
    if (npulse < 1 || npulse > NBX_DD_MAX_MESSAGES) { sprintf(err, "%d pulses: the library takes 1 to %d messages", npulse, NBX_DD_MAX_MESSAGES); return -1; }
    if (dd->bScrewPBC) { sprintf(err, "screw periodic boundaries rotate y and z in the pulse along x (domdec.c:710-728): not supported"); return -1; }
    for (k = 0; k < npulse; k++) { ns += pulse[k].nsend_at; nr += pulse[k].nrecv_at; }
    nat = dd->nat_home + nr;                                /* = dd->nat_tot once the set-up is complete */
    cell = (int *)malloc((size_t)(nat > 0 ? nat : 1) * sizeof(int));
    ss   = (int *)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int));
    rs   = (int *)malloc((size_t)(nr > 0 ? nr : 1) * sizeof(int));
    if (!cell || !ss || !rs) { sprintf(err, "out of memory"); goto fail; }
    for (i = 0; i < nat; i++) { cell[i] = -1; }
    for (s = 0; s < na; s++)
    {
        if (a[s] >= nat) { sprintf(err, "grid slot %d holds atom %d, but home + halo atoms are %d", s, a[s], nat); goto fail; }
        if (a[s] >= 0) { cell[a[s]] = s; }
    }
    ns = 0; nr = 0;
    nat_tot = dd->nat_home;
    for (k = 0; k < npulse; k++)
    {
        const nbx_dd_gmx_pulse_t *p   = &pulse[k];
        int                       d   = p->dim_index, dim, n0 = ns, r0 = nr;
        gmx_bool                  pbc;
        if (d < 0 || d >= dd->ndim || (k > 0 && d < pulse[k - 1].dim_index)) { sprintf(err, "pulse %d: dimension index %d out of order", k, d); goto fail; }
        dim = dd->dim[d];
        pbc = (dd->ci[dim] == 0);                           /* domdec.c:671: this rank's pulse crosses the boundary */
        /* dd_sendrecv_rvec(dd, d, dddirBackward, ...), domdec_network.c:123-124 */
        msg[k].send_to   = dd->neighbor[d][1];
        msg[k].recv_from = dd->neighbor[d][0];
        msg[k].n_send    = p->nsend_at;
        msg[k].n_recv    = p->nrecv_at;
        for (j = 0; j < DIM; j++) { msg[k].shift[j] = pbc ? box[dim][j] : 0.0f; }           /* :673-676, :705 */
        vis[XX] = vis[YY] = vis[ZZ] = 0; vis[dim] = 1;                                                         /* :791-793 */
        msg[k].shift_index = (pbc && with_fshift) ? IVEC2IS(vis) : -1;                          /* :786-789, :847 */
        for (i = 0; i < p->nsend_cg; i++)                                                       /* :685-694 */
        {
            for (j = dd->cgindex[p->index[i]]; j < dd->cgindex[p->index[i] + 1]; j++)
            {
                /* j may lie beyond the atoms counted so far: a pulse that is not in place scatters into zone blocks (:743-753) */
                if (ns - n0 >= p->nsend_at || j < 0 || j >= nat || cell[j] < 0) { sprintf(err, "pulse %d: atom %d of charge group %d cannot be sent (home + halo atoms %d, nsend %d)", k, j, p->index[i], nat, p->nsend_at); goto fail; }
                ss[ns++] = cell[j];
            }
        }
        if (ns - n0 != p->nsend_at) { sprintf(err, "pulse %d: its charge groups hold %d atoms, nsend says %d", k, ns - n0, p->nsend_at); goto fail; }
        if (p->in_place)                                                                        /* :731-734 */
        {
            for (j = nat_tot; j < nat_tot + p->nrecv_at; j++) { rs[nr++] = j; }
        }
        else                                                                                    /* :743-753 */
        {
            for (z = 0; z < p->nzone; z++)
            {
                for (j = p->cell2at0[z]; j < p->cell2at1[z]; j++)
                {
                    if (nr - r0 >= p->nrecv_at) { sprintf(err, "pulse %d: cell2at ranges hold more than the %d atoms received", k, p->nrecv_at); goto fail; }
                    rs[nr++] = j;
                }
            }
            if (nr - r0 != p->nrecv_at) { sprintf(err, "pulse %d: cell2at ranges hold %d atoms, nrecv says %d", k, nr - r0, p->nrecv_at); goto fail; }
        }
        nat_tot += p->nrecv_at;                                                                 /* :755 */
        for (j = r0; j < nr; j++)
        {
            if (rs[j] < dd->nat_home || rs[j] >= nat || cell[rs[j]] < 0) { sprintf(err, "pulse %d: received atom %d is not on the grid", k, rs[j]); goto fail; }
            rs[j] = cell[rs[j]];
        }
        phase_end[k] = k + 1;
    }
    free(cell);
    *send_slots = ss; *recv_slots = rs; *n_send_tot = ns; *n_recv_tot = nr;
    return 0;
fail:
    free(cell); free(ss); free(rs);
    return -1;
}

/*
 * Per search, after nbnxn_put_on_grid_nonlocal (sim_util.c:978) and before the first nbnxn_hip_dd_move_x: installs the
 * reference's pulses in the module.  Returns 0, -1 (translation; message in err) or the library's error code.
 */
static int nbnxn_hip_dd_gmx_bind(nbnxn_cuda_ptr_t cu_nb, const gmx_domdec_t *dd, matrix box, gmx_bool with_fshift, int npulse,
                                 const nbx_dd_gmx_pulse_t *pulse, const int *a, int na, char *err)
{
    nbx_dd_message_t msg[NBX_DD_MAX_MESSAGES];
    int              phase_end[NBX_DD_MAX_MESSAGES], *ss, *rs, ns, nr, rc;

    if (nbnxn_hip_dd_gmx_translate(dd, box, with_fshift, npulse, pulse, a, na, msg, phase_end, &ss, &ns, &rs, &nr, err) != 0) { return -1; }
    rc = nbnxn_hip_dd_set_messages((nbnxn_hip_ptr_t)cu_nb, npulse, msg, ss, rs);
    if (rc == 0) { rc = nbnxn_hip_dd_set_phases((nbnxn_hip_ptr_t)cu_nb, npulse, phase_end); }
    if (rc != 0) { sprintf(err, "libnbnxn_hip refused the pulses (code %d)", rc); }
    free(ss); free(rs);
    return rc;
}

#ifdef NBNXN_HIP_DD_GMX_IN_DOMDEC
/*
 * The accessor: for domdec.c only (define NBNXN_HIP_DD_GMX_IN_DOMDEC and include this header below the definition of
 * gmx_domdec_comm_t, domdec.c:380; declare it in legacyheaders/domdec.h).  The loop is dd_move_x's (domdec.c:667-758).
 * Returns the number of pulses, or -1 when there are more than max.
 */
int dd_get_nbnxn_hip_pulses(const gmx_domdec_t *dd, nbx_dd_gmx_pulse_t *pulse, int max)
{
    int n = 0, d, p, nzone = 1;
    for (d = 0; d < dd->ndim; d++)
    {
        const gmx_domdec_comm_dim_t *cd = &dd->comm->cd[d];
        for (p = 0; p < cd->np; p++, n++)
        {
            const gmx_domdec_ind_t *ind = &cd->ind[p];
            if (n == max) { return -1; }
            pulse[n].dim_index = d;                  pulse[n].nzone    = nzone;
            pulse[n].nsend_cg  = ind->nsend[nzone];  pulse[n].nsend_at = ind->nsend[nzone + 1];
            pulse[n].nrecv_at  = ind->nrecv[nzone + 1];
            pulse[n].index     = ind->index;         pulse[n].in_place = cd->bInPlace;
            pulse[n].cell2at0  = ind->cell2at0;      pulse[n].cell2at1 = ind->cell2at1;
        }
        nzone += nzone;
    }
    return n;
}
#endif

#ifdef __cplusplus
}
#endif

#endif /* NBNXN_HIP_DD_GMX_H */
