/*
 * binding_check_dd.c -- compiles the reference-side binding of the halo exchange (include/nbnxn_hip_dd_gmx.h) against
 * the reference's own headers (gmx_domdec_t, matrix, IVEC2IS, nbnxn_cuda_ptr_t), the way binding_check.c does for the
 * force calls, and runs the host-only translation on a hand-made decomposition.
 * TEST INFRASTRUCTURE: built by oracle/Makefile only where /root/reference exists (-Wall -Wextra
 * -Werror=incompatible-pointer-types ...), linked against libnbnxn_hip.so so that every nbnxn_hip_dd_* name the binding
 * uses is proven exported; run by tests/test_abi.py, which derives what it must print from dd_move_x's rules.
 * The run touches no GPU: it calls nbnxn_hip_dd_gmx_translate only.
 */
#define GMX_GPU
#include <stdio.h>

#include "typedefs.h"
#include "types/commrec.h"
#include "types/nb_verlet.h"
#include "types/ishift.h"
#include "domdec.h"

#include "nbnxn_hip_dd_gmx.h"

/* the calls a maintainer adds to do_force_cutsVERLET; never executed */
int use_dd_names(nonbonded_verlet_t *nbv, t_commrec *cr, matrix box, const nbx_dd_gmx_pulse_t *pulse, int npulse, const int *a, int na)
{
    char err[STRLEN];
    int  rc = nbnxn_hip_dd_gmx_bind(nbv->cu_nbv, cr->dd, box, TRUE, npulse, pulse, a, na, err);
    if (rc == 0) { rc = nbnxn_hip_dd_move_x((nbnxn_hip_ptr_t)nbv->cu_nbv); }
    if (rc == 0) { rc = nbnxn_hip_dd_move_f((nbnxn_hip_ptr_t)nbv->cu_nbv); }
    return rc;
}

/* the decomposition of tests/test_abi.py::test_dd_binding_translates_the_reference_pulses (keep the two in step):
 * two decomposed dimensions x and y, this rank at cell 0 of both (both pulses cross the periodic boundary), 3-atom
 * charge groups; x: one pulse in place; y: two pulses, not in place, the first forwards what x brought, the second
 * what the first brought.  Atom layout [home 0-5][zone 1: 6-8][zone 2: 9-11 | 12-14][zone 3: 15-17]. */
static int self_test(void)
{
    gmx_domdec_t       dd;
    matrix             box      = { { 4.0f, 0.0f, 0.0f }, { 0.5f, 3.0f, 0.0f }, { 0.0f, 0.0f, 2.0f } };
    int                cgindex[7] = { 0, 3, 6, 9, 12, 15, 18 };
    int                a[24]    = { 2, 0, 1, 5, 4, 3, -1, -1, 8, 7, 6, 11, 9, 10, -1, 14, 13, 12, 17, 16, 15, -1, -1, -1 };
    int                idx0[1]  = { 1 }, idx1[2] = { 0, 2 }, idx2[1] = { 3 };
    int                c0a[2]   = { 9, 15 }, c1a[2] = { 12, 18 }, c0b[2] = { 12, 18 }, c1b[2] = { 15, 18 };
    nbx_dd_gmx_pulse_t pulse[14];
    nbx_dd_message_t   msg[NBX_DD_MAX_MESSAGES];
    int                phase_end[NBX_DD_MAX_MESSAGES], *ss, *rs, ns, nr, k, i, rc;
    char               err[STRLEN] = "";

    memset(&dd, 0, sizeof(dd)); memset(pulse, 0, sizeof(pulse));
    dd.ndim = 2; dd.dim[0] = XX; dd.dim[1] = YY; dd.ci[XX] = 0; dd.ci[YY] = 0; dd.ci[ZZ] = 0;
    dd.neighbor[0][0] = 1; dd.neighbor[0][1] = 3; dd.neighbor[1][0] = 2; dd.neighbor[1][1] = 5;
    dd.cgindex = cgindex; dd.nat_home = 6; dd.nat_tot = 18;
    pulse[0].dim_index = 0; pulse[0].nzone = 1; pulse[0].nsend_cg = 1; pulse[0].nsend_at = 3; pulse[0].nrecv_at = 3; pulse[0].index = idx0; pulse[0].in_place = 1;
    pulse[1].dim_index = 1; pulse[1].nzone = 2; pulse[1].nsend_cg = 2; pulse[1].nsend_at = 6; pulse[1].nrecv_at = 6; pulse[1].index = idx1;
    pulse[1].cell2at0 = c0a; pulse[1].cell2at1 = c1a;
    pulse[2].dim_index = 1; pulse[2].nzone = 2; pulse[2].nsend_cg = 1; pulse[2].nsend_at = 3; pulse[2].nrecv_at = 3; pulse[2].index = idx2;
    pulse[2].cell2at0 = c0b; pulse[2].cell2at1 = c1b;

    rc = nbnxn_hip_dd_gmx_translate(&dd, box, TRUE, 3, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    if (rc != 0) { printf("translate failed: %s\n", err); return 1; }
    for (k = 0; k < 3; k++)
    {
        printf("msg %d to %d from %d n_send %d n_recv %d shift_index %d shift %g %g %g phase_end %d\n", k, msg[k].send_to, msg[k].recv_from,
               msg[k].n_send, msg[k].n_recv, msg[k].shift_index, msg[k].shift[0], msg[k].shift[1], msg[k].shift[2], phase_end[k]);
    }
    printf("send"); for (i = 0; i < ns; i++) { printf(" %d", ss[i]); } printf("\n");
    printf("recv"); for (i = 0; i < nr; i++) { printf(" %d", rs[i]); } printf("\n");
    free(ss); free(rs);

    /* without a shift-force array (the dd_move_f(dd, f_twin, NULL) of sim_util.c:1434): no shift index, same shift */
    rc = nbnxn_hip_dd_gmx_translate(&dd, box, FALSE, 3, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    printf("no fshift: rc %d shift_index %d %d %d\n", rc, msg[0].shift_index, msg[1].shift_index, msg[2].shift_index);
    free(ss); free(rs);
    /* a rank that is not at the boundary in y */
    dd.ci[YY] = 1;
    rc = nbnxn_hip_dd_gmx_translate(&dd, box, TRUE, 3, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    printf("inner in y: rc %d shift_index %d %d %d shift_y %g %g %g\n", rc, msg[0].shift_index, msg[1].shift_index, msg[2].shift_index,
           msg[0].shift[1], msg[1].shift[1], msg[2].shift[1]);
    free(ss); free(rs);
    dd.ci[YY] = 0;

    /* refusals */
    rc = nbnxn_hip_dd_gmx_translate(&dd, box, TRUE, 14, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    printf("refused %d: %s\n", rc, err);
    dd.bScrewPBC = TRUE;
    rc = nbnxn_hip_dd_gmx_translate(&dd, box, TRUE, 3, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    printf("refused %d: %s\n", rc, err);
    dd.bScrewPBC = FALSE;
    pulse[1].nsend_at = 5; pulse[1].nrecv_at = 7;           /* totals unchanged, the charge groups disagree */
    rc = nbnxn_hip_dd_gmx_translate(&dd, box, TRUE, 3, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    printf("refused %d: %s\n", rc, err);
    pulse[1].nsend_at = 6; pulse[1].nrecv_at = 6;
    a[12] = -1;                                             /* atom 9 left off the grid */
    rc = nbnxn_hip_dd_gmx_translate(&dd, box, TRUE, 3, pulse, a, 24, msg, phase_end, &ss, &ns, &rs, &nr, err);
    printf("refused %d: %s\n", rc, err);
    return 0;
}

int main(void) { return self_test(); }
