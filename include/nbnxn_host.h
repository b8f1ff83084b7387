/*
 * nbnxn_host.h -- C ABI of the host-side producers of the GPU list format: gridding,
 * atom-data fill and cluster-pair list construction.  The reference keeps these on the
 * host ("host-kept; GPU list format is the input contract"); this library is
 * the from-scratch C++ equivalent that bench.py, the tests and the multi-GPU driver use
 * to make nbnxn_atomdata_t / nbnxn_pairlist_t contents on a box without the reference.
 * No GPU code: it loads anywhere.
 *
 * Reference interfaces mirrored (src/gromacs/mdlib/...):
 *   nbnxn_put_on_grid                    nbnxn_search.c:1777
 *   nbnxn_atomdata_set                   nbnxn_atomdata.c:1094
 *   nbnxn_atomdata_copy_x_to_nbat_x      nbnxn_atomdata.c:1141
 *   nbnxn_make_pairlist                  nbnxn_search.c:5583
 *   nbnxn_atomdata_add_nbat_f_to_f       nbnxn_atomdata.c:1600
 *
 * Scope: rectangular boxes, pbc = xyz, one home grid plus optional halo grids of a 1-D
 * (z) domain decomposition.  All functions return 0 on success, <0 on error
 * (nbnxn_host_last_error() has the text).
 */
#ifndef NBNXN_HOST_H
#define NBNXN_HOST_H

#include "nbnxn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbnxn_host *nbnxn_host_ptr_t;

const char *nbnxn_host_last_error(void);

nbnxn_host_ptr_t nbnxn_host_create(void);
void             nbnxn_host_destroy(nbnxn_host_ptr_t h);

/* Sets the periodic box (rectangular, nm) and which dimensions get periodic shifts in the
 * pair search; a dimension that is domain-decomposed gets none (nbnxn_search.c:4906-4927). */
int nbnxn_host_set_box(nbnxn_host_ptr_t h, const float box[3], const int pbc_shift_dim[3]);

/* The same for a triclinic box: box[9] is the GROMACS matrix, rows are box vectors, lower triangular
 * and within the skew limits of check_box (gmxlib/pbc.c:121-156).  Atoms are expected in the
 * rectangular brick [0, box[d][d]) that put_atoms_in_box leaves them in (pbc.c:1562); the
 * search then visits up to 5 x 3 x 3 shift vectors (nbnxn_search.c:4906-4927, :5017-5098). */
int nbnxn_host_set_box_triclinic(nbnxn_host_ptr_t h, const float box[9], const int pbc_shift_dim[3]);

/* nbnxn_put_on_grid for zone `zone` (0 = home, >0 = halo zones, in order).  x is [n*3]
 * for the atoms a0..a0+n-1 of the caller's local numbering (home atoms first, halo atoms
 * after).  corner0/corner1 bound the zone.  atom_density <= 0: computed from n and the
 * zone volume.  Builds columns, z/y/x sorting into 8-atom clusters, bounding boxes, and
 * the grid-order index arrays. */
int nbnxn_host_put_on_grid(nbnxn_host_ptr_t h, int zone,
                           const float corner0[3], const float corner1[3],
                           int a0, int n, const float *x, float atom_density);

/* padded atom count of all grids (multiple of 64) and of the home grid */
int nbnxn_host_natoms_padded(nbnxn_host_ptr_t h);
int nbnxn_host_natoms_padded_local(nbnxn_host_ptr_t h);
int nbnxn_host_ncolumns(nbnxn_host_ptr_t h, int zone, int *ncx, int *ncy);
int nbnxn_host_nsupercells(nbnxn_host_ptr_t h);

/* grid order -> caller atom index (or -1 for filler), [natoms_padded]  (nbs->a) */
const int *nbnxn_host_atom_order(nbnxn_host_ptr_t h);
/* caller atom index -> grid index, [natoms]  (nbs->cell) */
const int *nbnxn_host_cell_index(nbnxn_host_ptr_t h);

/* nbnxn_atomdata_set: types and charges in grid order; fillers get type ntype-1 (the
 * zero-LJ dummy type, nbnxn_atomdata.c:561,873) and q = 0.  type/q are in caller order for
 * all atoms on the grids.  Fills xq[.][3] and type_out [natoms_padded]. */
int nbnxn_host_set_atomdata(nbnxn_host_ptr_t h, int ntype_with_dummy,
                            const int *type, const float *q,
                            float *xq /* [natoms_padded*4] */, int *type_out);

/* nbnxn_atomdata_copy_x_to_nbat_x: x (caller order, [natoms*3]) -> xq[.][0..2] in grid
 * order; fillers at -107 nm * (1 + cell coordinate) (nbnxn_atomdata.c:256-295). */
int nbnxn_host_copy_x_to_xq(nbnxn_host_ptr_t h, const float *x, float *xq);

/* nbnxn_atomdata_add_nbat_f_to_f: f[a] += fnb[cell[a]] for atoms a0..a0+n-1 */
int nbnxn_host_add_f(nbnxn_host_ptr_t h, const float *fnb /* [natoms_padded*3] */,
                     int a0, int n, float *f /* [natoms*3] */);

/* nbnxn_make_pairlist for the 8x8x8 GPU list.  iloc = NBX_LOCAL: home grid vs home grid
 * (half list); NBX_NONLOCAL: home grid vs every halo grid.  Topology exclusions as CSR in
 * caller atom numbering (t_blocka: index[natoms+1], a[]); may be NULL.
 * min_ci_balanced: the target from nbnxn_hip_min_ci_balanced; <= 0 disables splitting
 * (get_nsubpair_max, nbnxn_search.c:4324-4408).  xq must be the array filled by
 * set_atomdata/copy_x_to_xq (used for the single-pair atom distance check, :3077-3091). */
int nbnxn_host_make_pairlist(nbnxn_host_ptr_t h, int iloc, float rlist,
                             const int *excl_index, const int *excl_a,
                             int min_ci_balanced, const float *xq);

/* The same with the zone pairs of a domain decomposition: nzp triples {i-zone, first j-zone, end j-zone} as the
 * reference's dd_zp3 / dd_zp2 / dd_zp1 (mdlib/domdec.c:433-449) give them to the i-zone loop of nbnxn_make_pairlist
 * (nbnxn_search.c:5652-5699).  i-clusters of the list may then lie in halo zones (the eighth-shell decomposition
 * computes pairs between two imported zones).  {0, 0, 1} is the local list, {0, 1, nzones} the non-local list of
 * nbnxn_host_make_pairlist. */
int nbnxn_host_make_pairlist_zones(nbnxn_host_ptr_t h, int iloc, int nzp, const int *zone_pairs, float rlist,
                                   const int *excl_index, const int *excl_a, int min_ci_balanced, const float *xq);

/* Progressive balancing of split lists (default on): entries get one of four pair budgets around the average
 * that min_ci_balanced implies, by a hash of (i super-cell, shift), so that the smallest entries -- which the
 * GPU runs last, the list being ordered by size -- are a third of the average (the reference's progBal,
 * nbnxn_search.c:3928-3960, reaches the same end from a running entry count).  Off: every entry is split at
 * the average budget. */
int nbnxn_host_set_progressive_balance(nbnxn_host_ptr_t h, int on);

/* the list built last for iloc, as an nbnxn_pairlist_t-compatible view (owned by h) */
const nbx_pairlist_t *nbnxn_host_pairlist(nbnxn_host_ptr_t h, int iloc);

/* the 45 shift vectors of the box (calc_shifts, gmxlib/shift_util / pbc.c) */
int nbnxn_host_shift_vec(nbnxn_host_ptr_t h, float *shift_vec /* [45*3] */);

#ifdef __cplusplus
}
#endif

#endif /* NBNXN_HOST_H */
