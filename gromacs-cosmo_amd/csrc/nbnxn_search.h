// Internal interface of the device-side pair search (nbnxn_search.hip, nbnxn_search_prims.hip).
// Not part of the public ABI; the C entry points are in nbnxn_hip_api.hip.
#ifndef NBNXN_SEARCH_H
#define NBNXN_SEARCH_H

#include <hip/hip_runtime.h>

#include "nbnxn_hip.h"

struct nbx_search;

nbx_search *nbx_search_create();
void        nbx_search_destroy(nbx_search *s);
// progressive balancing of split lists (default on), see progressive_nsp_max in nbnxn_host.cpp
void        nbx_search_set_progressive_balance(nbx_search *s, bool on);
// text of the last failure of a call that returned non-zero
const char *nbx_search_error();

// Stage 1 of nbnxn_put_on_grid (nbnxn_search.c:1777) for one zone: column grid over [corner0, corner1) from the
// atom density, the zone's atoms d_x[a0 .. a0+n) bucketed per column and ordered by z.  Zones are gridded in order
// (0 = home, then the halo zones as the domain decomposition numbers them); zone 0 starts a new search and takes the
// box and the periodic dimensions.  d_x is [all atoms][3] in caller order, on the device.  Synchronises the stream
// once (the padded atom count sizes the caller's arrays).  ALL zones get their columns before any is filled.
int nbx_search_grid_columns(nbx_search *s, hipStream_t st, int zone, const float box[9], const int pbc_shift_dim[3],
                            const float corner0[3], const float corner1[3], int a0, int n, const float *d_x, float atom_density);
int nbx_search_natoms_padded(const nbx_search *s);         // all zones
int nbx_search_natoms_padded_local(const nbx_search *s);   // home zone
int nbx_search_nzones(const nbx_search *s);
// Stage 2, per zone in order: per super-cell ordering in y and x, bounding boxes, and the grid-ordered atom data
// (coordinates + charge with fillers, types with the dummy type): nbnxn_atomdata_set /
// nbnxn_atomdata_copy_x_to_nbat_x of the reference folded into the gridding pass.  d_x / d_q / d_type hold all
// atoms in caller order; d_xq / d_type_out are the grid-ordered arrays of all zones.
int nbx_search_grid_fill(nbx_search *s, hipStream_t st, int zone, const float *d_x, const float *d_q, const int *d_type,
                         int ntype_with_dummy, float4 *d_xq, int *d_type_out);
// device arrays: grid slot -> caller atom (-1 filler) [natoms_padded]; caller atom -> grid slot [atoms of all zones]
const int *nbx_search_atom_order(const nbx_search *s);
const int *nbx_search_cell_index(const nbx_search *s);

struct nbx_search_list_t {
    int               nsci, ncj4, nexcl, nci_tot;
    const nbx_sci_t  *sci;
    const nbx_cj4_t  *cj4;
    const nbx_excl_t *excl;
};

// nbnxn_make_pairlist (nbnxn_search.c:5583): iloc NBX_LOCAL = home zone against itself (half list), NBX_NONLOCAL =
// home zone against the halo zones (all shifts); the list in the GPU format, left on the device until the next call.
// d_excl_index [atoms+1] / d_excl_a: topology exclusions (t_blocka) in caller numbering, may be null.
// zone_pairs [3 * nzp] = {i-zone, first j-zone, end j-zone} (host memory), the reference's dd_zp tables
// (domdec.c:433-449, the i-zone loop of nbnxn_make_pairlist, nbnxn_search.c:5652-5699); nzp = 0: the pairs above.
int nbx_search_make_pairlist(nbx_search *s, hipStream_t st, int iloc, int nzp, const int *zone_pairs, float rlist, int min_ci_balanced,
                             const int *d_excl_index, const int *d_excl_a, const float4 *d_xq, nbx_search_list_t *out);

// verification: grid geometry and per-cluster data back on the host
int nbx_search_grid_info(const nbx_search *s, int zone, int *ncx, int *ncy, int *nc);
int nbx_search_download_grid(const nbx_search *s, hipStream_t st, int *a, int *cell, float *bb, float *bbcz, int *nsubc);

// --- nbnxn_search_prims.hip: library scans / sorts behind plain signatures (hipCUB is heavy to
// compile and must not see -ffp-contract=off) ---
struct nbx_prim_tmp { void *p = nullptr; size_t bytes = 0; };
hipError_t nbx_prim_exclusive_sum(nbx_prim_tmp &tmp, const int *in, int *out, int n, hipStream_t st);
hipError_t nbx_prim_sort_u64_int(nbx_prim_tmp &tmp, const unsigned long long *k_in, unsigned long long *k_out, const int *v_in,
                                 int *v_out, int n, int end_bit, hipStream_t st);
hipError_t nbx_prim_sort_u32_sci(nbx_prim_tmp &tmp, const unsigned *k_in, unsigned *k_out, const nbx_sci_t *v_in, nbx_sci_t *v_out,
                                 int n, int end_bit, hipStream_t st);
void       nbx_prim_free(nbx_prim_tmp &tmp);

#endif
