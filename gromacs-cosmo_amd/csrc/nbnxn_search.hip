// Pair search on the device: gridding and construction of the cluster-pair list
// in the GPU format (nbnxn_sci_t / nbnxn_cj4_t / nbnxn_excl_t), home zone, without the atoms or
// the list ever visiting the host.
//
// WHAT is built follows GROMACS 5.0.4: nbnxn_put_on_grid (nbnxn_search.c:1777: set_grid_size_xy
// :433-486, calc_column_indices :1493-1590, sort_columns_supersub :1374-1490, fill_cell :1164),
// nbnxn_make_pairlist (:5583: make_cluster_list_supersub :2955-3123, set_self_and_newton_excls
// :2759-2780, set_sci_top_excls :3673-3807, split_sci_entry :3936-4024, sort_sci :5517-5580).
// HOW is specific to the device: no per-thread lists that get combined, but count -> scan ->
// fill passes over a fixed enumeration of (i super-cell, shift) entries, one wave64 per entry,
// the 64 lanes testing 64 candidate j-clusters at a time and compacting the survivors with a
// ballot; every float decision (column index, bounding-box distance, atom distance) is made
// with the same IEEE operations in the same order as the host-side builder
// (nbnxn_host.cpp), and this file is compiled with -ffp-contract=off, so that both produce the
// SAME list -- which is how tests/test_gpu_search.py checks it, entry for entry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nbnxn_search.h"
#include "nbnxn_list_rules.h"

namespace {

constexpr int   NA_C     = NBX_CLUSTER_SIZE;
constexpr int   NSUBC    = NBX_NCL_PER_SUPERCL;
constexpr int   NA_SC    = NA_C * NSUBC;
constexpr float FAR_AWAY = 107.0f;   // nbnxn_atomdata.c:256
constexpr int   NSHIFT_HALF = 14;    // shifts <= CENTRAL with |t| <= 1 in every dimension
constexpr int   NSHIFT_HALF_WIDE = 23;   // shifts <= CENTRAL with |tx| <= 2: all shift indices 0..22

thread_local std::string g_err;

int sfail(const char *fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define S_TRY(call)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) { return sfail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } \
    } while (0)

template <typename T>
struct Buf {
    T     *p = nullptr;
    size_t alloc = 0;
    int reserve(size_t n)
    {
        if (n <= alloc && p) { return 0; }
        if (p) { (void)hipFree(p); p = nullptr; alloc = 0; }
        const size_t na = n + n / 4 + 64;
        if (hipMalloc((void **)&p, na * sizeof(T)) != hipSuccess) { return sfail("hipMalloc of %zu bytes failed", na * sizeof(T)); }
        alloc = na;
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); } p = nullptr; alloc = 0; }
};

struct GridDims {
    float c0[3];
    int   ncx, ncy;
    float sx, sy, inv_sx, inv_sy;
};

// home + halo zones: the reference's eighth-shell decomposition has at most 8 (domdec.c:433-449); the half-shell
// decomposition with direct peers of gromacs-cosmo_amd/dd.py has 1 + 13 in three dimensions
constexpr int MAX_ZONES = NBX_MAX_ZONES;

// what the list kernels need to know about one j-zone
struct ZoneParams {
    GridDims   g;
    const int *cxy_ind;   // column -> first super-cell, zone-local numbering
    int        cell0;     // first super-cell of the zone in the global numbering
    int        zr;        // index of the zone's z-range in zrange[]
};

constexpr int MAX_IZONES = 4;   // i-zones of a list: the reference's zone pairs have at most four (dd_zp3, domdec.c:440)

struct ListParams {
    int        nci;       // super-cells of the i-zones (zones 0 .. niz - 1, contiguous from cell 0; the home zone alone unless zone pairs say more)
    int        iz_end[MAX_IZONES];      // first super-cell behind i-zone k
    unsigned   allowed[MAX_IZONES];     // bit j: i-zone k is paired with the list's j-zone slot j
    float      rbb2[MAX_IZONES][MAX_ZONES];   // boundingbox_only_distance2 of (i-zone, j-zone slot)
    int        njz;       // j-zones of this list: 1 (the home zone itself) or the halo zones
    int        same;      // the j-zone is the i-zone: half list, 14 shifts; otherwise all 27 shifts
    int        wide;      // a skewed box that needs two box lengths in x (shp[0] == 2): 23 / 45 shifts instead
    float      box[3];    // the diagonal
    float      byx, bzx, bzy;   // box[YY][XX], box[ZZ][XX], box[ZZ][YY] of a triclinic box
    int        shp[3];
    int        tmp_cap;   // MODE 2 of k_list: scratch slots per entry
    float      rlist, rl2;
    ZoneParams zp[MAX_ZONES];
};

// Entry numbering, in the host builder's loop order: i super-cell, then j-zone, then shift (ascending shift
// index = tz, ty, tx loops).  Within one grid only the 14 shifts <= CENTRAL are visited (one-way list).
struct Entry { int sci, jzi, tx, ty, tz; };

// the 14 shifts of a half list in ascending shift index (XYZ2IS, types/ishift.h:45-55), which is
// the order the reference's tz / ty / tx loops visit them (nbnxn_search.c:5060-5096)
__device__ __constant__ signed char c_shift_t[NSHIFT_HALF][3] = {
    {-1, -1, -1}, {0, -1, -1}, {1, -1, -1}, {-1, 0, -1}, {0, 0, -1}, {1, 0, -1}, {-1, 1, -1}, {0, 1, -1}, {1, 1, -1},
    {-1, -1, 0},  {0, -1, 0},  {1, -1, 0},  {-1, 0, 0},  {0, 0, 0}};

__device__ __forceinline__ int shift_index(int tx, int ty, int tz) { return 5 * (3 * (tz + 1) + ty + 1) + tx + 2; }

// how entry numbers map to (i super-cell, j-zone, shift)
struct EntryMap { int same, njz, wide; };

__host__ __device__ __forceinline__ int entries_per_cell(EntryMap m)
{
    return m.same ? (m.wide ? NSHIFT_HALF_WIDE : NSHIFT_HALF) : (m.wide ? 45 : 27) * m.njz;
}

__device__ __forceinline__ Entry decode_entry(int e, EntryMap m)
{
    Entry     en;
    const int per = entries_per_cell(m);
    en.sci        = e / per;
    const int r   = e - en.sci * per;
    if (m.wide)
    {
        // tx = -2..2: the slot within a zone IS the shift index (XYZ2IS), 0..22 for a half list
        en.jzi = m.same ? 0 : r / 45;
        const int ss = r - en.jzi * 45;
        en.tx = ss % 5 - 2; en.ty = (ss / 5) % 3 - 1; en.tz = ss / 15 - 1;
    }
    else if (m.same)
    {
        en.jzi = 0;
        en.tx = c_shift_t[r][0]; en.ty = c_shift_t[r][1]; en.tz = c_shift_t[r][2];
    }
    else
    {
        en.jzi = r / 27;
        const int ss = r - en.jzi * 27;
        en.tx = ss % 3 - 1; en.ty = (ss / 3) % 3 - 1; en.tz = ss / 9 - 1;
    }
    return en;
}

// ------------------------------------------------------------------------------------------
// gridding
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned float_key_host_device(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// calc_column_indices: column of every atom (clamped: rounding at the upper edge), column
// populations, and the sort key (column, z) whose stable sort reproduces "z ascending, ties by
// atom index" of the host builder
__global__ void k_columns(int a0, int n, const float *__restrict__ x, GridDims g, unsigned long long *__restrict__ key,
                          int *__restrict__ val)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) { return; }
    const int   at = a0 + i;                // x holds all zones' atoms; this zone's start at a0
    const float px = x[3 * at], py = x[3 * at + 1];
    const float pz = x[3 * at + 2] + 0.0f;   // -0 -> +0: they compare equal and must sort as equal
    int         cx = (int)floorf((px - g.c0[0]) * g.inv_sx);
    int         cy = (int)floorf((py - g.c0[1]) * g.inv_sy);
    cx = min(max(cx, 0), g.ncx - 1);
    cy = min(max(cy, 0), g.ncy - 1);
    const int col = cx * g.ncy + cy;
    key[i]     = ((unsigned long long)(unsigned)col << 32) | float_key_host_device(pz);
    val[i]     = at;
}

// column populations from the sorted keys (no atomics, no contention on crowded columns): the
// atoms of column c are the keys in [c << 32, (c + 1) << 32).  Also the super-cells per column;
// entry ncxy is the zero that makes the exclusive scans yield the totals.
__device__ __forceinline__ int first_key_not_below(const unsigned long long *key, int n, unsigned long long v)
{
    int lo = 0, hi = n;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (key[mid] < v) { lo = mid + 1; } else { hi = mid; }
    }
    return lo;
}

__global__ void k_column_cells(int ncxy, int n, const unsigned long long *__restrict__ sorted_key, int *__restrict__ cxy_na,
                               int *__restrict__ ncz)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncxy) { return; }
    if (c == ncxy) { cxy_na[c] = 0; ncz[c] = 0; return; }
    const int na = first_key_not_below(sorted_key, n, (unsigned long long)(c + 1) << 32)
                 - first_key_not_below(sorted_key, n, (unsigned long long)c << 32);
    cxy_na[c] = na;
    ncz[c]    = (na + NA_SC - 1) / NA_SC;
}

// order-preserving map float -> unsigned (and back)
__device__ __forceinline__ unsigned float_key(float v)
{
    const unsigned u = __float_as_uint(v + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SortItem {
    float key;
    int   idx;
};

__device__ __forceinline__ bool item_less(const SortItem &a, const SortItem &b)
{
    return a.key < b.key || (a.key == b.key && a.idx < b.idx);
}

// ascending bitonic sort of one item per lane within aligned groups of N lanes
template <int N>
__device__ __forceinline__ void group_sort(SortItem &v, int lane)
{
    const int l = lane & (N - 1);
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
    {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
        {
            SortItem o;
            o.key = __shfl_xor(v.key, j);
            o.idx = __shfl_xor(v.idx, j);
            const bool up       = (l & k) == 0;
            const bool lower    = (l & j) == 0;
            const bool want_min = (up == lower);
            const bool o_less   = item_less(o, v);
            if (want_min == o_less) { v = o; }
        }
    }
}

// sort_columns_supersub + fill_cell for one super-cell per wave: the column's atoms arrive
// z-ordered; each half of 32 is ordered in y, each quarter of 16 in x; then bounding boxes per
// cluster, the atom-order arrays and the grid-ordered atom data with fillers
// (nbnxn_atomdata.c:288-295, 873, 937-951).
__global__ void __launch_bounds__(64)
k_supercells(int cell0, int nc, int ncxy, GridDims g, const int *__restrict__ cxy_ind, const int *__restrict__ cxy_na,
             const int *__restrict__ atom_start, const int *__restrict__ sorted_atom, const float *__restrict__ x,
             const float *__restrict__ q, const int *__restrict__ type, int dummy_type, int *__restrict__ a_out,
             int *__restrict__ cell_out, float4 *__restrict__ xq, int *__restrict__ type_out, float *__restrict__ bb,
             float *__restrict__ bbcz, float *__restrict__ bbsc, int *__restrict__ nsubc)
{
    const int sc   = blockIdx.x;          // zone-local; the per-cell arrays are indexed globally (cell0 + sc)
    const int scg  = cell0 + sc;
    const int lane = threadIdx.x;
    if (sc >= nc) { return; }
    // the column that holds this super-cell: the last c with cxy_ind[c] <= sc (empty columns repeat an index)
    int lo = 0, hi = ncxy;
    while (hi - lo > 1)
    {
        const int mid = (lo + hi) >> 1;
        if (cxy_ind[mid] <= sc) { lo = mid; } else { hi = mid; }
    }
    const int c    = lo;
    const int k    = sc - cxy_ind[c];
    const int na   = cxy_na[c];
    const int na_c = min(NA_SC, na - k * NA_SC);
    const int off  = k * NA_SC + lane;
    int       at   = off < na ? sorted_atom[atom_start[c] + off] : -1;

    // z-range of the super-cell from the z-ordered atoms
    const float pz_sorted = at >= 0 ? x[3 * at + 2] : 0.0f;
    const float z_first   = __shfl(pz_sorted, 0);
    const float z_last    = __shfl(pz_sorted, na_c - 1);

    // y within each z-half, then x within each y-half; every other sub-cell BACKWARDS, as the reference's
    // sort_atoms(..., Backwards) gives it (nbnxn_search.c:1448: sub_z & 1 for y; :1462: (cz * 2 + sub_y) & 1 = sub_y & 1
    // for x; :724-734: coordinate descending, ties by descending atom index, unused slots still last).  A backward
    // group sorts the negated coordinate and the mirrored index ascending.
    SortItem v;
    const bool back_y = ((lane >> 5) & 1) != 0;
    v.idx = at >= 0 ? (back_y ? INT_MAX - 1 - at : at) : INT_MAX;
    v.key = at >= 0 ? (back_y ? -x[3 * at + 1] : x[3 * at + 1]) : INFINITY;
    group_sort<32>(v, lane);
    at = v.idx != INT_MAX ? (back_y ? INT_MAX - 1 - v.idx : v.idx) : -1;
    const bool back_x = ((lane >> 4) & 1) != 0;
    v.idx = at >= 0 ? (back_x ? INT_MAX - 1 - at : at) : INT_MAX;
    v.key = at >= 0 ? (back_x ? -x[3 * at + 0] : x[3 * at + 0]) : INFINITY;
    group_sort<16>(v, lane);
    at = v.idx != INT_MAX ? (back_x ? INT_MAX - 1 - v.idx : v.idx) : -1;

    float px = 0, py = 0, pz = 0;
    if (at >= 0) { px = x[3 * at]; py = x[3 * at + 1]; pz = x[3 * at + 2]; }
    float blo[3] = {at >= 0 ? px : 1e30f, at >= 0 ? py : 1e30f, at >= 0 ? pz : 1e30f};
    float bhi[3] = {at >= 0 ? px : -1e30f, at >= 0 ? py : -1e30f, at >= 0 ? pz : -1e30f};
#pragma unroll
    for (int m = 1; m < NA_C; m <<= 1)
    {
#pragma unroll
        for (int d = 0; d < 3; d++)
        {
            blo[d] = fminf(blo[d], __shfl_xor(blo[d], m));
            bhi[d] = fmaxf(bhi[d], __shfl_xor(bhi[d], m));
        }
    }
    const int  cl_in_sc   = lane >> 3;
    const bool cl_has_atoms = (cl_in_sc * NA_C) < na_c;
    if ((lane & 7) == 0)
    {
        float *b = bb + ((size_t)scg * NSUBC + cl_in_sc) * 6;
#pragma unroll
        for (int d = 0; d < 3; d++)
        {
            b[d]     = cl_has_atoms ? blo[d] : 0.0f;
            b[3 + d] = cl_has_atoms ? bhi[d] : 0.0f;
        }
    }
    // box of the whole super-cell = union of its clusters' boxes (min / max are exact, so the order does not matter)
#pragma unroll
    for (int m = NA_C; m < NA_SC; m <<= 1)
    {
#pragma unroll
        for (int d = 0; d < 3; d++)
        {
            blo[d] = fminf(blo[d], __shfl_xor(blo[d], m));
            bhi[d] = fmaxf(bhi[d], __shfl_xor(bhi[d], m));
        }
    }
    if (lane == 0)
    {
        const int ns  = min(NSUBC, (na_c + NA_C - 1) / NA_C);
        nsubc[scg]     = ns;
        bbcz[2 * scg]     = z_first;
        bbcz[2 * scg + 1] = z_last;
#pragma unroll
        for (int d = 0; d < 3; d++) { bbsc[6 * scg + d] = blo[d]; bbsc[6 * scg + 3 + d] = bhi[d]; }
        // (the zone's cluster total is summed by k_zrange: one atomic per super-cell on a single counter cost
        // 240 us of this kernel's 290 at 1.53 M atoms)
    }

    const size_t gi = (size_t)scg * NA_SC + lane;
    a_out[gi]       = at;
    if (at >= 0)
    {
        cell_out[at] = (int)gi;
        xq[gi]       = make_float4(px, py, pz, q[at]);
        type_out[gi] = type[at];
    }
    else
    {
        // filler: its cluster's grid coordinates, distinct z per atom
        const int cx = c / g.ncy, cy = c - cx * g.ncy;
        const int gx = cx * 2 + ((lane >> 3) & 1), gy = cy * 2 + ((lane >> 4) & 1), gz = k * 2 + (lane >> 5);
        xq[gi]       = make_float4(-FAR_AWAY * (1 + NA_C * gx), -FAR_AWAY * (1 + NA_C * gy), -FAR_AWAY * (1 + NA_C * gz + (lane & 7)), 0.0f);
        type_out[gi] = dummy_type;
    }
}

// z-range of all atoms as order-preserving keys: the cells of a column are ordered in z, so the first and
// the last cell of every column suffice.  Also the number of clusters with atoms in the zone (nbnxn_search.c's
// grid->nsubc_tot, which sizes the list split).  One block; zrange was initialised to {max key, 0}.
__global__ void __launch_bounds__(256)
k_zrange(int cell0, int nc, int ncxy, const int *__restrict__ cxy_ind, const float *__restrict__ bbcz, const int *__restrict__ nsubc,
         unsigned *__restrict__ zrange, int *__restrict__ nsubc_tot)
{
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for (int c = threadIdx.x; c < ncxy; c += blockDim.x)
    {
        const int c0 = cxy_ind[c], c1 = cxy_ind[c + 1];
        if (c1 > c0)
        {
            kmin = min(kmin, float_key(bbcz[2 * (cell0 + c0)]));
            kmax = max(kmax, float_key(bbcz[2 * (cell0 + c1 - 1) + 1]));
        }
    }
    atomicMin(&zrange[0], kmin);
    atomicMax(&zrange[1], kmax);
    int sum = 0;
    for (int c = threadIdx.x; c < nc; c += blockDim.x) { sum += nsubc[cell0 + c]; }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { sum += __shfl_xor(sum, m); }
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) { *nsubc_tot = part[0] + part[1] + part[2] + part[3]; }
}

// ------------------------------------------------------------------------------------------
// list construction
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float sqr(float v) { return v * v; }

// distance^2 between two bounding boxes (nbnxn_search.c:2113-2144), a: {lo[3], hi[3]}
__device__ __forceinline__ float bb_dist2(const float *a, const float *b)
{
    float d2 = 0;
#pragma unroll
    for (int d = 0; d < 3; d++)
    {
        const float dl = a[d] - b[3 + d];
        const float dh = b[d] - a[3 + d];
        const float dm = fmaxf(fmaxf(dl, dh), 0.0f);
        d2 += dm * dm;
    }
    return d2;
}

// One wave per entry e = (i super-cell) * 14 + (shift slot).
// MODE 2 (the usual route): ONE walk.  ncj[e] = number of j-clusters the entry keeps; the j-clusters and their
//   i-masks go to the entry's own range of a scratch array (LIST_TMP_CAP slots per entry), the cluster pairs are
//   counted into nci_tot; k_list_compact then packs the ranges into cj4 groups once the scan has placed them.
//   An entry with more j-clusters than its range raises `overflow`, and the list is made the two-walk way:
// MODE 0: only ncj[e].  MODE 1: the cj4 groups of the entry are written at cj4_start[e] (the array was zeroed), the
//   groups holding the entry's own clusters are marked as needing exclusion words, cluster pairs counted.
constexpr int LIST_TMP_CAP = 512;   // default of nbx_search::list_cap (GMX_HIP_NB_LIST_TMP_CAP overrides it: tests of the other route)

template <int MODE>
__global__ void __launch_bounds__(64)
k_list(ListParams P, const int *__restrict__ nsubc, const float *__restrict__ bb,
       const float *__restrict__ bbcz, const float *__restrict__ bbsc, const unsigned *__restrict__ zrange,
       const float4 *__restrict__ xq, int *__restrict__ ncj,
       const int *__restrict__ cj4_start, nbx_cj4_t *__restrict__ cj4, int *__restrict__ need_excl,
       unsigned *__restrict__ nci_tot, int *__restrict__ tmp_cj, unsigned char *__restrict__ tmp_mb, int *__restrict__ overflow)
{
    constexpr bool FILL = (MODE == 1);
    constexpr bool TEMP = (MODE == 2);
    const int e    = blockIdx.x;
    const int lane = threadIdx.x;
    const Entry en = decode_entry(e, EntryMap{P.same, P.njz, P.wide});
    const int   sci = en.sci, tx = en.tx, ty = en.ty, tz = en.tz;
    const ZoneParams &Z = P.zp[en.jzi];
    const int *__restrict__ cxy_ind = Z.cxy_ind;
    const int   cell0 = Z.cell0;          // j super-cells: zone-local index + cell0 = global index
    const int nsub_i = nsubc[sci];
    int       iz     = 0;                 // the i-zone of this super-cell
    while (iz < MAX_IZONES - 1 && sci >= P.iz_end[iz]) { iz++; }
    const float rbb2_iz = P.rbb2[iz][en.jzi];
    const bool  active  = nsub_i > 0 && abs(tx) <= P.shp[0] && abs(ty) <= P.shp[1] && abs(tz) <= P.shp[2] && ((P.allowed[iz] >> en.jzi) & 1u);
    if (!active)
    {
        if (!FILL && lane == 0) { ncj[e] = 0; }
        return;
    }
    const int   shift = shift_index(tx, ty, tz);
    // shx, shy, shz (nbnxn_search.c:5019, :5054, :5098), the host builder's expressions term for term
    const float sh[3] = {tx * P.box[0] + ty * P.byx + tz * P.bzx, ty * P.box[1] + tz * P.bzy, tz * P.box[2]};
    const bool  central = P.same && (shift == NBX_CENTRAL);   // half list only within one grid

    // box of the whole i super-cell (unshifted union of its clusters' boxes, from the gridding pass), then shifted
    float bS[6];
#pragma unroll
    for (int d = 0; d < 3; d++) { bS[d] = bbsc[6 * sci + d] + sh[d]; bS[3 + d] = bbsc[6 * sci + 3 + d] + sh[d]; }
    const GridDims &g = Z.g;
    int cxf = (int)floorf((bS[0] - P.rlist - g.c0[0]) * g.inv_sx);
    int cxl = (int)floorf((bS[3] + P.rlist - g.c0[0]) * g.inv_sx);
    int cyf = (int)floorf((bS[1] - P.rlist - g.c0[1]) * g.inv_sy);
    int cyl = (int)floorf((bS[4] + P.rlist - g.c0[1]) * g.inv_sy);
    cxf = max(cxf, 0); cxl = min(cxl, g.ncx - 1);
    cyf = max(cyf, 0); cyl = min(cyl, g.ncy - 1);
    // Many (super-cell, shift) combinations of a large box reach nothing: leave before staging anything.  In x and y
    // that shows as an empty column window.  In z it is decided against the z-range of ALL atoms with a margin
    // (0.1 % of rlist + 1e-3 nm) far above float rounding: when the shifted cell is further than that from every
    // atom, the exact z-test below would reject every cell of every column as well, so the list is unchanged.
    const float zmargin = P.rlist * 1.001f + 1e-3f;
    const bool  none_in_z = (bS[5] + zmargin < key_float(zrange[2 * Z.zr])) || (bS[2] - zmargin > key_float(zrange[2 * Z.zr + 1]));
    if (cxf > cxl || cyf > cyl || none_in_z)
    {
        if (!FILL && lane == 0) { ncj[e] = 0; }
        return;
    }

    __shared__ float s_bbi[NSUBC][6];   // shifted i-cluster boxes
    __shared__ float s_xi[NA_SC][3];    // shifted i-atoms
    if (lane < NSUBC * 6)
    {
        const int ci = lane / 6, w = lane - ci * 6;
        s_bbi[ci][w] = bb[((size_t)sci * NSUBC + ci) * 6 + w] + sh[w % 3];
    }
    {
        const float4 xa = xq[(size_t)sci * NA_SC + lane];
        s_xi[lane][0] = xa.x + sh[0];
        s_xi[lane][1] = xa.y + sh[1];
        s_xi[lane][2] = xa.z + sh[2];
    }
    __syncthreads();

    int       running = 0;   // j-clusters kept so far (wave-uniform)
    unsigned  npairs  = 0;   // cluster pairs of this lane's j-clusters
    const int base4   = FILL ? cj4_start[e] : 0;

    // Stage 1, one lane per column of the (cxf..cxl) x (cyf..cyl) window, in the host builder's column order:
    // the column's distance in x and y and, since its cells are ordered in z, the run of cells the z-test
    // (the host builder's expression) lets through -- found by bisection on that very expression from both ends.
    // The lanes search their columns side by side, so the chain of dependent loads is one column long, not 25.
    // Stage 2 then walks the columns in order, the 64 lanes on 8 cells x 8 clusters at a time.
    const int ncoly = cyl - cyf + 1, ncol = (cxl - cxf + 1) * ncoly;
    for (int colbase = 0; colbase < ncol; colbase += 64)
    {
        int   my_lo = 0, my_hi = 0;
        float my_d2xy = 0;
        if (colbase + lane < ncol)
        {
            const int   ic = colbase + lane;
            const int   cx = cxf + ic / ncoly, cy = cyf + ic % ncoly;
            float       d2x   = 0;
            const float colx0 = g.c0[0] + cx * g.sx, colx1 = colx0 + g.sx;
            if (colx0 > bS[3]) { d2x = sqr(colx0 - bS[3]); }
            else if (colx1 < bS[0]) { d2x = sqr(bS[0] - colx1); }
            float       d2xy  = d2x;
            const float coly0 = g.c0[1] + cy * g.sy, coly1 = coly0 + g.sy;
            if (coly0 > bS[4]) { d2xy += sqr(coly0 - bS[4]); }
            else if (coly1 < bS[1]) { d2xy += sqr(bS[1] - coly1); }
            if (d2x < P.rl2 && d2xy < P.rl2)
            {
                int       c0 = cxy_ind[cx * g.ncy + cy];
                const int c1 = cxy_ind[cx * g.ncy + cy + 1];
                if (central) { c0 = max(c0, sci); }   // half list: j super-cells >= i (:5270-5283)
                int a = c0, b = c1;
                while (a < b)
                {
                    const int   mid = (a + b) >> 1;
                    const float z1  = bbcz[2 * (cell0 + mid) + 1];
                    if (z1 < bS[2] && d2xy + sqr(bS[2] - z1) >= P.rl2) { a = mid + 1; } else { b = mid; }
                }
                my_lo = a;
                b     = c1;
                while (a < b)
                {
                    const int   mid = (a + b) >> 1;
                    const float z0  = bbcz[2 * (cell0 + mid)];
                    if (z0 > bS[5] && d2xy + sqr(z0 - bS[5]) >= P.rl2) { b = mid; } else { a = mid + 1; }
                }
                my_hi   = a;
                my_d2xy = d2xy;
            }
        }
        const int nthis = min(64, ncol - colbase);
        for (int k = 0; k < nthis; k++)
        {
            const int   z_lo = __shfl(my_lo, k), z_hi = __shfl(my_hi, k);
            const float d2xy = __shfl(my_d2xy, k);
            for (int cb = z_lo; cb < z_hi; cb += 8)
            {
                const int scj = cb + (lane >> 3);
                const int cjo = lane & 7;
                unsigned  mb  = 0;
                if (scj < z_hi && cjo < nsubc[cell0 + scj])
                {
                    float       d2 = d2xy;
                    const float z0 = bbcz[2 * (cell0 + scj)], z1 = bbcz[2 * (cell0 + scj) + 1];
                    if (z0 > bS[5]) { d2 += sqr(z0 - bS[5]); }
                    else if (z1 < bS[2]) { d2 += sqr(bS[2] - z1); }
                    if (d2 < P.rl2)
                    {
                        const int cj = (cell0 + scj) * NSUBC + cjo;
                        float     bj[6];
#pragma unroll
                        for (int w = 0; w < 6; w++) { bj[w] = bb[(size_t)cj * 6 + w]; }
                        const int ci1 = (central && sci == scj) ? cjo + 1 : nsub_i;
                        int       np = 0, ci_last = -1;
                        float     d2_last = 0;
                        for (int ci = 0; ci < ci1; ci++)
                        {
                            const float d2c = bb_dist2(s_bbi[ci], bj);
                            if (d2c < P.rl2) { mb |= 1u << ci; ci_last = ci; d2_last = d2c; np++; }
                        }
                        // a lone cluster pair beyond the box-only radius gets the atom check (:3073-3092)
                        if (np == 1 && d2_last >= rbb2_iz)
                        {
                            // all eight j-atoms are requested at once (a loop with an early exit would wait for
                            // them one by one); "any pair in range" does not depend on the order of the tests
                            float4 xj[NA_C];
#pragma unroll
                            for (int j = 0; j < NA_C; j++) { xj[j] = xq[(size_t)cj * NA_C + j]; }
                            bool in_range = false;
                            for (int i = 0; i < NA_C; i++)
                            {
                                const float *xi = s_xi[ci_last * NA_C + i];
#pragma unroll
                                for (int j = 0; j < NA_C; j++)
                                {
                                    in_range |= (sqr(xi[0] - xj[j].x) + sqr(xi[1] - xj[j].y) + sqr(xi[2] - xj[j].z) < P.rl2);
                                }
                            }
                            if (!in_range) { mb = 0; }
                        }
                    }
                }
                const unsigned long long kept = __ballot(mb != 0);
                if (TEMP && mb != 0)
                {
                    const int slot = running + __popcll(kept & ((1ull << lane) - 1ull));
                    if (slot < P.tmp_cap)
                    {
                        tmp_cj[(size_t)e * P.tmp_cap + slot] = (cell0 + scj) * NSUBC + cjo;
                        tmp_mb[(size_t)e * P.tmp_cap + slot] = (unsigned char)mb;
                    }
                    npairs += __popc(mb);
                }
                if (FILL && mb != 0)
                {
                    const int slot = running + __popcll(kept & ((1ull << lane) - 1ull));
                    nbx_cj4_t *grp = cj4 + base4 + (slot >> 2);
                    grp->cj[slot & 3] = (cell0 + scj) * NSUBC + cjo;
                    reinterpret_cast<unsigned char *>(&grp->imei[0].imask)[slot & 3] = (unsigned char)mb;
                    reinterpret_cast<unsigned char *>(&grp->imei[1].imask)[slot & 3] = (unsigned char)mb;
                    if (central && sci == scj)
                    {
                        need_excl[2 * (base4 + (slot >> 2))]     = 1;
                        need_excl[2 * (base4 + (slot >> 2)) + 1] = 1;
                    }
                    npairs += __popc(mb);
                }
                running += __popcll(kept);
            }
        }
    }
    if (!FILL)
    {
        if (lane == 0)
        {
            ncj[e] = running;
            if (TEMP && running > P.tmp_cap) { atomicOr(overflow, 1); }
        }
    }
    if (FILL || TEMP)
    {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { npairs += __shfl_xor(npairs, m); }
        // 64 counters, not one: an atomic per entry on a single address serialises (hundreds of thousands of entries)
        if (lane == 0 && npairs) { atomicAdd(nci_tot + (e & 63), npairs); }
    }
}

// MODE 2's second half, one thread per cj4 group of the list: the group's entry is the last one that starts at or
// before it (entries without groups share their start with the next one that has some); slots 4g .. 4g+3 of that
// entry's scratch range become the group, whole (padding slots zero, no exclusion words yet), and the groups that
// hold the entry's own clusters are marked as needing exclusion words, exactly as MODE 1 leaves them.
__global__ void __launch_bounds__(256)
k_list_compact(EntryMap em, int cap, int nent, int ncj4, const int *__restrict__ ncj, const int *__restrict__ cj4_start,
               const int *__restrict__ tmp_cj, const unsigned char *__restrict__ tmp_mb, nbx_cj4_t *__restrict__ cj4,
               int *__restrict__ need_excl)
{
    const int g4 = blockIdx.x * 256 + threadIdx.x;
    if (g4 >= ncj4) { return; }
    int lo = 0, hi = nent - 1;
    while (lo < hi)
    {
        const int mid = (lo + hi + 1) >> 1;
        if (cj4_start[mid] <= g4) { lo = mid; } else { hi = mid - 1; }
    }
    const int   e       = lo;
    const int   n       = ncj[e];
    const int   gi      = g4 - cj4_start[e];
    const Entry en      = decode_entry(e, em);
    const bool  central = em.same && shift_index(en.tx, en.ty, en.tz) == NBX_CENTRAL;
    nbx_cj4_t   grp;
    unsigned    imask = 0;
    bool        own   = false;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int slot = 4 * gi + k;
        int       cj   = 0;
        if (slot < n)
        {
            cj = tmp_cj[(size_t)e * cap + slot];
            imask |= (unsigned)tmp_mb[(size_t)e * cap + slot] << (8 * k);
            own = own || (central && (cj / NSUBC) == en.sci);
        }
        grp.cj[k] = cj;
    }
    grp.imei[0].imask = imask; grp.imei[0].excl_ind = 0;
    grp.imei[1].imask = imask; grp.imei[1].excl_ind = 0;
    cj4[g4] = grp;
    if (own)
    {
        need_excl[2 * g4]     = 1;
        need_excl[2 * g4 + 1] = 1;
    }
}

__global__ void k_groups_of(int n, const int *__restrict__ ncj, int *__restrict__ ncj4)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > n) { return; }
    ncj4[e] = e < n ? (ncj[e] + 3) >> 2 : 0;
}

__device__ __forceinline__ int cj_at(const nbx_cj4_t *cj4, int ind) { return cj4[ind >> 2].cj[ind & 3]; }

// set_sci_top_excls, one wave per entry, one lane per i-atom.  APPLY = false marks the
// (cj4 group, half) pairs that need an exclusion word set; APPLY = true clears the bits (also
// the j > i masks of the entry's own clusters, set_self_and_newton_excls_supersub).
template <bool APPLY>
__device__ __forceinline__ void
excl_entry(int e, int lane, int natoms, EntryMap em, const int *__restrict__ ncj, const int *__restrict__ cj4_start,
           const nbx_cj4_t *__restrict__ cj4, const int *__restrict__ a, const int *__restrict__ cell,
           const int *__restrict__ excl_index, const int *__restrict__ excl_a, int *__restrict__ need_excl,
           nbx_excl_t *__restrict__ excl)
{
    const int n    = ncj[e];
    if (n == 0) { return; }
    const Entry en     = decode_entry(e, em);
    const int  sci     = en.sci;
    const bool central = em.same && shift_index(en.tx, en.ty, en.tz) == NBX_CENTRAL;
    const int  first   = cj4_start[e] * 4, last = first + n - 1;

    if (APPLY && central)
    {
        // the entry's own clusters are among its first slots (j super-cells start at i); lane -> (ej, ei >= ej)
        for (int s = first; s <= min(last, first + NSUBC - 1); s++)
        {
            const int cj = cj_at(cj4, s);
            if ((cj >> 3) != sci) { continue; }
            const int cjo = cj & 7;
            if (lane < NA_C * NA_C)
            {
                const int ej = lane >> 3, ei = lane & 7;
                if (ei >= ej)
                {
                    const int idx = cj4[s >> 2].imei[ej >> 2].excl_ind;
                    atomicAnd(&excl[idx].pair[(ej & 3) * NA_C + ei], ~(1u << ((s & 3) * NSUBC + cjo)));
                }
            }
        }
    }
    if (excl_index == nullptr) { return; }
    const int gi = sci * NA_SC + lane;
    const int ai = a[gi];
    if (ai < 0) { return; }
    const int cj_first = cj_at(cj4, first), cj_last = cj_at(cj4, last);
    const int si = lane >> 3;
    for (int k = excl_index[ai]; k < excl_index[ai + 1]; k++)
    {
        const int aj = excl_a[k];
        if (aj == ai || aj < 0 || aj >= natoms) { continue; }
        const int ge = cell[aj];
        if (ge < 0) { continue; }
        if (central && ge <= gi) { continue; }
        const int se = ge >> 3;
        if (se < cj_first || se > cj_last) { continue; }
        int lo = first, hi = last + 1, found = -1;
        while (lo < hi)
        {
            const int mid = (lo + hi) >> 1;
            const int cm  = cj_at(cj4, mid);
            if (cm == se) { found = mid; break; }
            if (se < cm) { hi = mid; } else { lo = mid + 1; }
        }
        if (found < 0) { continue; }
        const int      inner_i = lane & 7, inner_e = ge & 7;
        const unsigned bit     = 1u << ((found & 3) * NSUBC + si);
        if (cj4[found >> 2].imei[0].imask & bit)
        {
            if (!APPLY) { need_excl[2 * (found >> 2) + (inner_e >> 2)] = 1; }
            else
            {
                const int idx = cj4[found >> 2].imei[inner_e >> 2].excl_ind;
                atomicAnd(&excl[idx].pair[(inner_e & 3) * NA_C + inner_i], ~bit);
            }
        }
    }
}

// A wave takes EXCL_EPW consecutive entries: most (cell, shift) combinations of a large box hold nothing, and a wave
// per entry would be launched just to find that out (340 k waves at 1.53 M atoms, ~30 k of them with work).
constexpr int EXCL_EPW = 8;

template <bool APPLY>
__global__ void __launch_bounds__(64)
k_excl(int nent, int natoms, EntryMap em, const int *__restrict__ ncj, const int *__restrict__ cj4_start, const nbx_cj4_t *__restrict__ cj4,
       const int *__restrict__ a, const int *__restrict__ cell, const int *__restrict__ excl_index,
       const int *__restrict__ excl_a, int *__restrict__ need_excl, nbx_excl_t *__restrict__ excl)
{
    const int e0 = blockIdx.x * EXCL_EPW;
    const int e1 = min(nent, e0 + EXCL_EPW);
    for (int e = e0; e < e1; e++)
    {
        excl_entry<APPLY>(e, threadIdx.x, natoms, em, ncj, cj4_start, cj4, a, cell, excl_index, excl_a, need_excl, excl);
    }
}

// exclusion word sets: entry 0 is the shared all-ones set; flagged (group, half) pairs get
// 1 + their rank, written into the group and initialised to all ones
__global__ void k_excl_init(int n2, const int *__restrict__ need_excl, const int *__restrict__ rank, nbx_cj4_t *__restrict__ cj4,
                            nbx_excl_t *__restrict__ excl)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
    {
        for (int t = 0; t < 32; t++) { excl[0].pair[t] = 0xffffffffu; }
    }
    if (i >= n2 || !need_excl[i]) { return; }
    const int idx = 1 + rank[i];
    cj4[i >> 1].imei[i & 1].excl_ind = idx;
    for (int t = 0; t < 32; t++) { excl[idx].pair[t] = 0xffffffffu; }
}

// progressive balancing: the host builder's progressive_nsp_max (nbnxn_host.cpp), the same function
__device__ __forceinline__ int progressive_nsp_max(int nsp_max_av, int sci, int shift)
{
    if (nsp_max_av <= 0) { return nsp_max_av; }
    unsigned h = (unsigned)sci * 2654435761u + (unsigned)shift * 40503u;
    h ^= h >> 15;
    const int   cls = (int)(h & 7u);
    const float f   = cls < 6 ? 1.5f : (cls == 6 ? 0.75f : 0.4f);
    return max(NSUBC * 4 / 2 + 1, (int)(nsp_max_av * f));
}

// split_sci_entry, one thread per entry.  WRITE = false: nparts[e]; WRITE = true: the entry's
// nbnxn_sci_t parts at part_start[e].
template <bool WRITE>
__global__ void k_split(int nentries, EntryMap em, const int *__restrict__ ncj, const int *__restrict__ cj4_start,
                        const nbx_cj4_t *__restrict__ cj4, int nsp_max_av, int progressive, int *__restrict__ nparts,
                        const int *__restrict__ part_start, nbx_sci_t *__restrict__ sci_out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > nentries) { return; }
    if (e == nentries) { if (!WRITE) { nparts[e] = 0; } return; }
    const int n = ncj[e];
    if (n == 0) { if (!WRITE) { nparts[e] = 0; } return; }
    const Entry en  = decode_entry(e, em);
    const int sci   = en.sci;
    const int shift = shift_index(en.tx, en.ty, en.tz);
    const int nsp_max = progressive ? progressive_nsp_max(nsp_max_av, sci, shift) : nsp_max_av;
    const int start = cj4_start[e], end = start + ((n + 3) >> 2);
    const int j4len = end - start;
    int       count = 0;
    nbx_sci_t *out  = WRITE ? sci_out + part_start[e] : nullptr;
    auto emit = [&](int s0, int s1) {
        if (s1 > s0)
        {
            if (WRITE) { out[count].sci = sci; out[count].shift = shift; out[count].cj4_ind_start = s0; out[count].cj4_ind_end = s1; }
            count++;
        }
    };
    if (!(nsp_max > 0 && j4len > 1 && j4len * NSUBC * 4 > nsp_max))
    {
        emit(start, end);
    }
    else
    {
        // the reference's loop keeps all parts and balances the last two at the end; here the
        // second-to-last one is held back until it is known not to be the one that gets adjusted
        int  cur_start = start;
        int  pend0 = 0, pend1 = 0;
        bool have_pend = false;
        int  nsp = 0, nsp_sci = 0, nsp_cj4 = 0, nsp_cj4_e = 0;
        for (int c = start; c < end; c++)
        {
            const int nsp_cj4_p = nsp_cj4;
            nsp_cj4 = __popc(cj4[c].imei[0].imask);
            if (nsp_cj4 > 0 && nsp + nsp_cj4 > nsp_max)
            {
                if (have_pend) { emit(pend0, pend1); }
                pend0 = cur_start; pend1 = c; have_pend = true;
                cur_start = c;
                nsp_sci   = nsp;
                nsp_cj4_e = nsp_cj4_p;
                nsp       = 0;
            }
            nsp += nsp_cj4;
        }
        int last0 = cur_start;
        if (have_pend && nsp_sci - nsp_cj4_e >= nsp + nsp_cj4_e) { pend1--; last0--; }
        if (have_pend) { emit(pend0, pend1); }
        emit(last0, end);
    }
    if (!WRITE) { nparts[e] = count; }
}

// sort_sci (nbnxn_search.c:5517-5580) is a stable counting sort on a small key: entries with more cj4 groups first,
// counts clipped at m, m + 1 <= 64 keys.  One wave per 64 consecutive entries.  COUNT: hist[key][wave] = how many of
// the wave's entries have that key (key-major, so that ONE exclusive scan over the array gives every (key, wave) its
// first output position).  Otherwise: an entry goes to that position plus its rank among the wave's earlier entries
// with the same key.  For the few tens of thousands of entries of a list this replaces a library sort of ~60 us
// (five merge passes) by two short kernels and a scan.
__device__ __forceinline__ int sci_sort_key(const nbx_sci_t &e, int m) { return m - min(m, e.cj4_ind_end - e.cj4_ind_start); }

// second key (nbnxn_list_rules.h): the entry's cluster pairs in NBX_SCI_PAIR_KEYS buckets, most pairs first
__device__ __forceinline__ int sci_pair_key(const nbx_sci_t &e, const nbx_cj4_t *__restrict__ cj4, int m)
{
    int npair = 0;
    for (int j4 = e.cj4_ind_start; j4 < e.cj4_ind_end; j4++) { npair += __popc(cj4[j4].imei[0].imask | cj4[j4].imei[1].imask); }
    return nbx_sci_pair_key(e.cj4_ind_end - e.cj4_ind_start, npair, m);
}

// both keys in one for the library sort that takes over when m + 1 > 64
__global__ void k_sort_keys(int nsci, const nbx_sci_t *__restrict__ sci, const nbx_cj4_t *__restrict__ cj4, int m, int pair_key, unsigned *__restrict__ key)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsci) { return; }
    key[i] = (unsigned)sci_sort_key(sci[i], m) * NBX_SCI_PAIR_KEYS + (pair_key ? (unsigned)sci_pair_key(sci[i], cj4, m) : 0u);
}

// PAIRS: the pass on the second key (run FIRST: two stable passes, least significant key first, give the order by
// (group count, pair key)); kmax = the largest key of the pass (m, or NBX_SCI_PAIR_KEYS - 1)
template <bool COUNT, bool PAIRS>
__global__ void __launch_bounds__(64)
k_sort_sci(int nsci, int nwaves, int m, int kmax, const nbx_sci_t *__restrict__ sci, const nbx_cj4_t *__restrict__ cj4, int *__restrict__ hist,
           const int *__restrict__ first, nbx_sci_t *__restrict__ sci_out, unsigned *__restrict__ keybuf)
{
    const int  w    = blockIdx.x;
    const int  lane = threadIdx.x;
    const int  i    = w * 64 + lane;
    const bool have = i < nsci;
    nbx_sci_t  e;
    if (have) { e = sci[i]; }
    // (the pair key walks the entry's groups: the counting pass keeps it for the pass that moves the entries)
    int key = -1;
    if (have)
    {
        if (!PAIRS) { key = sci_sort_key(e, m); }
        else if (COUNT) { key = sci_pair_key(e, cj4, m); keybuf[i] = (unsigned)key; }
        else { key = (int)keybuf[i]; }
    }
    if (COUNT)
    {
        // lane k counts key k
        int mine = 0;
        for (int k = 0; k <= kmax; k++)
        {
            const unsigned long long b = __ballot(key == k);
            if (lane == k) { mine = __popcll(b); }
        }
        if (lane <= kmax) { hist[(size_t)lane * nwaves + w] = mine; }
    }
    else
    {
        unsigned long long same = 0;
        for (int k = 0; k <= kmax; k++)
        {
            const unsigned long long b = __ballot(key == k);
            if (key == k) { same = b; }
        }
        if (have)
        {
            const int rank = __popcll(same & ((1ull << lane) - 1ull));
            sci_out[first[(size_t)key * nwaves + w] + rank] = e;
        }
    }
}

inline int blocks_for(size_t n, int bs) { return (int)((n + bs - 1) / bs); }

} // namespace

struct Zone {
    GridDims g;
    int      a0 = 0, n = 0, cell0 = 0, nc = 0, ncxy = 0;
    float    density = 0;
    float    c1[3] = {0, 0, 0};
    Buf<int>                cxy_na, ncz, cxy_ind, atom_start, val_in, sorted_atom;
    Buf<unsigned long long> key_in, key_out;
    void release()
    {
        cxy_na.release(); ncz.release(); cxy_ind.release(); atom_start.release(); val_in.release(); sorted_atom.release();
        key_in.release(); key_out.release();
    }
};

struct nbx_search {
    float box[3] = {0, 0, 0};          // the diagonal
    float byx = 0, bzx = 0, bzy = 0;   // box[YY][XX], box[ZZ][XX], box[ZZ][YY]
    int   shp[3] = {1, 1, 1};
    int   nzones = 0, nzones_filled = 0;
    int   natoms = 0;      // atoms of all zones gridded so far
    int   nc     = 0;      // super-cells of all zones
    int   nsubc_tot0 = -1; // clusters with atoms in the home zone (read back when a list needs it)
    bool  progressive = true;   // progressive balancing of split lists, as in the host builder
    Zone  zone[MAX_ZONES];
    // per super-cell / cluster / atom, all zones, global numbering
    Buf<int>      a, cell, nsubc, counters;
    Buf<unsigned> nci_slots;   // 64 partial counts of the cluster pairs of the list being made
    Buf<float>    bb, bbcz, bbsc;
    Buf<unsigned> zrange;
    // list
    Buf<int>        ncj, ncj4_e, cj4_start, need_excl, excl_rank, nparts, part_start;
    Buf<nbx_cj4_t>  cj4;
    Buf<nbx_excl_t> excl;
    Buf<nbx_sci_t>  sci_a, sci_b;
    Buf<unsigned>   skey_in, skey_out;
    Buf<int>        sort_hist, sort_first;
    int                 list_cap = LIST_TMP_CAP;
    Buf<int>            tmp_cj;   // k_list MODE 2: list_cap j-clusters per entry ...
    Buf<unsigned char>  tmp_mb;   // ... and their i-cluster masks
    nbx_prim_tmp    tmp;
    // pinned landing area for the handful of integers the host needs (array sizes): a device-to-host copy into
    // pageable memory is staged and costs ~25 us each, into pinned memory a third of that
    int            *h_pin = nullptr;
};

nbx_search *nbx_search_create()
{
    nbx_search *s = new nbx_search();
    if (const char *env = getenv("GMX_HIP_NB_LIST_TMP_CAP")) { s->list_cap = std::max(4, atoi(env)); }
    if (hipHostMalloc((void **)&s->h_pin, 96 * sizeof(int), hipHostMallocPortable) != hipSuccess) { s->h_pin = nullptr; }
    return s;
}

void nbx_search_destroy(nbx_search *s)
{
    if (!s) { return; }
    for (Zone &z : s->zone) { z.release(); }
    s->a.release(); s->cell.release(); s->nsubc.release(); s->counters.release(); s->nci_slots.release();
    s->bb.release(); s->bbcz.release(); s->bbsc.release(); s->zrange.release();
    s->ncj.release(); s->ncj4_e.release(); s->cj4_start.release(); s->need_excl.release(); s->excl_rank.release();
    s->nparts.release(); s->part_start.release(); s->cj4.release(); s->excl.release(); s->sci_a.release(); s->sci_b.release();
    s->skey_in.release(); s->skey_out.release(); s->sort_hist.release(); s->sort_first.release(); s->tmp_cj.release(); s->tmp_mb.release();
    nbx_prim_free(s->tmp);
    if (s->h_pin) { (void)hipHostFree(s->h_pin); }
    delete s;
}

const char *nbx_search_error() { return g_err.c_str(); }

void nbx_search_set_progressive_balance(nbx_search *s, bool on) { if (s) { s->progressive = on; } }

int nbx_search_grid_columns(nbx_search *s, hipStream_t st, int zone, const float box[9], const int pbc_shift_dim[3],
                            const float corner0[3], const float corner1[3], int a0, int n, const float *d_x, float atom_density)
{
    if (!s) { return sfail("null search handle"); }
    if (zone < 0 || zone >= MAX_ZONES) { return sfail("zone %d outside [0,%d)", zone, MAX_ZONES); }
    if (zone == 0)
    {
        for (int d = 0; d < 3; d++)
        {
            if (!(box[4 * d] > 0)) { return sfail("box[%d][%d] = %g must be > 0", d, d, box[4 * d]); }
        }
        // check_box (gmxlib/pbc.c:121-156), as the host-side builder applies it
        if (box[1] != 0 || box[2] != 0 || box[5] != 0)
        {
            return sfail("Only triclinic boxes with the first vector parallel to the x-axis and the second vector in the xy-plane are supported.");
        }
        const float margin = 1.0010f;   // BOX_MARGIN, pbc.c:68
        if (std::fabs(box[3]) > margin * 0.5f * box[0] || std::fabs(box[6]) > margin * 0.5f * box[0] || std::fabs(box[7]) > margin * 0.5f * box[4])
        {
            return sfail("Triclinic box is too skewed.");
        }
        if ((box[3] != 0 && !(pbc_shift_dim[0] && pbc_shift_dim[1]))
            || ((box[6] != 0 || box[7] != 0) && !(pbc_shift_dim[0] && pbc_shift_dim[1] && pbc_shift_dim[2])))
        {
            return sfail("a triclinic box vector needs periodic shifts in the dimensions it couples");
        }
        for (int d = 0; d < 3; d++)
        {
            s->box[d] = box[4 * d];
            s->shp[d] = pbc_shift_dim[d] ? 1 : 0;
        }
        s->byx = box[3]; s->bzx = box[6]; s->bzy = box[7];
        s->nzones = 0; s->nzones_filled = 0; s->natoms = 0; s->nc = 0; s->nsubc_tot0 = -1;
        if (s->counters.reserve(32)) { return 1; }
        S_TRY(hipMemsetAsync(s->counters.p, 0, 32 * sizeof(int), st));
    }
    if (zone != s->nzones) { return sfail("zones must be gridded in order: got %d, expected %d", zone, s->nzones); }
    if (s->nzones_filled != 0) { return sfail("all zones get their columns before the first one is filled"); }
    if (n <= 0 || !d_x) { return sfail("zone %d has no atoms", zone); }
    if (a0 != s->natoms) { return sfail("zone %d starts at atom %d but %d atoms are gridded", zone, a0, s->natoms); }
    Zone &z = s->zone[zone];
    float size[3];
    for (int d = 0; d < 3; d++)
    {
        z.g.c0[d] = corner0[d];
        z.c1[d]   = corner1[d];
        size[d]   = corner1[d] - corner0[d];
        if (!(size[d] > 0)) { return sfail("zone %d has no extent along %d", zone, d); }
    }
    // set_grid_size_xy (:433-486): columns of 2x2 roughly cubic sub-cells, as the host builder sizes them
    if (atom_density <= 0) { atom_density = std::max(n, 1) / (size[0] * size[1] * size[2]); }
    z.density = atom_density;
    GridDims &g = z.g;
    if (n > NA_SC)
    {
        const float tlen = std::cbrt(NA_C / atom_density);
        g.ncx = std::max(1, (int)(size[0] / (tlen * 2)));
        g.ncy = std::max(1, (int)(size[1] / (tlen * 2)));
    }
    else { g.ncx = 1; g.ncy = 1; }
    g.sx = size[0] / g.ncx; g.inv_sx = 1 / g.sx;
    g.sy = size[1] / g.ncy; g.inv_sy = 1 / g.sy;
    const int ncxy = g.ncx * g.ncy;
    z.ncxy = ncxy; z.a0 = a0; z.n = n;

    if (z.cxy_na.reserve(ncxy + 1) || z.ncz.reserve(ncxy + 1) || z.cxy_ind.reserve(ncxy + 1) || z.atom_start.reserve(ncxy + 1)
        || z.key_in.reserve(n) || z.key_out.reserve(n) || z.val_in.reserve(n) || z.sorted_atom.reserve(n)) { return 1; }
    hipLaunchKernelGGL(k_columns, dim3(blocks_for(n, 256)), dim3(256), 0, st, a0, n, d_x, g, z.key_in.p, z.val_in.p);
    S_TRY(hipGetLastError());
    int col_bits = 1;
    while ((1 << col_bits) < ncxy) { col_bits++; }
    S_TRY(nbx_prim_sort_u64_int(s->tmp, z.key_in.p, z.key_out.p, z.val_in.p, z.sorted_atom.p, n, 32 + col_bits, st));
    hipLaunchKernelGGL(k_column_cells, dim3(blocks_for(ncxy + 1, 256)), dim3(256), 0, st, ncxy, n, z.key_out.p, z.cxy_na.p, z.ncz.p);
    S_TRY(hipGetLastError());
    S_TRY(nbx_prim_exclusive_sum(s->tmp, z.ncz.p, z.cxy_ind.p, ncxy + 1, st));
    S_TRY(nbx_prim_exclusive_sum(s->tmp, z.cxy_na.p, z.atom_start.p, ncxy + 1, st));
    int nc = 0;
    if (!s->h_pin) { return sfail("no pinned host memory for the search"); }
    S_TRY(hipMemcpyAsync(&s->h_pin[0], z.cxy_ind.p + ncxy, sizeof(int), hipMemcpyDeviceToHost, st));
    S_TRY(hipStreamSynchronize(st));
    nc = s->h_pin[0];
    z.nc    = nc;
    z.cell0 = s->nc;
    s->nc += nc;
    s->natoms += n;
    s->nzones = zone + 1;
    return 0;
}

int nbx_search_natoms_padded(const nbx_search *s) { return s && s->nzones > 0 ? s->nc * NA_SC : -1; }
int nbx_search_natoms_padded_local(const nbx_search *s) { return s && s->nzones > 0 ? s->zone[0].nc * NA_SC : -1; }
int nbx_search_nzones(const nbx_search *s) { return s ? s->nzones : 0; }

int nbx_search_grid_fill(nbx_search *s, hipStream_t st, int zone, const float *d_x, const float *d_q, const int *d_type,
                         int ntype_with_dummy, float4 *d_xq, int *d_type_out)
{
    if (!s || zone < 0 || zone >= s->nzones) { return sfail("grid_fill of zone %d before its grid_columns", zone); }
    if (zone != s->nzones_filled) { return sfail("zones are filled in order: got %d, expected %d", zone, s->nzones_filled); }
    if (zone == 0)
    {
        const size_t npad = (size_t)s->nc * NA_SC;
        if (s->a.reserve(npad) || s->cell.reserve(s->natoms) || s->nsubc.reserve(s->nc) || s->bb.reserve((size_t)s->nc * NSUBC * 6)
            || s->bbcz.reserve((size_t)s->nc * 2) || s->bbsc.reserve((size_t)s->nc * 6) || s->zrange.reserve(2 * MAX_ZONES)) { return 1; }
        static unsigned init[2 * MAX_ZONES];   // {min = all ones, max = 0} per zone
        for (int i = 0; i < MAX_ZONES; i++) { init[2 * i] = 0xffffffffu; init[2 * i + 1] = 0u; }
        S_TRY(hipMemcpyAsync(s->zrange.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    }
    Zone &z = s->zone[zone];
    hipLaunchKernelGGL(k_supercells, dim3(z.nc), dim3(64), 0, st, z.cell0, z.nc, z.ncxy, z.g, z.cxy_ind.p, z.cxy_na.p, z.atom_start.p,
                       z.sorted_atom.p, d_x, d_q, d_type, ntype_with_dummy - 1, s->a.p, s->cell.p, d_xq, d_type_out, s->bb.p,
                       s->bbcz.p, s->bbsc.p, s->nsubc.p);
    hipLaunchKernelGGL(k_zrange, dim3(1), dim3(256), 0, st, z.cell0, z.nc, z.ncxy, z.cxy_ind.p, s->bbcz.p, s->nsubc.p, s->zrange.p + 2 * zone,
                       s->counters.p + 2 + zone);
    S_TRY(hipGetLastError());
    s->nzones_filled = zone + 1;
    return 0;
}

const int *nbx_search_atom_order(const nbx_search *s) { return s->a.p; }
const int *nbx_search_cell_index(const nbx_search *s) { return s->cell.p; }

int nbx_search_grid_info(const nbx_search *s, int zone, int *ncx, int *ncy, int *nc)
{
    if (!s || zone < 0 || zone >= s->nzones) { return sfail("no such zone"); }
    *ncx = s->zone[zone].g.ncx; *ncy = s->zone[zone].g.ncy; *nc = s->zone[zone].nc;
    return 0;
}

int nbx_search_download_grid(const nbx_search *s, hipStream_t st, int *a, int *cell, float *bb, float *bbcz, int *nsubc)
{
    if (!s || s->nzones == 0 || s->nzones_filled != s->nzones) { return sfail("no grid"); }
    if (a) { S_TRY(hipMemcpyAsync(a, s->a.p, (size_t)s->nc * NA_SC * sizeof(int), hipMemcpyDeviceToHost, st)); }
    if (cell) { S_TRY(hipMemcpyAsync(cell, s->cell.p, (size_t)s->natoms * sizeof(int), hipMemcpyDeviceToHost, st)); }
    if (bb) { S_TRY(hipMemcpyAsync(bb, s->bb.p, (size_t)s->nc * NSUBC * 6 * sizeof(float), hipMemcpyDeviceToHost, st)); }
    if (bbcz) { S_TRY(hipMemcpyAsync(bbcz, s->bbcz.p, (size_t)s->nc * 2 * sizeof(float), hipMemcpyDeviceToHost, st)); }
    if (nsubc) { S_TRY(hipMemcpyAsync(nsubc, s->nsubc.p, (size_t)s->nc * sizeof(int), hipMemcpyDeviceToHost, st)); }
    S_TRY(hipStreamSynchronize(st));
    return 0;
}

namespace {

// tuning aid: NBNXN_NL_SPLIT_SCALE multiplies the pair budget of non-local list entries (both builders read it)
static double nonlocal_split_scale()
{
    static const double v = [] { const char *e = getenv("NBNXN_NL_SPLIT_SCALE"); const double x = e ? atof(e) : 1.0; return x > 0 ? x : 1.0; }();
    return v;
}

// get_nsubpair_max (nbnxn_search.c:4324-4408), estimate from the home grid, as in the host builder
int nsubpair_max(const nbx_search &s, float rlist, int min_ci_balanced, double frac_of_local)
{
    const Zone &z = s.zone[0];
    if (min_ci_balanced <= 0 || z.nc >= min_ci_balanced || z.nc == 0) { return -1; }
    double ls[3];
    ls[0] = (z.c1[0] - z.g.c0[0]) / (z.g.ncx * 2);
    ls[1] = (z.c1[1] - z.g.c0[1]) / (z.g.ncy * 2);
    ls[2] = (z.c1[2] - z.g.c0[2]) * z.g.ncx * z.g.ncy / (z.nc * 2.0);
    const double diag2 = ls[0] * ls[0] + ls[1] * ls[1] + ls[2] * ls[2];
    const float  f     = (NA_C - 1.0f) / NA_C;
    const double r_eff = rlist + 0.5 * (f * f) * std::sqrt(diag2 / 3);
    double       vol   = ls[0] * ls[1] * ls[2];
    vol += (ls[0] * ls[1] + ls[0] * ls[2] + ls[1] * ls[2]) * r_eff;
    vol += 2 * (ls[0] + ls[1] + ls[2]) * 0.25 * M_PI * r_eff * r_eff;
    vol += 0.5 * 4.0 / 3.0 * M_PI * r_eff * r_eff * r_eff;
    const double nsp_est = s.nsubc_tot0 * vol * z.density / NA_C * frac_of_local;
    const int    nsp     = std::max(1, (int)(nsp_est / min_ci_balanced + 0.5));
    return nsp + NSUBC * 4 / 2;
}

} // namespace

int nbx_search_make_pairlist(nbx_search *s, hipStream_t st, int iloc, int nzp, const int *zone_pairs, float rlist, int min_ci_balanced,
                             const int *d_excl_index, const int *d_excl_a, const float4 *d_xq, nbx_search_list_t *out)
{
    if (!s || s->nzones == 0 || s->nzones_filled != s->nzones) { return sfail("no grid: put the atoms on the grid first"); }
    if (!(rlist > 0)) { return sfail("rlist must be > 0"); }
    const bool same = (iloc == NBX_LOCAL);
    if (!same && s->nzones < 2)
    {
        // no halo zone: an empty non-local list
        out->nsci = out->ncj4 = out->nci_tot = 0; out->nexcl = 0;
        out->sci = nullptr; out->cj4 = nullptr; out->excl = nullptr;
        return 0;
    }
    for (int d = 0; d < 3; d++)
    {
        if (s->shp[d] && s->box[d] < 2 * rlist)
        {
            return sfail("box[%d] = %g is shorter than 2*rlist = %g: more than one periodic image in range", d, s->box[d], 2 * rlist);
        }
    }
    if (s->shp[1] && s->shp[2] && s->box[1] - std::fabs(s->bzy) < rlist)
    {
        return sfail("box[YY][YY] - |box[ZZ][YY]| = %g is shorter than rlist = %g", s->box[1] - std::fabs(s->bzy), rlist);
    }
    ListParams  P;
    memset(&P, 0, sizeof(P));
    P.same = same ? 1 : 0;
    P.njz  = same ? 1 : s->nzones - 1;
    // Which i-zones meet which j-zone slots.  Without zone pairs: the home zone against itself (local) or against
    // every halo zone (non-local).  With them (the reference's dd_zp tables, domdec.c:433-449, as the host builder
    // takes them): the local list is {0, 0, 1}; in the non-local list i-zones 0 .. 3 meet halo zones only, never
    // their own, so no pair of that list is within one grid and every entry visits all shifts.
    int niz = 1;
    if (nzp <= 0 || !zone_pairs)
    {
        P.allowed[0] = same ? 1u : ((1u << (s->nzones - 1)) - 1u);
    }
    else
    {
        for (int z = 0; z < nzp; z++)
        {
            const int iz = zone_pairs[3 * z], j0 = zone_pairs[3 * z + 1], j1 = zone_pairs[3 * z + 2];
            if (iz < 0 || iz >= s->nzones || j0 < 0 || j1 > s->nzones || j0 > j1)
            {
                return sfail("zone pair %d = {%d, %d, %d} outside the %d gridded zones", z, iz, j0, j1, s->nzones);
            }
            // one row per i-zone, ascending (domdec.c:433-449), as the host builder demands: both then give the same list
            if (z > 0 && iz <= zone_pairs[3 * (z - 1)])
            {
                return sfail("zone pair %d: i-zone %d after i-zone %d: the table lists every i-zone once, ascending", z, iz, zone_pairs[3 * (z - 1)]);
            }
            if (same)
            {
                if (nzp != 1 || iz != 0 || j0 != 0 || j1 != 1) { return sfail("the local list is the zone pair {0, 0, 1}"); }
                P.allowed[0] = 1u;
                continue;
            }
            if (iz >= MAX_IZONES) { return sfail("zone pair %d: i-zone %d, at most %d i-zones", z, iz, MAX_IZONES); }
            if (j1 > j0 && (j0 < 1 || (iz >= j0 && iz < j1)))
            {
                return sfail("zone pair %d = {%d, %d, %d}: the non-local list pairs i-zones with other halo zones only", z, iz, j0, j1);
            }
            for (int j = j0; j < j1; j++) { P.allowed[iz] |= 1u << (j - 1); }
            if (j1 > j0) { niz = std::max(niz, iz + 1); }
        }
    }
    for (int k = 0; k < MAX_IZONES; k++)
    {
        const int zk = std::min(k, niz - 1);
        P.iz_end[k]  = s->zone[zk].cell0 + s->zone[zk].nc;
    }
    P.nci = P.iz_end[niz - 1];
    for (int d = 0; d < 3; d++) { P.box[d] = s->box[d]; P.shp[d] = s->shp[d]; }
    P.byx = s->byx; P.bzx = s->bzx; P.bzy = s->bzy;
    // shift range (nbnxn_search.c:4906-4927): a skewed box can need two box lengths in x
    if (P.shp[0] && s->box[0] - std::fabs(s->byx) - std::fabs(s->bzx) < rlist) { P.shp[0] = 2; }
    P.wide = (P.shp[0] == 2) ? 1 : 0;
    const EntryMap em{P.same, P.njz, P.wide};
    P.rlist = rlist;
    P.rl2   = rlist * rlist;
    for (int k = 0; k < P.njz; k++)
    {
        const int   zid = same ? 0 : k + 1;
        const Zone &zj  = s->zone[zid];
        ZoneParams &zp  = P.zp[k];
        zp.g = zj.g; zp.cxy_ind = zj.cxy_ind.p; zp.cell0 = zj.cell0; zp.zr = zid;
        for (int iz = 0; iz < niz; iz++)
        {
            // boundingbox_only_distance2 (:4721-4757), float arithmetic as in the host builder
            const Zone &zi  = s->zone[iz];
            const float bbx = 0.5f * (zi.g.sx + zj.g.sx) / 2, bby = 0.5f * (zi.g.sy + zj.g.sy) / 2;
            const float r   = std::max(0.0f, rlist - 0.5f * std::sqrt(bbx * bbx + bby * bby));
            P.rbb2[iz][k]   = r * r;
        }
    }
    const int nent = P.nci * entries_per_cell(em);
    int      *cnt  = s->counters.p;   // [2 + z] clusters with atoms of zone z, [20] scratch overflow flag

    if (s->ncj.reserve(nent + 1) || s->ncj4_e.reserve(nent + 1) || s->cj4_start.reserve(nent + 1) || s->nparts.reserve(nent + 1)
        || s->part_start.reserve(nent + 1)) { return 1; }
    // one walk over the grid per entry, into a scratch range per entry (MODE 2 of k_list); the two-walk route only
    // when an entry outgrows its range.  The scratch is not cleared: the compaction reads what ncj[e] says was written.
    P.tmp_cap = s->list_cap;
    const bool one_walk = (size_t)nent * P.tmp_cap * (sizeof(int) + 1) <= ((size_t)4 << 30);
    if (one_walk && (s->tmp_cj.reserve((size_t)nent * P.tmp_cap) || s->tmp_mb.reserve((size_t)nent * P.tmp_cap))) { return 1; }
    int *overflow = cnt + 20;
    if (s->nci_slots.reserve(64)) { return 1; }
    S_TRY(hipMemsetAsync(s->nci_slots.p, 0, 64 * sizeof(unsigned), st));
    S_TRY(hipMemsetAsync(overflow, 0, sizeof(int), st));
    if (one_walk)
    {
        hipLaunchKernelGGL((k_list<2>), dim3(nent), dim3(64), 0, st, P, s->nsubc.p, s->bb.p, s->bbcz.p, s->bbsc.p, s->zrange.p, d_xq, s->ncj.p,
                           (const int *)nullptr, (nbx_cj4_t *)nullptr, (int *)nullptr, s->nci_slots.p, s->tmp_cj.p, s->tmp_mb.p, overflow);
    }
    else
    {
        hipLaunchKernelGGL((k_list<0>), dim3(nent), dim3(64), 0, st, P, s->nsubc.p, s->bb.p, s->bbcz.p, s->bbsc.p, s->zrange.p, d_xq, s->ncj.p,
                           (const int *)nullptr, (nbx_cj4_t *)nullptr, (int *)nullptr, (unsigned *)nullptr, (int *)nullptr,
                           (unsigned char *)nullptr, (int *)nullptr);
    }
    hipLaunchKernelGGL(k_groups_of, dim3(blocks_for(nent + 1, 256)), dim3(256), 0, st, nent, s->ncj.p, s->ncj4_e.p);
    S_TRY(hipGetLastError());
    S_TRY(nbx_prim_exclusive_sum(s->tmp, s->ncj4_e.p, s->cj4_start.p, nent + 1, st));
    int h1[2] = {0, 0};
    if (!s->h_pin) { return sfail("no pinned host memory for the search"); }
    S_TRY(hipMemcpyAsync(&s->h_pin[1], s->cj4_start.p + nent, sizeof(int), hipMemcpyDeviceToHost, st));
    S_TRY(hipMemcpyAsync(&s->h_pin[2], cnt + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    S_TRY(hipMemcpyAsync(&s->h_pin[7], overflow, sizeof(int), hipMemcpyDeviceToHost, st));
    S_TRY(hipStreamSynchronize(st));
    h1[0] = s->h_pin[1]; h1[1] = s->h_pin[2];
    const int ncj4  = h1[0];
    s->nsubc_tot0   = h1[1];
    const int nsp_max = nsubpair_max(*s, rlist, min_ci_balanced, same ? 1.0 : 0.25 * nonlocal_split_scale());

    if (s->cj4.reserve((size_t)ncj4 + 1) || s->need_excl.reserve(2 * (size_t)ncj4 + 1) || s->excl_rank.reserve(2 * (size_t)ncj4 + 1)) { return 1; }
    S_TRY(hipMemsetAsync(s->need_excl.p, 0, (2 * (size_t)ncj4 + 1) * sizeof(int), st));
    if (one_walk && s->h_pin[7] == 0)
    {
        if (ncj4 > 0)
        {
            hipLaunchKernelGGL(k_list_compact, dim3(blocks_for(ncj4, 256)), dim3(256), 0, st, em, P.tmp_cap, nent, ncj4, s->ncj.p, s->cj4_start.p,
                               s->tmp_cj.p, s->tmp_mb.p, s->cj4.p, s->need_excl.p);
        }
    }
    else
    {
        S_TRY(hipMemsetAsync(s->nci_slots.p, 0, 64 * sizeof(unsigned), st));   // the pair count of a walk that overflowed is not used
        S_TRY(hipMemsetAsync(s->cj4.p, 0, ((size_t)ncj4 + 1) * sizeof(nbx_cj4_t), st));
        hipLaunchKernelGGL((k_list<1>), dim3(nent), dim3(64), 0, st, P, s->nsubc.p, s->bb.p, s->bbcz.p, s->bbsc.p, s->zrange.p, d_xq, s->ncj.p,
                           s->cj4_start.p, s->cj4.p, s->need_excl.p, s->nci_slots.p, (int *)nullptr, (unsigned char *)nullptr, (int *)nullptr);
    }
    hipLaunchKernelGGL((k_excl<false>), dim3(blocks_for(nent, EXCL_EPW)), dim3(64), 0, st, nent, s->natoms, em, s->ncj.p, s->cj4_start.p, s->cj4.p, s->a.p, s->cell.p,
                       d_excl_index, d_excl_a, s->need_excl.p, (nbx_excl_t *)nullptr);
    hipLaunchKernelGGL((k_split<false>), dim3(blocks_for(nent + 1, 128)), dim3(128), 0, st, nent, em, s->ncj.p, s->cj4_start.p, s->cj4.p,
                       nsp_max, s->progressive ? 1 : 0, s->nparts.p, (const int *)nullptr, (nbx_sci_t *)nullptr);
    S_TRY(hipGetLastError());
    S_TRY(nbx_prim_exclusive_sum(s->tmp, s->need_excl.p, s->excl_rank.p, 2 * ncj4 + 1, st));
    S_TRY(nbx_prim_exclusive_sum(s->tmp, s->nparts.p, s->part_start.p, nent + 1, st));
    int h2[3] = {0, 0, 0};
    S_TRY(hipMemcpyAsync(&s->h_pin[3], s->excl_rank.p + 2 * ncj4, sizeof(int), hipMemcpyDeviceToHost, st));
    S_TRY(hipMemcpyAsync(&s->h_pin[4], s->part_start.p + nent, sizeof(int), hipMemcpyDeviceToHost, st));
    S_TRY(hipMemcpyAsync(&s->h_pin[16], s->nci_slots.p, 64 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    S_TRY(hipStreamSynchronize(st));
    h2[0] = s->h_pin[3]; h2[1] = s->h_pin[4]; h2[2] = 0;
    for (int k = 0; k < 64; k++) { h2[2] += s->h_pin[16 + k]; }
    const int nexcl = 1 + h2[0], nsci = h2[1], nci_tot = h2[2];

    if (s->excl.reserve(nexcl) || s->sci_a.reserve((size_t)nsci + 1) || s->sci_b.reserve((size_t)nsci + 1) || s->skey_in.reserve((size_t)nsci + 1)
        || s->skey_out.reserve((size_t)nsci + 1)) { return 1; }
    hipLaunchKernelGGL(k_excl_init, dim3(blocks_for(2 * (size_t)ncj4 + 1, 256)), dim3(256), 0, st, 2 * ncj4, s->need_excl.p, s->excl_rank.p,
                       s->cj4.p, s->excl.p);
    hipLaunchKernelGGL((k_excl<true>), dim3(blocks_for(nent, EXCL_EPW)), dim3(64), 0, st, nent, s->natoms, em, s->ncj.p, s->cj4_start.p, s->cj4.p, s->a.p, s->cell.p,
                       d_excl_index, d_excl_a, (int *)nullptr, s->excl.p);
    hipLaunchKernelGGL((k_split<true>), dim3(blocks_for(nent + 1, 128)), dim3(128), 0, st, nent, em, s->ncj.p, s->cj4_start.p, s->cj4.p,
                       nsp_max, s->progressive ? 1 : 0, (int *)nullptr, s->part_start.p, s->sci_a.p);
    S_TRY(hipGetLastError());
    const nbx_sci_t *sci_final = s->sci_a.p;
    // sort_sci (:5517-5580): only when the groups outnumber the entries
    if (ncj4 > nsci && nsci > 0)
    {
        const int m = (int)(2 * (size_t)ncj4 / (size_t)nsci);
        // second key (nbnxn_list_rules.h): only for lists that were split for the GPU's balance with the progressive classes
        const bool pair_key = nsp_max > 0 && s->progressive;
        if (m + 1 <= 64)
        {
            const int    nwaves = (nsci + 63) / 64;
            const size_t nh2    = (size_t)NBX_SCI_PAIR_KEYS * nwaves, nh = (size_t)(m + 1) * nwaves;
            if (s->sort_hist.reserve(nh2 + 1) || s->sort_first.reserve(nh2 + 1)) { return 1; }
            // the pair key first (sci_a -> sci_b), then the group count (sci_b -> sci_a): both passes are stable
            const nbx_sci_t *src = s->sci_a.p;
            nbx_sci_t       *dst = s->sci_b.p;
            if (pair_key)
            {
                hipLaunchKernelGGL((k_sort_sci<true, true>), dim3(nwaves), dim3(64), 0, st, nsci, nwaves, m, NBX_SCI_PAIR_KEYS - 1, src, s->cj4.p,
                                   s->sort_hist.p, (const int *)nullptr, (nbx_sci_t *)nullptr, s->skey_in.p);
                S_TRY(hipGetLastError());
                S_TRY(nbx_prim_exclusive_sum(s->tmp, s->sort_hist.p, s->sort_first.p, (int)nh2, st));
                hipLaunchKernelGGL((k_sort_sci<false, true>), dim3(nwaves), dim3(64), 0, st, nsci, nwaves, m, NBX_SCI_PAIR_KEYS - 1, src, s->cj4.p,
                                   (int *)nullptr, s->sort_first.p, dst, s->skey_in.p);
                S_TRY(hipGetLastError());
                src = s->sci_b.p; dst = s->sci_a.p;
            }
            hipLaunchKernelGGL((k_sort_sci<true, false>), dim3(nwaves), dim3(64), 0, st, nsci, nwaves, m, m, src, s->cj4.p, s->sort_hist.p,
                               (const int *)nullptr, (nbx_sci_t *)nullptr, (unsigned *)nullptr);
            S_TRY(hipGetLastError());
            S_TRY(nbx_prim_exclusive_sum(s->tmp, s->sort_hist.p, s->sort_first.p, (int)nh, st));
            hipLaunchKernelGGL((k_sort_sci<false, false>), dim3(nwaves), dim3(64), 0, st, nsci, nwaves, m, m, src, s->cj4.p, (int *)nullptr,
                               s->sort_first.p, dst, (unsigned *)nullptr);
            S_TRY(hipGetLastError());
            sci_final = dst;
        }
        else
        {
            int bits = 1;
            while ((1 << bits) <= (m + 1) * NBX_SCI_PAIR_KEYS) { bits++; }
            hipLaunchKernelGGL(k_sort_keys, dim3(blocks_for(nsci, 256)), dim3(256), 0, st, nsci, s->sci_a.p, s->cj4.p, m, pair_key ? 1 : 0, s->skey_in.p);
            S_TRY(hipGetLastError());
            S_TRY(nbx_prim_sort_u32_sci(s->tmp, s->skey_in.p, s->skey_out.p, s->sci_a.p, s->sci_b.p, nsci, bits, st));
            sci_final = s->sci_b.p;
        }
    }
    out->nsci = nsci; out->ncj4 = ncj4; out->nexcl = nexcl; out->nci_tot = nci_tot;
    out->sci = sci_final; out->cj4 = s->cj4.p; out->excl = s->excl.p;
    return 0;
}
