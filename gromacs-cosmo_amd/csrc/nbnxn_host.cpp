// Host-side producers of the GPU cluster-pair list format (see include/nbnxn_host.h).
//
// Written from scratch; what it must reproduce is the *format and semantics* of
// GROMACS 5.0.4's nbnxn gridding and list construction (src/gromacs/mdlib/nbnxn_search.c,
// nbnxn_atomdata.c), cited per function below, not its code structure.
#include "nbnxn_host.h"
#include "nbnxn_list_rules.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

constexpr int   NA_C      = NBX_CLUSTER_SIZE;                  // atoms per cluster
constexpr int   NSUBC     = NBX_NCL_PER_SUPERCL;               // clusters per super-cluster
constexpr int   NA_SC     = NA_C * NSUBC;                      // atoms per super-cluster
constexpr float FAR_AWAY  = 107.0f;                            // nbnxn_atomdata.c:256
constexpr unsigned ALL_BITS = 0xffffffffu;

struct BB { float lo[3], hi[3]; };

struct Grid {
    float            c0[3], c1[3];
    int              ncx = 0, ncy = 0;
    float            sx = 0, sy = 0, inv_sx = 0, inv_sy = 0;
    float            density = 0;
    int              cell0 = 0;       // first super-cell of this grid in the global numbering
    int              nc    = 0;       // number of super-cells
    std::vector<int> cxy_ind;         // column -> first super-cell (grid local), size ncx*ncy+1
    std::vector<int> cxy_na;          // atoms per column
    std::vector<int> nsubc;           // non-empty clusters per super-cell
    std::vector<BB>  bb;              // per cluster (nc*8)
    std::vector<float> bbcz;          // per super-cell z-range (2 each)
    std::vector<int> gx, gy, gz;      // per cluster: grid coordinates used for the filler positions
    int              nsubc_tot = 0;
};

struct List {
    std::vector<nbx_sci_t>  sci;
    std::vector<nbx_cj4_t>  cj4;
    std::vector<nbx_excl_t> excl;
    int                     nci_tot = 0;
    float                   rlist   = 0;
    nbx_pairlist_t          view;
};

} // namespace

struct nbnxn_host {
    float             box[3]      = {0, 0, 0};   // the diagonal
    // off-diagonal elements of a triclinic box (rows are box vectors, GROMACS layout: box[YY][XX],
    // box[ZZ][XX], box[ZZ][YY]; the upper triangle is zero by check_box, gmxlib/pbc.c:135)
    float             byx = 0, bzx = 0, bzy = 0;
    int               shp[3]      = {1, 1, 1};   // which dimensions get periodic shifts (1) or none (0)
    std::vector<Grid> grids;
    std::vector<int>  a;      // grid index -> atom
    std::vector<int>  cell;   // atom -> grid index
    List              lists[2];
    bool              progressive = true;   // progressive balancing of split lists (progressive_nsp_max)
};

namespace {

inline float sqr(float x) { return x * x; }

// distance^2 between two boxes (nbnxn_search.c:2113-2144, subc_bb_dist2)
inline float bb_dist2(const BB &a, const BB &b)
{
    float d2 = 0;
    for (int d = 0; d < 3; d++)
    {
        float dl = a.lo[d] - b.hi[d];
        float dh = b.lo[d] - a.hi[d];
        float dm = std::max(std::max(dl, dh), 0.0f);
        d2 += dm * dm;
    }
    return d2;
}

// Sort `idx[0..n)` by coordinate dim ascending; ties by atom index so that the order does
// not depend on the input order (the reference sorts for "binary reproducibility",
// nbnxn_search.c:679-687).  backwards: the reference's sort_atoms(..., Backwards = TRUE), which reads its
// sorted slots from the top down (:724-734): coordinate descending, ties by DEScending atom index.
void sort_dim(int *idx, int n, const float *x, int a0, int dim, bool backwards = false)
{
    if (backwards)
    {
        std::sort(idx, idx + n, [&](int p, int q) {
            float xp = x[3 * (p - a0) + dim], xq = x[3 * (q - a0) + dim];
            return xp > xq || (xp == xq && p > q);
        });
        return;
    }
    std::sort(idx, idx + n, [&](int p, int q) {
        float xp = x[3 * (p - a0) + dim], xq = x[3 * (q - a0) + dim];
        return xp < xq || (xp == xq && p < q);
    });
}

} // namespace

extern "C" {

const char *nbnxn_host_last_error(void) { return g_err.c_str(); }

nbnxn_host_ptr_t nbnxn_host_create(void) { return new nbnxn_host(); }

void nbnxn_host_destroy(nbnxn_host_ptr_t h) { delete h; }

int nbnxn_host_set_box(nbnxn_host_ptr_t h, const float box[3], const int pbc_shift_dim[3])
{
    if (!h) { return fail("null handle"); }
    const float m[9] = {box[0], 0, 0, 0, box[1], 0, 0, 0, box[2]};
    return nbnxn_host_set_box_triclinic(h, m, pbc_shift_dim);
}

int nbnxn_host_set_box_triclinic(nbnxn_host_ptr_t h, const float box[9], const int pbc_shift_dim[3])
{
    if (!h) { return fail("null handle"); }
    for (int d = 0; d < 3; d++)
    {
        if (!(box[4 * d] > 0)) { return fail("box[%d][%d] = %g must be > 0", d, d, box[4 * d]); }
    }
    // check_box (gmxlib/pbc.c:121-156)
    if (box[1] != 0 || box[2] != 0 || box[5] != 0)
    {
        return fail("Only triclinic boxes with the first vector parallel to the x-axis and the second vector in the xy-plane are supported.");
    }
    const float margin = 1.0010f;   // BOX_MARGIN, pbc.c:68
    if (std::fabs(box[3]) > margin * 0.5f * box[0] || std::fabs(box[6]) > margin * 0.5f * box[0] || std::fabs(box[7]) > margin * 0.5f * box[4])
    {
        return fail("Triclinic box is too skewed.");
    }
    if ((box[3] != 0 && !(pbc_shift_dim[0] && pbc_shift_dim[1])) || ((box[6] != 0 || box[7] != 0) && !(pbc_shift_dim[0] && pbc_shift_dim[1] && pbc_shift_dim[2])))
    {
        return fail("a triclinic box vector needs periodic shifts in the dimensions it couples");
    }
    for (int d = 0; d < 3; d++)
    {
        h->box[d] = box[4 * d];
        h->shp[d] = pbc_shift_dim[d] ? 1 : 0;
    }
    h->byx = box[3]; h->bzx = box[6]; h->bzy = box[7];
    h->grids.clear();
    h->a.clear();
    h->cell.clear();
    return 0;
}

// nbnxn_put_on_grid (nbnxn_search.c:1777) = set_grid_size_xy (:433-486) + calc_cell_indices
// (:1588) + sort_columns_supersub (:1374-1490) + fill_cell (:1164).
int nbnxn_host_put_on_grid(nbnxn_host_ptr_t h, int zone,
                           const float corner0[3], const float corner1[3],
                           int a0, int n, const float *x, float atom_density)
{
    if (!h) { return fail("null handle"); }
    if (zone != (int)h->grids.size()) { return fail("zones must be gridded in order: got %d, expected %zu", zone, h->grids.size()); }
    if (n < 0) { return fail("negative atom count"); }
    if (zone == 0 && a0 != 0) { return fail("home zone must start at atom 0"); }
    if ((int)h->cell.size() != a0) { return fail("zone %d starts at atom %d but %zu atoms are gridded", zone, a0, h->cell.size()); }

    Grid g;
    float size[3];
    for (int d = 0; d < 3; d++)
    {
        g.c0[d] = corner0[d];
        g.c1[d] = corner1[d];
        size[d] = corner1[d] - corner0[d];
        if (!(size[d] > 0)) { return fail("zone %d has no extent along %d", zone, d); }
    }
    if (atom_density <= 0)
    {
        atom_density = std::max(n, 1) / (size[0] * size[1] * size[2]);
    }
    g.density = atom_density;

    // set_grid_size_xy: columns of 2x2 approximately cubic sub-cells, rounded down
    if (n > NA_SC)
    {
        float tlen = std::cbrt(NA_C / atom_density);
        g.ncx = std::max(1, (int)(size[0] / (tlen * 2)));
        g.ncy = std::max(1, (int)(size[1] / (tlen * 2)));
    }
    else
    {
        g.ncx = 1;
        g.ncy = 1;
    }
    g.sx = size[0] / g.ncx; g.inv_sx = 1 / g.sx;
    g.sy = size[1] / g.ncy; g.inv_sy = 1 / g.sy;

    const int ncxy = g.ncx * g.ncy;
    g.cxy_na.assign(ncxy, 0);
    g.cxy_ind.assign(ncxy + 1, 0);

    // calc_column_indices (:1493-1590): clamp for rounding at the upper edge; halo atoms may
    // lie outside the zone in x/y and are clamped as well.
    std::vector<int> col(n);
    for (int i = 0; i < n; i++)
    {
        int cx = (int)std::floor((x[3 * i + 0] - g.c0[0]) * g.inv_sx);
        int cy = (int)std::floor((x[3 * i + 1] - g.c0[1]) * g.inv_sy);
        cx     = std::min(std::max(cx, 0), g.ncx - 1);
        cy     = std::min(std::max(cy, 0), g.ncy - 1);
        col[i] = cx * g.ncy + cy;
        g.cxy_na[col[i]]++;
    }
    for (int c = 0; c < ncxy; c++)
    {
        g.cxy_ind[c + 1] = g.cxy_ind[c] + (g.cxy_na[c] + NA_SC - 1) / NA_SC;
    }
    g.nc    = g.cxy_ind[ncxy];
    g.cell0 = h->grids.empty() ? 0 : h->grids.back().cell0 + h->grids.back().nc;

    const size_t gbase = (size_t)g.cell0 * NA_SC;
    h->a.resize(gbase + (size_t)g.nc * NA_SC, -1);
    h->cell.resize(a0 + n, -1);
    g.nsubc.assign(g.nc, 0);
    g.bb.assign((size_t)g.nc * NSUBC, BB{{0, 0, 0}, {0, 0, 0}});
    g.bbcz.assign((size_t)g.nc * 2, 0.0f);
    g.gx.assign((size_t)g.nc * NSUBC, 0);
    g.gy.assign((size_t)g.nc * NSUBC, 0);
    g.gz.assign((size_t)g.nc * NSUBC, 0);

    // bucket atoms per column
    std::vector<int> fillpos(ncxy);
    for (int c = 0; c < ncxy; c++) { fillpos[c] = (g.cell0 + g.cxy_ind[c]) * NA_SC; }
    for (int i = 0; i < n; i++) { h->a[fillpos[col[i]]++] = a0 + i; }

    // sort_columns_supersub: z over the column, then per 64: 2 z-halves, y into 2, x into 2
#pragma omp parallel for schedule(dynamic, 4)
    for (int c = 0; c < ncxy; c++)
    {
        const int cx  = c / g.ncy, cy = c % g.ncy;
        const int na  = g.cxy_na[c];
        const int ncz = g.cxy_ind[c + 1] - g.cxy_ind[c];
        int      *col_a = h->a.data() + (size_t)(g.cell0 + g.cxy_ind[c]) * NA_SC;
        sort_dim(col_a, na, x, a0, 2);
        for (int sub_z = 0; sub_z < ncz * 2; sub_z++)
        {
            const int ash_z = sub_z * 32;
            const int na_z  = std::min(32, na - ash_z);
            if (sub_z % 2 == 0)
            {
                const int sc   = g.cxy_ind[c] + sub_z / 2;
                const int na_c = std::min(NA_SC, na - ash_z);
                g.nsubc[sc]    = std::min(NSUBC, (na_c + NA_C - 1) / NA_C);
                g.bbcz[2 * sc]     = x[3 * (col_a[ash_z] - a0) + 2];
                g.bbcz[2 * sc + 1] = x[3 * (col_a[ash_z + na_c - 1] - a0) + 2];
            }
            if (na_z <= 0) { continue; }
            // y within a z-half, x within a y-half: every other sub-cell is sorted BACKWARDS, so that consecutive
            // sub-cells snake through the super-cell (sort_columns_supersub, nbnxn_search.c:1448 and :1462:
            // Backwards = sub_z & 1 for y, (cz * GPU_NSUBCELL_Y + sub_y) & 1 = sub_y & 1 for x)
            sort_dim(col_a + ash_z, na_z, x, a0, 1, (sub_z & 1) != 0);
            for (int sub_y = 0; sub_y < 2; sub_y++)
            {
                const int ash_y = ash_z + sub_y * 16;
                const int na_y  = std::min(16, na - ash_y);
                if (na_y <= 0) { continue; }
                sort_dim(col_a + ash_y, na_y, x, a0, 0, (sub_y & 1) != 0);
                for (int sub_x = 0; sub_x < 2; sub_x++)
                {
                    const int ash_x = ash_y + sub_x * NA_C;
                    const int na_x  = std::min(NA_C, na - ash_x);
                    const int cl    = (g.cxy_ind[c] * NA_SC + ash_x) / NA_C;   // grid-local cluster
                    g.gx[cl] = cx * 2 + sub_x;
                    g.gy[cl] = cy * 2 + sub_y;
                    g.gz[cl] = sub_z;
                    if (na_x <= 0) { continue; }
                    BB b;
                    for (int d = 0; d < 3; d++) { b.lo[d] = 1e30f; b.hi[d] = -1e30f; }
                    for (int k = 0; k < na_x; k++)
                    {
                        const float *xa = x + 3 * (col_a[ash_x + k] - a0);
                        for (int d = 0; d < 3; d++)
                        {
                            b.lo[d] = std::min(b.lo[d], xa[d]);
                            b.hi[d] = std::max(b.hi[d], xa[d]);
                        }
                    }
                    g.bb[cl] = b;
                }
            }
        }
        // filler clusters beyond the last atom of the column also need grid coordinates
        for (int k = 0; k < ncz * NSUBC; k++)
        {
            const int cl = g.cxy_ind[c] * NSUBC + k;
            const int sub_z = k / 4, sub_y = (k / 2) & 1, sub_x = k & 1;
            g.gx[cl] = cx * 2 + sub_x;
            g.gy[cl] = cy * 2 + sub_y;
            g.gz[cl] = sub_z;
        }
        for (int k = na; k < ncz * NA_SC; k++) { col_a[k] = -1; }
    }
    g.nsubc_tot = std::accumulate(g.nsubc.begin(), g.nsubc.end(), 0);

    for (size_t gi = gbase; gi < h->a.size(); gi++)
    {
        if (h->a[gi] >= 0) { h->cell[h->a[gi]] = (int)gi; }
    }
    h->grids.push_back(std::move(g));
    return 0;
}

int nbnxn_host_natoms_padded(nbnxn_host_ptr_t h) { return h ? (int)h->a.size() : -1; }

int nbnxn_host_natoms_padded_local(nbnxn_host_ptr_t h)
{
    if (!h || h->grids.empty()) { return -1; }
    return h->grids[0].nc * NA_SC;
}

int nbnxn_host_ncolumns(nbnxn_host_ptr_t h, int zone, int *ncx, int *ncy)
{
    if (!h || zone < 0 || zone >= (int)h->grids.size()) { return fail("bad zone"); }
    *ncx = h->grids[zone].ncx;
    *ncy = h->grids[zone].ncy;
    return 0;
}

int nbnxn_host_nsupercells(nbnxn_host_ptr_t h) { return h ? (int)h->a.size() / NA_SC : -1; }

const int *nbnxn_host_atom_order(nbnxn_host_ptr_t h) { return h->a.data(); }
const int *nbnxn_host_cell_index(nbnxn_host_ptr_t h) { return h->cell.data(); }

// nbnxn_atomdata_set (nbnxn_atomdata.c:1094): types (:873) and charges (:937-951)
int nbnxn_host_set_atomdata(nbnxn_host_ptr_t h, int ntype_with_dummy,
                            const int *type, const float *q, float *xq, int *type_out)
{
    if (!h) { return fail("null handle"); }
    const size_t np = h->a.size();
    for (size_t gi = 0; gi < np; gi++)
    {
        const int at = h->a[gi];
        if (at >= 0)
        {
            if (type[at] < 0 || type[at] >= ntype_with_dummy - 1)
            {
                return fail("atom %d has type %d outside [0,%d)", at, type[at], ntype_with_dummy - 1);
            }
            type_out[gi]   = type[at];
            xq[4 * gi + 3] = q[at];
        }
        else
        {
            type_out[gi]   = ntype_with_dummy - 1;
            xq[4 * gi + 3] = 0.0f;
        }
    }
    return 0;
}

// nbnxn_atomdata_copy_x_to_nbat_x (nbnxn_atomdata.c:1141); filler placement :288-295
int nbnxn_host_copy_x_to_xq(nbnxn_host_ptr_t h, const float *x, float *xq)
{
    if (!h) { return fail("null handle"); }
    for (const Grid &g : h->grids)
    {
#pragma omp parallel for schedule(static)
        for (int cl = 0; cl < g.nc * NSUBC; cl++)
        {
            const size_t gi0 = ((size_t)g.cell0 * NSUBC + cl) * NA_C;
            for (int k = 0; k < NA_C; k++)
            {
                const size_t gi = gi0 + k;
                const int    at = h->a[gi];
                if (at >= 0)
                {
                    xq[4 * gi + 0] = x[3 * at + 0];
                    xq[4 * gi + 1] = x[3 * at + 1];
                    xq[4 * gi + 2] = x[3 * at + 2];
                }
                else
                {
                    xq[4 * gi + 0] = -FAR_AWAY * (1 + NA_C * g.gx[cl]);
                    xq[4 * gi + 1] = -FAR_AWAY * (1 + NA_C * g.gy[cl]);
                    xq[4 * gi + 2] = -FAR_AWAY * (1 + NA_C * g.gz[cl] + k);
                }
            }
        }
    }
    return 0;
}

// nbnxn_atomdata_add_nbat_f_to_f (nbnxn_atomdata.c:1600, inner loop :1329-1336)
int nbnxn_host_add_f(nbnxn_host_ptr_t h, const float *fnb, int a0, int n, float *f)
{
    if (!h) { return fail("null handle"); }
    if (a0 < 0 || a0 + n > (int)h->cell.size()) { return fail("atom range out of bounds"); }
#pragma omp parallel for schedule(static)
    for (int at = a0; at < a0 + n; at++)
    {
        const size_t gi = h->cell[at];
        f[3 * at + 0] += fnb[3 * gi + 0];
        f[3 * at + 1] += fnb[3 * gi + 1];
        f[3 * at + 2] += fnb[3 * gi + 2];
    }
    return 0;
}

int nbnxn_host_shift_vec(nbnxn_host_ptr_t h, float *sv)
{
    if (!h) { return fail("null handle"); }
    // XYZ2IS ordering, legacyheaders/types/ishift.h:45-55: x fastest (-2..2), then y, then z
    int is = 0;
    for (int tz = -1; tz <= 1; tz++)
    {
        for (int ty = -1; ty <= 1; ty++)
        {
            for (int tx = -2; tx <= 2; tx++)
            {
                // calc_shifts (gmxlib/pbc.c:1326-1349): k box[XX] + l box[YY] + m box[ZZ], in float
                sv[3 * is + 0] = tx * h->box[0] + ty * h->byx + tz * h->bzx;
                sv[3 * is + 1] = ty * h->box[1] + tz * h->bzy;
                sv[3 * is + 2] = tz * h->box[2];
                is++;
            }
        }
    }
    return 0;
}

} // extern "C"

namespace {

// Work for one thread's slice of i super-cells: its own sci/cj4/excl arrays with local
// numbering, concatenated afterwards (the reference's combine_nblists, nbnxn_search.c:4466).
struct PartList {
    std::vector<nbx_sci_t>  sci;
    std::vector<nbx_cj4_t>  cj4;
    std::vector<nbx_excl_t> excl;   // excl[0] unused locally (shared all-ones entry is global 0)
    int                     nci_tot = 0;
    int                     cj_ind  = 0;   // running j-cluster slot (4 per cj4)
};

inline int new_excl(PartList &pl)
{
    nbx_excl_t e;
    for (int t = 0; t < 32; t++) { e.pair[t] = ALL_BITS; }
    pl.excl.push_back(e);
    return (int)pl.excl.size();   // local index + 1; 0 means "none"
}

inline nbx_excl_t &get_excl(PartList &pl, int cj4_ind, int warp)
{
    int &ei = pl.cj4[cj4_ind].imei[warp].excl_ind;
    if (ei == 0) { ei = new_excl(pl); }
    return pl.excl[ei - 1];
}

// make_cluster_list_supersub (nbnxn_search.c:2955-3123)
void make_cluster_list(const nbnxn_host &h, const Grid &gi, const Grid &gj, PartList &pl,
                       int sci, int scj, bool sci_equals_scj, const BB *bb_ci,
                       const float *x_ci, const float *xq, float rl2, float rbb2)
{
    (void)h;
    float d2l[NSUBC];
    for (int cjo = 0; cjo < gj.nsubc[scj]; cjo++)
    {
        const int cj4_ind   = pl.cj_ind >> 2;
        const int cj_offset = pl.cj_ind & 3;
        if (cj4_ind >= (int)pl.cj4.size())
        {
            nbx_cj4_t z;
            memset(&z, 0, sizeof(z));
            pl.cj4.push_back(z);
        }
        const int cj    = scj * NSUBC + cjo;              // grid-local j cluster
        const int cj_gl = gj.cell0 * NSUBC + cj;
        unsigned  imask = 0;
        const int ci1   = sci_equals_scj ? cjo + 1 : gi.nsubc[sci];
        int       npair = 0, ci_last = -1;
        for (int ci = 0; ci < ci1; ci++)
        {
            d2l[ci] = bb_dist2(bb_ci[ci], gj.bb[cj]);
            if (d2l[ci] < rl2)
            {
                imask |= 1u << (cj_offset * NSUBC + ci);
                ci_last = ci;
                npair++;
            }
        }
        // single-pair entries get the atom-distance check (PRUNE_LIST_CPU_ONE, :3073-3092)
        if (npair == 1 && d2l[ci_last] >= rbb2)
        {
            bool in_range = false;
            for (int i = 0; i < NA_C && !in_range; i++)
            {
                const float *xi = x_ci + (ci_last * NA_C + i) * 3;
                for (int j = 0; j < NA_C; j++)
                {
                    const float *xj = xq + 4 * ((size_t)cj_gl * NA_C + j);
                    if (sqr(xi[0] - xj[0]) + sqr(xi[1] - xj[1]) + sqr(xi[2] - xj[2]) < rl2)
                    {
                        in_range = true;
                        break;
                    }
                }
            }
            if (!in_range)
            {
                imask &= ~(1u << (cj_offset * NSUBC + ci_last));
                npair--;
            }
        }
        if (npair > 0)
        {
            nbx_cj4_t &e = pl.cj4[cj4_ind];
            e.cj[cj_offset] = cj_gl;
            if (sci_equals_scj)
            {
                // set_self_and_newton_excls_supersub (:2759-2780): keep only j > i
                nbx_excl_t *ex[2] = { nullptr, nullptr };
                // take both entries first: get_excl may grow the vector
                get_excl(pl, cj4_ind, 0);
                get_excl(pl, cj4_ind, 1);
                ex[0] = &pl.excl[pl.cj4[cj4_ind].imei[0].excl_ind - 1];
                ex[1] = &pl.excl[pl.cj4[cj4_ind].imei[1].excl_ind - 1];
                for (int ej = 0; ej < NA_C; ej++)
                {
                    for (int ei = ej; ei < NA_C; ei++)
                    {
                        ex[ej >> 2]->pair[(ej & 3) * NA_C + ei] &= ~(1u << (cj_offset * NSUBC + cjo));
                    }
                }
            }
            pl.cj4[cj4_ind].imei[0].imask |= imask;
            pl.cj4[cj4_ind].imei[1].imask |= imask;
            pl.cj_ind++;
            pl.nci_tot += npair;
            pl.sci.back().cj4_ind_end = (pl.cj_ind + 3) >> 2;
        }
    }
}

// set_sci_top_excls (nbnxn_search.c:3673-3807)
void set_top_excls(const nbnxn_host &h, PartList &pl, bool diag_removed,
                   const int *excl_index, const int *excl_a)
{
    const nbx_sci_t &e = pl.sci.back();
    if (e.cj4_ind_end == e.cj4_ind_start) { return; }
    const int sci          = e.sci;
    const int cj_ind_first = e.cj4_ind_start * 4;
    const int cj_ind_last  = pl.cj_ind - 1;
    auto      cj_at        = [&](int ind) { return pl.cj4[ind >> 2].cj[ind & 3]; };
    const int cj_first     = cj_at(cj_ind_first);
    const int cj_last      = cj_at(cj_ind_last);
    for (int i = 0; i < NA_SC; i++)
    {
        const int ai = h.a[(size_t)sci * NA_SC + i];
        if (ai < 0) { continue; }
        const int si = i >> 3;
        for (int eind = excl_index[ai]; eind < excl_index[ai + 1]; eind++)
        {
            const int aj = excl_a[eind];
            if (aj == ai) { continue; }
            if (aj < 0 || aj >= (int)h.cell.size()) { continue; }   // partner not on this rank's grids
            const int ge = h.cell[aj];
            if (ge < 0) { continue; }
            if (diag_removed && ge <= sci * NA_SC + i) { continue; }
            const int se = ge >> 3;
            if (se < cj_first || se > cj_last) { continue; }
            // the j-clusters of an entry are ascending: bisect
            int lo = cj_ind_first, hi = cj_ind_last + 1, found = -1;
            while (lo < hi)
            {
                const int mid = (lo + hi) >> 1;
                const int cm  = cj_at(mid);
                if (cm == se) { found = mid; break; }
                if (se < cm) { hi = mid; } else { lo = mid + 1; }
            }
            if (found < 0) { continue; }
            const int      inner_i = i - si * NA_C;
            const int      inner_e = ge - se * NA_C;
            const unsigned bit     = 1u << ((found & 3) * NSUBC + si);
            if (pl.cj4[found >> 2].imei[0].imask & bit)
            {
                nbx_excl_t &ex = get_excl(pl, found >> 2, inner_e >> 2);
                ex.pair[(inner_e & 3) * NA_C + inner_i] &= ~bit;
            }
        }
    }
}

// split_sci_entry (nbnxn_search.c:3936-4024), without the progressive-balance variant
void split_last_sci(PartList &pl, int nsp_max)
{
    const nbx_sci_t last = pl.sci.back();
    const int j4len = last.cj4_ind_end - last.cj4_ind_start;
    if (!(j4len > 1 && j4len * NSUBC * 4 > nsp_max)) { return; }
    pl.sci.pop_back();
    nbx_sci_t cur = last;
    cur.cj4_ind_end = cur.cj4_ind_start;
    int nsp = 0, nsp_sci = 0, nsp_cj4 = 0, nsp_cj4_e = 0;
    std::vector<nbx_sci_t> parts;
    for (int c = last.cj4_ind_start; c < last.cj4_ind_end; c++)
    {
        const int nsp_cj4_p = nsp_cj4;
        nsp_cj4 = __builtin_popcount(pl.cj4[c].imei[0].imask);
        if (nsp_cj4 > 0 && nsp + nsp_cj4 > nsp_max)
        {
            cur.cj4_ind_end = c;
            parts.push_back(cur);
            cur.cj4_ind_start = c;
            nsp_sci   = nsp;
            nsp_cj4_e = nsp_cj4_p;
            nsp       = 0;
        }
        nsp += nsp_cj4;
    }
    cur.cj4_ind_end = last.cj4_ind_end;
    parts.push_back(cur);
    // balance the last two parts by moving one cj4 (:4013-4020)
    if (parts.size() >= 2 && nsp_sci - nsp_cj4_e >= nsp + nsp_cj4_e)
    {
        parts[parts.size() - 2].cj4_ind_end--;
        parts[parts.size() - 1].cj4_ind_start--;
    }
    for (const nbx_sci_t &p : parts)
    {
        if (p.cj4_ind_end > p.cj4_ind_start) { pl.sci.push_back(p); }
    }
}

// Progressive balancing.  The reference makes the entries of a list "progressively smaller, which improves load
// balancing" (split_sci_entry with progBal, nbnxn_search.c:3928-3960) from a running entry count, which only a
// sequential builder has.  The same end is reached here with a size class per (i super-cell, shift): three
// quarters of the entries are split at 1.5x the average pair budget, an eighth at 0.75x and an eighth at 0.4x.
// The number of entries stays about what min_ci_balanced asks for, while the entries the GPU runs last
// (sort_sci orders by size) are well below the average, and with them the launch's tail, which lasts about one
// entry.  Measured on an MI355X at 97 k atoms: +2-3 % over uniform entries, and no more dependence on how the
// entry count happens to divide by the 4096 wave slots.  A function of (sci, shift) alone, so every builder and
// every thread agrees on it (nbnxn_search.hip has the same function).
inline int progressive_nsp_max(int nsp_max_av, int sci, int shift)
{
    if (nsp_max_av <= 0) { return nsp_max_av; }
    unsigned h = (unsigned)sci * 2654435761u + (unsigned)shift * 40503u;
    h ^= h >> 15;
    const int   cls = (int)(h & 7u);
    float       f   = cls < 6 ? 1.5f : (cls == 6 ? 0.75f : 0.4f);
    // tuning aid (tools/ab_prog.sh): NBNXN_PROG_FACTORS="f0,...,f7" replaces the eight class factors in THIS builder
    // only; the device search keeps the built-in ones, so a list made with it is for kernel timing, not for the search tests
    static const std::vector<float> env = [] {
        std::vector<float> v;
        if (const char *e = getenv("NBNXN_PROG_FACTORS")) { for (const char *q = e; *q;) { char *end; v.push_back(strtof(q, &end)); q = (*end == ',') ? end + 1 : end; if (end == q && *end) { break; } } }
        return v;
    }();
    if (env.size() == 8) { f = env[cls]; }
    return std::max(NSUBC * 4 / 2 + 1, (int)(nsp_max_av * f));
}

// tuning aid: NBNXN_NL_SPLIT_SCALE multiplies the pair budget of non-local list entries (both builders read it)
static double nonlocal_split_scale()
{
    static const double v = [] { const char *e = getenv("NBNXN_NL_SPLIT_SCALE"); const double x = e ? atof(e) : 1.0; return x > 0 ? x : 1.0; }();
    return v;
}

// get_nsubpair_max (nbnxn_search.c:4324-4408), local (single-grid) estimate
int nsubpair_max(const Grid &g, float rlist, int min_ci_balanced, double frac_of_local)
{
    if (min_ci_balanced <= 0 || g.nc >= min_ci_balanced || g.nc == 0) { return -1; }
    double ls[3];
    ls[0] = (g.c1[0] - g.c0[0]) / (g.ncx * 2);
    ls[1] = (g.c1[1] - g.c0[1]) / (g.ncy * 2);
    ls[2] = (g.c1[2] - g.c0[2]) * g.ncx * g.ncy / (g.nc * 2.0);
    const double diag2 = ls[0] * ls[0] + ls[1] * ls[1] + ls[2] * ls[2];
    const double r_eff = rlist + 0.5 * sqr((NA_C - 1.0f) / NA_C) * std::sqrt(diag2 / 3);   // nbnxn_rlist_inc_outside_fac
    double vol = ls[0] * ls[1] * ls[2];
    vol += (ls[0] * ls[1] + ls[0] * ls[2] + ls[1] * ls[2]) * r_eff;
    vol += 2 * (ls[0] + ls[1] + ls[2]) * 0.25 * M_PI * r_eff * r_eff;
    vol += 0.5 * 4.0 / 3.0 * M_PI * r_eff * r_eff * r_eff;
    const double nsp_est = g.nsubc_tot * vol * g.density / NA_C * frac_of_local;
    int nsp = std::max(1, (int)(nsp_est / min_ci_balanced + 0.5));
    return nsp + NSUBC * 4 / 2;
}

} // namespace

extern "C" {

// nbnxn_make_pairlist (nbnxn_search.c:5583) / nbnxn_make_pairlist_part (:4798-5417)
int nbnxn_host_make_pairlist(nbnxn_host_ptr_t h, int iloc, float rlist,
                             const int *excl_index, const int *excl_a,
                             int min_ci_balanced, const float *xq)
{
    if (!h) { return fail("null handle"); }
    if (h->grids.empty()) { return fail("no grid: call nbnxn_host_put_on_grid first"); }
    // the home zone against itself, or against every halo zone (nbnxn_search.c:5652-5699 with one i-zone)
    const int zp[3] = {0, iloc == NBX_LOCAL ? 0 : 1, iloc == NBX_LOCAL ? 1 : (int)h->grids.size()};
    return nbnxn_host_make_pairlist_zones(h, iloc, 1, zp, rlist, excl_index, excl_a, min_ci_balanced, xq);
}

// The general form: the list of zone pairs {i-zone, first j-zone, end j-zone} that the reference takes from its domain
// decomposition (dd_zp3 / dd_zp2 / dd_zp1, domdec.c:433-449; the i-zone loop of nbnxn_make_pairlist,
// nbnxn_search.c:5652-5699): i-clusters may lie in a halo zone.
int nbnxn_host_make_pairlist_zones(nbnxn_host_ptr_t h, int iloc, int nzp, const int *zone_pairs, float rlist,
                                   const int *excl_index, const int *excl_a, int min_ci_balanced, const float *xq)
{
    if (!h) { return fail("null handle"); }
    if (h->grids.empty()) { return fail("no grid: call nbnxn_host_put_on_grid first"); }
    if (iloc != NBX_LOCAL && iloc != NBX_NONLOCAL) { return fail("bad iloc %d", iloc); }
    if (nzp < 1 || !zone_pairs) { return fail("no zone pairs"); }
    for (int z = 0; z < nzp; z++)
    {
        const int iz = zone_pairs[3 * z], j0 = zone_pairs[3 * z + 1], j1 = zone_pairs[3 * z + 2];
        if (iz < 0 || iz >= (int)h->grids.size() || j0 < 0 || j1 > (int)h->grids.size() || j0 > j1)
        {
            return fail("zone pair %d = {%d, %d, %d} outside the %zu gridded zones", z, iz, j0, j1, h->grids.size());
        }
        // one row per i-zone, i-zones ascending, as in the reference's tables (domdec.c:433-449): the entries of the list
        // come out in table order here and in i-cell order in the device builder, which must give the same list
        if (z > 0 && iz <= zone_pairs[3 * (z - 1)])
        {
            return fail("zone pair %d: i-zone %d after i-zone %d: the table lists every i-zone once, ascending", z, iz, zone_pairs[3 * (z - 1)]);
        }
    }
    for (int d = 0; d < 3; d++)
    {
        if (h->shp[d] && h->box[d] < 2 * rlist)
        {
            return fail("box[%d] = %g is shorter than 2*rlist = %g: more than one periodic image in range", d, h->box[d], 2 * rlist);
        }
    }
    // the second limit of max_cutoff2 (gmxlib/pbc.c:171-185): only single box vectors (two in x) are tried
    if (h->shp[1] && h->shp[2] && h->box[1] - std::fabs(h->bzy) < rlist)
    {
        return fail("box[YY][YY] - |box[ZZ][YY]| = %g is shorter than rlist = %g", h->box[1] - std::fabs(h->bzy), rlist);
    }
    // shift range (nbnxn_search.c:4906-4927): a skewed box can need two box lengths in x
    int shp[3] = {h->shp[0], h->shp[1], h->shp[2]};
    if (shp[0] && h->box[0] - std::fabs(h->byx) - std::fabs(h->bzx) < rlist) { shp[0] = 2; }
    const float rl2 = rlist * rlist;
    // the pair budget of an entry comes from the home grid (get_nsubpair_max, nbnxn_search.c:4324-4408)
    const int nsp_max = nsubpair_max(h->grids[0], rlist, min_ci_balanced, iloc == NBX_LOCAL ? 1.0 : 0.25 * nonlocal_split_scale());

    std::vector<PartList> parts;
    for (int zpi = 0; zpi < nzp; zpi++)
    {
    const int   izone = zone_pairs[3 * zpi];
    const Grid &gi    = h->grids[izone];
    const int   jz0 = zone_pairs[3 * zpi + 1], jz1 = zone_pairs[3 * zpi + 2];
    if (jz1 <= jz0 || gi.nc == 0) { continue; }

    // contiguous slices of i super-cells per part; parts are concatenated in order, so the
    // result does not depend on the thread count
    const int    npart = std::max(1, std::min(64, gi.nc / 8));
    const size_t part0 = parts.size();
    parts.resize(part0 + npart);

#pragma omp parallel for schedule(dynamic, 1)
    for (int p = 0; p < npart; p++)
    {
        PartList &pl = parts[part0 + p];
        const int s0 = (int)((long long)gi.nc * p / npart);
        const int s1 = (int)((long long)gi.nc * (p + 1) / npart);
        BB    bb_ci[NSUBC];
        float x_ci[NA_SC * 3];
        for (int sci = s0; sci < s1; sci++)
        {
            if (gi.nsubc[sci] == 0) { continue; }
            // i super-cell bounding box (unshifted)
            BB bI;
            for (int d = 0; d < 3; d++) { bI.lo[d] = 1e30f; bI.hi[d] = -1e30f; }
            for (int c = 0; c < gi.nsubc[sci]; c++)
            {
                const BB &b = gi.bb[sci * NSUBC + c];
                for (int d = 0; d < 3; d++)
                {
                    bI.lo[d] = std::min(bI.lo[d], b.lo[d]);
                    bI.hi[d] = std::max(bI.hi[d], b.hi[d]);
                }
            }
            for (int jz = jz0; jz < jz1; jz++)
            {
                const Grid &gj       = h->grids[jz];
                const bool same_grid = (jz == izone);
                const float bbx = 0.5f * (gi.sx + gj.sx) / 2, bby = 0.5f * (gi.sy + gj.sy) / 2;
                // boundingbox_only_distance2 (:4721-4757)
                const float rbb2 = sqr(std::max(0.0f, rlist - 0.5f * std::sqrt(bbx * bbx + bby * bby)));
                for (int tz = -shp[2]; tz <= shp[2]; tz++)
                {
                    for (int ty = -shp[1]; ty <= shp[1]; ty++)
                    {
                        for (int tx = -shp[0]; tx <= shp[0]; tx++)
                        {
                            const int shift = 5 * (3 * (tz + 1) + ty + 1) + tx + 2;   // XYZ2IS
                            // one-way lists: only backward shifts within one grid (:5091-5096)
                            if (same_grid && shift > NBX_CENTRAL) { continue; }
                            // shx, shy, shz (:5019, :5054, :5098)
                            const float sh[3] = { tx * h->box[0] + ty * h->byx + tz * h->bzx, ty * h->box[1] + tz * h->bzy, tz * h->box[2] };
                            BB bS = bI;
                            for (int d = 0; d < 3; d++) { bS.lo[d] += sh[d]; bS.hi[d] += sh[d]; }
                            // j column range
                            int cxf = (int)std::floor((bS.lo[0] - rlist - gj.c0[0]) * gj.inv_sx);
                            int cxl = (int)std::floor((bS.hi[0] + rlist - gj.c0[0]) * gj.inv_sx);
                            int cyf = (int)std::floor((bS.lo[1] - rlist - gj.c0[1]) * gj.inv_sy);
                            int cyl = (int)std::floor((bS.hi[1] + rlist - gj.c0[1]) * gj.inv_sy);
                            cxf = std::max(cxf, 0); cxl = std::min(cxl, gj.ncx - 1);
                            cyf = std::max(cyf, 0); cyl = std::min(cyl, gj.ncy - 1);
                            if (cxf > cxl || cyf > cyl) { continue; }

                            // new_sci_entry (:3845)
                            nbx_sci_t e;
                            e.sci           = gi.cell0 + sci;
                            e.shift         = shift;
                            e.cj4_ind_start = (pl.cj_ind + 3) >> 2;
                            e.cj4_ind_end   = e.cj4_ind_start;
                            pl.cj_ind       = e.cj4_ind_start * 4;
                            pl.sci.push_back(e);
                            bool any = false;

                            for (int c = 0; c < gi.nsubc[sci]; c++)
                            {
                                bb_ci[c] = gi.bb[sci * NSUBC + c];
                                for (int d = 0; d < 3; d++) { bb_ci[c].lo[d] += sh[d]; bb_ci[c].hi[d] += sh[d]; }
                            }
                            for (int k = 0; k < NA_SC; k++)
                            {
                                const float *xa = xq + 4 * ((size_t)(gi.cell0 + sci) * NA_SC + k);
                                x_ci[3 * k + 0] = xa[0] + sh[0];
                                x_ci[3 * k + 1] = xa[1] + sh[1];
                                x_ci[3 * k + 2] = xa[2] + sh[2];
                            }
                            const bool central_same = same_grid && shift == NBX_CENTRAL;
                            for (int cx = cxf; cx <= cxl; cx++)
                            {
                                float d2x = 0;
                                const float colx0 = gj.c0[0] + cx * gj.sx, colx1 = colx0 + gj.sx;
                                if (colx0 > bS.hi[0]) { d2x = sqr(colx0 - bS.hi[0]); }
                                else if (colx1 < bS.lo[0]) { d2x = sqr(bS.lo[0] - colx1); }
                                if (d2x >= rl2) { continue; }
                                for (int cy = cyf; cy <= cyl; cy++)
                                {
                                    float d2xy = d2x;
                                    const float coly0 = gj.c0[1] + cy * gj.sy, coly1 = coly0 + gj.sy;
                                    if (coly0 > bS.hi[1]) { d2xy += sqr(coly0 - bS.hi[1]); }
                                    else if (coly1 < bS.lo[1]) { d2xy += sqr(bS.lo[1] - coly1); }
                                    if (d2xy >= rl2) { continue; }
                                    int c0 = gj.cxy_ind[cx * gj.ncy + cy];
                                    const int c1 = gj.cxy_ind[cx * gj.ncy + cy + 1];
                                    // half list: only j super-cells >= i (:5270-5283)
                                    if (central_same) { c0 = std::max(c0, sci); }
                                    for (int scj = c0; scj < c1; scj++)
                                    {
                                        if (gj.nsubc[scj] == 0) { continue; }
                                        float d2 = d2xy;
                                        const float z0 = gj.bbcz[2 * scj], z1 = gj.bbcz[2 * scj + 1];
                                        if (z0 > bS.hi[2]) { d2 += sqr(z0 - bS.hi[2]); }
                                        else if (z1 < bS.lo[2]) { d2 += sqr(bS.lo[2] - z1); }
                                        if (d2 >= rl2) { continue; }
                                        make_cluster_list(*h, gi, gj, pl, sci, scj,
                                                          central_same && sci == scj,
                                                          bb_ci, x_ci, xq, rl2, rbb2);
                                        any = true;
                                    }
                                }
                            }
                            (void)any;
                            if (excl_index != nullptr && excl_a != nullptr)
                            {
                                set_top_excls(*h, pl, central_same, excl_index, excl_a);
                            }
                            // close_ci_entry_supersub (:4027-4055)
                            if (pl.sci.back().cj4_ind_end > pl.sci.back().cj4_ind_start)
                            {
                                pl.cj_ind = ((pl.cj_ind + 3) >> 2) * 4;
                                // drop a trailing cj4 that was opened but never used
                                pl.cj4.resize(pl.cj_ind >> 2);
                                if (nsp_max > 0)
                                {
                                    const nbx_sci_t &le = pl.sci.back();
                                    split_last_sci(pl, h->progressive ? progressive_nsp_max(nsp_max, le.sci, le.shift) : nsp_max);
                                }
                            }
                            else
                            {
                                pl.cj_ind = pl.sci.back().cj4_ind_start * 4;
                                pl.cj4.resize(pl.cj_ind >> 2);
                                pl.sci.pop_back();
                            }
                        }
                    }
                }
            }
        }
    }

    }   // zone pairs

    // combine_nblists (:4466-4560): concatenate, renumber cj4 and excl indices
    List &L = h->lists[iloc];
    L.sci.clear(); L.cj4.clear(); L.excl.clear();
    L.nci_tot = 0;
    L.rlist   = rlist;
    nbx_excl_t ones;
    for (int t = 0; t < 32; t++) { ones.pair[t] = ALL_BITS; }
    L.excl.push_back(ones);   // shared "no exclusions" entry (:2532-2534)
    for (PartList &pl : parts)
    {
        const int cj4_off  = (int)L.cj4.size();
        const int excl_off = (int)L.excl.size() - 1;   // local index is 1-based
        for (nbx_sci_t s : pl.sci)
        {
            s.cj4_ind_start += cj4_off;
            s.cj4_ind_end   += cj4_off;
            L.sci.push_back(s);
        }
        for (nbx_cj4_t c : pl.cj4)
        {
            for (int w = 0; w < 2; w++)
            {
                if (c.imei[w].excl_ind > 0) { c.imei[w].excl_ind += excl_off; }
            }
            L.cj4.push_back(c);
        }
        L.excl.insert(L.excl.end(), pl.excl.begin(), pl.excl.end());
        L.nci_tot += pl.nci_tot;
    }

    // sort_sci (:5517-5580): descending cj4 count, stable, counts clipped at twice the mean.  Second key (not in the
    // reference, nbx_sci_pair_key): among entries of one group count the ones with more cluster pairs come first, so that
    // what the GPU starts last is the lightest work of the smallest size class.
    if (L.cj4.size() > L.sci.size() && !L.sci.empty())
    {
        const int m = (int)(2 * L.cj4.size() / L.sci.size());
        std::vector<std::pair<int, nbx_sci_t>> keyed(L.sci.size());
        for (size_t i = 0; i < L.sci.size(); i++)
        {
            const nbx_sci_t &e = L.sci[i];
            int npair = 0;
            for (int j4 = e.cj4_ind_start; j4 < e.cj4_ind_end; j4++) { npair += __builtin_popcount(L.cj4[j4].imei[0].imask | L.cj4[j4].imei[1].imask); }
            const int k1 = m - std::min(m, e.cj4_ind_end - e.cj4_ind_start);
            // (only for lists that were split for the GPU's balance, with the progressive classes on: a list that was
            // not split keeps the reference's order exactly)
            const int k2 = (nsp_max > 0 && h->progressive) ? nbx_sci_pair_key(e.cj4_ind_end - e.cj4_ind_start, npair, m) : 0;
            keyed[i] = {k1 * NBX_SCI_PAIR_KEYS + k2, e};
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<int, nbx_sci_t> &a, const std::pair<int, nbx_sci_t> &b) { return a.first < b.first; });
        for (size_t i = 0; i < L.sci.size(); i++) { L.sci[i] = keyed[i].second; }
    }

    nbx_pairlist_t &v = L.view;
    memset(&v, 0, sizeof(v));
    v.bSimple = 0;
    v.na_ci = NA_C; v.na_cj = NA_C; v.na_sc = NA_SC;
    v.rlist = rlist;
    v.nsci = (int)L.sci.size();   v.sci = L.sci.data();   v.sci_nalloc = (int)L.sci.capacity();
    v.ncj4 = (int)L.cj4.size();   v.cj4 = L.cj4.data();   v.cj4_nalloc = (int)L.cj4.capacity();
    v.nexcl = (int)L.excl.size(); v.excl = L.excl.data(); v.excl_nalloc = (int)L.excl.capacity();
    v.nci_tot = L.nci_tot;
    return 0;
}

int nbnxn_host_set_progressive_balance(nbnxn_host_ptr_t h, int on)
{
    if (!h) { return fail("null handle"); }
    h->progressive = on != 0;
    return 0;
}

const nbx_pairlist_t *nbnxn_host_pairlist(nbnxn_host_ptr_t h, int iloc)
{
    if (!h || iloc < 0 || iloc > 1) { return nullptr; }
    return &h->lists[iloc].view;
}

} // extern "C"
