// C-ABI layer of the MI355X nonbonded module: device data management and per-step calls
// (include/nbnxn_hip.h).  Replaces mdlib/nbnxn_cuda/nbnxn_cuda_data_mgmt.cu and the host
// part of nbnxn_cuda.cu; structure is our own: one handle, two HIP streams, plain
// hipMalloc'd arrays with 20 % over-allocation, hipEvent timing.
#include <hip/hip_runtime.h>

#include <chrono>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound at run time (dd_bind)
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "nbnxn_hip.h"
#include "nbnxn_kernel.h"
#include "nbnxn_search.h"

namespace {

thread_local std::string g_last_error;
int                      g_abort_on_error = 1;

// gmx_fatal equivalent (cudautils.cuh:65-90): message, then abort unless switched off
int nbx_fail(const char *fmt, ...)
{
    char    buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (g_abort_on_error)
    {
        fprintf(stderr, "nbnxn_hip fatal error: %s\n", buf);
        abort();
    }
    return 1;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { return nbx_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } \
    } while (0)

template <typename T>
struct DevBuf {
    T     *p     = nullptr;
    size_t n     = 0;   // elements in use
    size_t alloc = 0;   // elements allocated
};

// cu_realloc_buffered equivalent (cudautils.cu): grow with head-room, never shrink
template <typename T>
int dev_reserve(DevBuf<T> &b, size_t n, bool *realloced = nullptr)
{
    if (realloced) { *realloced = false; }
    if (n > b.alloc)
    {
        if (b.p) { HIP_TRY(hipFree(b.p)); b.p = nullptr; }
        const size_t na = n + n / 5 + 64;
        HIP_TRY(hipMalloc((void **)&b.p, na * sizeof(T)));
        b.alloc = na;
        if (realloced) { *realloced = true; }
    }
    b.n = n;
    return 0;
}

template <typename T>
void dev_free(DevBuf<T> &b)
{
    if (b.p) { (void)hipFree(b.p); }
    b.p = nullptr; b.n = b.alloc = 0;
}

struct PList {
    DevBuf<nbx_sci_t>  sci;
    DevBuf<nbx_cj4_t>  cj4;
    DevBuf<nbx_excl_t> excl;
    int                nsci = 0, ncj4 = 0, nexcl = 0, nci_tot = 0;
    float              rlist = 0;
    bool               do_prune = false;
    // dynamic pruning (nbnxn_hip_launch_kernel_pruneonly): outer masks beside the list, rolling position
    DevBuf<unsigned>   imask_outer;
    bool               have_outer = false;
    int                rolling_part = 0, rolling_nparts = 0;
};

} // namespace

// ---- the entry point that allocates nothing (nbnxn_hip_launch_kernel_noalloc) ----
// Its kernels need the same scratch buffer the other launches get from the handle's allocation: the NBX_OUT_SLOTS
// partial-sum slots {e_lj, e_el, fshift[45][3]} that the waves add into and a fold kernel sums.  Here they live in a
// file-scope device array: 2 x 64 x 144 floats = 73,728 bytes, whatever the system size, held by every GPU the library
// runs on until the process ends; the local and the non-local launch of a two-stream handle each use their own half,
// so the two may overlap (they run on different streams in the decomposed step).  `scratch` is a host-side stand-in
// for the symbol, NOT a device address: host code only ever uses the per-device address hipGetSymbolAddress returned
// (g_scratch_ptr), as kernel argument and for the clearing fill.
#define NBX_NOALLOC_SCRATCH_FLOATS (NBX_OUT_SLOTS * NBX_OUT_STRIDE)
#define NBX_MAX_DEVICES 64
static __device__ __align__(256) float scratch[2 * NBX_NOALLOC_SCRATCH_FLOATS];

namespace {
float     *g_scratch_ptr[NBX_MAX_DEVICES] = {};   // each GPU has its own copy of the array at its own address
std::mutex g_scratch_mutex;

// once per device, with that device current (nbnxn_hip_init calls it right after hipSetDevice)
int bind_scratch(int device)
{
    if (device < 0 || device >= NBX_MAX_DEVICES) { return nbx_fail("device %d outside [0, %d)", device, NBX_MAX_DEVICES); }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) { return 0; }
    float *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) { return nbx_fail("hipGetSymbolAddress returned no address for the scratch array on device %d", device); }
    g_scratch_ptr[device] = ptr;
    return 0;
}

// sums the slots in double, as fold_outputs does on the host, and ASSIGNS {e_lj, e_el, fshift[45][3]} to out
__global__ void nbx_fold_slots_kernel(const float *__restrict__ slots, float *__restrict__ out)
{
    const int k = threadIdx.x;
    if (k < 2 + NBX_SHIFTS * 3)
    {
        double sum = 0;
        for (int s = 0; s < NBX_OUT_SLOTS; s++) { sum += (double)slots[s * NBX_OUT_STRIDE + k]; }
        out[k] = (float)sum;
    }
}
} // namespace

struct nbnxn_hip {
    int         device = 0;
    bool        two_streams = false;
    int         reserved_cus = 0;   // CUs the local stream leaves to the non-local stream's work (CU mask)
    bool        nonlocal_high_priority = true;   // priority of the non-local stream (GMX_HIP_NB_NONLOCAL_PRIORITY)
    hipStream_t stream[2] = {nullptr, nullptr};
    hipEvent_t  misc_ops_done = nullptr, nonlocal_done = nullptr;
    hipEvent_t  stream_mark[2] = {nullptr, nullptr};   // nbnxn_hip_stream_wait
    // home coordinates written on the local stream (H2D of xq, x gather) and not yet seen by the non-local stream:
    // the halo pack of nbnxn_hip_dd_move_x / nbnxn_hip_halo_pack_x reads them there
    hipEvent_t  x_written = nullptr;
    bool        x_written_pending = false;
    hipEvent_t  k_start[2] = {nullptr, nullptr}, k_stop[2] = {nullptr, nullptr};
    hipEvent_t  h2d_start[2] = {nullptr, nullptr}, h2d_stop[2] = {nullptr, nullptr};
    hipEvent_t  d2h_start[2] = {nullptr, nullptr}, d2h_stop[2] = {nullptr, nullptr};
    bool        k_timed[2] = {false, false}, h2d_timed[2] = {false, false}, d2h_timed[2] = {false, false};
    bool        last_ener[2] = {false, false}, last_prune[2] = {false, false};
    bool        do_time = true;

    // atom data
    DevBuf<float4> xq;
    DevBuf<float>  f;
    DevBuf<int>    type;
    float         *shift_vec = nullptr;   // 45*3
    float         *out = nullptr;         // [NBX_OUT_SLOTS][NBX_OUT_STRIDE]: {e_lj, e_el, fshift[45][3]} per slot
    int            natoms = 0, natoms_local = 0, ntype = 0;
    bool           shift_vec_uploaded = false;

    // parameters
    DevBuf<float2> nbfp;
    int            el_type = NBX_EL_RF;
    float          rcoulomb = 0, rvdw = 0, rlist = 0;
    float          epsfac = 0, k_rf = 0, c_rf = 0, beta = 0, sh_ewald = 0, disp_cpot = 0, rep_cpot = 0;
    // LJ modifier: 0 cut-off (+ potential shift through the cpot values), 1 force switch, 2 potential switch
    int            vdw_mod = 0;
    float          rvdw_switch = 0, fsw[4] = {0, 0, 0, 0}, psw[3] = {0, 0, 0};
    // LJ-PME (vdw_mod 3): grid C6 per type pair, Ewald coefficient and potential shift of the dispersion mesh
    DevBuf<float>  c6grid;
    float          ewaldcoeff_lj = 0, sh_lj_ewald = 0;
    // tabulated Ewald (el_type NBX_EL_EWALD_TAB): force table divided by beta^2
    DevBuf<float>  coulomb_tab;
    int            tab_size = 0;
    float          tab_scale = 0, tab_beta = 0, tab_rc = 0;
    bool           have_const = false;

    PList plist[2];

    // kernel-ready constants (nbnxn_kernel.h), derived from the parameters above when they change, not per launch
    nbx_kernel_k_t kc;
    bool           kc_valid = false;

    // pinned staging for the small outputs (nb_staging, nbnxn_cuda_types.h:104-110)
    float *fbase   = nullptr;   // one allocation: 2 x {[out slots][f]}, so one memset clears slots and forces
    // In-kernel clear (nbnxn_hip_set_clear_in_kernel): the two halves of fbase alternate as the
    // output buffer; every LOCAL launch zeroes the half that is NOT in use, and clear_outputs
    // then only flips to it.  `cur` is the half in use, `clean` says the other half is all zero.
    bool   clear_in_kernel = false;
    int    cur             = 0;
    bool   clean_other     = false;
    size_t half_floats     = 0;      // floats per half: OUT_FLOATS + f.alloc
    // device-side atom-data moves (nbnxn_hip_set_atom_order / copy_x_to_xq / add_f)
    DevBuf<int>   d_a, d_cell;      // grid slot -> caller atom (-1: filler); caller atom -> grid slot
    DevBuf<float> d_x, d_fa;        // staging for a host x / host f in caller order
    float        *h_fa = nullptr;   // pinned landing area of d_fa
    size_t        h_fa_alloc = 0;
    int           natoms_real = 0;
    // device-side pair search (nbnxn_hip_put_on_grid / nbnxn_hip_make_pairlist), created on first use
    nbx_search   *search = nullptr;
    bool          grid_on_device = false;   // the atom arrays in use came from nbnxn_hip_put_on_grid
    bool          excl_held = false;        // d_excl_index / d_excl_a hold a topology
    DevBuf<int>   d_type_in, d_excl_index, d_excl_a;   // caller-order types; topology exclusions (CSR)
    DevBuf<float> d_q_in;
    float *h_stage = nullptr;   // host copy of `out`, folded in double by fold_outputs()
    void  *h_stage_dev = nullptr;   // the device's view of h_stage (pinned + mapped), for the copy kernels
    bool   copy_kernels = true;     // nbnxn_hip_set_copy_kernels
    double *h_fold = nullptr;       // pinned + mapped: the slots summed by the copy-back kernel, {e_lj, e_el, fshift[45][3]}
    void   *h_fold_dev = nullptr;
    bool    slots_folded = false;   // the last local copy-back left its sums in h_fold (else: raw slots in h_stage)
    // domain decomposition over RCCL (nbnxn_hip_dd_*): one pulse along one dimension
    ncclComm_t    dd_comm = nullptr;
    int           dd_rank = -1, dd_nranks = 0;
    // the halo messages of a step (nbnxn_hip_dd_set_messages): rows of message k start at dd_send_off[k] / dd_recv_off[k]
    // (multiples of 64) in the slot lists and buffers; coordinates go to send_to, forces come back from it
    int              dd_nmsg = 0;
    // phases of an exchange (nbnxn_hip_dd_set_phases): messages [dd_phase_end[p - 1], dd_phase_end[p]) travel together, the
    // phases one after the other (the reference's pulses per dimension, whose later ones forward what earlier ones brought)
    int              dd_nphase = 0;
    int              dd_phase_end[NBX_DD_MAX_MESSAGES];
    nbx_dd_message_t dd_msg[NBX_DD_MAX_MESSAGES];
    int              dd_send_off[NBX_DD_MAX_MESSAGES], dd_recv_off[NBX_DD_MAX_MESSAGES];
    int              dd_send_rows = 0, dd_recv_rows = 0;   // padded totals
    DevBuf<int>      dd_send_slots, dd_recv_slots, dd_send_wave_msg;
    DevBuf<float>    dd_msg_shift;                         // float4 per message: shift, shift index as int bits
    DevBuf<float> dd_x_out, dd_x_in, dd_f_out, dd_f_in;

    nbx_wallclock_gpu_t timings;
};

namespace {

int check_handle(nbnxn_hip_ptr_t nb)
{
    if (!nb) { return nbx_fail("null nbnxn_hip handle"); }
    return 0;
}

int check_loc(int loc)
{
    if (loc != NBX_LOCAL && loc != NBX_NONLOCAL)
    {
        return nbx_fail("Invalid locality passed (%d); valid here is only local (%d) or nonlocal (%d)", loc, NBX_LOCAL, NBX_NONLOCAL);
    }
    return 0;
}

// Home coordinates were just written on the local stream: remember it, so that the halo pack on the non-local
// stream waits for exactly that write (and not for the local kernel behind it, which it is meant to run beside).
int mark_x_written(nbnxn_hip *nb, int iloc)
{
    if (nb->two_streams && iloc == NBX_LOCAL)
    {
        HIP_TRY(hipEventRecord(nb->x_written, nb->stream[NBX_LOCAL]));
        nb->x_written_pending = true;
    }
    return 0;
}

int wait_x_written(nbnxn_hip *nb, hipStream_t st)
{
    if (nb->two_streams && nb->x_written_pending && st != nb->stream[NBX_LOCAL])
    {
        HIP_TRY(hipStreamWaitEvent(st, nb->x_written, 0));
        nb->x_written_pending = false;
    }
    return 0;
}

// atom range of a locality (nbnxn_cuda.cu:311-321)
void loc_range(const nbnxn_hip *nb, int loc, int *begin, int *len)
{
    if (loc == NBX_LOCAL) { *begin = 0; *len = nb->natoms_local; }
    else { *begin = nb->natoms_local; *len = nb->natoms - nb->natoms_local; }
}

// Ewald long-range force table F(r) = -d/dr [erf(beta r) / r] at ntab points of spacing dx, made
// the way table_spline3_fill_ewald_lr does for the reference's GPU table (tables.c:165-311,
// called from nbnxn_cuda_data_mgmt.cu:134): a quadratic spline is laid through the potential
// at the table points and interval midpoints, and the tabulated force at a point is the mean of
// the spline derivatives of the two intervals that meet there -- which makes LINEAR
// interpolation of the force table consistent with the quadratic spline energy.  erf(beta r) / r
// is finite everywhere, so the reference's out-of-range continuation (:218-222, for the r^-6
// LJ table) has nothing to do here.
void make_ewald_force_table(std::vector<float> &tab, int ntab, double dx, double beta)
{
    auto v_lr = [beta](double r) { return r == 0 ? beta * 2 / std::sqrt(M_PI) : std::erf(beta * r) / r; };
    tab.assign(ntab, 0.0f);
    for (int hi = ntab - 1; hi >= 1; hi--)
    {
        // interval (hi-1, hi): spline s(x) = v_hi + d_hi (x - x_hi) + curvature term, with the
        // derivative at the upper end chosen so that the quadratic reproduces the potential at
        // both ends and at the midpoint on average (tables.c:244-252), and the derivative at
        // the lower end following from the two end values (:268-277)
        const double x_hi = hi * dx;
        const double v_hi = v_lr(x_hi), v_lo = v_lr((hi - 1) * dx), v_mid = v_lr(x_hi - 0.5 * dx);
        const double d_hi = (v_hi - v_lo) / dx + 2.0 * (v_hi + v_lo - 2.0 * v_mid) / dx;
        const double d_lo = d_hi - 2.0 * (d_hi * dx + v_lo - v_hi) / dx;
        // each inner point gets half of either side; the float roundings are those of the
        // reference's real-typed table (store the half from above first, then add the half from below)
        if (hi == ntab - 1) { tab[hi] = (float)(-d_hi); }
        else { tab[hi] = (float)((double)tab[hi] - 0.5 * d_hi); }
        tab[hi - 1] = (float)(-0.5 * d_lo);
    }
    tab[0] *= 2;   // the first point only has an interval above it
}

int upload_coulomb_tab(nbnxn_hip *nb, float beta, float rcoulomb)
{
    if (nb->coulomb_tab.p && nb->tab_beta == beta && nb->tab_rc == rcoulomb) { return 0; }
    // GPU_EWALD_COULOMB_FORCE_TABLE_SIZE and the "- 2" of nbnxn_cuda_data_mgmt.cu:128-130
    const int    ntab  = 1536;
    const double scale = (ntab - 2) / (double)rcoulomb;
    std::vector<float> tab;
    make_ewald_force_table(tab, ntab, 1.0 / scale, beta);
    // the kernel works in beta-scaled coordinates: F / beta^2, index = (beta r) * (scale / beta)
    for (float &t : tab) { t = (float)((double)t / ((double)beta * (double)beta)); }
    if (dev_reserve(nb->coulomb_tab, (size_t)ntab)) { return 1; }
    HIP_TRY(hipMemcpyAsync(nb->coulomb_tab.p, tab.data(), ntab * sizeof(float), hipMemcpyHostToDevice, nb->stream[0]));
    HIP_TRY(hipStreamSynchronize(nb->stream[0]));
    nb->tab_size = ntab; nb->tab_scale = (float)scale; nb->tab_beta = beta; nb->tab_rc = rcoulomb;
    return 0;
}

int set_params(nbnxn_hip *nb, int el_type, float rcoulomb, float rvdw, float rlist, float epsfac,
               float k_rf, float c_rf, float beta, float sh_ewald, float disp_cpot, float rep_cpot)
{
    if (el_type != NBX_EL_RF && el_type != NBX_EL_EWALD_ANA && el_type != NBX_EL_EWALD_TAB)
    {
        return nbx_fail("unsupported electrostatics flavour %d", el_type);
    }
    if (!(rcoulomb > 0)) { return nbx_fail("rcoulomb must be > 0"); }
    // twin-range: only rvdw < rcoulomb with Ewald (what PME load balancing produces and the only
    // combination the Verlet scheme allows, readir.c / nbnxn_cuda_data_mgmt.cu:303-357)
    if (rvdw > rcoulomb || !(rvdw > 0)) { return nbx_fail("rvdw (%g) must be in (0, rcoulomb = %g]", rvdw, rcoulomb); }
    if (rvdw != rcoulomb && el_type == NBX_EL_RF)
    {
        return nbx_fail("rvdw (%g) != rcoulomb (%g) with reaction field: the Verlet scheme has no such flavour", rvdw, rcoulomb);
    }
    if (el_type != NBX_EL_RF && !(beta > 0)) { return nbx_fail("ewaldcoeff_q must be > 0 for the Ewald flavours"); }
    // the energies of the tabulated flavour are analytical too (as in the reference), so the fit's range binds both
    if (el_type != NBX_EL_RF && beta * beta * rcoulomb * rcoulomb > 16.0f)
    {
        return nbx_fail("(ewaldcoeff_q*rcoulomb)^2 = %g exceeds the range (16) of the analytical Ewald fit", beta * beta * rcoulomb * rcoulomb);
    }
    nb->el_type = el_type;
    nb->rcoulomb = rcoulomb; nb->rvdw = rvdw; nb->rlist = rlist;
    nb->epsfac = epsfac; nb->k_rf = k_rf; nb->c_rf = c_rf;
    nb->beta = beta; nb->sh_ewald = sh_ewald;
    nb->disp_cpot = disp_cpot; nb->rep_cpot = rep_cpot;
    // (re)tabulate like init_ewald_coulomb_force_table, also on PME load-balancing updates
    // (nbnxn_cuda_data_mgmt.cu:597-606)
    if (el_type == NBX_EL_EWALD_TAB && upload_coulomb_tab(nb, beta, rcoulomb)) { return 1; }
    nb->have_const = true;
    nb->kc_valid   = false;
    return 0;
}

int upload_nbfp(nbnxn_hip *nb, int ntype, const float *nbfp)
{
    if (ntype <= 0 || !nbfp) { return nbx_fail("bad LJ table"); }
    if (dev_reserve(nb->nbfp, (size_t)ntype * ntype)) { return 1; }
    HIP_TRY(hipMemcpyAsync(nb->nbfp.p, nbfp, (size_t)ntype * ntype * sizeof(float2), hipMemcpyHostToDevice, nb->stream[0]));
    HIP_TRY(hipStreamSynchronize(nb->stream[0]));
    nb->ntype = ntype;
    return 0;
}

constexpr size_t OUT_FLOATS = (size_t)NBX_OUT_SLOTS * NBX_OUT_STRIDE;

// the force array lives behind the energy / shift-force slots in one allocation
int reserve_f(nbnxn_hip *nb, size_t n, bool *realloced)
{
    *realloced = false;
    if (n > nb->f.alloc || !nb->fbase)
    {
        if (nb->fbase) { HIP_TRY(hipFree(nb->fbase)); nb->fbase = nullptr; }
        const size_t na = n + n / 5 + 64;
        nb->half_floats = (OUT_FLOATS + na + 63) / 64 * 64;
        HIP_TRY(hipMalloc((void **)&nb->fbase, 2 * nb->half_floats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(nb->fbase, 0, OUT_FLOATS * sizeof(float), nb->stream[0]));
        nb->cur         = 0;
        nb->clean_other = false;
        nb->out         = nb->fbase;
        nb->f.p         = nb->fbase + OUT_FLOATS;
        nb->f.alloc     = na;
        *realloced  = true;
    }
    nb->f.n = n;
    return 0;
}

int clear_e_fshift(nbnxn_hip *nb)
{
    HIP_TRY(hipMemsetAsync(nb->out, 0, NBX_OUT_SLOTS * NBX_OUT_STRIDE * sizeof(float), nb->stream[0]));
    return 0;
}

// sums the accumulation slots in double: e[0] = e_lj, e[1] = e_el, e[2..136] = fshift
void fold_outputs(const nbnxn_hip *nb, double *e)
{
    for (int k = 0; k < 2 + NBX_SHIFTS * 3; k++) { e[k] = 0; }
    for (int s = 0; s < NBX_OUT_SLOTS; s++)
    {
        const float *p = nb->h_stage + (size_t)s * NBX_OUT_STRIDE;
        for (int k = 0; k < 2 + NBX_SHIFTS * 3; k++) { e[k] += p[k]; }
    }
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { return 0; }
    return ms;
}

int init_pairlist_impl(nbnxn_hip *nb, int iloc, float rlist, int nsci, const nbx_sci_t *sci,
                       int ncj4, const nbx_cj4_t *cj4, int nexcl, const nbx_excl_t *excl, int nci_tot)
{
    if (check_loc(iloc)) { return 1; }
    if (nsci < 0 || ncj4 < 0 || nexcl < 0) { return nbx_fail("negative list size"); }
    if (nsci > 0 && (!sci || !cj4 || !excl || nexcl < 1)) { return nbx_fail("list arrays missing"); }
    // host-side shape check before anything reaches a kernel: every index the kernel will
    // follow must stay inside the uploaded arrays and the atom arrays
    const int ncl = nb->natoms / NBX_CLUSTER_SIZE;
    for (int s = 0; s < nsci; s++)
    {
        if (sci[s].cj4_ind_start < 0 || sci[s].cj4_ind_end < sci[s].cj4_ind_start || sci[s].cj4_ind_end > ncj4)
        {
            return nbx_fail("sci[%d] has cj4 range [%d,%d) outside [0,%d)", s, sci[s].cj4_ind_start, sci[s].cj4_ind_end, ncj4);
        }
        if (sci[s].shift < 0 || sci[s].shift >= NBX_SHIFTS) { return nbx_fail("sci[%d] has shift %d", s, sci[s].shift); }
        if (sci[s].sci < 0 || (sci[s].sci + 1) * NBX_NCL_PER_SUPERCL > ncl)
        {
            return nbx_fail("sci[%d] = %d lies outside the %d atoms set by init_atomdata", s, sci[s].sci, nb->natoms);
        }
    }
    for (int c = 0; c < ncj4; c++)
    {
        for (int w = 0; w < 2; w++)
        {
            if (cj4[c].imei[w].excl_ind < 0 || cj4[c].imei[w].excl_ind >= nexcl) { return nbx_fail("cj4[%d] has excl_ind %d outside [0,%d)", c, cj4[c].imei[w].excl_ind, nexcl); }
        }
        const unsigned m = cj4[c].imei[0].imask | cj4[c].imei[1].imask;
        for (int jm = 0; jm < 4; jm++)
        {
            if (((m >> (8 * jm)) & 0xffu) && (cj4[c].cj[jm] < 0 || cj4[c].cj[jm] >= ncl))
            {
                return nbx_fail("cj4[%d].cj[%d] = %d lies outside the %d clusters set by init_atomdata", c, jm, cj4[c].cj[jm], ncl);
            }
        }
    }
    PList      &pl = nb->plist[iloc];
    hipStream_t st = nb->stream[iloc];
    if (dev_reserve(pl.sci, nsci) || dev_reserve(pl.cj4, ncj4) || dev_reserve(pl.excl, nexcl))
    {
        return 1;
    }
    hipEvent_t t0 = nb->h2d_start[iloc], t1 = nb->h2d_stop[iloc];
    if (nb->do_time) { HIP_TRY(hipEventRecord(t0, st)); }
    if (nsci > 0)
    {
        HIP_TRY(hipMemcpyAsync(pl.sci.p, sci, (size_t)nsci * sizeof(nbx_sci_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(pl.cj4.p, cj4, (size_t)ncj4 * sizeof(nbx_cj4_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(pl.excl.p, excl, (size_t)nexcl * sizeof(nbx_excl_t), hipMemcpyHostToDevice, st));
    }
    if (nb->do_time)
    {
        HIP_TRY(hipEventRecord(t1, st));
        HIP_TRY(hipEventSynchronize(t1));
        nb->timings.pl_h2d_t += elapsed(t0, t1);
        if (iloc == NBX_LOCAL) { nb->timings.pl_h2d_c++; }
    }
    else
    {
        HIP_TRY(hipStreamSynchronize(st));   // the host arrays may be reused by the caller
    }
    pl.nsci = nsci; pl.ncj4 = ncj4; pl.nexcl = nexcl; pl.nci_tot = nci_tot; pl.rlist = rlist;
    pl.do_prune = true; pl.have_outer = false;   // a fresh list gets pruned by its first kernel (nbnxn_cuda_data_mgmt.cu:792-794)
    return 0;
}

int init_atomdata_impl(nbnxn_hip *nb, int natoms, int natoms_local, int ntype, const int *type)
{
    if (natoms < 0 || natoms_local < 0 || natoms_local > natoms) { return nbx_fail("bad atom counts %d/%d", natoms_local, natoms); }
    if (natoms % (NBX_CLUSTER_SIZE * NBX_NCL_PER_SUPERCL) != 0) { return nbx_fail("natoms %d is not a multiple of 64 (padded super-clusters)", natoms); }
    if (nb->ntype != 0 && ntype != nb->ntype) { return nbx_fail("atomdata has %d types, LJ table has %d", ntype, nb->ntype); }
    for (int i = 0; i < natoms; i++)
    {
        if (type[i] < 0 || type[i] >= ntype) { return nbx_fail("atom %d has type %d outside [0,%d)", i, type[i], ntype); }
    }
    bool re = false;
    if (dev_reserve(nb->xq, natoms) || dev_reserve(nb->type, natoms) || reserve_f(nb, (size_t)natoms * 3, &re)) { return 1; }
    nb->natoms = natoms; nb->natoms_local = natoms_local;
    nb->clean_other = false;   // the idle half was zeroed for the old atom count
    nb->d_a.n = 0; nb->d_cell.n = 0;   // a new grid: the atom order has to be set again
    nb->grid_on_device = false;
    hipStream_t st = nb->stream[0];
    if (re && natoms > 0)
    {
        // a new force array starts from zero (nbnxn_cuda_data_mgmt.cu:893-898)
        HIP_TRY(hipMemsetAsync(nb->f.p, 0, nb->f.alloc * sizeof(float), st));
    }
    if (natoms > 0)
    {
        HIP_TRY(hipMemcpyAsync(nb->type.p, type, (size_t)natoms * sizeof(int), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    // lists refer to the old atom arrays
    nb->plist[0].nsci = 0; nb->plist[1].nsci = 0;
    return 0;
}

// Kernel-ready constants (nbnxn_kernel.h: nbx_kernel_k_t), rounded once from double.  They change with
// init_const / set_vdw_modifier / set_ljpme / pme_loadbal_update_param only, so the ~25 pow / divisions are
// spent there (kc_valid) and a launch copies the struct and sets the one value that belongs to the list.
void compute_kernel_consts(const nbnxn_hip *nb, nbx_kernel_k_t &k)
{
    const bool   ew = (nb->el_type != NBX_EL_RF);
    const double b  = ew ? (double)nb->beta : 1.0;
    k.xscale   = (float)b;
    k.qscale   = (float)((double)nb->epsfac * b * b);
    k.c6scale  = (float)std::pow(b, 7);
    k.c12scale = (float)std::pow(b, 13);
    k.escale   = (float)(1.0 / b);
    k.rc2      = (float)(b * b * (double)nb->rcoulomb * (double)nb->rcoulomb);
    k.rlist2   = 0;   // belongs to the list: fill_kernel_consts
    k.rvdw2    = (float)(b * b * (double)nb->rvdw * (double)nb->rvdw);
    k.r2inc    = (float)(b * b * (double)NBX_AVOID_SING_R2_INC);
    // switch constants in scaled units: rsw' = beta rsw; a force-switch term c6' (c2 + c3 rsw') rsw'^2 / r'
    // with c6' = c6 beta^7 must equal (true term) / beta, which gives c2 / beta^9, c3 / beta^10 for
    // dispersion and c2 / beta^15, c3 / beta^16 for repulsion; potential switch c_n / beta^n
    k.vdw_rsw = (float)(b * (double)nb->rvdw_switch);
    k.fsw_d2  = (float)((double)nb->fsw[0] / std::pow(b, 9));
    k.fsw_d3  = (float)((double)nb->fsw[1] / std::pow(b, 10));
    k.fsw_r2  = (float)((double)nb->fsw[2] / std::pow(b, 15));
    k.fsw_r3  = (float)((double)nb->fsw[3] / std::pow(b, 16));
    k.psw_c3  = (float)((double)nb->psw[0] / std::pow(b, 3));
    k.psw_c4  = (float)((double)nb->psw[1] / std::pow(b, 4));
    k.psw_c5  = (float)((double)nb->psw[2] / std::pow(b, 5));
    k.sh       = (float)((double)nb->sh_ewald / b);
    k.rep12    = (float)((double)nb->rep_cpot / 12.0 / std::pow(b, 12));
    k.ndisp6   = (float)(-(double)nb->disp_cpot / 6.0 / std::pow(b, 6));
    // E_self = -c q_raw^2 epsfac with c = c_rf / 2 | beta / sqrt(pi); the kernel has q' = qscale q_raw
    // and multiplies the sum by escale at the end
    // LJ-PME: cl = ewaldcoeff_lj in scaled units
    const double cl = (double)nb->ewaldcoeff_lj / b;
    k.lje2        = (float)(cl * cl);
    k.lje2_nlog2e = (float)(-cl * cl * 1.4426950408889634);
    k.lje6_6      = (float)(std::pow(cl, 6) / 6.0);
    k.lje_sh6     = (float)((double)nb->sh_lj_ewald / std::pow(b, 6) / 6.0);
    k.tab_scale   = (float)((double)nb->tab_scale / b);
    k.tab_max     = (float)(nb->tab_size - 2);
    const double cself = ew ? (double)nb->beta * 0.5641895835477563 : 0.5 * (double)nb->c_rf;
    k.two_k_rf = 2 * nb->k_rf;
    k.c_rf     = nb->c_rf;
    k.self     = (float)(-cself * (double)nb->epsfac / ((double)k.qscale * (double)k.qscale) * b);
}

void fill_kernel_consts(nbnxn_hip *nb, float rlist, nbx_kernel_k_t &k)
{
    if (!nb->kc_valid) { compute_kernel_consts(nb, nb->kc); nb->kc_valid = true; }
    k = nb->kc;
    const double b = (nb->el_type != NBX_EL_RF) ? (double)nb->beta : 1.0;
    k.rlist2 = (float)(b * b * (double)rlist * (double)rlist);
}

// The CUDA module's benchmarking switches (nbnxn_cuda.cu:116-119, applied at :303, :367 and :585): always / never run
// the energy flavour, always run the pruning flavour, whatever the step's flags and the list's age.  Read once, like
// the reference's file-scope statics.  They change which kernel runs and what launch_cpyback copies, not what
// wait_gpu hands out (that follows the caller's flags, as in the reference).
struct BenchSwitches { bool always_ener, never_ener, always_prune; };
const BenchSwitches &bench_switches()
{
    static const BenchSwitches s = {getenv("GMX_GPU_ALWAYS_ENER") != nullptr, getenv("GMX_GPU_NEVER_ENER") != nullptr,
                                    getenv("GMX_GPU_ALWAYS_PRUNE") != nullptr};
    return s;
}
inline bool kernel_calc_ener(int flags)
{
    const BenchSwitches &b = bench_switches();
    return (((flags & NBX_FORCE_VIRIAL) != 0) || b.always_ener) && !b.never_ener;
}

// ---- host <-> device moves of the per-step arrays by a kernel instead of the copy engine ----
// The reference's calling convention moves xq in and f (+ energies, shift forces) out every step (nbnxn_cuda.cu:347-348,
// 492-522).  Through hipMemcpyAsync each of those copies has a floor of ~10 us on this stack and the payload then moves
// at 32-36 GB/s (profiles/r3_c_copy_split_probe.txt): 43 + 37 + 16 us around a 63 us kernel.  When the caller's arrays
// are pinned AND mapped into the device's address space -- what nbnxn_hip_pmalloc / the reference's nbat->alloc hook
// give, and what hipHostRegister gives -- a kernel can read and write them directly: no engine hand-over, and the
// force copy and the copy of the partial-sum slots become ONE launch.  Pageable or unaligned arrays keep the copy engine.
// NBNXN_HIP_COPY_KERNELS=0 switches the kernels off (A/B measurements).
typedef float nbx_copy_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) nbx_copy16_kernel(nbx_copy_f32x4 *__restrict__ dst0, const nbx_copy_f32x4 *__restrict__ src0, const int n0,
                                                         nbx_copy_f32x4 *__restrict__ dst1, const nbx_copy_f32x4 *__restrict__ src1, const int n1)
{
    // (nontemporal: the rows are touched once; the force array's lines need not stay in L2 on their way out)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n0) { __builtin_nontemporal_store(__builtin_nontemporal_load(&src0[i]), &dst0[i]); }
    else if (i - n0 < n1) { __builtin_nontemporal_store(__builtin_nontemporal_load(&src1[i - n0]), &dst1[i - n0]); }
}

// launch_cpyback's kernel: blocks [0, ncopy) move the forces to the caller's mapped array; one more block folds the
// partial-sum slots in double, in slot order -- bit for bit what fold_outputs() does on the host -- and writes the 137
// sums to the mapped staging words: the host then reads 1.1 KB instead of summing 36 KB that the device has just
// written (every line of them a miss in the host's caches, ~6 us per step)
__global__ void __launch_bounds__(256) nbx_cpyback_kernel(nbx_copy_f32x4 *__restrict__ dst, const nbx_copy_f32x4 *__restrict__ src, const int n,
                                                          const float *__restrict__ slots, double *__restrict__ folded, const int ncopy)
{
    if ((int)blockIdx.x < ncopy)
    {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < n) { __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]); }
    }
    else
    {
        const int k = threadIdx.x;
        if (k < 2 + NBX_SHIFTS * 3)
        {
            double sum = 0;
            for (int s = 0; s < NBX_OUT_SLOTS; s++) { sum += (double)slots[s * NBX_OUT_STRIDE + k]; }
            folded[k] = sum;
        }
    }
}

bool copy_kernels_default()
{
    static const bool on = !(getenv("NBNXN_HIP_COPY_KERNELS") && atoi(getenv("NBNXN_HIP_COPY_KERNELS")) == 0);
    return on;
}

// the device's view of a pinned, mapped host array, or nullptr (pageable memory, device memory, unknown pointers)
void *device_view_of_host(const nbnxn_hip *nb, const void *p)
{
    if (!p || !nb->copy_kernels) { return nullptr; }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) { return nullptr; }
    return at.devicePointer;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// dst0[0, bytes0) = src0, dst1[0, bytes1) = src1 in one launch on st; all pointers device-visible and 16-byte aligned,
// byte counts multiples of 16 (the callers check)
int copy16(hipStream_t st, void *dst0, const void *src0, size_t bytes0, void *dst1 = nullptr, const void *src1 = nullptr, size_t bytes1 = 0)
{
    const int n0 = (int)(bytes0 / 16), n1 = (int)(bytes1 / 16);
    if (n0 + n1 == 0) { return 0; }
    hipLaunchKernelGGL(nbx_copy16_kernel, dim3((n0 + n1 + 255) / 256), dim3(256), 0, st, (nbx_copy_f32x4 *)dst0, (const nbx_copy_f32x4 *)src0, n0,
                       (nbx_copy_f32x4 *)dst1, (const nbx_copy_f32x4 *)src1, n1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_impl(nbnxn_hip *nb, const float *h_xq, int flags, int iloc)
{
    if (check_loc(iloc)) { return 1; }
    if (!nb->have_const) { return nbx_fail("launch before init_const"); }
    PList &pl = nb->plist[iloc];
    // don't launch the kernel if there is no work to do (nbnxn_cuda.cu:305-309)
    if (pl.nsci == 0) { return 0; }
    if (!nb->shift_vec_uploaded) { return nbx_fail("launch before upload_shiftvec"); }
    hipStream_t st = nb->stream[iloc];
    const bool  ener = kernel_calc_ener(flags);   // energies and shift forces go together (:298-299)
    int         begin, len;
    loc_range(nb, iloc, &begin, &len);

    if (nb->two_streams)
    {
        if (iloc == NBX_LOCAL) { HIP_TRY(hipEventRecord(nb->misc_ops_done, st)); }
        else { HIP_TRY(hipStreamWaitEvent(st, nb->misc_ops_done, 0)); }
    }
    if (h_xq)
    {
        if (nb->do_time) { HIP_TRY(hipEventRecord(nb->h2d_start[iloc], st)); }
        if (len > 0)
        {
            const float *src  = h_xq + (size_t)begin * 4;
            const void  *dsrc = aligned16(src) ? device_view_of_host(nb, src) : nullptr;
            if (dsrc) { if (copy16(st, nb->xq.p + begin, dsrc, (size_t)len * sizeof(float4))) { return 1; } }
            else { HIP_TRY(hipMemcpyAsync(nb->xq.p + begin, src, (size_t)len * sizeof(float4), hipMemcpyHostToDevice, st)); }
        }
        if (nb->do_time) { HIP_TRY(hipEventRecord(nb->h2d_stop[iloc], st)); nb->h2d_timed[iloc] = true; }
        if (mark_x_written(nb, iloc)) { return 1; }
    }

    nbx_kernel_args_t a;
    a.xq = nb->xq.p; a.f = nb->f.p; a.type = nb->type.p; a.shift_vec = nb->shift_vec; a.ntype = nb->ntype;
    a.nbfp = nb->nbfp.p;
    a.c6grid = nb->c6grid.p; a.coulomb_tab = nb->coulomb_tab.p;
    if (nb->vdw_mod == 3 && !a.c6grid) { return nbx_fail("LJ-PME launch without the grid C6 table"); }
    if (nb->el_type == NBX_EL_EWALD_TAB && !a.coulomb_tab) { return nbx_fail("tabulated Ewald launch without the force table"); }
    a.sci = pl.sci.p; a.cj4 = pl.cj4.p; a.excl = pl.excl.p; a.out = nb->out;
    a.zero = nullptr; a.zero_n4 = 0;
    if (nb->clear_in_kernel && iloc == NBX_LOCAL)
    {
        // this launch also zeroes the idle half; clear_outputs will flip to it
        a.zero    = reinterpret_cast<float4 *>(nb->fbase + (size_t)(1 - nb->cur) * nb->half_floats);
        a.zero_n4 = (int)((OUT_FLOATS + (size_t)nb->natoms * 3 + 3) / 4);
    }
    fill_kernel_consts(nb, pl.rlist, a.k);

    if (nb->do_time) { HIP_TRY(hipEventRecord(nb->k_start[iloc], st)); }
    const bool prune = pl.do_prune || bench_switches().always_prune;   // nbnxn_cuda.cu:367
    HIP_TRY(nbx_launch_nonbonded(a, pl.nsci, nb->el_type, ener, prune, nb->rvdw < nb->rcoulomb, nb->vdw_mod, st));
    if (nb->do_time) { HIP_TRY(hipEventRecord(nb->k_stop[iloc], st)); nb->k_timed[iloc] = true; }
    nb->last_ener[iloc] = ener; nb->last_prune[iloc] = pl.do_prune;
    if (a.zero) { nb->clean_other = true; }
    return 0;
}

int cpyback_impl(nbnxn_hip *nb, float *h_f, int flags, int aloc)
{
    if (check_loc(aloc)) { return 1; }
    const int iloc = aloc;
    if (nb->plist[iloc].nsci == 0) { return 0; }
    hipStream_t st = nb->stream[iloc];
    const bool  ener = kernel_calc_ener(flags);   // nbnxn_cuda.cu:585
    int         begin, len;
    loc_range(nb, aloc, &begin, &len);
    if (nb->do_time) { HIP_TRY(hipEventRecord(nb->d2h_start[iloc], st)); }
    // with two streams the local copy-back waits for the non-local kernel, which also
    // writes local forces (nbnxn_cuda.cu:483-503)
    if (iloc == NBX_LOCAL && nb->two_streams) { HIP_TRY(hipStreamWaitEvent(st, nb->nonlocal_done, 0)); }
    const bool   slots       = iloc == NBX_LOCAL && ener;
    const size_t slots_bytes = NBX_OUT_SLOTS * NBX_OUT_STRIDE * sizeof(float);
    bool         slots_done  = false;
    if (len > 0 && h_f)
    {
        float       *dst   = h_f + (size_t)begin * 3;
        const float *src   = nb->f.p + (size_t)begin * 3;
        const size_t bytes = (size_t)len * 3 * sizeof(float);
        void        *ddst  = (aligned16(dst) && aligned16(src) && bytes % 16 == 0) ? device_view_of_host(nb, dst) : nullptr;
        if (ddst && slots && nb->h_fold_dev)
        {
            // forces and, with them, the folded partial sums (h_fold is pinned and mapped by construction): one launch
            const int n = (int)(bytes / 16), ncopy = (n + 255) / 256;
            hipLaunchKernelGGL(nbx_cpyback_kernel, dim3(ncopy + 1), dim3(256), 0, st, (nbx_copy_f32x4 *)ddst, (const nbx_copy_f32x4 *)src, n,
                               (const float *)nb->out, (double *)nb->h_fold_dev, ncopy);
            HIP_TRY(hipGetLastError());
            slots_done = true;
            nb->slots_folded = true;
        }
        else if (ddst) { if (copy16(st, ddst, src, bytes)) { return 1; } }
        else { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st)); }
    }
    // (one-stream handles run both localities in order on the same stream: nothing to signal)
    if (iloc == NBX_NONLOCAL && nb->two_streams) { HIP_TRY(hipEventRecord(nb->nonlocal_done, st)); }
    if (slots && !slots_done)
    {
        nb->slots_folded = false;
        if (nb->h_stage_dev && nb->copy_kernels) { if (copy16(st, nb->h_stage_dev, nb->out, slots_bytes)) { return 1; } }
        else { HIP_TRY(hipMemcpyAsync(nb->h_stage, nb->out, slots_bytes, hipMemcpyDeviceToHost, st)); }
    }
    if (nb->do_time) { HIP_TRY(hipEventRecord(nb->d2h_stop[iloc], st)); nb->d2h_timed[iloc] = true; }
    return 0;
}

int wait_impl(nbnxn_hip *nb, int flags, int aloc, float *e_lj, float *e_el, float *fshift)
{
    if (check_loc(aloc)) { return 1; }
    const int iloc = aloc;
    PList    &pl   = nb->plist[iloc];
    if (pl.nsci == 0) { return 0; }
    const bool ener = (flags & NBX_FORCE_VIRIAL) != 0;
    // The reference waits either in cudaStreamSynchronize or, with GMX_NO_CUDA_STREAMSYNC, by polling host memory for the
    // end of the force copy (nbnxn_cuda.cu:607-624, nbnxn_cuda_data_mgmt.cu:618-705).  Here GMX_NO_HIP_STREAMSYNC polls the
    // stream's state instead of a memory pattern: the host thread spins, and sees the end of the step a few microseconds
    // before a blocked thread would be woken.
    static const bool poll = getenv("GMX_NO_HIP_STREAMSYNC") != nullptr;
    if (poll)
    {
        hipError_t q;
        while ((q = hipStreamQuery(nb->stream[iloc])) == hipErrorNotReady) { }
        HIP_TRY(q);
    }
    else { HIP_TRY(hipStreamSynchronize(nb->stream[iloc])); }
    if (nb->do_time)
    {
        nbx_wallclock_gpu_t &t = nb->timings;
        const int pi = nb->last_prune[iloc] ? 1 : 0, ei = nb->last_ener[iloc] ? 1 : 0;
        if (iloc == NBX_LOCAL) { t.nb_c++; t.ktime[pi][ei].c += 1; }
        if (nb->k_timed[iloc]) { t.ktime[pi][ei].t += elapsed(nb->k_start[iloc], nb->k_stop[iloc]); }
        if (nb->h2d_timed[iloc]) { t.nb_h2d_t += elapsed(nb->h2d_start[iloc], nb->h2d_stop[iloc]); nb->h2d_timed[iloc] = false; }
        if (nb->d2h_timed[iloc]) { t.nb_d2h_t += elapsed(nb->d2h_start[iloc], nb->d2h_stop[iloc]); nb->d2h_timed[iloc] = false; }
    }
    // add up energies and shift forces, once, at the local wait (nbnxn_cuda.cu:662-680)
    if (iloc == NBX_LOCAL && ener)
    {
        double e[2 + NBX_SHIFTS * 3];
        if (nb->slots_folded) { memcpy(e, nb->h_fold, sizeof(e)); }   // the copy-back kernel summed the slots
        else { fold_outputs(nb, e); }
        if (e_lj) { *e_lj += (float)e[0]; }
        if (e_el) { *e_el += (float)e[1]; }
        if (fshift)
        {
            for (int i = 0; i < NBX_SHIFTS * 3; i++) { fshift[i] += (float)e[2 + i]; }
        }
    }
    pl.do_prune = false;
    return 0;
}

} // namespace

extern "C" {

void nbnxn_hip_set_abort_on_error(int abort_on_error) { g_abort_on_error = abort_on_error; }
const char *nbnxn_hip_last_error(void) { return g_last_error.c_str(); }

int nbnxn_hip_detect_gpus(int *ngpu, char *err_str, int err_len)
{
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (ngpu) { *ngpu = (e == hipSuccess) ? n : 0; }
    if (e != hipSuccess)
    {
        if (err_str && err_len > 0) { snprintf(err_str, err_len, "%s", hipGetErrorString(e)); }
        return 1;
    }
    return 0;
}

int nbnxn_hip_init_gpu(int device_id, char *result_str, int result_len)
{
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) { e = hipFree(nullptr); }   // force context creation, as init_gpu does
    if (result_str && result_len > 0) { snprintf(result_str, result_len, "%s", e == hipSuccess ? "" : hipGetErrorString(e)); }
    return e == hipSuccess ? 0 : 1;
}

int nbnxn_hip_free_gpu(char *result_str, int result_len)
{
    hipError_t e = hipDeviceReset();
    if (result_str && result_len > 0) { snprintf(result_str, result_len, "%s", e == hipSuccess ? "" : hipGetErrorString(e)); }
    return e == hipSuccess ? 0 : 1;
}

int nbnxn_hip_get_current_gpu_device_id(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { return -1; }
    return d;
}

int nbnxn_hip_get_gpu_device_info_string(char *s, int len, int device_id)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device_id) != hipSuccess) { return nbx_fail("no device %d", device_id); }
    snprintf(s, len, "#%d: %s (%s), %d CUs, %.0f GiB, clock %d MHz", device_id, p.name, p.gcnArchName,
             p.multiProcessorCount, p.totalGlobalMem / 1073741824.0, p.clockRate / 1000);
    return 0;
}

void nbnxn_hip_pmalloc(void **h_ptr, size_t nbytes)
{
    *h_ptr = nullptr;
    if (nbytes == 0) { return; }
    // (coherent, non-coherent and write-combined pinned memory move xq / f at the same rates under the copy kernels:
    // measured)
    if (hipHostMalloc(h_ptr, nbytes, hipHostMallocDefault) != hipSuccess)
    {
        *h_ptr = nullptr;
        nbx_fail("hipHostMalloc of %zu bytes failed", nbytes);
    }
}

void nbnxn_hip_pfree(void *h_ptr)
{
    if (h_ptr && hipHostFree(h_ptr) != hipSuccess) { nbx_fail("hipHostFree failed"); }
}

int nbnxn_hip_init(FILE *fplog, nbnxn_hip_ptr_t *p_nb, int device_id, int bLocalAndNonlocal)
{
    if (!p_nb) { return nbx_fail("null output pointer"); }
    *p_nb = nullptr;
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        return nbx_fail("device %d is %s; this library carries gfx950 (MI355X) code objects only", device_id, prop.gcnArchName);
    }
    if (bind_scratch(device_id)) { return 1; }
    nbnxn_hip *nb = new nbnxn_hip();
    nb->device      = device_id;
    nb->two_streams = bLocalAndNonlocal != 0;
    // With two streams the local kernel fills every wave slot of the chip for the whole step (one-wave workgroups
    // at 4 waves per SIMD, all 512 VGPRs of a SIMD taken), and RCCL's send / receive kernel -- two or three
    // workgroups of 256 threads, 132 VGPRs, 20 KB LDS -- then finds no room until the local kernel drains (a 6 us
    // kernel took 58 us, profiles/README r1_j).  So the local stream leaves one CU of every shader engine alone:
    // CU-mask bit b is XCC b mod 8, engine (b / 8) mod 4, CU b / 32 of that engine (tools/ubench/cu_mask_probe.hip),
    // i.e. the first 32 bits.  Workgroups are dealt to the engines in turn, so fewer than one CU per engine does not
    // help (8 and 16 were tried).  The local kernel loses 12 % of its CUs and the exchange runs beside it: the
    // decomposed step went from 0.105 to 0.098 ms on one rank.  GMX_HIP_NB_RESERVE_CUS overrides the count (0: off).
    int reserve = nb->two_streams ? 32 : 0;
    if (const char *env = getenv("GMX_HIP_NB_RESERVE_CUS")) { reserve = nb->two_streams ? atoi(env) : 0; }
    reserve = std::max(0, std::min(reserve, prop.multiProcessorCount / 2));
    nb->reserved_cus = reserve;
    if (reserve > 0)
    {
        std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
        for (int b = 0; b < reserve; b++) { mask[b / 32] &= ~(1u << (b % 32)); }
        if (prop.multiProcessorCount % 32) { mask.back() &= (1u << (prop.multiProcessorCount % 32)) - 1; }
        // a platform that does not take CU masks gets an ordinary stream: the run is correct without the reservation.
        // hipExtStreamCreateWithCUMask has no flags argument: the stream it makes is a BLOCKING one (it orders itself
        // against the null stream), where every other stream of the module is hipStreamNonBlocking.  The module never
        // uses the null stream and no ordering in it relies on that implicit synchronisation (every cross-stream
        // dependency is an event); a host that puts work on the null stream sees it serialise with the masked local
        // stream only.  The two-stream GPU tests run with the reservation on and off (tests/test_gpu_z_dd.py).
        if (hipExtStreamCreateWithCUMask(&nb->stream[0], (uint32_t)mask.size(), mask.data()) != hipSuccess)
        {
            (void)hipGetLastError();
            nb->stream[0]    = nullptr;
            nb->reserved_cus = 0;
        }
    }
    if (nb->stream[0] == nullptr)
    {
        HIP_TRY(hipStreamCreateWithFlags(&nb->stream[0], hipStreamNonBlocking));
    }
    if (nb->two_streams)
    {
        // The reference gives the non-local stream the high priority (nbnxn_cuda_data_mgmt.cu:569-594).  Here BOTH
        // streams get the normal one: the two streams wait for each other through hipStreamWaitEvent, which is a
        // barrier packet at the head of a hardware queue, and when several processes share the GPU a queue of HIGHER
        // priority parked on such a packet starved the lower-priority queue it was waiting for -- for good (round 3,
        // tools/stall_ab.sh with four processes on one GPU: non-local high, local normal: the local stream's first
        // operation never ran, 1 run of 1 and about half the runs of round 2; non-local LOW: the mirror image, the
        // non-local stream never ran, run 4; both normal: 10 of 10 runs clean).  The CU-masked local stream cannot be
        // given a priority (hipExtStreamCreateWithCUMask takes none), so equal means normal.  DESIGN section 6.
        // GMX_HIP_NB_NONLOCAL_PRIORITY=high|low|normal overrides (one process per GPU never showed the stall).
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char *prio = getenv("GMX_HIP_NB_NONLOCAL_PRIORITY");
        nb->nonlocal_high_priority = prio ? (strcmp(prio, "high") == 0) : false;
        // hipDeviceGetStreamPriorityRange gives {least, greatest} = {1, -1} here: low, NORMAL = 0 in between, high
        int pr = nb->nonlocal_high_priority ? hi : (lo + hi) / 2;
        if (prio && strcmp(prio, "low") == 0) { pr = lo; }
        HIP_TRY(hipStreamCreateWithPriority(&nb->stream[1], hipStreamNonBlocking, pr));
        HIP_TRY(hipEventCreateWithFlags(&nb->misc_ops_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&nb->nonlocal_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&nb->stream_mark[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&nb->stream_mark[1], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&nb->x_written, hipEventDisableTiming));
    }
    else
    {
        nb->stream[1] = nb->stream[0];
    }
    // event timing is off with two streams, as in the reference (nbnxn_cuda_data_mgmt.cu:714-715):
    // the events serialise the streams; on MI355X a pair also costs ~7 us of device time per launch
    nb->do_time = !nb->two_streams && (getenv("GMX_DISABLE_HIP_TIMING") == nullptr);
    for (int l = 0; l < 2; l++)
    {
        HIP_TRY(hipEventCreate(&nb->k_start[l]));   HIP_TRY(hipEventCreate(&nb->k_stop[l]));
        HIP_TRY(hipEventCreate(&nb->h2d_start[l])); HIP_TRY(hipEventCreate(&nb->h2d_stop[l]));
        HIP_TRY(hipEventCreate(&nb->d2h_start[l])); HIP_TRY(hipEventCreate(&nb->d2h_stop[l]));
    }
    HIP_TRY(hipMalloc((void **)&nb->shift_vec, NBX_SHIFTS * 3 * sizeof(float)));
    { bool re; if (reserve_f(nb, 0, &re)) { return 1; } }
    HIP_TRY(hipHostMalloc((void **)&nb->h_stage, NBX_OUT_SLOTS * NBX_OUT_STRIDE * sizeof(float), hipHostMallocDefault));
    if (hipHostGetDevicePointer(&nb->h_stage_dev, nb->h_stage, 0) != hipSuccess) { (void)hipGetLastError(); nb->h_stage_dev = nullptr; }
    nb->copy_kernels = copy_kernels_default();
    HIP_TRY(hipHostMalloc((void **)&nb->h_fold, (2 + NBX_SHIFTS * 3) * sizeof(double), hipHostMallocDefault));
    if (hipHostGetDevicePointer(&nb->h_fold_dev, nb->h_fold, 0) != hipSuccess) { (void)hipGetLastError(); nb->h_fold_dev = nullptr; }
    memset(&nb->timings, 0, sizeof(nb->timings));
    if (clear_e_fshift(nb)) { return 1; }
    HIP_TRY(hipStreamSynchronize(nb->stream[0]));
    if (fplog)
    {
        fprintf(fplog, "Initialized HIP nonbonded module on device %d: %s (%s), %d CUs\n", device_id, prop.name,
                prop.gcnArchName, prop.multiProcessorCount);
    }
    *p_nb = nb;
    return 0;
}

int nbnxn_hip_init_const_flat(nbnxn_hip_ptr_t nb, int el_type, float rcoulomb, float rvdw, float rlist,
                              float epsfac, float k_rf, float c_rf, float ewaldcoeff_q, float sh_ewald,
                              float disp_cpot, float rep_cpot, int ntype, const float *nbfp)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    if (set_params(nb, el_type, rcoulomb, rvdw, rlist, epsfac, k_rf, c_rf, ewaldcoeff_q, sh_ewald, disp_cpot, rep_cpot)) { return 1; }
    nb->vdw_mod = 0; nb->rvdw_switch = 0;   // plain cut-off until nbnxn_hip_set_vdw_modifier_flat says otherwise
    nb->kc_valid = false;
    if (upload_nbfp(nb, ntype, nbfp)) { return 1; }
    if (clear_e_fshift(nb)) { return 1; }
    HIP_TRY(hipStreamSynchronize(nb->stream[0]));
    return 0;
}

int nbnxn_hip_set_vdw_modifier_flat(nbnxn_hip_ptr_t nb, int vdw_modifier, float rvdw_switch,
                                    float disp_c2, float disp_c3, float rep_c2, float rep_c3,
                                    float sw_c3, float sw_c4, float sw_c5)
{
    if (check_handle(nb)) { return 1; }
    if (!nb->have_const) { return nbx_fail("set_vdw_modifier before init_const"); }
    int mod;
    if (vdw_modifier == NBX_MOD_NONE || vdw_modifier == NBX_MOD_POTSHIFT) { mod = 0; }
    else if (vdw_modifier == NBX_MOD_FORCESWITCH) { mod = 1; }
    else if (vdw_modifier == NBX_MOD_POTSWITCH) { mod = 2; }
    else { return nbx_fail("VdW modifier %d is not built (cut-off, potential shift, force switch, potential switch are)", vdw_modifier); }
    if (mod != 0 && !(rvdw_switch >= 0 && rvdw_switch < nb->rvdw))
    {
        return nbx_fail("rvdw_switch (%g) must be in [0, rvdw = %g)", rvdw_switch, nb->rvdw);
    }
    nb->vdw_mod     = mod;
    nb->kc_valid    = false;
    nb->rvdw_switch = mod ? rvdw_switch : 0.0f;
    nb->fsw[0] = disp_c2; nb->fsw[1] = disp_c3; nb->fsw[2] = rep_c2; nb->fsw[3] = rep_c3;
    nb->psw[0] = sw_c3; nb->psw[1] = sw_c4; nb->psw[2] = sw_c5;
    return 0;
}

int nbnxn_hip_set_ljpme_flat(nbnxn_hip_ptr_t nb, int comb_rule, float ewaldcoeff_lj, float sh_lj_ewald, const float *nbfp_comb)
{
    if (check_handle(nb)) { return 1; }
    if (!nb->have_const) { return nbx_fail("set_ljpme before init_const"); }
    if (comb_rule != NBX_LJCR_GEOM && comb_rule != NBX_LJCR_LB) { return nbx_fail("LJ-PME needs the geometric or the Lorentz-Berthelot combination rule, got %d", comb_rule); }
    if (!(ewaldcoeff_lj > 0) || !nbfp_comb) { return nbx_fail("LJ-PME needs ewaldcoeff_lj > 0 and the per-type combination parameters"); }
    HIP_TRY(hipSetDevice(nb->device));
    // grid C6 of every type pair, in float and in the reference's operation order
    // (nbnxn_cuda_kernel_utils.cuh:202, :276-283): the combination rule is spent here, the kernel only reads the table
    const int          nt = nb->ntype;
    std::vector<float> g((size_t)nt * nt);
    for (int i = 0; i < nt; i++)
    {
        for (int j = 0; j < nt; j++)
        {
            if (comb_rule == NBX_LJCR_GEOM) { g[(size_t)i * nt + j] = nbfp_comb[2 * i] * nbfp_comb[2 * j]; }
            else
            {
                const float sigma  = nbfp_comb[2 * i] + nbfp_comb[2 * j];
                const float eps    = nbfp_comb[2 * i + 1] * nbfp_comb[2 * j + 1];
                const float sigma2 = sigma * sigma;
                g[(size_t)i * nt + j] = eps * sigma2 * sigma2 * sigma2;
            }
        }
    }
    if (dev_reserve(nb->c6grid, g.size())) { return 1; }
    HIP_TRY(hipMemcpyAsync(nb->c6grid.p, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice, nb->stream[0]));
    HIP_TRY(hipStreamSynchronize(nb->stream[0]));
    nb->vdw_mod       = 3;
    nb->kc_valid      = false;
    nb->rvdw_switch   = 0;
    nb->ewaldcoeff_lj = ewaldcoeff_lj;
    nb->sh_lj_ewald   = sh_lj_ewald;
    return 0;
}

int nbnxn_hip_init_const(nbnxn_hip_ptr_t nb, const nbx_interaction_const_t *ic, const nbx_verlet_group_t *nbv_group)
{
    if (check_handle(nb)) { return 1; }
    if (!ic || !nbv_group || !nbv_group[0].nbat) { return nbx_fail("init_const: missing interaction constants or atom data"); }
    const nbx_atomdata_t *nbat = nbv_group[0].nbat;
    int                   el;
    // flavour choice of nbnxn_cuda_data_mgmt.cu:303-357, reduced to what is built here
    if (ic->vdwtype != NBX_VDW_CUT && ic->vdwtype != NBX_VDW_PME)
    {
        return nbx_fail("The requested VdW type %d is not implemented in the HIP GPU accelerated kernels!", ic->vdwtype);
    }
    if (ic->eeltype == NBX_EEL_CUT || ic->eeltype == NBX_EEL_RF || ic->eeltype == NBX_EEL_GRF) { el = NBX_EL_RF; }
    else if (ic->eeltype == NBX_EEL_PME || ic->eeltype == NBX_EEL_EWALD)
    {
        // pick_ewald_kernel_type (nbnxn_cuda_data_mgmt.cu:217-265): analytical unless the tabulated
        // flavour is forced through the environment
        const bool force_ana = getenv("GMX_HIP_NB_ANA_EWALD") != nullptr, force_tab = getenv("GMX_HIP_NB_TAB_EWALD") != nullptr;
        if (force_ana && force_tab) { return nbx_fail("Both analytical and tabulated Ewald HIP non-bonded kernels requested through environment variables."); }
        el = force_tab ? NBX_EL_EWALD_TAB : NBX_EL_EWALD_ANA;
    }
    else { return nbx_fail("The requested electrostatics type %d is not implemented in the HIP GPU accelerated kernels!", ic->eeltype); }
    // eelCUT runs the RF kernel with k_rf = 0, c_rf = 1/rc as the reference sets them (forcerec.c:2072-2079)
    if (nbnxn_hip_init_const_flat(nb, el, ic->rcoulomb, ic->rvdw, ic->rlist, ic->epsfac, ic->k_rf, ic->c_rf,
                                  ic->ewaldcoeff_q, ic->sh_ewald, ic->dispersion_shift.cpot, ic->repulsion_shift.cpot,
                                  nbat->ntype, nbat->nbfp)) { return 1; }
    if (ic->vdwtype == NBX_VDW_PME)
    {
        if (nbat->comb_rule != ic->ljpme_comb_rule) { return nbx_fail("LJ-PME combination rule %d differs from the atom data's (%d)", ic->ljpme_comb_rule, nbat->comb_rule); }
        return nbnxn_hip_set_ljpme_flat(nb, ic->ljpme_comb_rule, ic->ewaldcoeff_lj, ic->sh_lj_ewald, nbat->nbfp_comb);
    }
    return nbnxn_hip_set_vdw_modifier_flat(nb, ic->vdw_modifier, ic->rvdw_switch, ic->dispersion_shift.c2, ic->dispersion_shift.c3,
                                           ic->repulsion_shift.c2, ic->repulsion_shift.c3, ic->vdw_switch.c3, ic->vdw_switch.c4,
                                           ic->vdw_switch.c5);
}

int nbnxn_hip_pme_loadbal_update_param(nbnxn_hip_ptr_t nb, const nbx_interaction_const_t *ic)
{
    if (check_handle(nb)) { return 1; }
    if (!nb->have_const) { return nbx_fail("pme_loadbal_update_param before init_const"); }
    return set_params(nb, nb->el_type, ic->rcoulomb, ic->rvdw, ic->rlist, ic->epsfac, ic->k_rf, ic->c_rf,
                      ic->ewaldcoeff_q, ic->sh_ewald, ic->dispersion_shift.cpot, ic->repulsion_shift.cpot);
}

int nbnxn_hip_init_pairlist_flat(nbnxn_hip_ptr_t nb, int iloc, float rlist, int nsci, const nbx_sci_t *sci,
                                 int ncj4, const nbx_cj4_t *cj4, int nexcl, const nbx_excl_t *excl, int nci_tot)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    return init_pairlist_impl(nb, iloc, rlist, nsci, sci, ncj4, cj4, nexcl, excl, nci_tot);
}

int nbnxn_hip_init_pairlist(nbnxn_hip_ptr_t nb, const nbx_pairlist_t *l, int iloc)
{
    if (check_handle(nb)) { return 1; }
    if (!l) { return nbx_fail("null pair list"); }
    if (l->bSimple || l->na_ci != NBX_CLUSTER_SIZE || l->na_cj != NBX_CLUSTER_SIZE)
    {
        return nbx_fail("The pair list is not in the 8x8x8 GPU format (bSimple %d, cluster size %d)", l->bSimple, l->na_ci);
    }
    return nbnxn_hip_init_pairlist_flat(nb, iloc, l->rlist, l->nsci, l->sci, l->ncj4, l->cj4, l->nexcl, l->excl, l->nci_tot);
}

int nbnxn_hip_init_atomdata_flat(nbnxn_hip_ptr_t nb, int natoms, int natoms_local, int ntype, const int *type)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    return init_atomdata_impl(nb, natoms, natoms_local, ntype, type);
}

int nbnxn_hip_init_atomdata(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat)
{
    if (check_handle(nb)) { return 1; }
    if (!nbat) { return nbx_fail("null atom data"); }
    if (nbat->xstride != 4 || nbat->fstride != 3) { return nbx_fail("atom data must be XYZQ/XYZ (xstride 4, fstride 3), got %d/%d", nbat->xstride, nbat->fstride); }
    return nbnxn_hip_init_atomdata_flat(nb, nbat->natoms, nbat->natoms_local, nbat->ntype, nbat->type);
}

int nbnxn_hip_upload_shiftvec_flat(nbnxn_hip_ptr_t nb, const float *shift_vec)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    HIP_TRY(hipMemcpyAsync(nb->shift_vec, shift_vec, NBX_SHIFTS * 3 * sizeof(float), hipMemcpyHostToDevice, nb->stream[0]));
    HIP_TRY(hipStreamSynchronize(nb->stream[0]));
    nb->shift_vec_uploaded = true;
    return 0;
}

int nbnxn_hip_upload_shiftvec(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat)
{
    if (check_handle(nb)) { return 1; }
    // only if the box changes or nothing was uploaded yet (nbnxn_cuda_data_mgmt.cu:803-814)
    if (nbat->bDynamicBox || !nb->shift_vec_uploaded)
    {
        return nbnxn_hip_upload_shiftvec_flat(nb, &nbat->shift_vec[0][0]);
    }
    return 0;
}

int nbnxn_hip_clear_outputs(nbnxn_hip_ptr_t nb, int flags)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    if (nb->clear_in_kernel && nb->clean_other)
    {
        // the last LOCAL launch left the idle half zeroed (stream order makes that visible to
        // everything enqueued from here on): use it
        nb->cur ^= 1;
        nb->clean_other = false;
        nb->out = nb->fbase + (size_t)nb->cur * nb->half_floats;
        nb->f.p = nb->out + OUT_FLOATS;
        return 0;
    }
    if (flags & NBX_FORCE_VIRIAL)
    {
        // slots and forces are adjacent: one fill
        HIP_TRY(hipMemsetAsync(nb->out, 0, (OUT_FLOATS + (size_t)nb->natoms * 3) * sizeof(float), nb->stream[0]));
    }
    else if (nb->natoms > 0) { HIP_TRY(hipMemsetAsync(nb->f.p, 0, (size_t)nb->natoms * 3 * sizeof(float), nb->stream[0])); }
    return 0;
}

int nbnxn_hip_free(nbnxn_hip_ptr_t nb)
{
    if (!nb) { return 0; }
    (void)hipSetDevice(nb->device);
    (void)hipDeviceSynchronize();
    dev_free(nb->xq); (void)hipFree(nb->fbase); nb->fbase = nullptr; nb->f = DevBuf<float>(); nb->out = nullptr; dev_free(nb->type); dev_free(nb->nbfp);
    for (int l = 0; l < 2; l++)
    {
        dev_free(nb->plist[l].sci); dev_free(nb->plist[l].cj4); dev_free(nb->plist[l].excl); dev_free(nb->plist[l].imask_outer);
        (void)hipEventDestroy(nb->k_start[l]);   (void)hipEventDestroy(nb->k_stop[l]);
        (void)hipEventDestroy(nb->h2d_start[l]); (void)hipEventDestroy(nb->h2d_stop[l]);
        (void)hipEventDestroy(nb->d2h_start[l]); (void)hipEventDestroy(nb->d2h_stop[l]);
    }
    (void)hipFree(nb->shift_vec);
    (void)hipHostFree(nb->h_stage);
    if (nb->h_fold) { (void)hipHostFree(nb->h_fold); }
    dev_free(nb->d_a); dev_free(nb->d_cell); dev_free(nb->d_x); dev_free(nb->d_fa);
    dev_free(nb->c6grid); dev_free(nb->coulomb_tab);
    dev_free(nb->d_type_in); dev_free(nb->d_excl_index); dev_free(nb->d_excl_a); dev_free(nb->d_q_in);
    nbx_search_destroy(nb->search); nb->search = nullptr;
    (void)nbnxn_hip_dd_finish(nb);
    if (nb->h_fa) { (void)hipHostFree(nb->h_fa); }
    if (nb->two_streams)
    {
        (void)hipEventDestroy(nb->misc_ops_done); (void)hipEventDestroy(nb->nonlocal_done);
        (void)hipEventDestroy(nb->stream_mark[0]); (void)hipEventDestroy(nb->stream_mark[1]);
        (void)hipEventDestroy(nb->x_written);
        (void)hipStreamDestroy(nb->stream[1]);
    }
    (void)hipStreamDestroy(nb->stream[0]);
    delete nb;
    return 0;
}

nbx_wallclock_gpu_t *nbnxn_hip_get_timings(nbnxn_hip_ptr_t nb)
{
    return (nb && nb->do_time) ? &nb->timings : nullptr;
}

int nbnxn_hip_reset_timings(nbnxn_hip_ptr_t nb)
{
    if (check_handle(nb)) { return 1; }
    memset(&nb->timings, 0, sizeof(nb->timings));
    return 0;
}

int nbnxn_hip_min_ci_balanced(nbnxn_hip_ptr_t nb)
{
    // The reference asks for 40 entries per multiprocessor (nbnxn_cuda_data_mgmt.cu:76), and 40 per CU is also where
    // this kernel's trade between the launch's tail and the per-entry prologue / epilogue lies on MI355X: one entry is
    // one wave64, a CU holds 16 of them.  Measured on the 97,200-atom water box (round 2 kernel, Ewald V+F / RF F
    // kernel time): 20 per CU 71.6 / 45.8 us, 30: 66.7 / 42.1, 40: 63.7 / 42.9, 60: 65.1 / 45.3, 80: 66.5 / 46.9.
    int        ncu = 256;
    hipDeviceProp_t p;
    if (nb && hipGetDeviceProperties(&p, nb->device) == hipSuccess) { ncu = p.multiProcessorCount; }
    return 40 * ncu;
}

int nbnxn_hip_is_kernel_ewald_analytical(const nbnxn_hip_ptr_t nb) { return nb && nb->el_type == NBX_EL_EWALD_ANA; }

int nbnxn_hip_launch_kernel_flat(nbnxn_hip_ptr_t nb, const float *xq, int flags, int iloc)
{
    if (check_handle(nb)) { return 1; }
    if (!xq) { return nbx_fail("null coordinates"); }
    HIP_TRY(hipSetDevice(nb->device));
    return launch_impl(nb, xq, flags, iloc);
}

int nbnxn_hip_launch_kernel(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat, int flags, int iloc)
{
    if (check_handle(nb)) { return 1; }
    if (!nbat || nbat->natoms != nb->natoms) { return nbx_fail("atom data does not match init_atomdata"); }
    return nbnxn_hip_launch_kernel_flat(nb, nbat->x, flags, iloc);
}

int nbnxn_hip_launch_kernel_resident(nbnxn_hip_ptr_t nb, int flags, int iloc)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    return launch_impl(nb, nullptr, flags, iloc);
}

int nbnxn_hip_launch_kernel_noalloc(nbnxn_hip_ptr_t nb, int flags, int iloc, int natoms, const float *d_xq, float *d_f,
                                    float *d_e_fshift)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    if (!nb->have_const) { return nbx_fail("launch before init_const"); }
    if (!nb->shift_vec_uploaded) { return nbx_fail("launch before upload_shiftvec"); }
    const bool ener = (flags & NBX_FORCE_VIRIAL) != 0;
    if (natoms > NBX_NOALLOC_MAX_ATOMS)
    {
        return nbx_fail("launch_kernel_noalloc: %d atoms exceed the maximum of %d this entry point takes", natoms, NBX_NOALLOC_MAX_ATOMS);
    }
    if (natoms != nb->natoms) { return nbx_fail("launch_kernel_noalloc: %d atoms passed, the installed atom data and list are for %d", natoms, nb->natoms); }
    if (!d_xq || !d_f || (ener && !d_e_fshift)) { return nbx_fail("launch_kernel_noalloc: null coordinate, force or energy array"); }
    PList &pl = nb->plist[iloc];
    if (pl.nsci == 0) { return 0; }
    if (pl.nsci > NBX_NOALLOC_MAX_NSCI)
    {
        return nbx_fail("launch_kernel_noalloc: %d list entries exceed the maximum of %d this entry point takes", pl.nsci, NBX_NOALLOC_MAX_NSCI);
    }
    if (!g_scratch_ptr[nb->device]) { return nbx_fail("launch_kernel_noalloc: the scratch array of device %d was never bound", nb->device); }
    float *slots = g_scratch_ptr[nb->device] + (size_t)iloc * NBX_NOALLOC_SCRATCH_FLOATS;   // a half per locality
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = nb->stream[iloc];
    // the ordering launch_impl keeps (nbnxn_cuda.cu:325-337): the non-local launch starts after whatever the local
    // stream held when the local launch was issued (clears, coordinate copies)
    if (nb->two_streams)
    {
        if (iloc == NBX_LOCAL) { HIP_TRY(hipEventRecord(nb->misc_ops_done, st)); }
        else { HIP_TRY(hipStreamWaitEvent(st, nb->misc_ops_done, 0)); }
    }

    nbx_kernel_args_t a;
    a.xq = reinterpret_cast<const float4 *>(d_xq); a.f = d_f; a.type = nb->type.p; a.shift_vec = nb->shift_vec; a.ntype = nb->ntype;
    a.nbfp = nb->nbfp.p; a.c6grid = nb->c6grid.p; a.coulomb_tab = nb->coulomb_tab.p;
    if (nb->vdw_mod == 3 && !a.c6grid) { return nbx_fail("LJ-PME launch without the grid C6 table"); }
    if (nb->el_type == NBX_EL_EWALD_TAB && !a.coulomb_tab) { return nbx_fail("tabulated Ewald launch without the force table"); }
    a.sci = pl.sci.p; a.cj4 = pl.cj4.p; a.excl = pl.excl.p;
    a.out = slots;              // the device address from hipGetSymbolAddress, never the symbol's name
    a.zero = nullptr; a.zero_n4 = 0;
    fill_kernel_consts(nb, pl.rlist, a.k);
    // the array keeps what the last call on this GPU left in it
    if (ener) { HIP_TRY(hipMemsetAsync(slots, 0, NBX_NOALLOC_SCRATCH_FLOATS * sizeof(float), st)); }
    HIP_TRY(nbx_launch_nonbonded(a, pl.nsci, nb->el_type, ener, pl.do_prune, nb->rvdw < nb->rcoulomb, nb->vdw_mod, st));
    if (ener)
    {
        hipLaunchKernelGGL(nbx_fold_slots_kernel, dim3(1), dim3(192), 0, st, slots, d_e_fshift);
        HIP_TRY(hipGetLastError());
    }
    // the list was pruned by this launch; launches after it on this stream see the pruned masks
    pl.do_prune = false;
    return 0;
}

int nbnxn_hip_launch_cpyback_flat(nbnxn_hip_ptr_t nb, float *f, int flags, int aloc)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    return cpyback_impl(nb, f, flags, aloc);
}

int nbnxn_hip_launch_cpyback(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat, int flags, int aloc)
{
    if (check_handle(nb)) { return 1; }
    if (!nbat || !nbat->out || !nbat->out[0].f) { return nbx_fail("atom data has no force output"); }
    return nbnxn_hip_launch_cpyback_flat(nb, nbat->out[0].f, flags, aloc);
}

int nbnxn_hip_wait_gpu_flat(nbnxn_hip_ptr_t nb, int flags, int aloc, float *e_lj, float *e_el, float *fshift)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    return wait_impl(nb, flags, aloc, e_lj, e_el, fshift);
}

int nbnxn_hip_wait_gpu(nbnxn_hip_ptr_t nb, const nbx_atomdata_t *nbat, int flags, int aloc,
                       float *e_lj, float *e_el, float (*fshift)[3])
{
    (void)nbat;
    return nbnxn_hip_wait_gpu_flat(nb, flags, aloc, e_lj, e_el, fshift ? &fshift[0][0] : nullptr);
}

int nbnxn_hip_get_device_buffers(nbnxn_hip_ptr_t nb, void **d_xq, void **d_f, void **d_type)
{
    if (check_handle(nb)) { return 1; }
    if (d_xq) { *d_xq = nb->xq.p; }
    if (d_f) { *d_f = nb->f.p; }
    if (d_type) { *d_type = nb->type.p; }
    return 0;
}

int nbnxn_hip_download_coulomb_tab(nbnxn_hip_ptr_t nb, float *tab, int tab_alloc, int *tab_size, float *tab_scale)
{
    if (check_handle(nb)) { return 1; }
    if (nb->el_type != NBX_EL_EWALD_TAB || !nb->coulomb_tab.p) { return nbx_fail("no Ewald force table: the flavour in use is not NBX_EL_EWALD_TAB"); }
    if (tab_size) { *tab_size = nb->tab_size; }
    if (tab_scale) { *tab_scale = nb->tab_scale; }
    if (tab)
    {
        if (tab_alloc < nb->tab_size) { return nbx_fail("download_coulomb_tab: room for %d entries, table has %d", tab_alloc, nb->tab_size); }
        HIP_TRY(hipSetDevice(nb->device));
        HIP_TRY(hipMemcpy(tab, nb->coulomb_tab.p, nb->tab_size * sizeof(float), hipMemcpyDeviceToHost));
        // undo the kernel's unit scaling (exact to half an ulp; the comparison with the reference table allows for it)
        for (int i = 0; i < nb->tab_size; i++) { tab[i] = (float)((double)tab[i] * (double)nb->beta * (double)nb->beta); }
    }
    return 0;
}

int nbnxn_hip_get_output_slots(nbnxn_hip_ptr_t nb, void **d_out, int *nslots, int *stride)
{
    if (check_handle(nb)) { return 1; }
    if (d_out) { *d_out = nb->out; }
    if (nslots) { *nslots = NBX_OUT_SLOTS; }
    if (stride) { *stride = NBX_OUT_STRIDE; }
    return 0;
}

void *nbnxn_hip_get_stream(nbnxn_hip_ptr_t nb, int iloc)
{
    if (!nb || iloc < 0 || iloc > 1) { return nullptr; }
    return (void *)nb->stream[iloc];
}

int nbnxn_hip_read_outputs(nbnxn_hip_ptr_t nb, float *e_lj, float *e_el, float *fshift)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = nb->stream[0];
    if (nb->two_streams) { HIP_TRY(hipStreamSynchronize(nb->stream[1])); }
    HIP_TRY(hipMemcpyAsync(nb->h_stage, nb->out, NBX_OUT_SLOTS * NBX_OUT_STRIDE * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double e[2 + NBX_SHIFTS * 3];
    fold_outputs(nb, e);
    if (e_lj) { *e_lj = (float)e[0]; }
    if (e_el) { *e_el = (float)e[1]; }
    if (fshift)
    {
        for (int i = 0; i < NBX_SHIFTS * 3; i++) { fshift[i] = (float)e[2 + i]; }
    }
    return 0;
}

int nbnxn_hip_download_cj4(nbnxn_hip_ptr_t nb, int iloc, int ncj4, nbx_cj4_t *cj4)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    if (ncj4 != nb->plist[iloc].ncj4) { return nbx_fail("list on the device has %d cj4 entries, caller asked for %d", nb->plist[iloc].ncj4, ncj4); }
    HIP_TRY(hipSetDevice(nb->device));
    HIP_TRY(hipStreamSynchronize(nb->stream[iloc]));
    if (ncj4 > 0) { HIP_TRY(hipMemcpy(cj4, nb->plist[iloc].cj4.p, (size_t)ncj4 * sizeof(nbx_cj4_t), hipMemcpyDeviceToHost)); }
    return 0;
}

int nbnxn_hip_verify_pairlist_flat(nbnxn_hip_ptr_t nb, int iloc, int nsci, const nbx_sci_t *sci,
                                   int ncj4, const nbx_cj4_t *cj4, int nexcl, const nbx_excl_t *excl)
{
    if (check_handle(nb) || check_loc(iloc)) { return -1; }
    PList &pl = nb->plist[iloc];
    if (nsci != pl.nsci || ncj4 != pl.ncj4 || nexcl != pl.nexcl)
    {
        nbx_fail("list sizes differ: device %d/%d/%d, host %d/%d/%d", pl.nsci, pl.ncj4, pl.nexcl, nsci, ncj4, nexcl);
        return -1;
    }
    if (hipSetDevice(nb->device) != hipSuccess || hipStreamSynchronize(nb->stream[iloc]) != hipSuccess) { return -1; }
    std::vector<nbx_sci_t>  dsci(nsci);
    std::vector<nbx_cj4_t>  dcj4(ncj4);
    std::vector<nbx_excl_t> dexcl(nexcl);
    if ((nsci && hipMemcpy(dsci.data(), pl.sci.p, (size_t)nsci * sizeof(nbx_sci_t), hipMemcpyDeviceToHost) != hipSuccess) ||
        (ncj4 && hipMemcpy(dcj4.data(), pl.cj4.p, (size_t)ncj4 * sizeof(nbx_cj4_t), hipMemcpyDeviceToHost) != hipSuccess) ||
        (nexcl && hipMemcpy(dexcl.data(), pl.excl.p, (size_t)nexcl * sizeof(nbx_excl_t), hipMemcpyDeviceToHost) != hipSuccess))
    {
        nbx_fail("read-back of the pair list failed");
        return -1;
    }
    int ndiff = 0;
    for (int s = 0; s < nsci; s++) { ndiff += memcmp(&dsci[s], &sci[s], sizeof(nbx_sci_t)) != 0; }
    for (int c = 0; c < ncj4; c++)
    {
        for (int w = 0; w < 2; w++)
        {
            ndiff += dcj4[c].imei[w].imask != cj4[c].imei[w].imask;
            ndiff += dcj4[c].imei[w].excl_ind != cj4[c].imei[w].excl_ind;
        }
        const unsigned m = cj4[c].imei[0].imask | cj4[c].imei[1].imask;
        for (int jm = 0; jm < 4; jm++)
        {
            // slots without interaction bits are padding with undefined cj (SURVEY 8a trap 1)
            if ((m >> (8 * jm)) & 0xffu) { ndiff += dcj4[c].cj[jm] != cj4[c].cj[jm]; }
        }
    }
    for (int x = 0; x < nexcl; x++)
    {
        for (int t = 0; t < 32; t++) { ndiff += dexcl[x].pair[t] != excl[x].pair[t]; }
    }
    return ndiff;
}

float nbnxn_hip_last_kernel_ms(nbnxn_hip_ptr_t nb, int iloc)
{
    if (!nb || iloc < 0 || iloc > 1 || !nb->do_time || !nb->k_timed[iloc]) { return -1.0f; }
    if (hipEventSynchronize(nb->k_stop[iloc]) != hipSuccess) { return -1.0f; }
    return elapsed(nb->k_start[iloc], nb->k_stop[iloc]);
}

/* ---- device-side atom-data moves ---- */

int nbnxn_hip_set_atom_order(nbnxn_hip_ptr_t nb, int natoms_real, const int *a, const int *cell)
{
    if (check_handle(nb)) { return 1; }
    if (nb->natoms <= 0) { return nbx_fail("set_atom_order before init_atomdata"); }
    if (!a || !cell || natoms_real < 0 || natoms_real > nb->natoms) { return nbx_fail("bad atom order arguments"); }
    // the two index arrays must be inverse permutations of each other (nbs->a / nbs->cell)
    int nreal = 0;
    for (int s = 0; s < nb->natoms; s++)
    {
        if (a[s] < -1 || a[s] >= natoms_real) { return nbx_fail("grid slot %d holds atom %d outside [-1,%d)", s, a[s], natoms_real); }
        if (a[s] >= 0) { nreal++; if (cell[a[s]] != s) { return nbx_fail("cell[a[%d]] = %d is not %d", s, cell[a[s]], s); } }
    }
    if (nreal != natoms_real) { return nbx_fail("%d grid slots hold atoms, expected %d", nreal, natoms_real); }
    HIP_TRY(hipSetDevice(nb->device));
    if (dev_reserve(nb->d_a, nb->natoms) || dev_reserve(nb->d_cell, natoms_real) ||
        dev_reserve(nb->d_x, (size_t)natoms_real * 3) || dev_reserve(nb->d_fa, (size_t)natoms_real * 3)) { return 1; }
    if ((size_t)natoms_real * 3 > nb->h_fa_alloc)
    {
        if (nb->h_fa) { HIP_TRY(hipHostFree(nb->h_fa)); nb->h_fa = nullptr; }
        nb->h_fa_alloc = (size_t)natoms_real * 3 + (size_t)natoms_real * 3 / 5 + 64;
        HIP_TRY(hipHostMalloc((void **)&nb->h_fa, nb->h_fa_alloc * sizeof(float), hipHostMallocDefault));
    }
    hipStream_t st = nb->stream[0];
    HIP_TRY(hipMemcpyAsync(nb->d_a.p, a, (size_t)nb->natoms * sizeof(int), hipMemcpyHostToDevice, st));
    if (natoms_real > 0) { HIP_TRY(hipMemcpyAsync(nb->d_cell.p, cell, (size_t)natoms_real * sizeof(int), hipMemcpyHostToDevice, st)); }
    HIP_TRY(hipStreamSynchronize(st));
    nb->natoms_real = natoms_real;
    return 0;
}

// ---- pair search on the device ----

int nbnxn_hip_put_on_grid_zones(nbnxn_hip_ptr_t nb, const float box[9], const int pbc_shift_dim[3], int nzones,
                                const nbx_grid_zone_t *zones, int natoms, const float *x, int x_on_device, const int *type,
                                const float *q)
{
    if (check_handle(nb)) { return 1; }
    if (!nb->have_const) { return nbx_fail("put_on_grid before init_const (the number of atom types comes from there)"); }
    if (natoms <= 0 || !box || !x || !zones) { return nbx_fail("put_on_grid: missing atoms, box or zones"); }
    if (nzones < 1 || nzones > NBX_MAX_ZONES) { return nbx_fail("put_on_grid: %d zones (a domain decomposition has 1 to %d)", nzones, NBX_MAX_ZONES); }
    if ((type == nullptr) != (q == nullptr)) { return nbx_fail("put_on_grid: types and charges come together"); }
    // the topology does not change between searches: NULL type / q reuse the arrays of the last call
    if (!type && ((int)nb->d_type_in.n != natoms || (int)nb->d_q_in.n != natoms))
    {
        return nbx_fail("put_on_grid without types / charges, but none are held for %d atoms", natoms);
    }
    {
        int a = 0;
        for (int z = 0; z < nzones; z++)
        {
            if (zones[z].a0 != a || zones[z].n <= 0) { return nbx_fail("zone %d covers atoms [%d,%d): zones must tile [0,%d) in order", z, zones[z].a0, zones[z].a0 + zones[z].n, natoms); }
            a += zones[z].n;
        }
        if (a != natoms) { return nbx_fail("the zones hold %d atoms, the call %d", a, natoms); }
    }
    HIP_TRY(hipSetDevice(nb->device));
    for (int i = 0; type && i < natoms; i++)
    {
        if (type[i] < 0 || type[i] >= nb->ntype - 1) { return nbx_fail("atom %d has type %d outside [0,%d)", i, type[i], nb->ntype - 1); }
    }
    if (!nb->search) { nb->search = nbx_search_create(); }
    nb->grid_on_device = false;
    hipStream_t st = nb->stream[0];
    const float *d_x = x;
    if (type && (dev_reserve(nb->d_type_in, natoms) || dev_reserve(nb->d_q_in, natoms))) { return 1; }
    if (!x_on_device)
    {
        if (dev_reserve(nb->d_x, (size_t)natoms * 3)) { return 1; }
        HIP_TRY(hipMemcpyAsync(nb->d_x.p, x, (size_t)natoms * 3 * sizeof(float), hipMemcpyHostToDevice, st));
        d_x = nb->d_x.p;
    }
    if (type)
    {
        HIP_TRY(hipMemcpyAsync(nb->d_type_in.p, type, (size_t)natoms * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(nb->d_q_in.p, q, (size_t)natoms * sizeof(float), hipMemcpyHostToDevice, st));
    }
    const int all_pbc[3] = {1, 1, 1};
    for (int z = 0; z < nzones; z++)
    {
        if (nbx_search_grid_columns(nb->search, st, z, box, pbc_shift_dim ? pbc_shift_dim : all_pbc, zones[z].corner0, zones[z].corner1,
                                    zones[z].a0, zones[z].n, d_x, zones[z].atom_density))
        {
            return nbx_fail("put_on_grid: %s", nbx_search_error());
        }
    }
    const int npad = nbx_search_natoms_padded(nb->search);
    // what nbnxn_cuda_init_atomdata does for a new grid (nbnxn_cuda_data_mgmt.cu:853-912)
    bool re = false;
    if (dev_reserve(nb->xq, npad) || dev_reserve(nb->type, npad) || reserve_f(nb, (size_t)npad * 3, &re)) { return 1; }
    nb->natoms = npad; nb->natoms_local = nbx_search_natoms_padded_local(nb->search);
    nb->clean_other = false;
    if (re) { HIP_TRY(hipMemsetAsync(nb->f.p, 0, nb->f.alloc * sizeof(float), st)); }
    nb->plist[0].nsci = 0; nb->plist[1].nsci = 0;
    for (int z = 0; z < nzones; z++)
    {
        if (nbx_search_grid_fill(nb->search, st, z, d_x, nb->d_q_in.p, nb->d_type_in.p, nb->ntype, nb->xq.p, nb->type.p))
        {
            return nbx_fail("put_on_grid: %s", nbx_search_error());
        }
    }
    // the atom order for copy_x_to_xq / add_f, without a trip through the host
    if (dev_reserve(nb->d_a, npad) || dev_reserve(nb->d_cell, natoms) || dev_reserve(nb->d_x, (size_t)natoms * 3)
        || dev_reserve(nb->d_fa, (size_t)natoms * 3)) { return 1; }
    if ((size_t)natoms * 3 > nb->h_fa_alloc)
    {
        if (nb->h_fa) { HIP_TRY(hipHostFree(nb->h_fa)); nb->h_fa = nullptr; }
        nb->h_fa_alloc = (size_t)natoms * 3 + (size_t)natoms * 3 / 5 + 64;
        HIP_TRY(hipHostMalloc((void **)&nb->h_fa, nb->h_fa_alloc * sizeof(float), hipHostMallocDefault));
    }
    HIP_TRY(hipMemcpyAsync(nb->d_a.p, nbx_search_atom_order(nb->search), (size_t)npad * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(nb->d_cell.p, nbx_search_cell_index(nb->search), (size_t)natoms * sizeof(int), hipMemcpyDeviceToDevice, st));
    nb->natoms_real = natoms;
    // shift vectors of the box, XYZ2IS order (types/ishift.h:45-55), summed as calc_shifts does (gmxlib/pbc.c:1326-1349)
    float sv[NBX_SHIFTS * 3];
    int   is = 0;
    for (int tz = -1; tz <= 1; tz++)
    {
        for (int ty = -1; ty <= 1; ty++)
        {
            for (int tx = -2; tx <= 2; tx++, is++)
            {
                sv[3 * is]     = tx * box[0] + ty * box[3] + tz * box[6];
                sv[3 * is + 1] = ty * box[4] + tz * box[7];
                sv[3 * is + 2] = tz * box[8];
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(nb->shift_vec, sv, sizeof(sv), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    nb->shift_vec_uploaded = true;
    nb->grid_on_device     = true;
    return 0;
}

int nbnxn_hip_put_on_grid(nbnxn_hip_ptr_t nb, const float box[9], const int pbc_shift_dim[3], int natoms, const float *x,
                          int x_on_device, const int *type, const float *q, float atom_density)
{
    if (!box) { return nbx_fail("put_on_grid: missing box"); }
    nbx_grid_zone_t z;
    for (int d = 0; d < 3; d++) { z.corner0[d] = 0; z.corner1[d] = box[4 * d]; }
    z.a0 = 0; z.n = natoms; z.atom_density = atom_density;
    return nbnxn_hip_put_on_grid_zones(nb, box, pbc_shift_dim, 1, &z, natoms, x, x_on_device, type, q);
}

int nbnxn_hip_make_pairlist(nbnxn_hip_ptr_t nb, int iloc, float rlist, const int *excl_index, const int *excl_a, int min_ci_balanced)
{
    return nbnxn_hip_make_pairlist_zones(nb, iloc, 0, nullptr, rlist, excl_index, excl_a, min_ci_balanced);
}

int nbnxn_hip_make_pairlist_zones(nbnxn_hip_ptr_t nb, int iloc, int nzp, const int *zone_pairs, float rlist, const int *excl_index,
                                  const int *excl_a, int min_ci_balanced)
{
    if (check_handle(nb)) { return 1; }
    if (nzp < 0 || (nzp > 0 && !zone_pairs)) { return nbx_fail("make_pairlist: %d zone pairs without a table", nzp); }
    if (check_loc(iloc)) { return 1; }
    if (iloc == NBX_NONLOCAL && nb->search && nbx_search_nzones(nb->search) < 2) { return nbx_fail("a non-local list needs halo zones on the grid (nbnxn_hip_put_on_grid_zones)"); }
    if (!nb->search || !nb->grid_on_device || nbx_search_natoms_padded(nb->search) != nb->natoms) { return nbx_fail("make_pairlist before put_on_grid (or after a new init_atomdata)"); }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = nb->stream[0];
    const int  *d_ei = nullptr, *d_ea = nullptr;
    if ((excl_index == nullptr) != (excl_a == nullptr)) { return nbx_fail("make_pairlist: exclusion index and atoms come together"); }
    if (!excl_index && nb->excl_held && (int)nb->d_excl_index.n == nb->natoms_real + 1)
    {
        // the exclusions of the last call that passed them (same topology)
        d_ei = nb->d_excl_index.p; d_ea = nb->d_excl_a.p;
    }
    if (excl_index && excl_a)
    {
        const int n = nb->natoms_real, ne = excl_index[n];
        if (excl_index[0] != 0 || ne < 0) { return nbx_fail("exclusion index does not start at 0 / ends at %d", ne); }
        if (dev_reserve(nb->d_excl_index, (size_t)n + 1) || dev_reserve(nb->d_excl_a, (size_t)ne + 1)) { return 1; }
        HIP_TRY(hipMemcpyAsync(nb->d_excl_index.p, excl_index, ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        if (ne > 0) { HIP_TRY(hipMemcpyAsync(nb->d_excl_a.p, excl_a, (size_t)ne * sizeof(int), hipMemcpyHostToDevice, st)); }
        d_ei = nb->d_excl_index.p; d_ea = nb->d_excl_a.p;
        nb->excl_held = true;
    }
    nbx_search_list_t L;
    if (nbx_search_make_pairlist(nb->search, st, iloc, nzp, zone_pairs, rlist, min_ci_balanced, d_ei, d_ea, nb->xq.p, &L))
    {
        return nbx_fail("make_pairlist: %s", nbx_search_error());
    }
    // what nbnxn_cuda_init_pairlist does (nbnxn_cuda_data_mgmt.cu:759-800), device to device
    PList &pl = nb->plist[iloc];
    if (dev_reserve(pl.sci, L.nsci) || dev_reserve(pl.cj4, L.ncj4) || dev_reserve(pl.excl, L.nexcl)) { return 1; }
    if (L.nsci > 0)
    {
        HIP_TRY(hipMemcpyAsync(pl.sci.p, L.sci, (size_t)L.nsci * sizeof(nbx_sci_t), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(pl.cj4.p, L.cj4, (size_t)L.ncj4 * sizeof(nbx_cj4_t), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(pl.excl.p, L.excl, (size_t)L.nexcl * sizeof(nbx_excl_t), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    pl.nsci = L.nsci; pl.ncj4 = L.ncj4; pl.nexcl = L.nexcl; pl.nci_tot = L.nci_tot; pl.rlist = rlist;
    pl.do_prune = true; pl.have_outer = false;
    return 0;
}

int nbnxn_hip_set_progressive_balance(nbnxn_hip_ptr_t nb, int on)
{
    if (check_handle(nb)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    if (!nb->search) { nb->search = nbx_search_create(); }
    nbx_search_set_progressive_balance(nb->search, on != 0);
    return 0;
}

int nbnxn_hip_forget_exclusions(nbnxn_hip_ptr_t nb)
{
    if (check_handle(nb)) { return 1; }
    nb->excl_held = false;
    return 0;
}

int nbnxn_hip_download_grid(nbnxn_hip_ptr_t nb, int *dims, int *a, int *cell, float *bb, float *bbcz, int *nsubc, float *xq, int *type)
{
    if (check_handle(nb)) { return 1; }
    if (!nb->search || !nb->grid_on_device || nbx_search_natoms_padded(nb->search) != nb->natoms) { return nbx_fail("download_grid before put_on_grid"); }
    HIP_TRY(hipSetDevice(nb->device));
    if (dims)
    {
        if (nbx_search_grid_info(nb->search, 0, &dims[0], &dims[1], &dims[2])) { return nbx_fail("download_grid: %s", nbx_search_error()); }
        dims[2] = nb->natoms / 64;   // super-cells of all zones
        dims[3] = nb->natoms;
    }
    if (nbx_search_download_grid(nb->search, nb->stream[0], a, cell, bb, bbcz, nsubc)) { return nbx_fail("download_grid: %s", nbx_search_error()); }
    if (xq) { HIP_TRY(hipMemcpy(xq, nb->xq.p, (size_t)nb->natoms * sizeof(float4), hipMemcpyDeviceToHost)); }
    if (type) { HIP_TRY(hipMemcpy(type, nb->type.p, (size_t)nb->natoms * sizeof(int), hipMemcpyDeviceToHost)); }
    return 0;
}

int nbnxn_hip_download_pairlist(nbnxn_hip_ptr_t nb, int iloc, int *counts, nbx_sci_t *sci, nbx_cj4_t *cj4, nbx_excl_t *excl)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    const PList &pl = nb->plist[iloc];
    if (counts) { counts[0] = pl.nsci; counts[1] = pl.ncj4; counts[2] = pl.nexcl; counts[3] = pl.nci_tot; }
    HIP_TRY(hipStreamSynchronize(nb->stream[iloc]));
    if (sci && pl.nsci > 0) { HIP_TRY(hipMemcpy(sci, pl.sci.p, (size_t)pl.nsci * sizeof(nbx_sci_t), hipMemcpyDeviceToHost)); }
    if (cj4 && pl.ncj4 > 0) { HIP_TRY(hipMemcpy(cj4, pl.cj4.p, (size_t)pl.ncj4 * sizeof(nbx_cj4_t), hipMemcpyDeviceToHost)); }
    if (excl && pl.nexcl > 0) { HIP_TRY(hipMemcpy(excl, pl.excl.p, (size_t)pl.nexcl * sizeof(nbx_excl_t), hipMemcpyDeviceToHost)); }
    return 0;
}

int nbnxn_hip_upload_xq_flat(nbnxn_hip_ptr_t nb, const float *xq, int iloc)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    if (!xq) { return nbx_fail("null xq"); }
    int begin, len;
    loc_range(nb, iloc, &begin, &len);
    HIP_TRY(hipSetDevice(nb->device));
    if (len > 0)
    {
        HIP_TRY(hipMemcpyAsync(nb->xq.p + begin, xq + (size_t)begin * 4, (size_t)len * sizeof(float4), hipMemcpyHostToDevice, nb->stream[iloc]));
    }
    return mark_x_written(nb, iloc);
}

int nbnxn_hip_copy_x_to_xq(nbnxn_hip_ptr_t nb, const float *x, int x_on_device, int iloc)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    if (nb->d_a.n == 0 || (int)nb->d_a.n != nb->natoms) { return nbx_fail("copy_x_to_xq before set_atom_order (or after a new init_atomdata)"); }
    if (!x) { return nbx_fail("null x"); }
    int begin, len;
    loc_range(nb, iloc, &begin, &len);
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t  st = nb->stream[iloc];
    const float *dx = x;
    if (!x_on_device)
    {
        // the whole caller-order array goes up on the LOCAL call; the non-local call reuses it
        if (iloc == NBX_LOCAL || !nb->two_streams)
        {
            HIP_TRY(hipMemcpyAsync(nb->d_x.p, x, (size_t)nb->natoms_real * 3 * sizeof(float), hipMemcpyHostToDevice, st));
            if (nb->two_streams) { HIP_TRY(hipEventRecord(nb->misc_ops_done, st)); }
        }
        else { HIP_TRY(hipStreamWaitEvent(st, nb->misc_ops_done, 0)); }
        dx = nb->d_x.p;
    }
    HIP_TRY(nbx_launch_gather_x(nb->d_a.p, dx, nb->xq.p, begin, len, st));
    return mark_x_written(nb, iloc);
}

int nbnxn_hip_add_f(nbnxn_hip_ptr_t nb, float *f, int f_on_device, int a0, int n, int aloc)
{
    if (check_handle(nb) || check_loc(aloc)) { return 1; }
    if (nb->d_cell.n == 0 || (int)nb->d_a.n != nb->natoms) { return nbx_fail("add_f before set_atom_order (or after a new init_atomdata)"); }
    if (!f || a0 < 0 || n < 0 || a0 + n > nb->natoms_real) { return nbx_fail("add_f: atom range [%d,%d) outside [0,%d)", a0, a0 + n, nb->natoms_real); }
    if (n == 0) { return 0; }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = nb->stream[aloc];
    if (f_on_device)
    {
        HIP_TRY(nbx_launch_scatter_f(nb->d_cell.p, nb->f.p, f, a0, n, true, st));
        return 0;
    }
    // host f: permute on the device, bring the dense block down, add on the host (blocking, like
    // the reference's host loop)
    HIP_TRY(nbx_launch_scatter_f(nb->d_cell.p, nb->f.p, nb->d_fa.p, a0, n, false, st));
    HIP_TRY(hipMemcpyAsync(nb->h_fa, nb->d_fa.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float *o = f + (size_t)a0 * 3;
    for (size_t i = 0; i < (size_t)n * 3; i++) { o[i] += nb->h_fa[i]; }
    return 0;
}

// ---- DD halo moves on the device ----

namespace {
int check_halo(nbnxn_hip *nb, const int *d_slots, int n, const void *d_buf, int iloc)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    if (n < 0 || n > nb->natoms) { return nbx_fail("halo of %d atoms on a grid of %d slots", n, nb->natoms); }
    if (n > 0 && (!d_slots || !d_buf)) { return nbx_fail("null halo slot list or buffer"); }
    if (!nb->xq.p || !nb->f.p) { return nbx_fail("halo move before the atom data exists"); }
    return 0;
}
} // namespace

int nbnxn_hip_halo_pack_x(nbnxn_hip_ptr_t nb, const int *d_slots, int n, const float *shift, float *d_buf, int iloc)
{
    if (check_halo(nb, d_slots, n, d_buf, iloc)) { return 1; }
    const float zero[3] = {0, 0, 0};
    HIP_TRY(hipSetDevice(nb->device));
    if (wait_x_written(nb, nb->stream[nb->two_streams ? iloc : 0])) { return 1; }
    HIP_TRY(nbx_launch_halo_pack_x(d_slots, n, nb->xq.p, shift ? shift : zero, reinterpret_cast<float4 *>(d_buf), nb->stream[nb->two_streams ? iloc : 0]));
    return 0;
}

int nbnxn_hip_halo_unpack_x(nbnxn_hip_ptr_t nb, const int *d_slots, int n, const float *d_buf, int iloc)
{
    if (check_halo(nb, d_slots, n, d_buf, iloc)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    HIP_TRY(nbx_launch_halo_unpack_x(d_slots, n, reinterpret_cast<const float4 *>(d_buf), nb->xq.p, nb->stream[nb->two_streams ? iloc : 0]));
    return 0;
}

int nbnxn_hip_halo_pack_f(nbnxn_hip_ptr_t nb, const int *d_slots, int n, float *d_buf, int iloc)
{
    if (check_halo(nb, d_slots, n, d_buf, iloc)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    HIP_TRY(nbx_launch_halo_pack_f(d_slots, n, nb->f.p, d_buf, nb->stream[nb->two_streams ? iloc : 0]));
    return 0;
}

int nbnxn_hip_halo_unpack_f(nbnxn_hip_ptr_t nb, const int *d_slots, int n, const float *d_buf, int shift_index, int iloc)
{
    if (check_halo(nb, d_slots, n, d_buf, iloc)) { return 1; }
    if (shift_index < -1 || shift_index >= 45) { return nbx_fail("shift index %d outside [-1, 45)", shift_index); }
    HIP_TRY(hipSetDevice(nb->device));
    // the entry in accumulation slot 0 of the output in use, {e_lj, e_el, fshift[45][3]}; the kernel's waves spread over the slots
    float *entry = shift_index >= 0 ? nb->out + 2 + 3 * shift_index : nullptr;
    HIP_TRY(nbx_launch_halo_unpack_f(d_slots, n, d_buf, nb->f.p, entry, nb->stream[nb->two_streams ? iloc : 0]));
    return 0;
}

int nbnxn_hip_reserved_cus(nbnxn_hip_ptr_t nb) { return nb ? nb->reserved_cus : 0; }

int nbnxn_hip_stream_wait(nbnxn_hip_ptr_t nb, int iloc_waits, int iloc_done)
{
    if (check_handle(nb) || check_loc(iloc_waits) || check_loc(iloc_done)) { return 1; }
    if (!nb->two_streams || iloc_waits == iloc_done) { return 0; }   // one stream: already in order
    HIP_TRY(hipSetDevice(nb->device));
    HIP_TRY(hipEventRecord(nb->stream_mark[iloc_done], nb->stream[iloc_done]));
    HIP_TRY(hipStreamWaitEvent(nb->stream[iloc_waits], nb->stream_mark[iloc_done], 0));
    return 0;
}

int nbnxn_hip_launch_kernel_pruneonly(nbnxn_hip_ptr_t nb, int iloc, float rlist_inner, int num_parts)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    PList &pl = nb->plist[iloc];
    if (pl.nsci == 0) { return 0; }
    if (!nb->shift_vec_uploaded) { return nbx_fail("pruneonly before upload_shiftvec"); }
    if (!(rlist_inner > 0) || rlist_inner > pl.rlist) { return nbx_fail("inner list radius %g must be in (0, rlist = %g]", rlist_inner, pl.rlist); }
    if (num_parts < 1 || num_parts > pl.nsci) { return nbx_fail("cannot prune in %d parts a list of %d entries", num_parts, pl.nsci); }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = nb->stream[iloc];
    if (!pl.have_outer)
    {
        // first pass over a new list: the outer masks (what the first force kernel would have left, had it
        // pruned) are set aside, the list itself gets the inner ones; all of it, whatever num_parts says
        if (dev_reserve(pl.imask_outer, pl.ncj4)) { return 1; }
        HIP_TRY(nbx_launch_prune(nb->xq.p, nb->shift_vec, pl.sci.p, pl.cj4.p, pl.imask_outer.p, 0, pl.nsci, true, pl.rlist, rlist_inner, st));
        pl.have_outer     = true;
        pl.do_prune       = false;
        pl.rolling_part   = 0;
        pl.rolling_nparts = num_parts;
        return 0;
    }
    if (pl.do_prune) { return nbx_fail("the list was replaced without its outer masks being reset"); }
    if (num_parts != pl.rolling_nparts) { pl.rolling_part = 0; pl.rolling_nparts = num_parts; }
    const int part  = pl.rolling_part;
    const int begin = (int)((long long)pl.nsci * part / num_parts), end = (int)((long long)pl.nsci * (part + 1) / num_parts);
    HIP_TRY(nbx_launch_prune(nb->xq.p, nb->shift_vec, pl.sci.p, pl.cj4.p, pl.imask_outer.p, begin, end - begin, false, pl.rlist, rlist_inner, st));
    pl.rolling_part = (part + 1) % num_parts;
    return 0;
}

int nbnxn_hip_set_prune(nbnxn_hip_ptr_t nb, int iloc, int do_prune)
{
    if (check_handle(nb) || check_loc(iloc)) { return 1; }
    nb->plist[iloc].do_prune = do_prune != 0;
    return 0;
}

int nbnxn_hip_set_copy_kernels(nbnxn_hip_ptr_t nb, int on)
{
    if (check_handle(nb)) { return 1; }
    nb->copy_kernels = on != 0;
    return 0;
}

int nbnxn_hip_set_clear_in_kernel(nbnxn_hip_ptr_t nb, int on)
{
    if (check_handle(nb)) { return 1; }
    nb->clear_in_kernel = on != 0;
    nb->clean_other     = false;
    return 0;
}

int nbnxn_hip_set_timing(nbnxn_hip_ptr_t nb, int on)
{
    if (check_handle(nb)) { return 1; }
    nb->do_time = on != 0;
    for (int l = 0; l < 2; l++) { nb->k_timed[l] = nb->h2d_timed[l] = nb->d2h_timed[l] = false; }
    return 0;
}

int nbnxn_hip_set_kernel_variant(nbnxn_hip_ptr_t nb, int variant)
{
    if (check_handle(nb)) { return 1; }
    if (variant != 0) { return nbx_fail("kernel variant %d does not exist (this build has variant 0 only)", variant); }
    return 0;
}

} // extern "C"

// ---- domain decomposition over RCCL ----
//
// The halo exchange of one decomposed dimension, issued from this library on its own streams: what
// dd_move_x / dd_move_f do with MPI_Sendrecv on host buffers (mdlib/domdec.c:651-875,
// dd_sendrecv_rvec domdec_network.c:114) becomes pack kernel -> one grouped ncclSend/ncclRecv pair
// between devices -> unpack kernel, all on the non-local stream, with no host copy and no
// stream hand-over to another runtime.  RCCL is bound at run time (dlopen), so hosts that run one
// GPU need no RCCL; a process that already holds an RCCL (PyTorch brings its own) shares that copy.

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId)    GetUniqueId    = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank   = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart)     GroupStart     = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
    decltype(&ncclSend)           Send           = nullptr;
    decltype(&ncclRecv)           Recv           = nullptr;
    decltype(&ncclAllReduce)      AllReduce      = nullptr;
    decltype(&ncclCommCount)      CommCount      = nullptr;
    bool                          bound          = false;
};
RcclApi g_rccl;

int dd_bind()
{
    if (g_rccl.bound) { return 0; }
    // the copy the process already has, if any; then the system's
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) { h = dlopen("librccl.so.1", RTLD_NOW); }
    if (!h) { h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW); }
    if (!h) { return nbx_fail("RCCL not found: %s", dlerror()); }
#define NBX_BIND(name)                                                                  \
    g_rccl.name = reinterpret_cast<decltype(g_rccl.name)>(dlsym(h, "nccl" #name));      \
    if (!g_rccl.name) { return nbx_fail("RCCL lacks nccl" #name); }
    NBX_BIND(GetUniqueId) NBX_BIND(CommInitRank) NBX_BIND(CommDestroy) NBX_BIND(GetErrorString)
    NBX_BIND(GroupStart) NBX_BIND(GroupEnd) NBX_BIND(Send) NBX_BIND(Recv) NBX_BIND(AllReduce) NBX_BIND(CommCount)
#undef NBX_BIND
    g_rccl.bound = true;
    return 0;
}

#define RCCL_TRY(call)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) { return nbx_fail("RCCL: %s failed: %s", #call, g_rccl.GetErrorString(r_)); } \
    } while (0)

int check_dd(nbnxn_hip *nb)
{
    if (check_handle(nb)) { return 1; }
    if (!nb->dd_comm) { return nbx_fail("domain decomposition call before nbnxn_hip_dd_init"); }
    return 0;
}

hipStream_t dd_stream(nbnxn_hip *nb, int iloc) { return nb->stream[nb->two_streams ? iloc : 0]; }

} // namespace

int nbnxn_hip_dd_get_unique_id(char *id128)
{
    if (!id128) { return nbx_fail("null id buffer"); }
    if (dd_bind()) { return 1; }
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == NBX_DD_UNIQUE_ID_BYTES, "RCCL unique id size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int nbnxn_hip_dd_init(nbnxn_hip_ptr_t nb, const char *id128, int rank, int nranks)
{
    if (check_handle(nb)) { return 1; }
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) { return nbx_fail("dd_init: rank %d of %d", rank, nranks); }
    if (nb->dd_comm) { return nbx_fail("dd_init: this handle already has a communicator"); }
    if (dd_bind()) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    RCCL_TRY(g_rccl.CommInitRank(&nb->dd_comm, nranks, id, rank));
    nb->dd_rank = rank; nb->dd_nranks = nranks;
    return 0;
}

int nbnxn_hip_dd_comm_count(nbnxn_hip_ptr_t nb, int *nranks)
{
    if (check_dd(nb)) { return 1; }
    if (!nranks) { return nbx_fail("null result pointer"); }
    RCCL_TRY(g_rccl.CommCount(nb->dd_comm, nranks));
    return 0;
}

int nbnxn_hip_dd_finish(nbnxn_hip_ptr_t nb)
{
    if (!nb) { return 0; }
    if (nb->dd_comm)
    {
        (void)hipSetDevice(nb->device);
        (void)hipDeviceSynchronize();
        (void)g_rccl.CommDestroy(nb->dd_comm);
        nb->dd_comm = nullptr;
    }
    dev_free(nb->dd_send_slots); dev_free(nb->dd_recv_slots); dev_free(nb->dd_send_wave_msg); dev_free(nb->dd_msg_shift);
    dev_free(nb->dd_x_out); dev_free(nb->dd_x_in); dev_free(nb->dd_f_out); dev_free(nb->dd_f_in);
    nb->dd_nmsg = 0;
    nb->dd_nphase = 0;
    return 0;
}

int nbnxn_hip_dd_set_messages(nbnxn_hip_ptr_t nb, int nmsg, const nbx_dd_message_t *msg, const int *send_slots, const int *recv_slots)
{
    if (check_dd(nb)) { return 1; }
    if (nmsg < 1 || nmsg > NBX_DD_MAX_MESSAGES || !msg) { return nbx_fail("dd_set_messages: %d messages (1 to %d)", nmsg, NBX_DD_MAX_MESSAGES); }
    if (!nb->xq.p || !nb->f.p) { return nbx_fail("dd_set_messages before the atom data exists"); }
    // padded layouts: every message starts at a multiple of 64 rows, so that one wave of the pack / unpack kernels
    // belongs to one message
    // Everything is validated and laid out in locals first and only copied into the handle once the device arrays
    // hold it: a rejected call leaves the installed messages as they were (the Python binding runs with abort-on-error
    // off and goes on after a failure).
    std::vector<int> h_send, h_recv, h_wave;
    std::vector<float> h_shift((size_t)nmsg * 4);
    std::vector<int> send_off(nmsg), recv_off(nmsg);
    size_t si = 0, ri = 0;
    for (int k = 0; k < nmsg; k++)
    {
        const nbx_dd_message_t &m = msg[k];
        if (m.send_to < 0 || m.send_to >= nb->dd_nranks || m.recv_from < 0 || m.recv_from >= nb->dd_nranks)
        {
            return nbx_fail("dd_set_messages: message %d has neighbours %d / %d outside [0, %d)", k, m.send_to, m.recv_from, nb->dd_nranks);
        }
        if (m.n_send < 0 || m.n_recv < 0 || m.n_send > nb->natoms || m.n_recv > nb->natoms || (m.n_send > 0 && !send_slots) || (m.n_recv > 0 && !recv_slots))
        {
            return nbx_fail("dd_set_messages: message %d moves %d / %d halo atoms on a grid of %d slots", k, m.n_send, m.n_recv, nb->natoms);
        }
        if (m.shift_index < -1 || m.shift_index >= NBX_SHIFTS) { return nbx_fail("dd_set_messages: shift index %d outside [-1, %d)", m.shift_index, NBX_SHIFTS); }
        send_off[k] = (int)h_send.size(); recv_off[k] = (int)h_recv.size();
        for (int i = 0; i < m.n_send; i++, si++)
        {
            if (send_slots[si] < 0 || send_slots[si] >= nb->natoms) { return nbx_fail("dd_set_messages: send slot %d outside the grid", send_slots[si]); }
            h_send.push_back(send_slots[si]);
        }
        for (int i = 0; i < m.n_recv; i++, ri++)
        {
            if (recv_slots[ri] < 0 || recv_slots[ri] >= nb->natoms) { return nbx_fail("dd_set_messages: receive slot %d outside the grid", recv_slots[ri]); }
            h_recv.push_back(recv_slots[ri]);
        }
        while (h_send.size() % 64) { h_send.push_back(-1); }
        while (h_recv.size() % 64) { h_recv.push_back(-1); }
        h_wave.resize(h_send.size() / 64, k);
        int is = m.shift_index;
        memcpy(&h_shift[4 * k], m.shift, 3 * sizeof(float));
        memcpy(&h_shift[4 * k + 3], &is, sizeof(int));
    }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = dd_stream(nb, NBX_NONLOCAL);
    const size_t ns = std::max<size_t>(h_send.size(), 64), nr = std::max<size_t>(h_recv.size(), 64);
    if (dev_reserve(nb->dd_send_slots, ns) || dev_reserve(nb->dd_recv_slots, nr) || dev_reserve(nb->dd_send_wave_msg, ns / 64)
        || dev_reserve(nb->dd_msg_shift, (size_t)NBX_DD_MAX_MESSAGES * 4)
        || dev_reserve(nb->dd_x_out, ns * 4) || dev_reserve(nb->dd_x_in, nr * 4) || dev_reserve(nb->dd_f_out, nr * 3) || dev_reserve(nb->dd_f_in, ns * 3)) { return 1; }
    // the steps in flight still read the old lists
    HIP_TRY(hipStreamSynchronize(st));
    // from here on the device arrays change: the old messages are gone whatever happens, so no exchange may run on a
    // half-written set if an upload fails
    nb->dd_nmsg = 0; nb->dd_nphase = 0;
    if (!h_send.empty()) { HIP_TRY(hipMemcpyAsync(nb->dd_send_slots.p, h_send.data(), h_send.size() * sizeof(int), hipMemcpyHostToDevice, st)); }
    if (!h_recv.empty()) { HIP_TRY(hipMemcpyAsync(nb->dd_recv_slots.p, h_recv.data(), h_recv.size() * sizeof(int), hipMemcpyHostToDevice, st)); }
    if (!h_wave.empty()) { HIP_TRY(hipMemcpyAsync(nb->dd_send_wave_msg.p, h_wave.data(), h_wave.size() * sizeof(int), hipMemcpyHostToDevice, st)); }
    HIP_TRY(hipMemcpyAsync(nb->dd_msg_shift.p, h_shift.data(), h_shift.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < nmsg; k++) { nb->dd_send_off[k] = send_off[k]; nb->dd_recv_off[k] = recv_off[k]; nb->dd_msg[k] = msg[k]; }
    nb->dd_send_rows = (int)h_send.size(); nb->dd_recv_rows = (int)h_recv.size();
    nb->dd_nmsg = nmsg;
    nb->dd_nphase = 1; nb->dd_phase_end[0] = nmsg;      // all at once until nbnxn_hip_dd_set_phases says otherwise
    return 0;
}

int nbnxn_hip_dd_set_phases(nbnxn_hip_ptr_t nb, int nphase, const int *phase_end)
{
    if (check_dd(nb)) { return 1; }
    if (nb->dd_nmsg == 0) { return nbx_fail("dd_set_phases before dd_set_messages"); }
    if (nphase < 1 || nphase > nb->dd_nmsg || !phase_end) { return nbx_fail("dd_set_phases: %d phases for %d messages", nphase, nb->dd_nmsg); }
    for (int p = 0; p < nphase; p++)
    {
        if (phase_end[p] <= (p ? phase_end[p - 1] : 0)) { return nbx_fail("dd_set_phases: phase %d ends at message %d, not behind the phase before it", p, phase_end[p]); }
    }
    if (phase_end[nphase - 1] != nb->dd_nmsg) { return nbx_fail("dd_set_phases: the last phase ends at message %d of %d", phase_end[nphase - 1], nb->dd_nmsg); }
    HIP_TRY(hipSetDevice(nb->device));
    HIP_TRY(hipStreamSynchronize(dd_stream(nb, NBX_NONLOCAL)));     // no step in flight reads the phases, but keep the order of set-up calls simple
    nb->dd_nphase = nphase;
    for (int p = 0; p < nphase; p++) { nb->dd_phase_end[p] = phase_end[p]; }
    return 0;
}

namespace {
// rows [r0, r1) of the padded send / receive layouts that the messages [k0, k1) occupy
struct PhaseRows { int k0, k1, s0, s1, r0, r1; };
PhaseRows phase_rows(const nbnxn_hip *nb, int p)
{
    PhaseRows R;
    R.k0 = p ? nb->dd_phase_end[p - 1] : 0;
    R.k1 = nb->dd_phase_end[p];
    R.s0 = nb->dd_send_off[R.k0]; R.s1 = R.k1 < nb->dd_nmsg ? nb->dd_send_off[R.k1] : nb->dd_send_rows;
    R.r0 = nb->dd_recv_off[R.k0]; R.r1 = R.k1 < nb->dd_nmsg ? nb->dd_recv_off[R.k1] : nb->dd_recv_rows;
    return R;
}
}

int nbnxn_hip_dd_set_pulse(nbnxn_hip_ptr_t nb, int send_to, int recv_from, const int *send_slots, int n_send,
                           const int *recv_slots, int n_recv, const float *shift, int shift_index)
{
    // one pulse along one dimension = one message
    nbx_dd_message_t m;
    m.send_to = send_to; m.recv_from = recv_from; m.n_send = n_send; m.n_recv = n_recv; m.shift_index = shift_index;
    for (int d = 0; d < 3; d++) { m.shift[d] = shift ? shift[d] : 0.0f; }
    return nbnxn_hip_dd_set_messages(nb, 1, &m, send_slots, recv_slots);
}

int nbnxn_hip_dd_move_x(nbnxn_hip_ptr_t nb)
{
    if (check_dd(nb)) { return 1; }
    if (nb->dd_nmsg == 0) { return nbx_fail("dd_move_x before dd_set_pulse / dd_set_messages"); }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = dd_stream(nb, NBX_NONLOCAL);
    // the home coordinates the pack kernel reads may still be on their way on the local stream (x gather, H2D)
    if (wait_x_written(nb, st)) { return 1; }
    // phase by phase (one phase unless nbnxn_hip_dd_set_phases split the messages): a later phase may send what an
    // earlier one brought (dd_move_x, domdec.c:667-758: the pulses of the later dimensions forward the earlier zones)
    for (int p = 0; p < nb->dd_nphase; p++)
    {
        const PhaseRows R = phase_rows(nb, p);
        HIP_TRY(nbx_launch_halo_pack_x_msgs(nb->dd_send_slots.p + R.s0, R.s1 - R.s0, nb->xq.p, nb->dd_send_wave_msg.p + R.s0 / 64,
                                            reinterpret_cast<const float4 *>(nb->dd_msg_shift.p),
                                            reinterpret_cast<float4 *>(nb->dd_x_out.p) + R.s0, st));
        RCCL_TRY(g_rccl.GroupStart());
        for (int k = R.k0; k < R.k1; k++)
        {
            const nbx_dd_message_t &m = nb->dd_msg[k];
            if (m.n_send > 0) { RCCL_TRY(g_rccl.Send(nb->dd_x_out.p + (size_t)nb->dd_send_off[k] * 4, (size_t)m.n_send * 4, ncclFloat, m.send_to, nb->dd_comm, st)); }
            if (m.n_recv > 0) { RCCL_TRY(g_rccl.Recv(nb->dd_x_in.p + (size_t)nb->dd_recv_off[k] * 4, (size_t)m.n_recv * 4, ncclFloat, m.recv_from, nb->dd_comm, st)); }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(nbx_launch_halo_unpack_x(nb->dd_recv_slots.p + R.r0, R.r1 - R.r0, reinterpret_cast<const float4 *>(nb->dd_x_in.p) + R.r0, nb->xq.p, st));
    }
    return 0;
}

int nbnxn_hip_dd_move_f(nbnxn_hip_ptr_t nb)
{
    if (check_dd(nb)) { return 1; }
    if (nb->dd_nmsg == 0) { return nbx_fail("dd_move_f before dd_set_pulse / dd_set_messages"); }
    HIP_TRY(hipSetDevice(nb->device));
    hipStream_t st = dd_stream(nb, NBX_NONLOCAL);
    // the reverse path: forces on imported atoms go back to where the coordinates came from; the phases in reverse
    // order, so that what a later phase returns to a forwarded atom travels on with the earlier one (dd_move_f,
    // domdec.c:784-870)
    for (int p = nb->dd_nphase - 1; p >= 0; p--)
    {
        const PhaseRows R = phase_rows(nb, p);
        HIP_TRY(nbx_launch_halo_pack_f(nb->dd_recv_slots.p + R.r0, R.r1 - R.r0, nb->f.p, nb->dd_f_out.p + (size_t)R.r0 * 3, st));
        RCCL_TRY(g_rccl.GroupStart());
        for (int k = R.k0; k < R.k1; k++)
        {
            const nbx_dd_message_t &m = nb->dd_msg[k];
            if (m.n_recv > 0) { RCCL_TRY(g_rccl.Send(nb->dd_f_out.p + (size_t)nb->dd_recv_off[k] * 3, (size_t)m.n_recv * 3, ncclFloat, m.recv_from, nb->dd_comm, st)); }
            if (m.n_send > 0) { RCCL_TRY(g_rccl.Recv(nb->dd_f_in.p + (size_t)nb->dd_send_off[k] * 3, (size_t)m.n_send * 3, ncclFloat, m.send_to, nb->dd_comm, st)); }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        // into the accumulation slots of the output in use for the messages that crossed a periodic boundary
        HIP_TRY(nbx_launch_halo_unpack_f_msgs(nb->dd_send_slots.p + R.s0, R.s1 - R.s0, nb->dd_f_in.p + (size_t)R.s0 * 3, nb->f.p,
                                              nb->dd_send_wave_msg.p + R.s0 / 64, reinterpret_cast<const float4 *>(nb->dd_msg_shift.p), nb->out, st));
    }
    return 0;
}

int nbnxn_hip_dd_sum_outputs(nbnxn_hip_ptr_t nb, int iloc)
{
    if (check_dd(nb) || check_loc(iloc)) { return 1; }
    HIP_TRY(hipSetDevice(nb->device));
    RCCL_TRY(g_rccl.AllReduce(nb->out, nb->out, (size_t)NBX_OUT_SLOTS * NBX_OUT_STRIDE, ncclFloat, ncclSum, nb->dd_comm, dd_stream(nb, iloc)));
    return 0;
}

int nbnxn_hip_dd_step(nbnxn_hip_ptr_t nb, int flags, int sum_last_outputs, int exchange_beside_local, void *ev_before_local,
                      void *ev_after_local)
{
    // One decomposed step with resident coordinates, issued from ONE call: the host's time to issue a step is what
    // limits a short step, and every call through a foreign-function layer adds to it.
    if (check_dd(nb)) { return 1; }
    if (!nb->two_streams) { return nbx_fail("dd_step needs a handle with a local and a non-local stream"); }
    if (nb->dd_nmsg == 0) { return nbx_fail("dd_step before dd_set_pulse / dd_set_messages"); }
    HIP_TRY(hipSetDevice(nb->device));
    // NBNXN_DD_STEP_PROFILE=1: host time of every part, printed every 1000 steps (a measurement aid)
    static const bool prof = getenv("NBNXN_DD_STEP_PROFILE") != nullptr;
    static double     acc[7];
    static long       nstep = 0;
    auto              now   = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double            t     = prof ? now() : 0;
    auto              lap   = [&](int k) { if (prof) { const double u = now(); acc[k] += u - t; t = u; } };
    if (!exchange_beside_local && nbnxn_hip_dd_move_x(nb)) { return 1; }     // behind the last step's returned forces
    if (!exchange_beside_local) { lap(3); }
    if (sum_last_outputs && nbnxn_hip_dd_sum_outputs(nb, NBX_LOCAL)) { return 1; }
    lap(0);
    if (nbnxn_hip_clear_outputs(nb, flags)) { return 1; }
    lap(1);
    if (ev_before_local) { HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev_before_local), nb->stream[NBX_LOCAL])); }
    if (launch_impl(nb, nullptr, flags, NBX_LOCAL)) { return 1; }
    if (ev_after_local) { HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev_after_local), nb->stream[NBX_LOCAL])); }
    lap(2);
    if (exchange_beside_local && nbnxn_hip_dd_move_x(nb)) { return 1; }     // beside the local kernel
    if (exchange_beside_local) { lap(3); }
    if (launch_impl(nb, nullptr, flags, NBX_NONLOCAL)) { return 1; }
    lap(4);
    if (nbnxn_hip_dd_move_f(nb)) { return 1; }
    lap(5);
    const int rc = nbnxn_hip_stream_wait(nb, NBX_LOCAL, NBX_NONLOCAL);
    lap(6);
    if (prof && ++nstep % 1000 == 0)
    {
        fprintf(stderr, "dd_step host us/step: sum_outputs %.1f clear %.1f local launch %.1f move_x %.1f non-local launch %.1f move_f %.1f stream_wait %.1f\n",
                acc[0] / 1000, acc[1] / 1000, acc[2] / 1000, acc[3] / 1000, acc[4] / 1000, acc[5] / 1000, acc[6] / 1000);
        for (double &v : acc) { v = 0; }
    }
    return rc;
}
