// gfx950 (MI355X) cluster-pair nonbonded kernels.
//
// One wave64 per i-super-cluster entry (nbnxn_sci_t); the 8x8 atom pairs of an
// (i-cluster, j-cluster) pair map onto the 64 lanes: lane = 8*jatom + iatom, so lanes
// 0-31 / 32-63 read the exclusion words of list half 0 / 1 (types/nbnxn_pairlist.h:109-123).
// What is computed per pair follows the reference kernels (nbnxn_cuda_kernel.cuh:270-506,
// nbnxn_kernel_ref_inner.h:117-372); how it is mapped to the hardware is gfx950-specific:
//   * i-atom xyz, epsfac*q and type rows are staged once per entry in LDS;
//   * j-atom xq/type are read straight from HBM/L2, 8 distinct 16-B rows per wave load;
//   * j-forces: 3-step DPP reduction over the 8 i-lanes, then ONE 24-lane fp32 atomic
//     instruction covering the cluster's 96 contiguous bytes of f;
//   * i-forces: 24 register accumulators per lane, reduce-scattered over the 8 j-lanes at
//     the end (21 cross-lane moves), transposed through LDS and written with three
//     256-byte contiguous atomic instructions;
//   * energies / shift forces: wave-reduced, then added into one of NBX_OUT_SLOTS
//     accumulation slots (slot = entry index mod slots), so no long same-address fp32 atomic
//     chain builds up; the host folds the slots in double precision after the copy-back.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "nbnxn_kernel.h"

#ifdef NBX_TIMELINE
// debug builds only (make timeline): per list entry {start, first pair, last pair done, end} in 100 MHz ticks and the
// hardware id of the wave with the entry's group count
#define NBX_TL_WORDS 6
#if NBX_UNIT_EL == 0
#define NBX_TL_SYM nbx_timeline_rf
#define NBX_TL_FN nbnxn_hip_debug_timeline_rf
#else
#define NBX_TL_SYM nbx_timeline
#define NBX_TL_FN nbnxn_hip_debug_timeline
#endif
__device__ unsigned long long NBX_TL_SYM[NBX_TL_WORDS * 65536];
extern "C" int NBX_TL_FN(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(NBX_TL_SYM), sizeof(unsigned long long) * NBX_TL_WORDS * (size_t)(n < 65536 ? n : 65536)) == hipSuccess ? 0 : 1;
}
#endif

// raw buffer loads (the builtin of this compiler release loads a single dword whatever its width; the intrinsic
// declared by its LLVM name does what it says)
typedef int   nbx_i32x4 __attribute__((ext_vector_type(4)));
typedef float nbx_f32x4 __attribute__((ext_vector_type(4)));
__device__ nbx_f32x4 nbx_buffer_load_f32x4(nbx_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ int       nbx_buffer_load_i32(nbx_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
// v_writelane_b32: lane `lane` of the result is val, the others keep old
__device__ int       nbx_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// a 16-byte LDS read the compiler neither moves nor splits (volatile, in the LDS address space)
__device__ __forceinline__ nbx_f32x4 lds_load_f32x4_volatile(const float4 *p)
{
    typedef const volatile nbx_f32x4 __attribute__((address_space(3))) *lds_ptr_t;
    return *(lds_ptr_t)(p);
}

#ifndef NBX_KERNEL_WAVES
#define NBX_KERNEL_WAVES 4
#endif
#ifndef NBX_KERNEL_WAVES_F
#define NBX_KERNEL_WAVES_F 4
#endif
#ifndef NBX_ZERO_AT_END
#define NBX_ZERO_AT_END 1
#endif
#ifndef NBX_HDR2
#define NBX_HDR2 1     // the cj4 group header is requested two groups ahead and decoded one group ahead
#endif
#ifndef NBX_PRO2
#define NBX_PRO2 1     // the type-pair table is requested before anything else in the prologue
#endif
#ifndef NBX_FJ_CAP
#define NBX_FJ_CAP 32
#endif
#ifndef NBX_CPAIR
#define NBX_CPAIR 2      // the LJ entry of a pair read from the LDS table when the pair is computed: 0 never, 1 always, 2 Ewald flavours
#endif
#ifndef NBX_JRED2
#define NBX_JRED2 1
#endif
#ifndef NBX_XI_LDS
#define NBX_XI_LDS 1     // i-atoms read from LDS per pair instead of held in 32 registers: 0 never, 1 always, 2 Ewald flavours
#endif
#ifndef NBX_PLAIN_GROUPS
#define NBX_PLAIN_GROUPS 1   // groups without exclusion words and diagonal cluster pairs take a pair body without the mask work
#endif
#ifndef NBX_RF_VGPR
#define NBX_RF_VGPR 0        // reaction field: -2 k_rf in a VGPR (the pair's force fma then has no SGPR operand): measured, no gain
#endif
#ifndef NBX_XI_PREFETCH
#define NBX_XI_PREFETCH 2    // i-atoms in LDS: the next pair's row is requested a pair ahead, unconditionally: 0 never, 1 always,
#endif                       // 2 where it was measured to pay (reaction field; Ewald force-only: -2 %; Ewald with energies: +1 %)
#ifndef NBX_JAHEAD3
#define NBX_JAHEAD3 2    // j-atoms requested three j-clusters ahead instead of one: 0 never, 1 always, 2 reaction field force-only
#endif
#ifndef NBX_KERNEL_WAVES_EW
#define NBX_KERNEL_WAVES_EW 5    // waves per SIMD the flavours with their i-atoms in LDS are compiled for (5 needs <= 96 registers)
#endif
#if NBX_JAHEAD3 && !NBX_HDR2
#error "NBX_JAHEAD3 needs the next group's header at the top of the iteration (NBX_HDR2)"
#endif
// (not with the LJ-PME grid column: two registers would spill)
constexpr bool j_ahead3(int el, bool ener, int vmod) { return NBX_JAHEAD3 == 1 || (NBX_JAHEAD3 == 2 && el == NBX_EL_RF && !ener && vmod != 3); }
constexpr bool xi_in_lds(int el) { return NBX_XI_LDS == 1 || (NBX_XI_LDS == 2 && el != NBX_EL_RF); }
// Per-pair LJ entries (lj_per_pair): instead of the eight entries of a j-cluster fetched ahead into 16 registers, each
// pair reads its own 8-byte entry when it is computed: 14 registers fewer.  Measured (profiles/r3_r_cpair_ab.txt): Ewald F
// -2.5 %, Ewald V+F -1.2 % (with the i-atom prefetch, which the freed registers make pay there too), RF F +4.6 %: so
// for the Ewald flavours only, and never with the LJ-PME grid column (16-byte entries).
constexpr bool lj_per_pair(int el, int vmod, bool lds_table) { return lds_table && vmod != 3 && (NBX_CPAIR == 1 || (NBX_CPAIR == 2 && el != NBX_EL_RF)); }
constexpr bool xi_prefetch(int el, bool ener, bool per_pair_lj)
{
    return xi_in_lds(el) && (NBX_XI_PREFETCH == 1 || (NBX_XI_PREFETCH == 2 && (el == NBX_EL_RF || !ener || per_pair_lj)));
}
// (the global-table instantiations, more than 16 atom types, keep 4 waves: at 96 registers some of them spill)
constexpr int  kernel_waves(int el, bool ener, bool lds_table)
{
    return xi_in_lds(el) && lds_table ? NBX_KERNEL_WAVES_EW : (ener ? NBX_KERNEL_WAVES : NBX_KERNEL_WAVES_F);
}


namespace {


// The list, coordinate and parameter arrays are passed as separate __restrict__ kernel
// arguments: that gives them noalias, so the force atomics do not "clobber" them in the
// compiler's eyes and the wave-uniform list words (sci, cj4, excl_ind) stay on the scalar
// load path (s_load) inside the loops instead of vector loads + v_readfirstlane with a
// full vmcnt(0) round trip each.
#define NBX_KERNEL_PARAMS                                                                          \
    const float4 *__restrict__ p_xq, float *__restrict__ p_f, const int *__restrict__ p_type,      \
        const float *__restrict__ p_shift_vec, const float2 *__restrict__ p_nbfp,                   \
        const nbx_sci_t *__restrict__ p_sci, nbx_cj4_t *__restrict__ p_cj4,                         \
        const nbx_excl_t *__restrict__ p_excl, float *__restrict__ p_out, const int p_ntype,        \
        float4 *__restrict__ p_zero, const int p_zero_n4, const float *__restrict__ p_c6grid,       \
        const float *__restrict__ p_tab, const nbx_kernel_k_t p_k
#define NBX_KERNEL_ARGS_FROM_PARAMS                                                                \
    const nbx_kernel_view_t a = {p_xq, p_f, p_type, p_shift_vec, p_ntype, p_nbfp, p_sci, p_cj4, p_excl, p_out, p_c6grid, p_tab, p_k};
#define NBX_KERNEL_CALL_ARGS(A)                                                                    \
    (A).xq, (A).f, (A).type, (A).shift_vec, (A).nbfp, (A).sci, (A).cj4, (A).excl, (A).out, (A).ntype, (A).zero,   \
        (A).zero_n4, (A).c6grid, (A).coulomb_tab, (A).k

// device-side view with the same member names as nbx_kernel_args_t
struct nbx_kernel_view_t {
    const float4 *__restrict__ xq;
    float *__restrict__        f;
    const int *__restrict__    type;
    const float *__restrict__  shift_vec;
    int                        ntype;
    const float2 *__restrict__ nbfp;
    const nbx_sci_t *__restrict__  sci;
    nbx_cj4_t *__restrict__        cj4;
    const nbx_excl_t *__restrict__ excl;
    float *__restrict__            out;
    const float *__restrict__      c6grid;
    const float *__restrict__      tab;
    nbx_kernel_k_t                 k;
};

// Ewald real-space correction g(w) = -(erf(z)/z - 2/sqrt(pi) exp(-z^2))/z^2, w = z^2, as
// P6(w)/Q4(w); coefficients derived by tools/fit_pmecorr.py (max abs error 1.6e-7 in
// float on 0 <= w <= 16).
#define NBX_PMEF_P0 -7.522527264e-01f
#define NBX_PMEF_P1 8.071402458e-02f
#define NBX_PMEF_P2 -2.010053827e-02f
#define NBX_PMEF_P3 1.231729042e-03f
#define NBX_PMEF_P4 -6.630095574e-05f
#define NBX_PMEF_P5 1.910048934e-06f
#define NBX_PMEF_P6 -2.368872454e-08f
#define NBX_PMEF_Q1 4.927010142e-01f
#define NBX_PMEF_Q2 1.080706215e-01f
#define NBX_PMEF_Q3 1.314706087e-02f
#define NBX_PMEF_Q4 8.677762527e-04f

__device__ __forceinline__ float pme_corr_f(float w)
{
    float p = NBX_PMEF_P6;
    p = fmaf(p, w, NBX_PMEF_P5);
    p = fmaf(p, w, NBX_PMEF_P4);
    p = fmaf(p, w, NBX_PMEF_P3);
    p = fmaf(p, w, NBX_PMEF_P2);
    p = fmaf(p, w, NBX_PMEF_P1);
    p = fmaf(p, w, NBX_PMEF_P0);
    float q = NBX_PMEF_Q4;
    q = fmaf(q, w, NBX_PMEF_Q3);
    q = fmaf(q, w, NBX_PMEF_Q2);
    q = fmaf(q, w, NBX_PMEF_Q1);
    q = fmaf(q, w, 1.0f);
    return p * __builtin_amdgcn_rcpf(q);
}

// Force and potential corrections over ONE denominator (tools/fit_pmecorr.py jointfit: P6/Q5
// and P5/Q5, float evaluation errors 1.7e-7 / 2.3e-7 absolute on 0 <= w <= 16): the energy
// kernels need both, and the reciprocal is a quarter-rate instruction.
#define NBX_PMEJ_F0 -7.522528659e-01f
#define NBX_PMEJ_F1 7.207905358e-02f
#define NBX_PMEJ_F2 -1.981625525e-02f
#define NBX_PMEJ_F3 1.028096825e-03f
#define NBX_PMEJ_F4 -6.587890910e-05f
#define NBX_PMEJ_F5 1.812734519e-06f
#define NBX_PMEJ_F6 -2.231291589e-08f
#define NBX_PMEJ_V0 1.128379098e+00f
#define NBX_PMEJ_V1 1.927866508e-01f
#define NBX_PMEJ_V2 5.246387815e-02f
#define NBX_PMEJ_V3 3.745746373e-03f
#define NBX_PMEJ_V4 2.629633691e-04f
#define NBX_PMEJ_V5 1.725974995e-06f
#define NBX_PMEJ_Q1 5.041848465e-01f
#define NBX_PMEJ_Q2 1.145604868e-01f
#define NBX_PMEJ_Q3 1.489169305e-02f
#define NBX_PMEJ_Q4 1.120914681e-03f
#define NBX_PMEJ_Q5 3.363388774e-05f

__device__ __forceinline__ void pme_corr_fv(const float w, float &gf, float &gv)
{
    float q = NBX_PMEJ_Q5;
    q = fmaf(q, w, NBX_PMEJ_Q4);
    q = fmaf(q, w, NBX_PMEJ_Q3);
    q = fmaf(q, w, NBX_PMEJ_Q2);
    q = fmaf(q, w, NBX_PMEJ_Q1);
    q = fmaf(q, w, 1.0f);
    float pf = NBX_PMEJ_F6;
    pf = fmaf(pf, w, NBX_PMEJ_F5);
    pf = fmaf(pf, w, NBX_PMEJ_F4);
    pf = fmaf(pf, w, NBX_PMEJ_F3);
    pf = fmaf(pf, w, NBX_PMEJ_F2);
    pf = fmaf(pf, w, NBX_PMEJ_F1);
    pf = fmaf(pf, w, NBX_PMEJ_F0);
    float pv = NBX_PMEJ_V5;
    pv = fmaf(pv, w, NBX_PMEJ_V4);
    pv = fmaf(pv, w, NBX_PMEJ_V3);
    pv = fmaf(pv, w, NBX_PMEJ_V2);
    pv = fmaf(pv, w, NBX_PMEJ_V1);
    pv = fmaf(pv, w, NBX_PMEJ_V0);
    const float rq = __builtin_amdgcn_rcpf(q);
    gf = pf * rq;
    gv = pv * rq;
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    // bound_ctrl = true: all lanes have a source for the permutations used here, and with it the compiler folds
    // the move into one v_add_f32_dpp (without it row_half_mirror costs v_mov 0 + v_mov_dpp + v_add)
    const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, m);
}

// sum over the 8 lanes that share a j-atom (lane bits 0..2)
__device__ __forceinline__ float reduce8(float v)
{
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    return v;
}

constexpr int MAX_LDS_TYPES = NBX_KERNEL_MAX_LDS_TYPES;

// A third per-type-pair value rides with {c6, c12} for LJ-PME: the grid C6.
constexpr bool has_third(int vmod, bool) { return vmod == 3; }
// bytes per type pair in the LDS parameter table: {c6, c12} or {c6, c12, third, -}
constexpr int lds_entry_bytes(int vmod, bool ener) { return has_third(vmod, ener) ? 16 : 8; }

// Energy accumulators of a lane.  The plain cut-off flavours sum the potential-shift terms apart from the pair
// energies: s12 / s6 = sum of c12 / c6 and sq = sum of the charge products over the pairs that are inside the
// cut-off and not excluded; the constants (rep_cpot, disp_cpot, sh_ewald) multiply the wave's sums once.
struct nbx_eacc_t {
    float lj = 0, el = 0, s12 = 0, s6 = 0, sq = 0;
};

// ---------------------------------------------------------------------------------------
// The kernel.  Its shape came out of reading the gfx950 code of two earlier variants (kept in
// the git history: i-atoms re-read from LDS per pair; separate pair bodies for groups with
// and without exclusions, cj4-granular prefetch):
//  * each lane keeps its eight i-atoms (one per i-cluster) and their 24 force accumulators
//    in registers for the whole entry; the cut-off is a select, not a divergent branch;
//  * ONE pair body for groups with and without exclusion words.  An if/else over two bodies
//    merges the 24 i-force accumulators at its end; the register allocator then keeps two
//    copies of them (24 more live registers, three moves per skipped pair, 24 per skipped
//    j-cluster).  The single body pays two instructions per pair for exclusions instead: the
//    pair's exclusion bit is sign-extended to a 0 / ~0 word and ANDed onto the terms that
//    vanish for excluded pairs.  Excluded pairs get the reference's r^2 increment (see
//    pair_interaction): r = 0 occurs for the self pair, whose result the cut-off select
//    discards, and for distinct excluded atoms on one point, whose corrections must stay finite.
//  * the software pipeline works per j-cluster (two rolling j-atom register sets instead of
//    two four-cluster batches);
//  * the eight LJ parameter pairs of a j-cluster are fetched from LDS before its first pair
//    instead of one dependent LDS read per pair.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float and_mask(const float v, const int m)
{
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & m);
}

// constants of the pair body kept in VGPRs: a VALU instruction with an SGPR operand issues at half rate
// (tools/ubench/valu_mix.hip)
struct nbx_vconst_t {
    float r2inc;       // the r^2 increment (see pair_interaction)
    float m_two_k_rf;  // -2 k_rf (reaction-field flavours)
};

template <int EL, bool ENER, bool PRUNE, bool VDWC, int VMOD, bool PLAIN>
__device__ __forceinline__ void pair_interaction(const nbx_kernel_view_t &a, const float4 xi, const float4 xj,
                                        const float2 c, const float c6g, const unsigned wexcl, const int bitpos, const bool allowed,
                                        const nbx_vconst_t &vc,
                                        float &fix, float &fiy, float &fiz, float &fjx, float &fjy, float &fjz,
                                        nbx_eacc_t &acc, bool &any_in_rlist)
{
    const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
    // 0 for an excluded pair, all ones otherwise.  PLAIN: the group has no exclusion word and no diagonal cluster pair
    // (nbx_jhdr_t::plain), so the mask is a constant and every and_mask() below folds away
    int         m = -1;
    // (as asm: left to itself the compiler turns it into v_and + v_cmp + v_cndmask)
    if (!PLAIN) { __asm__("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(wexcl), "n"(bitpos)); }
    // the reference's r^2 += (1 - int_bit) * NBNXN_AVOID_SING_R2_INC (nbnxn_consts.h:66-79,
    // nbnxn_cuda_kernel.cuh:371, nbnxn_kernel_ref_inner.h:132): two distinct, mutually excluded atoms may sit on
    // one point (shells, virtual sites); their Ewald / reaction-field / LJ-PME correction must stay finite.  Here the
    // increment (in the kernel's scaled units, from a VGPR) starts the fma chain of EVERY pair: for a pair that
    // interacts it is below half an ulp of r^2 from r = 0.005 nm on (1e-12 / r^2 < 2^-25), far inside any LJ core, so
    // those pairs get the same float r^2 as without it and the mask instruction of the selective form is saved; it is
    // also far below an ulp of rc^2 / rlist^2, so cut-off and pruning decisions are the reference's.
    const float r2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, vc.r2inc)));
    if (PRUNE) { any_in_rlist = __any(r2 < a.k.rlist2); }
    // The cut-off and the j > i rule of a diagonal cluster pair go into the EXEC mask: lanes outside take no part in
    // what follows, so nothing below selects on the cut-off or multiplies by a 0 / 1 factor, and values that would
    // overflow there (filler atoms sit thousands of nm away) are never computed.
    if (!((r2 < a.k.rc2) && (PLAIN || allowed))) { return; }
    const float inv_r  = __builtin_amdgcn_rsqf(r2);
    if (VMOD == 0 && !VDWC && ENER)
    {
        // The energy kernels of the plain cut-off flavours (the headline configuration).  The exclusion mask goes onto
        // 1/r ONCE: r^-2, r^-6, the LJ force and energy and the plain Coulomb term then vanish for excluded pairs by
        // themselves.  The cut-off is a 0 / 1 factor: one select, and the two energy adds become fused multiply-adds.
        // The potential-shift constants do not enter here at all (nbx_eacc_t).  55 instructions per pair for
        // analytical Ewald where masking the results and selecting each of them took 60.
        const float inv_rm = and_mask(inv_r, m);
        const float u2     = inv_rm * inv_rm;
        const float u6     = u2 * u2 * u2;
        const float qq     = xi.w * xj.w;
        const float lj6    = u6 * fmaf(c.y, u6, -c.x);              // F_lj * r = c12 r^-12 - c6 r^-6
        const float t      = fmaf(qq, inv_rm, lj6) * u2;
        float       F, ec;
        if (EL == NBX_EL_RF)
        {
            F  = fmaf(qq, vc.m_two_k_rf, t);
            ec = inv_rm + fmaf(0.5f * a.k.two_k_rf, r2, -a.k.c_rf);   // c_rf also goes with excluded pairs
        }
        else
        {
            float gf, gv;
            pme_corr_fv(r2, gf, gv);
            if (EL == NBX_EL_EWALD_TAB)
            {
                const float rs  = fminf(r2 * inv_r * a.k.tab_scale, a.k.tab_max);
                const int   idx = (int)rs;
                const float fr  = rs - (float)idx;
                const float t0  = a.tab[idx], t1 = a.tab[idx + 1];
                F = fmaf(qq, -fmaf(fr, t1, fmaf(-fr, t0, t0)) * inv_r, t);
            }
            else { F = fmaf(qq, gf, t); }
            ec = inv_rm - gv;
        }
        const float ws = and_mask(1.0f, m);
        acc.lj  += fmaf(-c.x, u6, lj6);                               // 12 x [c12 r^-12 / 12 - c6 r^-6 / 6]
        acc.s12 = fmaf(c.y, ws, acc.s12);
        acc.s6  = fmaf(c.x, ws, acc.s6);
        acc.el  = fmaf(qq, ec, acc.el);
        if (EL != NBX_EL_RF) { acc.sq = fmaf(qq, ws, acc.sq); }
        fjx = fmaf(-dx, F, fjx); fjy = fmaf(-dy, F, fjy); fjz = fmaf(-dz, F, fjz);
        fix = fmaf(dx, F, fix);  fiy = fmaf(dy, F, fiy);  fiz = fmaf(dz, F, fiz);
        return;
    }
    const float inv_r2 = inv_r * inv_r;
    const float inv_r6 = inv_r2 * inv_r2 * inv_r2;
    // twin-range flavour (VDW_CUTOFF_CHECK, nbnxn_cuda_kernel.cuh:405-409, 436-441): rvdw < rcoulomb
    const bool  vdw_in_range = VDWC ? (r2 < a.k.rvdw2) : true;
    const float qq = xi.w * xj.w;
    const float lj6 = inv_r6 * fmaf(c.y, inv_r6, -c.x);     // F_lj * r = c12 r^-12 - c6 r^-6
    float       t, elj = 0.0f;
    if (VMOD == 0)
    {
        // (F_lj + F_coulomb) * r^2, as r^-6 (c12 r^-6 - c6) + qq / r.  LJ + plain Coulomb vanish for
        // excluded pairs; the reaction-field / Ewald correction stays
        const float lj6v = VDWC ? (vdw_in_range ? lj6 : 0.0f) : lj6;
        t = and_mask(fmaf(qq, inv_r, lj6v) * inv_r2, m);
        if (ENER)
        {
            // (twin-range flavours only: the others returned above) 12 x [c12 (r^-12 + rep_cpot) / 12 - c6 (r^-6 + disp_cpot) / 6];
            // the 1/12 is applied to the wave's sum
            elj = and_mask(fmaf(-c.x, inv_r6, lj6 + 12.0f * fmaf(c.y, a.k.rep12, c.x * a.k.ndisp6)), m);
        }
    }
    else if (VMOD == 3)
    {
        // LJ-PME: the mesh part of the dispersion is taken off the pair force; like the Ewald
        // correction it also acts between excluded pairs (nbnxn_cuda_kernel_utils.cuh:186-302,
        // nbnxn_kernel_ref_inner.h:207-251).  c6g is the grid C6 of the type pair.
        const float cr2  = r2 * a.k.lje2;
        const float ex   = __builtin_amdgcn_exp2f(r2 * a.k.lje2_nlog2e);
        const float poly = fmaf(fmaf(0.5f, cr2, 1.0f), cr2, 1.0f);
        const float g    = c6g * fmaf(-ex, fmaf(inv_r6, poly, a.k.lje6_6), inv_r6);   // grid force * r
        if (VDWC)
        {
            const float ljv = vdw_in_range ? and_mask(lj6, m) + g : 0.0f;
            t = (and_mask(qq * inv_r, m) + ljv) * inv_r2;
        }
        else { t = (and_mask(fmaf(qq, inv_r, lj6), m) + g) * inv_r2; }
        if (ENER)
        {
            // the shift sh_lj_ewald only goes with real LJ pairs
            const float shift = fmaf(c.y, a.k.rep12, fmaf(c.x, a.k.ndisp6, c6g * a.k.lje_sh6));
            elj = and_mask(fmaf(c.x * inv_r6, -1.0f / 12.0f, fmaf(lj6, 1.0f / 12.0f, shift)), m)
                + c6g * (1.0f / 6.0f) * inv_r6 * fmaf(-ex, poly, 1.0f);
        }
    }
    else
    {
        // LJ force switch / potential switch (nbnxn_kernel_ref_inner.h:170-205,
        // nbnxn_cuda_kernel_utils.cuh:84-183), in the kernel's scaled units
        const float r   = r2 * inv_r;
        const float rsw = fmaxf(r - a.k.vdw_rsw, 0.0f);
        float       flj = and_mask(lj6 * inv_r2, m);         // F_lj / r, masked like inv_r6 in the reference
        const float shift = fmaf(c.y, a.k.rep12, c.x * a.k.ndisp6);
        elj = fmaf(c.x * inv_r6, -1.0f / 12.0f, fmaf(lj6, 1.0f / 12.0f, shift));
        if (VMOD == 1)
        {
            // the switch force is NOT masked by the exclusion bit in the reference kernels
            const float rsw2 = rsw * rsw;
            flj += (c.y * fmaf(a.k.fsw_r3, rsw, a.k.fsw_r2) - c.x * fmaf(a.k.fsw_d3, rsw, a.k.fsw_d2)) * rsw2 * inv_r;
            if (ENER)
            {
                elj += (c.x * fmaf(a.k.fsw_d3 * 0.25f, rsw, a.k.fsw_d2 * (1.0f / 3.0f))
                        - c.y * fmaf(a.k.fsw_r3 * 0.25f, rsw, a.k.fsw_r2 * (1.0f / 3.0f))) * rsw2 * rsw;
            }
            elj = and_mask(elj, m);
        }
        else
        {
            elj = and_mask(elj, m);
            const float sw  = fmaf(fmaf(fmaf(a.k.psw_c5, rsw, a.k.psw_c4), rsw, a.k.psw_c3), rsw * rsw * rsw, 1.0f);
            const float dsw = fmaf(fmaf(5.0f * a.k.psw_c5, rsw, 4.0f * a.k.psw_c4), rsw, 3.0f * a.k.psw_c3) * rsw * rsw;
            flj = fmaf(flj, sw, -elj * dsw * inv_r);
            elj *= sw;
        }
        if (VDWC) { flj = vdw_in_range ? flj : 0.0f; }
        t = and_mask(qq * inv_r * inv_r2, m) + flj;
    }
    if (VDWC && ENER) { elj = vdw_in_range ? elj : 0.0f; }
    float F, gv = 0.0f;
    if (EL == NBX_EL_RF) { F = fmaf(qq, vc.m_two_k_rf, t); }
    else if (EL == NBX_EL_EWALD_TAB)
    {
        // force table with linear interpolation (nbnxn_cuda_kernel_utils.cuh:304-317); the table
        // was divided by beta^2 at upload and is indexed by the scaled distance.  The index is
        // clamped: pairs beyond the cut-off (and the r = 0 self pair, NaN -> last entry) are
        // discarded below but must not read outside the table.  Energies stay analytical, as
        // in the reference (nbnxn_cuda_kernel.cuh:468-471).
        const float rs  = fminf(r2 * inv_r * a.k.tab_scale, a.k.tab_max);
        const int   idx = (int)rs;
        const float fr  = rs - (float)idx;
        const float t0  = a.tab[idx], t1 = a.tab[idx + 1];
        const float T   = fmaf(fr, t1, fmaf(-fr, t0, t0));
        if (ENER) { float unused; pme_corr_fv(r2, unused, gv); }
        F = fmaf(qq, -T * inv_r, t);
    }
    else
    {
        float gf;
        // scaled coordinates: r2 is (beta r)^2 and qq carries beta^2
        if (ENER) { pme_corr_fv(r2, gf, gv); }
        else { gf = pme_corr_f(r2); }
        F = fmaf(qq, gf, t);
    }
    if (ENER)
    {
        float eel;
        if (EL == NBX_EL_RF) { eel = qq * (and_mask(inv_r, m) + fmaf(0.5f * a.k.two_k_rf, r2, -a.k.c_rf)); }
        else { eel = qq * (and_mask(inv_r - a.k.sh, m) - gv); }
        acc.lj += elj;
        acc.el += eel;
    }
    fjx = fmaf(-dx, F, fjx); fjy = fmaf(-dy, F, fjy); fjz = fmaf(-dz, F, fjz);
    fix = fmaf(dx, F, fix);  fiy = fmaf(dy, F, fiy);  fiz = fmaf(dz, F, fiz);
}

struct nbx_jhdr_t {
    unsigned imask;
    int      cj[4];
    unsigned wexcl;   // per lane
    bool     plain;   // wave-uniform: no exclusion word and no j-cluster of the entry's own super-cluster in a central entry
};

struct nbx_jslot_t {
    float4 x;
    int    t;
};

// buffer resources of the coordinate and type arrays plus this lane's byte offsets into a cluster's rows
struct nbx_jsrc_t {
    nbx_i32x4 rx, rt;
    int       voff16, voff4;
};

__device__ __forceinline__ nbx_i32x4 make_rsrc(const void *p)
{
    // raw buffer (stride 0), no range limit: the offsets come from the list, which the API checks against natoms
    const unsigned long long a = (unsigned long long)p;
    nbx_i32x4                r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xffffu);
    r.z = -1;
    r.w = 0x00020000;
    return r;
}

// Ordering of LDS accesses between the lanes of ONE wave (the workgroup is one wave).  __syncthreads() would do, but
// as a workgroup-scope fence it also drains every outstanding global operation (s_waitcnt vmcnt(0)): the j-force
// atomics just issued, ~1-3 thousand cycles under load, several times per entry.  The LDS processes one wave's
// instructions in order, so a write followed by a read needs no wait at all; what is needed is that the COMPILER keeps
// the order, which a wavefront-scope fence does without emitting an instruction.
#ifndef NBX_WAVE_SYNC
#define NBX_WAVE_SYNC 1
#endif
__device__ __forceinline__ void wave_lds_sync()
{
#if NBX_WAVE_SYNC
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __syncthreads();
#endif
}

// a cj4 group as it lies in memory: w0 = cj[4], w1 = {imei[0].imask, imei[0].excl_ind, imei[1].imask, imei[1].excl_ind}
struct nbx_jraw_t {
    int4 w0, w1;
};

__device__ __forceinline__ void load_jraw(const nbx_kernel_view_t &a, int j4, nbx_jraw_t &r)
{
    // the whole 32-byte group in one scalar load
    const int4 *g = reinterpret_cast<const int4 *>(a.cj4 + j4);
    r.w0 = g[0];
    r.w1 = g[1];
}

__device__ __forceinline__ void decode_jhdr(const nbx_kernel_view_t &a, const nbx_jraw_t &r, int lane, const bool central, const int sci, nbx_jhdr_t &h)
{
    // slots without interaction bits hold undefined j-clusters (nbnxn_search.c:2460-2472): cluster 0 is read for them
    const int4 w0 = r.w0, w1 = r.w1;
    h.imask = (unsigned)w1.x | (unsigned)w1.z;
    const int ei0 = w1.y, ei1 = w1.w;
    // excl[0] is the shared all-ones entry (types/nbnxn_pairlist.h:197-199): no load for it
    h.wexcl = 0xffffffffu;
    if ((ei0 | ei1) != 0) { h.wexcl = a.excl[lane < 32 ? ei0 : ei1].pair[lane & 31]; }
    h.cj[0] = (h.imask & 0x000000ffu) ? w0.x : 0;
    h.cj[1] = (h.imask & 0x0000ff00u) ? w0.y : 0;
    h.cj[2] = (h.imask & 0x00ff0000u) ? w0.z : 0;
    h.cj[3] = (h.imask & 0xff000000u) ? w0.w : 0;
#if NBX_PLAIN_GROUPS
    // Most groups carry no exclusion word and no diagonal cluster pair: they take the pair body without the exclusion
    // mask and without the j > i rule (PLAIN).  Decided here, one group ahead, on wave-uniform words only.
    const bool own = central && (((h.cj[0] >> 3) == sci) | ((h.cj[1] >> 3) == sci) | ((h.cj[2] >> 3) == sci) | ((h.cj[3] >> 3) == sci));
    h.plain = ((ei0 | ei1) == 0) && !own;
#else
    h.plain = false;
#endif
}

__device__ __forceinline__ void load_jhdr(const nbx_kernel_view_t &a, int j4, int lane, nbx_jhdr_t &h)
{
    // the whole 32-byte group in one scalar load; slots without interaction bits hold undefined j-clusters
    // (nbnxn_search.c:2460-2472): cluster 0 is read for them instead
    const int4 *g  = reinterpret_cast<const int4 *>(a.cj4 + j4);
    const int4  w0 = g[0];   // cj[4]
    const int4  w1 = g[1];   // imei[0].imask, imei[0].excl_ind, imei[1].imask, imei[1].excl_ind
    h.imask = (unsigned)w1.x | (unsigned)w1.z;
    const int ei0 = w1.y, ei1 = w1.w;
    // excl[0] is the shared all-ones entry (types/nbnxn_pairlist.h:197-199): no load for it
    h.wexcl = 0xffffffffu;
    if ((ei0 | ei1) != 0) { h.wexcl = a.excl[lane < 32 ? ei0 : ei1].pair[lane & 31]; }
    h.cj[0] = (h.imask & 0x000000ffu) ? w0.x : 0;
    h.cj[1] = (h.imask & 0x0000ff00u) ? w0.y : 0;
    h.cj[2] = (h.imask & 0x00ff0000u) ? w0.z : 0;
    h.cj[3] = (h.imask & 0xff000000u) ? w0.w : 0;
    h.plain = false;
}

__device__ __forceinline__ void load_jslot(const nbx_jsrc_t &js, int cj, nbx_jslot_t &s)
{
    // buffer loads: the wave-uniform cluster offset rides in an SGPR, the lane's row offset is a constant VGPR --
    // no address arithmetic per j-cluster
#ifdef NBX_DBG_JFIX      // timing experiments only (wrong forces): every j-cluster is one of clusters 0-3, always in L1
    cj &= 3;
#endif
    const nbx_f32x4 v = nbx_buffer_load_f32x4(js.rx, js.voff16, cj * 128, 0);
    s.x = make_float4(v.x, v.y, v.z, v.w);
    s.t = nbx_buffer_load_i32(js.rt, js.voff4, cj * 32, 0);
}

// Per-entry LJ table in LDS (NBFP_LDS): for every j-type a row of [tidxi][i-cluster] parameter entries of THIS entry's
// 64 i-atoms, so a lane finds the eight entries it needs for a j-atom of type tj at consecutive addresses
// tj * row + tidxi * 8 * E + i * E: one multiply-add for the address and four (E = 8) or eight (E = 16) 16-byte reads
// with immediate offsets per j-cluster, where a shared [tj][ti] table costs an address per pair.  Rows are padded by
// Bank layout: a 16-byte read is served in groups of 16 lanes that hold positions 0-3 of two j-atoms and positions
// 4-7 of two others (MI355X_MICROARCH.md, LDS), each lane taking 4 of the 64 banks.  With 8-byte entries a position's
// eight entries are 64 bytes; positions 4-7 are shifted by 16 bytes and rows by 32 bytes (mod 256), so that the four
// positions of a half and the rows of two j-types tile the banks exactly.  With 16-byte entries positions are 144 bytes
// apart (9 bank quads: the eight positions fall into eight different quads).
constexpr int ljt_pos_bytes(int vmod, bool ener, int pos)
{
    return lds_entry_bytes(vmod, ener) == 8 ? pos * 64 + (pos >> 2) * 16 : pos * 144;
}
constexpr int ljt_row_bytes(int vmod, bool ener) { return lds_entry_bytes(vmod, ener) == 8 ? 544 : 1168; }

template <int EL, bool ENER, bool PRUNE, bool VDWC, int VMOD, bool NBFP_LDS, int JM, bool PLAIN>
__device__ __forceinline__ void jcluster(const nbx_kernel_view_t &a, const float4 (&xi)[8], const unsigned tlo,
                                            const unsigned thi, float (&fix)[8], float (&fiy)[8], float (&fiz)[8],
                                            const nbx_jslot_t &s, const int cj, const unsigned wexcl,
                                            unsigned &imask, const bool central, const int sci, const int tidxi,
                                            const int tidxj, const char *ljt_lane, float *s_fj_lane,
                                            int &cjs, int &nfj, nbx_eacc_t &acc, const float ox, const float oy,
                                            const float oz, const nbx_vconst_t &vc, const float4 *s_xi_lane)
{
    if (!(imask & (0xffu << (JM * 8)))) { return; }
    float2 c[8];
    float  cg[8];
    constexpr int  TS    = lds_entry_bytes(VMOD, ENER);
    constexpr bool THIRD = has_third(VMOD, ENER);
#ifdef NBX_DBG_NOLJROWS   // timing experiments only (wrong forces): no LDS reads for the LJ rows
    if (true)
    {
#pragma unroll
        for (int i = 0; i < 8; i++) { c[i] = make_float2(xi[i].w, xi[i].w * 0.5f); cg[i] = 0.0f; }
    }
    else
#endif
    constexpr bool CPAIR = lj_per_pair(EL, VMOD, NBFP_LDS);
    constexpr bool XIPF  = xi_prefetch(EL, ENER, CPAIR);
    if (CPAIR) { }
    else if (NBFP_LDS)
    {
        const char *row = ljt_lane + __umul24((unsigned)s.t, (unsigned)ljt_row_bytes(VMOD, ENER));
        if (THIRD)
        {
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const float4 v = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(row + 16 * i, 16));
                c[i]  = make_float2(v.x, v.y);
                cg[i] = v.z;
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const float4 v = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(row + 16 * k, 16));
                c[2 * k]     = make_float2(v.x, v.y);
                c[2 * k + 1] = make_float2(v.z, v.w);
                cg[2 * k] = cg[2 * k + 1] = 0.0f;
            }
        }
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const unsigned tb = ((i < 4 ? tlo : thi) >> (8 * (i & 3))) & 0xffu;
            c[i]  = a.nbfp[tb * a.ntype + s.t];
            cg[i] = VMOD == 3 ? a.c6grid[tb * a.ntype + s.t] : 0.0f;
            if (EL != NBX_EL_RF) { c[i].x *= a.k.c6scale; c[i].y *= a.k.c12scale; cg[i] *= a.k.c6scale; }
        }
    }
    // lj_per_pair: the LJ entry of a pair is read when the pair is computed (2 live registers instead of 16)
    // (the row's LDS address is pinned in a register: left alone the compiler re-adds lane base and row offset in front
    // of every pair's read)
    typedef const char __attribute__((address_space(3))) *lds_cptr_t;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef const f32x2_t __attribute__((address_space(3))) *lds_f2ptr_t;
    unsigned crow_addr = (unsigned)(unsigned long long)(lds_cptr_t)(ljt_lane + __umul24((unsigned)s.t, (unsigned)ljt_row_bytes(VMOD, ENER)));
    if (CPAIR) { __asm__ volatile("" : "+v"(crow_addr)); }
    const lds_cptr_t crow = (lds_cptr_t)(unsigned long long)crow_addr;
    // i-atoms from LDS (xi_in_lds), xi_prefetch: the row of pair i + 1 is requested at the head of pair i, outside the test of pair
    // i + 1's mask bit -- as a volatile access, which the compiler neither sinks into the conditional body (where its
    // whole latency stood between the branch and the first subtraction) nor splits into a 12- and a 4-byte read (8 + 2
    // LDS cycles where the 16-byte read takes 4, MI355X_MICROARCH.md)
    nbx_f32x4 xin;
    if (XIPF) { xin = lds_load_f32x4_volatile(s_xi_lane); }
    float4 xj = s.x;
    // Ewald: scaled coordinates relative to the entry's origin, see the kernel
    if (EL != NBX_EL_RF) { xj.x = fmaf(xj.x, a.k.xscale, ox); xj.y = fmaf(xj.y, a.k.xscale, oy); xj.z = fmaf(xj.z, a.k.xscale, oz); }
    float fjx = 0, fjy = 0, fjz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        const unsigned bit = 1u << (JM * 8 + i);
        float4 xiv = xi[i];
        if (XIPF)
        {
            xiv = make_float4(xin.x, xin.y, xin.z, xin.w);
            if (i < 7) { xin = lds_load_f32x4_volatile(s_xi_lane + (i + 1) * 8); }
        }
        if (imask & bit)
        {
            const bool diag = !PLAIN && central && (cj == sci * 8 + i);
            bool       any  = true;
            if (CPAIR) { const f32x2_t v = *(lds_f2ptr_t)(crow + 8 * i); c[i] = make_float2(v.x, v.y); cg[i] = 0.0f; }
            // Ewald flavours: the i-atoms stay in LDS and are read per pair (xi_in_lds)
            if (xi_in_lds(EL) && !XIPF) { xiv = s_xi_lane[i * 8]; }
            pair_interaction<EL, ENER, PRUNE, VDWC, VMOD, PLAIN>(a, xiv, xj, c[i], cg[i], wexcl, JM * 8 + i, !diag || tidxj > tidxi, vc,
                                                 fix[i], fiy[i], fiz[i], fjx, fjy, fjz, acc, any);
            if (PRUNE && !any) { imask &= ~bit; }
        }
    }
#if NBX_JRED2
    // Sums over the 8 lanes of a j-atom, delivered as x / y / z to lanes 0 / 1 / 2 of the group.  x and y share ONE
    // reduction tree: in the first step an even lane keeps its x and takes its neighbour's x, an odd lane keeps its y
    // and takes its neighbour's y, so that after it even lanes carry x sums and odd lanes y sums; the two later steps
    // (lane ^ 2, lane + 4) keep the parity.  6 cross-lane adds and 3 selects where three separate trees took 9 and 2
    // (a DPP add costs the SIMD twice a plain instruction, tools/ubench/valu_mix.hip).
    const bool  odd  = (tidxi & 1) != 0;
    const float keep = odd ? fjy : fjx, give = odd ? fjx : fjy;
    float       rxy  = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
    rxy = dpp_add<0x4E>(rxy);     // quad_perm [2,3,0,1]
    rxy = dpp_add<0x104>(rxy);    // row_shl:4: lane k takes lane k + 4 (lanes 0 / 1 of the group complete x / y)
    fjz = dpp_add<0xB1>(fjz);
    fjz = dpp_add<0x4E>(fjz);
    fjz = dpp_add<0x104>(fjz);    // lanes 0-3 of the group complete z
    // keep the last reduction step out of the lane-masked store below: sunk in there, its cross-lane move
    // can no longer be folded into the add (a DPP source lane must be active)
    __asm__ volatile("" : "+v"(rxy), "+v"(fjz));
    const float fj = tidxi == 2 ? fjz : rxy;
    if (tidxi < 3) { s_fj_lane[nfj * 24] = fj; }
#else
    fjx = reduce8(fjx); fjy = reduce8(fjy); fjz = reduce8(fjz);
    // keep the last reduction step out of the lane-masked store below: sunk in there, its cross-lane move
    // can no longer be folded into the add (a DPP source lane must be active)
    __asm__ volatile("" : "+v"(fjx), "+v"(fjy), "+v"(fjz));
    const float fj = tidxi == 0 ? fjx : (tidxi == 1 ? fjy : fjz);
    if (tidxi < 3) { s_fj_lane[nfj * 24] = fj; }
#endif
    // the parked cluster's index goes into lane nfj of a register (one v_writelane, no LDS traffic, no exec change)
    cjs = nbx_writelane(cj, nfj, cjs);
    nfj++;
}

// j-forces are not sent to memory one cluster at a time: on CDNA the wave's loads, stores
// and atomics retire in order through one counter (vmcnt), so a load issued after an atomic
// cannot be waited for without also waiting for that atomic's round trip to the memory side
// (~1-3 us under load).  The reduced j-forces of up to FJ_CAP clusters are parked in LDS and
// flushed as full 64-lane atomic instructions; the kernel's loads then only wait for loads.
__device__ __forceinline__ void flush_fj(float *f, const float *s_fj, const int cjs, int n, int lane)
{
    wave_lds_sync();
    // (the lane read takes place with every lane active: a cross-lane read from a lane that has left the loop returns 0)
    for (int base = 0; base < n * 24; base += 64)
    {
        const int idx  = base + lane;
        const int cl   = (idx * 2731) >> 16;   // idx / 24 for idx < 4096
        const int comp = idx - cl * 24;
        const int cj   = __shfl(cjs, cl);
#ifdef NBX_DBG_NO_JATOMICS      // timing experiments only (wrong forces): what the j-force atomics cost
        if (idx < n * 24 && s_fj[idx] == 1.2345e33f) { atomicAdd(&f[(unsigned)(cj * 24 + comp)], s_fj[idx]); }
#else
        if (idx < n * 24) { atomicAdd(&f[(unsigned)(cj * 24 + comp)], s_fj[idx]); }
#endif
    }
    wave_lds_sync();
}

// sum over the wave with DPP moves only (no LDS round trips); the result is valid in lanes 48..63
__device__ __forceinline__ float wave_sum_hi(float v)
{
    v = reduce8(v);
    v = dpp_add<0x140>(v);   // row_mirror: every lane of a row holds the row's sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
    return v;
}

template <int EL, bool ENER, bool PRUNE, bool VDWC, int VMOD, bool NBFP_LDS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kernel_waves(EL, ENER, NBFP_LDS), 8))) nbnxn_kernel(NBX_KERNEL_PARAMS)
{
    NBX_KERNEL_ARGS_FROM_PARAMS
    const int lane  = threadIdx.x;
    const int tidxi = lane & 7;
    const int tidxj = lane >> 3;

    __shared__ float4 s_xqi[64];
    __shared__ int    s_ti[NBFP_LDS ? 1 : 64];
    constexpr int     TS     = lds_entry_bytes(VMOD, ENER);
    constexpr int     ROW    = ljt_row_bytes(VMOD, ENER);
    constexpr int     FJ_CAP = NBX_FJ_CAP;
    // j-force parking area; at the start of an entry it holds the staged LJ table (MAX_LDS_TYPES^2 x 12 bytes = 3 KB),
    // at its end the transposed i-forces
    __shared__ __attribute__((aligned(16))) float s_fj[FJ_CAP * 24];
    static_assert(MAX_LDS_TYPES * MAX_LDS_TYPES * 12 <= FJ_CAP * 24 * 4, "staged LJ table must fit the parking area");
    // per-entry LJ table, ntype rows of ROW bytes (dynamic: sized by the launch)
    extern __shared__ __attribute__((aligned(16))) char s_ljt[];
    int               nfj = 0;
    int               cjs = 0;

#ifdef NBX_TIMELINE
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl_c0 = __builtin_amdgcn_s_memtime();
#endif
    // the sci entry heads the longest chain of dependent requests of the prologue (entry -> first group -> j-atoms): it is
    // the first thing asked for, in the same batch of kernel arguments as the table below
#ifdef NBX_DBG_SAME_ENTRY   // timing experiments only: every wave walks one of the first 64 (largest) entries: everything in cache
    const nbx_sci_t e     = a.sci[blockIdx.x & 63];
#else
    const nbx_sci_t e     = a.sci[blockIdx.x];
#endif
    const int       ntype = a.ntype;
#if NBX_PRO2
    // the type-pair table depends on nothing but the kernel arguments: requested first, it arrives while the entry,
    // its first group and its i-atoms (three dependent round trips) are on their way
    float2 st_c[(MAX_LDS_TYPES * MAX_LDS_TYPES + 63) / 64];
    float  st_g[(MAX_LDS_TYPES * MAX_LDS_TYPES + 63) / 64];
    if (NBFP_LDS)
    {
#pragma unroll
        for (int m = 0; m < (MAX_LDS_TYPES * MAX_LDS_TYPES + 63) / 64; m++)
        {
            const int k = lane + 64 * m;
            st_c[m] = make_float2(0.0f, 0.0f); st_g[m] = 0.0f;
            if (k < ntype * ntype)
            {
                st_c[m] = a.nbfp[k];
                if (VMOD == 3) { st_g[m] = a.c6grid[k]; }
            }
        }
    }
#endif
    const int       sci   = e.sci;
    const bool      central = (e.shift == NBX_CENTRAL);

#if !NBX_ZERO_AT_END
    // in-kernel clear of the idle output buffer (nbnxn_hip_set_clear_in_kernel): the next step
    // accumulates into it, so no fill has to be dispatched between two steps
    for (int i = blockIdx.x * 64 + lane; i < p_zero_n4; i += gridDim.x * 64) { p_zero[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
#endif

    nbx_jsrc_t js;
    js.rx     = make_rsrc(a.xq);
    js.rt     = make_rsrc(a.type);
    js.voff16 = tidxj * 16;
    js.voff4  = tidxj * 4;

    nbx_jhdr_t  cur;
    nbx_jslot_t sa, sb;
    // The first group's words are REQUESTED here and decoded below: the entry's own atoms, their types and the shift
    // vector hang on the sci entry alone, so they go out before the wave waits for the group (round 3; they used to be
    // requested behind that wait, one round trip later, and the LDS staging of the i-atoms and of the LJ rows then stood
    // between the arrival of the first j-atoms and the first pair instead of beside their flight).
    // (the two kernel arguments the decode below needs are fetched early: left alone the compiler fetched them in a round
    // trip of their own between the first group's arrival and the request for its j-atoms.  Only by-value uses: naming a
    // pointer argument in an asm operand takes its loads off the scalar path.)
    {
        const unsigned long long excl_bits = reinterpret_cast<unsigned long long>(p_excl);
        __asm__ volatile("" : : "s"(excl_bits), "s"(p_k.qscale));
    }
    const nbx_f32x4 vi  = nbx_buffer_load_f32x4(js.rx, lane * 16, sci * 1024, 0);
    int             ti  = nbx_buffer_load_i32(js.rt, lane * 4, sci * 256, 0);
    const float     shx = a.shift_vec[3 * e.shift + 0], shy = a.shift_vec[3 * e.shift + 1], shz = a.shift_vec[3 * e.shift + 2];
    nbx_jraw_t raw0;
    raw0.w0 = make_int4(0, 0, 0, 0); raw0.w1 = make_int4(0, 0, 0, 0);
    if (e.cj4_ind_start < e.cj4_ind_end) { load_jraw(a, e.cj4_ind_start, raw0); }
    decode_jhdr(a, raw0, lane, central, sci, cur);
    load_jslot(js, cur.cj[0], sa);
    constexpr bool JA3 = j_ahead3(EL, ENER, VMOD);
    nbx_jslot_t    sc, sd;
    if (JA3)
    {
        load_jslot(js, cur.cj[1], sb);
        load_jslot(js, cur.cj[2], sc);
    }

    // Ewald: the scaled coordinates are x' = fma(x, beta, C) with C = -fl(beta * o), o the entry's first (shifted)
    // i-atom.  The product is exact inside the fma and C is the SAME number for every i- and j-atom of the entry, so
    // it cancels in x_i' - x_j': what is left is the rounding of two numbers of local magnitude (ulp(beta * 3 nm)),
    // where scaling the raw coordinates would round at ulp(beta |x|) and cost ~1e-5 of force accuracy.
    float ox = 0, oy = 0, oz = 0;
    {
        float4 x;
        x.x = vi.x + shx;
        x.y = vi.y + shy;
        x.z = vi.z + shz;
        x.w = vi.w * a.k.qscale;
        if (EL != NBX_EL_RF)
        {
            // o is lane 0's atom: taken from its registers, no trip through LDS
            ox = -(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x.x))) * a.k.xscale);
            oy = -(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x.y))) * a.k.xscale);
            oz = -(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x.z))) * a.k.xscale);
            // the pair loop reads the rows from LDS (xi_in_lds): they are staged scaled, once
            if (xi_in_lds(EL)) { x.x = fmaf(x.x, a.k.xscale, ox); x.y = fmaf(x.y, a.k.xscale, oy); x.z = fmaf(x.z, a.k.xscale, oz); }
        }
        s_xqi[lane] = x;
        if (NBFP_LDS)
        {
            // the type-pair table as it lies in memory ([ti][tj]), scaled, staged in the parking area
            float2 *stage  = reinterpret_cast<float2 *>(s_fj);
            float  *stage3 = s_fj + 2 * MAX_LDS_TYPES * MAX_LDS_TYPES;
#if NBX_PRO2
#pragma unroll
            for (int m = 0; m < (MAX_LDS_TYPES * MAX_LDS_TYPES + 63) / 64; m++)
            {
                const int k = lane + 64 * m;
                if (k < ntype * ntype)
                {
                    float2 c = st_c[m];
                    if (EL != NBX_EL_RF) { c.x *= a.k.c6scale; c.y *= a.k.c12scale; }
                    stage[k] = c;
                    if (VMOD == 3) { stage3[k] = st_g[m] * (EL != NBX_EL_RF ? a.k.c6scale : 1.0f); }
                }
            }
#else
            for (int k = lane; k < ntype * ntype; k += 64)
            {
                float2 c = a.nbfp[k];
                if (EL != NBX_EL_RF) { c.x *= a.k.c6scale; c.y *= a.k.c12scale; }
                stage[k] = c;
                if (VMOD == 3) { stage3[k] = a.c6grid[k] * (EL != NBX_EL_RF ? a.k.c6scale : 1.0f); }
            }
#endif
        }
        else { s_ti[lane] = ti; }
    }
    wave_lds_sync();

    float c6ii = 0.0f;   // LJ-PME self term
    if (NBFP_LDS)
    {
        // this lane as i-atom (cluster lane / 8, position lane % 8) fills its column of every j-type's row
        const float2 *stage  = reinterpret_cast<const float2 *>(s_fj);
        const float  *stage3 = s_fj + 2 * MAX_LDS_TYPES * MAX_LDS_TYPES;
        char         *dst    = s_ljt + (TS == 8 ? tidxi * 64 + (tidxi >> 2) * 16 : tidxi * 144) + tidxj * TS;
        const int     base   = ti * ntype;
#pragma unroll 4
        for (int tj = 0; tj < ntype; tj++)
        {
            const float2 c = stage[base + tj];
            if (has_third(VMOD, ENER))
            {
                *reinterpret_cast<float4 *>(__builtin_assume_aligned(dst + tj * ROW, 16)) = make_float4(c.x, c.y, stage3[base + tj], 0.0f);
            }
            else { *reinterpret_cast<float2 *>(dst + tj * ROW) = c; }
        }
        if (ENER && VMOD == 3) { c6ii = stage[base + ti].x; }
    }
    else if (ENER && VMOD == 3) { c6ii = a.nbfp[ti * ntype + ti].x * (EL != NBX_EL_RF ? a.k.c6scale : 1.0f); }

    // this lane's eight i-atoms (one per i-cluster) stay in registers for the whole entry, or are read from LDS per
    // pair (xi_in_lds); (global-table path: their types are packed one per byte)
    float4   xi[8];
    unsigned tlo = 0, thi = 0;
    float    fix[8], fiy[8], fiz[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        if (xi_in_lds(EL)) { xi[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
        else
        {
            xi[i] = s_xqi[i * 8 + tidxi];
            if (EL != NBX_EL_RF)
            {
                xi[i].x = fmaf(xi[i].x, a.k.xscale, ox); xi[i].y = fmaf(xi[i].y, a.k.xscale, oy); xi[i].z = fmaf(xi[i].z, a.k.xscale, oz);
            }
        }
        if (!NBFP_LDS)
        {
            const unsigned t = (unsigned)s_ti[i * 8 + tidxi];
            if (i < 4) { tlo |= t << (8 * i); } else { thi |= t << (8 * (i - 4)); }
        }
        fix[i] = 0; fiy[i] = 0; fiz[i] = 0;
    }

    // the r^2 increment lives in a VGPR: as the addend of the distance's first fma it would otherwise be an SGPR
    // operand, and a VALU instruction with one issues at half rate (tools/ubench/valu_mix.hip)
    nbx_vconst_t vc;
    __asm__ volatile("v_mov_b32 %0, %1" : "=v"(vc.r2inc) : "s"(a.k.r2inc));
    vc.m_two_k_rf = 0.0f;
#if NBX_RF_VGPR
    if (EL == NBX_EL_RF) { __asm__ volatile("v_mov_b32 %0, %1" : "=v"(vc.m_two_k_rf) : "s"(-a.k.two_k_rf)); }
#else
    if (EL == NBX_EL_RF) { vc.m_two_k_rf = -a.k.two_k_rf; }
#endif

    nbx_eacc_t acc;
    if (ENER)
    {
        if (central && e.cj4_ind_end > e.cj4_ind_start && a.cj4[e.cj4_ind_start].cj[0] == sci * 8)
        {
            // self term (nbnxn_cuda_kernel.cuh:239-262) in the scaled charges
            const float q = s_xqi[lane].w;
            acc.el = a.k.self * q * q;
            // LJ-PME self term: 0.5 * C6_ii / 6 * cl^6 / 6 per atom (nbnxn_cuda_kernel.cuh:232-246)
            if (VMOD == 3) { acc.lj = c6ii * (0.5f / 6.0f) * a.k.lje6_6; }
        }
    }
    // the per-entry table is complete and the staged copy no longer needed: the parking area is free
    wave_lds_sync();

    const char *ljt_lane  = s_ljt + (TS == 8 ? tidxi * 64 + (tidxi >> 2) * 16 : tidxi * 144);
    float      *s_fj_lane = s_fj + tidxj * 3 + (tidxi < 3 ? tidxi : 0);

#ifdef NBX_TIMELINE
    const unsigned long long tl_t1 = __builtin_amdgcn_s_memrealtime();
#endif
#define NBX_JCLUSTER_P(JM, SLOT, PLAIN)                                                                    \
    jcluster<EL, ENER, PRUNE, VDWC, VMOD, NBFP_LDS, JM, PLAIN>(a, xi, tlo, thi, fix, fiy, fiz, SLOT, cur.cj[JM], cur.wexcl,    \
                                               imask, central, sci, tidxi, tidxj, ljt_lane, s_fj_lane, cjs,  \
                                               nfj, acc, ox, oy, oz, vc, s_xqi + tidxi)
#if NBX_HDR2
    // the raw words of the group after the current one, requested a whole group ahead (zeros: no such group)
    nbx_jraw_t raw1;
    raw1.w0 = make_int4(0, 0, 0, 0); raw1.w1 = make_int4(0, 0, 0, 0);
    if (e.cj4_ind_start + 1 < e.cj4_ind_end) { load_jraw(a, e.cj4_ind_start + 1, raw1); }
#endif
    // One iteration of the group loop.  Software pipeline: the next group's list words are requested before this
    // group's arithmetic starts, each j-cluster's atoms one j-cluster ahead.
    // A scalar load that is consumed right away (the exclusion word's address and the j-cluster indices hang on it)
    // parks the wave for a scalar-cache or L2 round trip once per group.  So the group after next is REQUESTED here and
    // the next group, requested one iteration ago, is decoded from registers (all-zero words give an empty group: no
    // interaction bits, cluster 0, no exclusion word).
#if NBX_HDR2
#define NBX_ITER_HEAD                                                                                      \
        decode_jhdr(a, raw1, lane, central, sci, nxt);                                                     \
        nbx_jraw_t raw2;                                                                                   \
        raw2.w0 = make_int4(0, 0, 0, 0); raw2.w1 = make_int4(0, 0, 0, 0);                                  \
        if (j4 + 2 < e.cj4_ind_end) { load_jraw(a, j4 + 2, raw2); }
#define NBX_ITER_TAIL raw1 = raw2;
#else
#define NBX_ITER_HEAD                                                                                      \
        nxt.imask = 0; nxt.wexcl = 0xffffffffu; nxt.plain = false;                                         \
        nxt.cj[0] = nxt.cj[1] = nxt.cj[2] = nxt.cj[3] = 0;                                                 \
        if (j4 + 1 < e.cj4_ind_end) { load_jhdr(a, j4 + 1, lane, nxt); }
#define NBX_ITER_TAIL
#endif
    // j-atoms are requested one j-cluster ahead (two rolling register sets) or, where it was measured to pay (j_ahead3: the
    // reaction-field force-only kernels, -1.4 %; Ewald and the energy kernels 0 ... +0.5 %), three ahead (four sets)
#define NBX_GROUP_BODY(PLAIN)                                                                              \
        if (JA3)                                                                                           \
        {                                                                                                  \
            load_jslot(js, cur.cj[3], sd);                                                                 \
            NBX_JCLUSTER_P(0, sa, PLAIN);                                                                  \
            load_jslot(js, nxt.cj[0], sa);                                                                 \
            NBX_JCLUSTER_P(1, sb, PLAIN);                                                                  \
            load_jslot(js, nxt.cj[1], sb);                                                                 \
            NBX_JCLUSTER_P(2, sc, PLAIN);                                                                  \
            load_jslot(js, nxt.cj[2], sc);                                                                 \
            NBX_JCLUSTER_P(3, sd, PLAIN);                                                                  \
        }                                                                                                  \
        else                                                                                               \
        {                                                                                                  \
            load_jslot(js, cur.cj[1], sb);                                                                 \
            NBX_JCLUSTER_P(0, sa, PLAIN);                                                                  \
            load_jslot(js, cur.cj[2], sa);                                                                 \
            NBX_JCLUSTER_P(1, sb, PLAIN);                                                                  \
            load_jslot(js, cur.cj[3], sb);                                                                 \
            NBX_JCLUSTER_P(2, sa, PLAIN);                                                                  \
            load_jslot(js, nxt.cj[0], sa);                                                                 \
            NBX_JCLUSTER_P(3, sb, PLAIN);                                                                  \
        }
#define NBX_ITER(PLAIN)                                                                                    \
    {                                                                                                      \
        nbx_jhdr_t nxt;                                                                                    \
        NBX_ITER_HEAD                                                                                      \
        unsigned imask = cur.imask;                                                                        \
        NBX_GROUP_BODY(PLAIN)                                                                              \
        if (PRUNE)                                                                                         \
        {                                                                                                  \
            if (lane == 0)                                                                                 \
            {                                                                                              \
                a.cj4[j4].imei[0].imask = imask;                                                           \
                a.cj4[j4].imei[1].imask = imask;                                                           \
            }                                                                                              \
        }                                                                                                  \
        cur = nxt;                                                                                         \
        NBX_ITER_TAIL                                                                                      \
        if (nfj > FJ_CAP - 4) { flush_fj(a.f, s_fj, cjs, nfj, lane); nfj = 0; }                            \
    }
#if NBX_PLAIN_GROUPS
    // Runs of groups with (cur.plain false) and without exclusion words / diagonal cluster pairs alternate: two loops
    // in sequence inside an outer loop, NOT an if / else inside one loop -- over a diamond the register allocator kept
    // two copies of the 24 i-force accumulators and spilled; through loops that follow each other they stay put.
    for (int j4 = e.cj4_ind_start; j4 < e.cj4_ind_end;)
    {
        for (; j4 < e.cj4_ind_end && !cur.plain; j4++) NBX_ITER(false)
        for (; j4 < e.cj4_ind_end && cur.plain; j4++) NBX_ITER(true)
    }
#else
    for (int j4 = e.cj4_ind_start; j4 < e.cj4_ind_end; j4++) NBX_ITER(false)
#endif
#undef NBX_ITER
#undef NBX_ITER_HEAD
#undef NBX_GROUP_BODY
#undef NBX_ITER_TAIL
#undef NBX_JCLUSTER_P
    if (nfj > 0) { flush_fj(a.f, s_fj, cjs, nfj, lane); }
#ifdef NBX_TIMELINE
    const unsigned long long tl_t2 = __builtin_amdgcn_s_memrealtime();
#endif

    // i-forces: every lane holds 24 partial sums (8 i-clusters x xyz) that have to be added over the 8 lanes sharing
    // an i-atom.  They go through LDS transposed, half of the i-clusters at a time (the j-force parking area is free
    // by now and holds exactly 96 x 8 floats): a lane writes its value of (component c, atom a) to word
    // ((3 a + c) * 8 + tidxj) -- bank (24 tidxi + tidxj) mod 64, conflict-free -- and then sums the eight contiguous
    // words of one or two outputs.  24 LDS writes, 8 wide reads and 28 adds per entry where the reduce-scatter over
    // lane shuffles took 21 round trips through the LDS crossbar with a wait each, and 63 selects.
    float s1 = 0, s2 = 0;   // per lane: sums of the outputs it owned
    {
        float *fi = a.f + (size_t)sci * 192;
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int at = k * 8 + tidxi;
                s_fj[(at * 3) * 8 + tidxj]     = fix[4 * h + k];
                s_fj[(at * 3 + 1) * 8 + tidxj] = fiy[4 * h + k];
                s_fj[(at * 3 + 2) * 8 + tidxj] = fiz[4 * h + k];
            }
            wave_lds_sync();
            {
                const float4 p0 = *reinterpret_cast<const float4 *>(&s_fj[lane * 8]);
                const float4 p1 = *reinterpret_cast<const float4 *>(&s_fj[lane * 8 + 4]);
                const float  o1 = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w));
#ifdef NBX_DBG_NO_IATOMICS      // timing experiments only (wrong forces)
                if (o1 == 1.2345e33f)
#endif
                atomicAdd(&fi[h * 96 + lane], o1);
                s1 += o1;
            }
            if (lane < 32)
            {
                const float4 p0 = *reinterpret_cast<const float4 *>(&s_fj[(64 + lane) * 8]);
                const float4 p1 = *reinterpret_cast<const float4 *>(&s_fj[(64 + lane) * 8 + 4]);
                const float  o2 = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w));
#ifdef NBX_DBG_NO_IATOMICS
                if (o2 == 1.2345e33f)
#endif
                atomicAdd(&fi[h * 96 + 64 + lane], o2);
                s2 += o2;
            }
            wave_lds_sync();
        }
    }

#if NBX_ZERO_AT_END
    // in-kernel clear of the idle output buffer (nbnxn_hip_set_clear_in_kernel): the next step accumulates into it, so
    // no fill has to be dispatched between two steps.  Issued last: loads, stores and atomics of a wave retire in order,
    // so at the head of the kernel these stores stood in front of every load of the prologue.
    for (int i = blockIdx.x * 64 + lane; i < p_zero_n4; i += gridDim.x * 64) { p_zero[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
#endif
#ifdef NBX_TIMELINE
    if (lane == 0 && blockIdx.x < 65536)
    {
        unsigned hwid;
        __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        NBX_TL_SYM[NBX_TL_WORDS * blockIdx.x]     = tl_t0;
        NBX_TL_SYM[NBX_TL_WORDS * blockIdx.x + 1] = tl_t1;
        NBX_TL_SYM[NBX_TL_WORDS * blockIdx.x + 2] = tl_t2;
        NBX_TL_SYM[NBX_TL_WORDS * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
        NBX_TL_SYM[NBX_TL_WORDS * blockIdx.x + 4] = hwid | ((unsigned long long)(e.cj4_ind_end - e.cj4_ind_start) << 32) | ((unsigned long long)(xcc & 0xf) << 56);
        NBX_TL_SYM[NBX_TL_WORDS * blockIdx.x + 5] = __builtin_amdgcn_s_memtime() - tl_c0;     // shader-clock cycles of the wave
    }
#endif
    if (ENER)
    {
        if (VMOD == 0 && !VDWC)
        {
            // the potential-shift terms, from the sums over the pairs they go with (nbx_eacc_t)
            acc.lj = fmaf(12.0f, fmaf(acc.s12, a.k.rep12, acc.s6 * a.k.ndisp6), acc.lj);
            if (EL != NBX_EL_RF) { acc.el = fmaf(-a.k.sh, acc.sq, acc.el); }
        }
        const float E_lj = wave_sum_hi(acc.lj) * a.k.escale * (VMOD == 0 ? 1.0f / 12.0f : 1.0f);
        const float E_el = wave_sum_hi(acc.el) * a.k.escale;
        // the sums are complete in the wave's last lanes: lanes 59..61 add the shift force, 62 / 63 the energies
        // into this entry's accumulation slot
        float *slot = a.out + (size_t)(blockIdx.x & (NBX_OUT_SLOTS - 1)) * NBX_OUT_STRIDE;
        if (!central)
        {
            // s1 holds component lane % 3, s2 (lanes 0-31) component (lane + 1) % 3
            const int   c1 = lane % 3, c2 = (lane + 1) % 3;
            const float sx = wave_sum_hi((c1 == 0 ? s1 : 0.0f) + (c2 == 0 ? s2 : 0.0f));
            const float sy = wave_sum_hi((c1 == 1 ? s1 : 0.0f) + (c2 == 1 ? s2 : 0.0f));
            const float sz = wave_sum_hi((c1 == 2 ? s1 : 0.0f) + (c2 == 2 ? s2 : 0.0f));
            if (lane >= 59 && lane < 62) { atomicAdd(&slot[2 + 3 * e.shift + (lane - 59)], lane == 59 ? sx : (lane == 60 ? sy : sz)); }
        }
        if (lane >= 62) { atomicAdd(&slot[lane - 62], lane == 62 ? E_lj : E_el); }
    }
}

// Up to how many atom types the per-entry LJ table in LDS is used (beyond: the global table through L1 / L2).  The
// table costs ntype x 544 bytes of LDS per workgroup (1168 with the LJ-PME grid column) on top of 4.1 KB, and the
// workgroups a CU holds follow from that; GMX_HIP_NB_LDS_TABLE_MAX_TYPES overrides (0: always the global table), for
// the measurement of where the switch belongs (tools/lds_threshold_scan.py).
inline int lds_table_max_types(int vmod)
{
    static const int env = getenv("GMX_HIP_NB_LDS_TABLE_MAX_TYPES") ? atoi(getenv("GMX_HIP_NB_LDS_TABLE_MAX_TYPES")) : -1;
    if (env >= 0) { return env < MAX_LDS_TYPES ? env : MAX_LDS_TYPES; }
    // Measured on the 100 k-atom chain mixture, Ewald V+F, LDS table / global table (profiles/r3_b_lds_threshold.txt):
    // cut-off LJ (8-byte entries) 0.87-0.92 for 3 ... 16 types: the LDS table always pays, although from 11 types on
    // it costs resident workgroups (16 -> 11 per CU).  LJ-PME (16-byte entries, 1168 B per type): 0.94 at 3 and 5
    // types, 1.07 at 7, 1.11 at 9, 1.17 at 15: beyond 6 types the table takes the CU below 16 workgroups and the
    // global table wins.
    return vmod == 3 ? 6 : MAX_LDS_TYPES;
}

template <int EL, bool ENER, bool PRUNE, bool VDWC, int VMOD>
void launch_flavour(const nbx_kernel_args_t &a, int nsci, hipStream_t stream)
{
    if (a.ntype <= lds_table_max_types(VMOD))
    {
        // NBX_DBG_EXTRA_LDS: bytes of unused LDS per workgroup, to lower the occupancy in timing experiments
        static const size_t extra_lds = getenv("NBX_DBG_EXTRA_LDS") ? (size_t)atoi(getenv("NBX_DBG_EXTRA_LDS")) : 0;
        hipLaunchKernelGGL((nbnxn_kernel<EL, ENER, PRUNE, VDWC, VMOD, true>), dim3(nsci), dim3(64), (size_t)a.ntype * ljt_row_bytes(VMOD, ENER) + extra_lds, stream,
                           NBX_KERNEL_CALL_ARGS(a));
    }
    else { hipLaunchKernelGGL((nbnxn_kernel<EL, ENER, PRUNE, VDWC, VMOD, false>), dim3(nsci), dim3(64), 0, stream, NBX_KERNEL_CALL_ARGS(a)); }
}

template <int EL, bool VDWC, int VMOD>
void launch_ener_prune(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, hipStream_t stream)
{
    if (ener) { if (prune) { launch_flavour<EL, true, true, VDWC, VMOD>(a, nsci, stream); }
                else       { launch_flavour<EL, true, false, VDWC, VMOD>(a, nsci, stream); } }
    else      { if (prune) { launch_flavour<EL, false, true, VDWC, VMOD>(a, nsci, stream); }
                else       { launch_flavour<EL, false, false, VDWC, VMOD>(a, nsci, stream); } }
}

template <int EL, bool VDWC>
void launch_vmod(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, int vdw_mod, hipStream_t stream)
{
    if (vdw_mod == 1) { launch_ener_prune<EL, VDWC, 1>(a, nsci, ener, prune, stream); }
    else if (vdw_mod == 2) { launch_ener_prune<EL, VDWC, 2>(a, nsci, ener, prune, stream); }
    else if (vdw_mod == 3) { launch_ener_prune<EL, VDWC, 3>(a, nsci, ener, prune, stream); }
    else { launch_ener_prune<EL, VDWC, 0>(a, nsci, ener, prune, stream); }
}

} // namespace

// One translation unit per electrostatics flavour (Makefile: -DNBX_UNIT_EL=0|1|2), so that the
// flavour table (nbnxn_cuda_kernels.cuh has 6 x 5 x 4 entries) builds in parallel.
#if NBX_UNIT_EL == 0
hipError_t nbx_launch_nonbonded_rf(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, bool, int vdw_mod, hipStream_t stream)
{
    // no twin-range flavour for reaction field: rvdw == rcoulomb there (PME load balancing scales rcoulomb only)
    launch_vmod<NBX_EL_RF, false>(a, nsci, ener, prune, vdw_mod, stream);
    return hipGetLastError();
}

hipError_t nbx_launch_nonbonded(const nbx_kernel_args_t &a, int nsci, int el_type, bool ener, bool prune, bool vdw_cutoff,
                                int vdw_mod, hipStream_t stream)
{
    if (nsci <= 0) { return hipSuccess; }
    if (el_type == NBX_EL_RF) { return nbx_launch_nonbonded_rf(a, nsci, ener, prune, vdw_cutoff, vdw_mod, stream); }
    if (el_type == NBX_EL_EWALD_TAB) { return nbx_launch_nonbonded_ewald_tab(a, nsci, ener, prune, vdw_cutoff, vdw_mod, stream); }
    return nbx_launch_nonbonded_ewald_ana(a, nsci, ener, prune, vdw_cutoff, vdw_mod, stream);
}
#elif NBX_UNIT_EL == 1
hipError_t nbx_launch_nonbonded_ewald_ana(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, bool vdw_cutoff, int vdw_mod, hipStream_t stream)
{
    if (vdw_cutoff) { launch_vmod<NBX_EL_EWALD_ANA, true>(a, nsci, ener, prune, vdw_mod, stream); }
    else { launch_vmod<NBX_EL_EWALD_ANA, false>(a, nsci, ener, prune, vdw_mod, stream); }
    return hipGetLastError();
}
#elif NBX_UNIT_EL == 2
hipError_t nbx_launch_nonbonded_ewald_tab(const nbx_kernel_args_t &a, int nsci, bool ener, bool prune, bool vdw_cutoff, int vdw_mod, hipStream_t stream)
{
    if (vdw_cutoff) { launch_vmod<NBX_EL_EWALD_TAB, true>(a, nsci, ener, prune, vdw_mod, stream); }
    else { launch_vmod<NBX_EL_EWALD_TAB, false>(a, nsci, ener, prune, vdw_mod, stream); }
    return hipGetLastError();
}
#else
#error "NBX_UNIT_EL must be 0 (reaction field), 1 (analytical Ewald) or 2 (tabulated Ewald)"
#endif
