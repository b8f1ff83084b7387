"""Python loader of the CPU checker (oracle/nbnxn_oracle.c) and, where it was built, of the
reference's own kernels (oracle/_ref/libnbnxn_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gromacs-cosmo_amd/ imports this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OracleParams(C.Structure):
    _fields_ = [("el_type", C.c_int),
                ("rcoulomb", C.c_double), ("rvdw", C.c_double),
                ("epsfac", C.c_double), ("k_rf", C.c_double), ("c_rf", C.c_double),
                ("beta", C.c_double), ("sh_ewald", C.c_double),
                ("disp_cpot", C.c_double), ("rep_cpot", C.c_double),
                ("vdw_modifier", C.c_int), ("rvdw_switch", C.c_double),
                ("disp_c2", C.c_double), ("disp_c3", C.c_double), ("rep_c2", C.c_double), ("rep_c3", C.c_double),
                ("sw_c3", C.c_double), ("sw_c4", C.c_double), ("sw_c5", C.c_double),
                ("ljpme", C.c_int), ("ewaldcoeff_lj", C.c_double), ("sh_lj_ewald", C.c_double),
                ("nbfp_comb", C.c_void_p),
                ("coul_tab", C.c_void_p), ("tab_size", C.c_int), ("tab_scale", C.c_double)]


class RefParams(C.Structure):
    _fields_ = [("el_type", C.c_int),
                ("rcoulomb", C.c_float), ("rvdw", C.c_float), ("rlist", C.c_float),
                ("epsfac", C.c_float), ("k_rf", C.c_float), ("c_rf", C.c_float),
                ("ewaldcoeff_q", C.c_float), ("sh_ewald", C.c_float),
                ("disp_cpot", C.c_float), ("rep_cpot", C.c_float),
                ("vdw_modifier", C.c_int), ("rvdw_switch", C.c_float),
                ("disp_c2", C.c_float), ("disp_c3", C.c_float), ("rep_c2", C.c_float), ("rep_c3", C.c_float),
                ("sw_c3", C.c_float), ("sw_c4", C.c_float), ("sw_c5", C.c_float),
                ("ljpme", C.c_int), ("ewaldcoeff_lj", C.c_float), ("sh_lj_ewald", C.c_float)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


GPU_EWALD_TAB_SIZE = 1536   # GPU_EWALD_COULOMB_FORCE_TABLE_SIZE


def ewald_force_table(beta, rc, ntab=GPU_EWALD_TAB_SIZE):
    """(table, scale) as the reference's GPU module makes them (nbnxn_cuda_data_mgmt.cu:120-183)."""
    lib().oracle_ewald_force_table.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p]
    scale = (ntab - 2) / float(np.float32(rc))
    tab = np.zeros(ntab, dtype=np.float32)
    lib().oracle_ewald_force_table(ntab, 1.0 / scale, float(np.float32(beta)), _p(tab))
    return tab, scale


def _ljpme_code(ip):
    """0 off, 1 geometric, 2 Lorentz-Berthelot (oracle_params_t.ljpme)."""
    if getattr(ip, "vdwtype", 0) != 5:
        return 0
    return 1 + int(ip.ljpme_comb_rule)


def nbfp_comb(nbfp, ntype, ljpme):
    """nbat->nbfp_comb from the diagonal of nbat->nbfp (nbnxn_atomdata.c:397-429): ljpme 1 ->
    sqrt(6 C6), sqrt(12 C12); 2 -> 0.5 (C12'/C6')^(1/6), sqrt(C6'^2/C12') (0, 0 without LJ)."""
    t = np.asarray(nbfp, dtype=np.float32).reshape(ntype, ntype, 2).astype(np.float64)
    out = np.zeros((ntype, 2), dtype=np.float32)
    for i in range(ntype):
        c6, c12 = t[i, i]
        if ljpme == 1:
            out[i] = np.sqrt(c6), np.sqrt(c12)
        elif c6 > 0 and c12 > 0:
            out[i] = 0.5 * (c12 / c6) ** (1.0 / 6.0), np.sqrt(c6 * c6 / c12)
    return out


def oracle_params(ip, ntype=None, nbfp=None):
    """ip: gromacs_cosmo_amd.system.InteractionParams (float32-rounded like the C structs).
    LJ-PME needs the LJ table (for nbfp_comb); the arrays the struct points to are kept alive
    on the returned object."""
    p = _oracle_params_scalar(ip)
    p.ljpme = _ljpme_code(ip)
    if p.ljpme:
        f = lambda v: float(np.float32(v))
        p.ewaldcoeff_lj, p.sh_lj_ewald = f(ip.ewaldcoeff_lj), f(ip.sh_lj_ewald)
        p._comb = nbfp_comb(nbfp, ntype, p.ljpme)
        p.nbfp_comb = p._comb.ctypes.data
    if ip.el_type == 2:
        p._tab, sc = ewald_force_table(ip.ewaldcoeff_q, ip.rcoulomb)
        p.coul_tab, p.tab_size, p.tab_scale = p._tab.ctypes.data, len(p._tab), sc
    return p


def _oracle_params_scalar(ip):
    f = lambda v: float(np.float32(v))
    return OracleParams(ip.el_type, f(ip.rcoulomb), f(ip.rvdw), f(ip.epsfac), f(ip.k_rf), f(ip.c_rf),
                        f(ip.ewaldcoeff_q), f(ip.sh_ewald), f(ip.disp_cpot), f(ip.rep_cpot),
                        int(getattr(ip, "vdw_modifier", 1)), f(getattr(ip, "rvdw_switch", 0.0)),
                        f(getattr(ip, "disp_c2", 0.0)), f(getattr(ip, "disp_c3", 0.0)),
                        f(getattr(ip, "rep_c2", 0.0)), f(getattr(ip, "rep_c3", 0.0)),
                        f(getattr(ip, "sw_c3", 0.0)), f(getattr(ip, "sw_c4", 0.0)), f(getattr(ip, "sw_c5", 0.0)))


def ref_params(ip):
    return RefParams(ip.el_type, ip.rcoulomb, ip.rvdw, ip.rlist, ip.epsfac, ip.k_rf, ip.c_rf,
                     ip.ewaldcoeff_q, ip.sh_ewald, ip.disp_cpot, ip.rep_cpot,
                     int(getattr(ip, "vdw_modifier", 1)), getattr(ip, "rvdw_switch", 0.0),
                     getattr(ip, "disp_c2", 0.0), getattr(ip, "disp_c3", 0.0), getattr(ip, "rep_c2", 0.0),
                     getattr(ip, "rep_c3", 0.0), getattr(ip, "sw_c3", 0.0), getattr(ip, "sw_c4", 0.0),
                     getattr(ip, "sw_c5", 0.0), _ljpme_code(ip), getattr(ip, "ewaldcoeff_lj", 0.0),
                     getattr(ip, "sh_lj_ewald", 0.0))


_oracle = None
_ref = None


def lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _oracle = C.CDLL(path)
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libnbnxn_ref.so"))


def reflib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(_HERE, "_ref", "libnbnxn_ref.so"))
        _ref.ref_kernel_ref_setup.restype = C.c_void_p
        _ref.ref_kernel_ref_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _ref.ref_kernel_ref_free.argtypes = [C.c_void_p]
        _ref.ref_kernel_ref_nci.argtypes = [C.c_void_p]
        _ref.ref_kernel_ref_ncj.argtypes = [C.c_void_p]
    return _ref


def kernel_gpu_list(xq, type_, ntype, nbfp, shift_vec, pl, ip, calc_ener=True):
    """Own CPU restatement over the GPU list.  Returns dict(f [n,3] f64, fshift [45,3], Vc, Vvdw, counts)."""
    n = len(xq)
    f = np.zeros((n, 3), dtype=np.float64)
    fshift = np.zeros((45, 3), dtype=np.float64)
    vc, vv = C.c_double(0), C.c_double(0)
    counts = (C.c_longlong * 3)()
    p = oracle_params(ip, ntype, nbfp)
    nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
    rc = lib().oracle_kernel_gpu_list(C.c_int(n), _p(xq), _p(type_), C.c_int(ntype), _p(nbfp), _p(shift_vec),
                                      C.c_int(pl.nsci), _p(pl.sci), _p(pl.cj4), _p(pl.excl),
                                      C.c_double(float(np.float32(pl.rlist))), C.byref(p), C.c_int(int(calc_ener)),
                                      _p(f), _p(fshift), C.byref(vc), C.byref(vv), counts)
    if rc != 0:
        raise RuntimeError("oracle_kernel_gpu_list: list refers to atoms outside the arrays")
    return dict(f=f, fshift=fshift, Vc=vc.value, Vvdw=vv.value, counts=tuple(counts))


def kernel_gpu_list_f32(xq, type_, ntype, nbfp, shift_vec, pl, ip):
    n = len(xq)
    f = np.zeros((n, 3), dtype=np.float32)
    p = oracle_params(ip, ntype, nbfp)
    nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
    lib().oracle_kernel_gpu_list_f32(C.c_int(n), _p(xq), _p(type_), C.c_int(ntype), _p(nbfp), _p(shift_vec),
                                     C.c_int(pl.nsci), _p(pl.sci), _p(pl.cj4), _p(pl.excl), C.byref(p), _p(f))
    return f


def prune_list(xq, shift_vec, pl, rlist):
    """Returns the cj4 array after the GPU kernels' pruning rule."""
    cj4 = pl.cj4.copy()
    lib().oracle_prune_list(_p(xq), _p(shift_vec), C.c_int(pl.nsci), _p(pl.sci), _p(cj4), C.c_float(rlist))
    return cj4


def brute_force(x, q, type_, ntype, nbfp, box, excl_index, excl_a, ip, calc_ener=True):
    n = len(x)
    f = np.zeros((n, 3), dtype=np.float64)
    vir = np.zeros(9, dtype=np.float64)
    vc, vv = C.c_double(0), C.c_double(0)
    p = oracle_params(ip, ntype, nbfp)
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    type_ = np.ascontiguousarray(type_, dtype=np.int32)
    nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
    box = np.ascontiguousarray(box, dtype=np.float32)
    # a [3] box is rectangular, a [3, 3] box the GROMACS matrix (rows are box vectors)
    fn = lib().oracle_brute_force_tric if box.size == 9 else lib().oracle_brute_force
    fn(C.c_int(n), _p(x), _p(q), _p(type_), C.c_int(ntype), _p(nbfp), _p(box),
                             _p(excl_index), _p(excl_a), C.byref(p), C.c_int(int(calc_ener)),
                             _p(f), _p(vir), C.byref(vc), C.byref(vv))
    return dict(f=f, vir=vir.reshape(3, 3), Vc=vc.value, Vvdw=vv.value)


def virial(xq, f, shift_vec, fshift):
    vir = np.zeros(9, dtype=np.float64)
    f = np.ascontiguousarray(f, dtype=np.float64)
    fshift = np.ascontiguousarray(fshift, dtype=np.float64)
    lib().oracle_virial(C.c_int(len(xq)), _p(xq), _p(f), _p(shift_vec), _p(fshift), _p(vir))
    return vir.reshape(3, 3)


def ref_kernel_gpu_ref(xq, type_, ntype, nbfp, shift_vec, pl, ip, calc_ener=True):
    """The reference's nbnxn_kernel_gpu_ref on the same list."""
    n = len(xq)
    f = np.zeros((n, 3), dtype=np.float32)
    fshift = np.zeros((45, 3), dtype=np.float32)
    vc, vv = C.c_float(0), C.c_float(0)
    p = ref_params(ip)
    nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
    reflib().ref_run_kernel_gpu_ref(C.c_int(n), _p(xq), _p(type_), C.c_int(ntype), _p(nbfp), _p(shift_vec),
                                    C.c_int(pl.nsci), _p(pl.sci), C.c_int(pl.ncj4), _p(pl.cj4),
                                    C.c_int(pl.nexcl), _p(pl.excl), C.byref(p), C.c_int(int(calc_ener)),
                                    _p(f), _p(fshift), C.byref(vc), C.byref(vv))
    return dict(f=f, fshift=fshift, Vc=vc.value, Vvdw=vv.value)


def ref_ewald_table(ip):
    """(FDV0 [n, 4] float32, scale) of the Coulomb table the compiled reference kernels consume for `ip`
    (oracle/ref_harness.c make_ic: the harness's restatement of tables.c:165-369 / forcerec.c:1865-1905)."""
    p = ref_params(ip)
    sc = C.c_float(0)
    n = reflib().ref_ewald_table(C.byref(p), C.byref(sc), None, 0)
    out = np.zeros((n, 4), dtype=np.float32)
    reflib().ref_ewald_table(C.byref(p), C.byref(sc), _p(out), out.size)
    return out, sc.value


class RefKernelRef:
    """The reference's nbnxn_kernel_ref (plain-C 4x4) on the simple list derived from the GPU list."""

    def __init__(self, xq, type_, ntype, nbfp, shift_vec, pl, ip, nlist=1):
        self.n = len(xq)
        p = ref_params(ip)
        nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
        self.nlist = nlist
        self.ctx = reflib().ref_kernel_ref_setup(C.c_int(self.n), _p(xq), _p(type_), C.c_int(ntype), _p(nbfp),
                                                 _p(shift_vec), C.c_int(pl.nsci), _p(pl.sci), _p(pl.cj4),
                                                 _p(pl.excl), C.byref(p), C.c_int(nlist))
        self.nci = reflib().ref_kernel_ref_nci(self.ctx)
        self.ncj = reflib().ref_kernel_ref_ncj(self.ctx)

    def run(self, calc_ener=True, nthreads=1, outputs=True):
        if not outputs:
            reflib().ref_kernel_ref_run(self.ctx, int(calc_ener), nthreads, None, None, None, None)
            return None
        f = np.zeros((self.n, 3), dtype=np.float32)
        fshift = np.zeros((45, 3), dtype=np.float32)
        vc, vv = C.c_float(0), C.c_float(0)
        reflib().ref_kernel_ref_run(self.ctx, int(calc_ener), nthreads, _p(f), _p(fshift),
                                    C.cast(C.byref(vc), C.c_void_p), C.cast(C.byref(vv), C.c_void_p))
        return dict(f=f, fshift=fshift, Vc=vc.value, Vvdw=vv.value)

    def close(self):
        if self.ctx:
            reflib().ref_kernel_ref_free(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()


_simd = None


def have_ref_simd():
    return os.path.exists(os.path.join(_HERE, "_ref", "libnbnxn_ref_simd.so"))


def simdlib():
    global _simd
    if _simd is None:
        _simd = C.CDLL(os.path.join(_HERE, "_ref", "libnbnxn_ref_simd.so"))
        _simd.ref_simd_setup.restype = C.c_void_p
        _simd.ref_simd_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for fn in (_simd.ref_simd_free, _simd.ref_simd_nci, _simd.ref_simd_ncj, _simd.ref_simd_comb_rule):
            fn.argtypes = [C.c_void_p]
    return _simd


class RefKernelSimd4xN:
    """The reference's SIMD 4xN OpenMP kernel (AVX2_256: 4x8 cluster pairs, analytical Ewald or RF) on the simple list
    derived from the GPU list (oracle/ref_simd_harness.c).  The TIMING yardstick of bench.py's cpu_baseline; pins
    nothing.  nlist lists, one OpenMP thread each at most (nbnxn_kernel_simd_4xn.c:309-311)."""

    def __init__(self, xq, type_, ntype, nbfp, shift_vec, pl, ip, nlist=1):
        self.n = len(xq)
        p = ref_params(ip)
        nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
        self.nlist = nlist
        self.ctx = simdlib().ref_simd_setup(C.c_int(self.n), _p(xq), _p(type_), C.c_int(ntype), _p(nbfp), _p(shift_vec),
                                            C.c_int(pl.nsci), _p(pl.sci), _p(pl.cj4), _p(pl.excl), C.byref(p), C.c_int(nlist))
        if not self.ctx:
            raise RuntimeError("ref_simd_setup refused the input (atoms not a multiple of 8, LJ-PME, or nlist outside 1..32)")
        self.nci = simdlib().ref_simd_nci(self.ctx)
        self.ncj = simdlib().ref_simd_ncj(self.ctx)
        self.comb_rule = simdlib().ref_simd_comb_rule(self.ctx)

    def run(self, calc_ener=True, nthreads=1, outputs=True):
        if not outputs:
            simdlib().ref_simd_run(self.ctx, int(calc_ener), nthreads, None, None, None, None)
            return None
        f = np.zeros((self.n, 3), dtype=np.float32)
        fshift = np.zeros((45, 3), dtype=np.float32)
        vc, vv = C.c_float(0), C.c_float(0)
        simdlib().ref_simd_run(self.ctx, int(calc_ener), nthreads, _p(f), _p(fshift),
                               C.cast(C.byref(vc), C.c_void_p), C.cast(C.byref(vv), C.c_void_p))
        return dict(f=f, fshift=fshift, Vc=vc.value, Vvdw=vv.value)

    def close(self):
        if self.ctx:
            simdlib().ref_simd_free(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()
