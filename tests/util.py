"""Shared helpers of the test-suite."""
import os
from types import SimpleNamespace

import numpy as np

import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
from gromacs_cosmo_amd.host import PairList
from gromacs_cosmo_amd.system import InteractionParams

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = ["spc216_rf09", "spc3240_rf09", "spce3240_ewald09", "ljchain2000_rf", "ljchain2000_ewald"]
LARGE = ["spc97k_rf10", "spce97k_ewald10"]
PFIELDS = ["el_type", "rcoulomb", "rvdw", "rlist", "epsfac", "k_rf", "c_rf", "ewaldcoeff_q", "sh_ewald",
           "disp_cpot", "rep_cpot"]

# Tolerances of the north_star: forces within 1e-5 relative RMS, energies/virial within 1e-5 relative.
FORCE_RTOL = 1e-5
ENERGY_RTOL = 1e-5


def relrms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


XFIELDS = ["vdw_modifier", "rvdw_switch", "disp_c2", "disp_c3", "rep_c2", "rep_c3", "sw_c3", "sw_c4", "sw_c5",
           "vdwtype", "ljpme_comb_rule", "ewaldcoeff_lj", "sh_lj_ewald"]     # older fixtures stop at sw_c5
SWITCH = ["spc648_rf09_fswitch07", "ljchain1000_ewald09_pswitch07", "ljchain1000_ewald09_fswitch07"]
LJPME = ["ljchain1000_ewald09_ljpme_geom", "ljchain1000_ewald09_ljpme_lb", "spc648_rf09_ljpme_geom"]
# two distinct, mutually excluded atoms on one point (coincident_excluded_system below); the first two also hold
# nbnxn_kernel_gpu_ref outputs
COINCIDENT = ["ljchain1000_coincident_rf09", "ljchain1000_coincident_ewald09", "ljchain1000_coincident_ewald09_fswitch07",
              "ljchain1000_coincident_ewald09_pswitch07", "ljchain1000_coincident_ewald09_ljpme_geom",
              "ljchain1000_coincident_rf09_ljpme_lb"]


def params_from_array(arr, ext=None):
    kw = {k: float(v) for k, v in zip(PFIELDS, arr)}
    kw["el_type"] = int(kw["el_type"])
    if ext is not None:
        kw.update({k: float(v) for k, v in zip(XFIELDS, ext)})
        for k in ("vdw_modifier", "vdwtype", "ljpme_comb_rule"):
            if k in kw:
                kw[k] = int(kw[k])
    return InteractionParams(**kw)


def load_small(name):
    """A golden fixture as (prepared-like namespace, ip, dict of reference outputs)."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    ip = params_from_array(d["params"], d["params_ext"] if "params_ext" in d.files else None)
    pl = PairList(d["sci"].copy(), d["cj4"].copy(), d["excl"].copy(), int(d["nci_tot"]), ip.rlist)
    p = SimpleNamespace(xq=d["xq"].copy(), type=d["type"].copy(), ntype=int(d["ntype"]), nbfp=d["nbfp"].copy(),
                        shift_vec=d["shift_vec"].copy(), pl=pl, natoms=len(d["xq"]))
    return p, ip, d


def load_large(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    ip = params_from_array(d["params"])
    system = g.system.water_box(*[int(t) for t in d["tiles"]], model=str(d["model"]))
    return system, ip, d


def virial_from(xq, f, shift_vec, fshift):
    """-0.5 * (sum x (x) f + sum shift_vec (x) fshift) in double (mdlib/calcvir.c:55-93)."""
    x = xq[:, :3].astype(np.float64)
    v = -0.5 * (x.T @ np.asarray(f, dtype=np.float64))
    v += -0.5 * (shift_vec.astype(np.float64).T @ np.asarray(fshift, dtype=np.float64))
    return v


def brute_force_sample(system, ip, idx):
    """Forces on the atoms `idx` of a System in double, from ALL atoms by minimum image: the
    list-independent truth for a sample of a box too large for the full oracle (pair arithmetic
    of nbnxn_kernel_ref_inner.h:117-372 with analytical Ewald; exclusions from the topology CSR)."""
    from scipy.special import erf
    x = system.x.astype(np.float64)
    box = system.box.astype(np.float64)
    q = system.q.astype(np.float64)
    out = np.zeros((len(idx), 3))
    rc2 = ip.rcoulomb ** 2
    for n, i in enumerate(idx):
        d = x[i] - x
        d -= box * np.floor(d / box + 0.5)
        r2 = (d * d).sum(1)
        near = np.nonzero(r2 < rc2)[0]
        near = near[near != i]
        d, r2 = d[near], r2[near]
        bit = (~np.isin(near, system.excl_a[system.excl_index[i]:system.excl_index[i + 1]])).astype(np.float64)
        r = np.sqrt(r2)
        qq = ip.epsfac * q[i] * q[near]
        c6 = 6.0 * system.c6[system.type[i], system.type[near]]
        c12 = 12.0 * system.c12[system.type[i], system.type[near]]
        fs = bit * (c12 / r2 ** 6 - c6 / r2 ** 3) / r2
        if ip.el_type == abi.EL_RF:
            fs += qq * (bit / (r2 * r) - 2.0 * ip.k_rf)
        else:
            b = ip.ewaldcoeff_q
            fs += qq * (bit / (r2 * r) - (erf(b * r) / r - 2.0 * b / np.sqrt(np.pi) * np.exp(-b * b * r2)) / r2)
        out[n] = (fs[:, None] * d).sum(0)
    return out


def coincident_excluded_system(natoms=1000, every=23, seed=2026):
    """The 8-type chain mixture with every `every`-th bonded (1-2, mutually excluded) pair of atoms put on ONE point:
    what shell / Drude particles and virtual sites on atoms look like to the kernels.  r = 0 between two DISTINCT
    atoms is what the reference's r^2 += (1 - int_bit) * NBNXN_AVOID_SING_R2_INC exists for (nbnxn_consts.h:66-79).
    Returns the System and the list of (k, k + 1) caller-order pairs."""
    s = g.system.lj_chain_mixture(natoms, ntype=8, seed=seed)
    x = s.x.copy()
    pairs = []
    for k in range(5, natoms - 1, every):
        if k + 1 in s.excl_a[s.excl_index[k]:s.excl_index[k + 1]]:
            x[k + 1] = x[k]
            pairs.append((k, k + 1))
    s.x = x
    s.name = f"ljchain{natoms}_coincident"
    return s, pairs


def ljpme_grid_c6(nbfp, ntype, comb_rule, ti, tj):
    """grid C6 of a type pair as the kernels combine it (nbnxn_kernel_ref_inner.h:211-226)"""
    c = g.system.nbfp_comb(nbfp, ntype, comb_rule).astype(np.float64)
    if comb_rule == abi.LJCR_GEOM:
        return c[ti, 0] * c[tj, 0]
    sigma, eps = c[ti, 0] + c[tj, 0], c[ti, 1] * c[tj, 1]
    return eps * sigma ** 6


def write_step_file(path, p, ip):
    """The input of `oracle/_ref/binding_check <in> <out>` (oracle/binding_check.c, step_header_t): one prepared
    system and its list as raw arrays, for the run of a step through the reference's own entry-point names."""
    f32 = lambda v: np.float32(v)
    head_i = np.zeros(16, dtype=np.int32)
    head_i[:10] = [0x4e42584e, p.natoms, p.ntype, p.pl.nsci, p.pl.ncj4, p.pl.nexcl, p.pl.nci_tot,
                   abi.EEL_RF if ip.el_type == abi.EL_RF else abi.EEL_PME, int(getattr(ip, "vdwtype", 0)), int(getattr(ip, "vdw_modifier", 1))]
    head_f = np.zeros(16, dtype=np.float32)
    head_f[:12] = [f32(ip.rcoulomb), f32(ip.rvdw), f32(ip.rlist), f32(ip.epsfac), f32(ip.k_rf), f32(ip.c_rf), f32(ip.ewaldcoeff_q),
                   f32(ip.sh_ewald), f32(ip.disp_cpot), f32(ip.rep_cpot), f32(ip.epsilon_r), f32(ip.epsilon_rf)]
    with open(path, "wb") as fh:
        fh.write(head_i.tobytes())
        fh.write(head_f.tobytes())
        for a, dt in ((p.xq, np.float32), (p.type, np.int32), (p.nbfp, np.float32), (p.shift_vec, np.float32)):
            fh.write(np.ascontiguousarray(a, dtype=dt).tobytes())
        fh.write(p.pl.sci[:p.pl.nsci].tobytes())
        fh.write(p.pl.cj4[:p.pl.ncj4].tobytes())
        fh.write(p.pl.excl[:p.pl.nexcl].tobytes())


def read_step_result(path, natoms):
    raw = open(path, "rb").read()
    info = np.frombuffer(raw, dtype=np.int32, count=8)
    e = np.frombuffer(raw, dtype=np.float32, count=2, offset=32)
    fshift = np.frombuffer(raw, dtype=np.float32, count=135, offset=40).reshape(45, 3)
    f1 = np.frombuffer(raw, dtype=np.float32, count=natoms * 3, offset=40 + 540).reshape(natoms, 3)
    f2 = np.frombuffer(raw, dtype=np.float32, count=natoms * 3, offset=40 + 540 + natoms * 12).reshape(natoms, 3)
    return SimpleNamespace(min_ci_balanced=int(info[0]), analytical=int(info[1]), nb_c=int(info[2]), c_prune_ener=int(info[3]),
                           c_plain=int(info[4]), device=int(info[5]), ndev=int(info[6]), dev_info_size=int(info[7]),
                           e_lj=float(e[0]), e_el=float(e[1]), fshift=fshift, f1=f1, f2=f2)
