"""Synthetic workloads of BASELINE.json and the interaction constants the reference's
init_interaction_const (mdlib/forcerec.c:1976-2079) would hand to the kernels.

Everything here is deterministic and self-contained (no file outside the package):
the water boxes tile the 216-molecule SPC box shipped as data/spc216.npz.
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np

from . import abi

ONE_4PI_EPS0 = 332.0636930 * 4.184 * 0.1   # legacyheaders/physics.h:82-83 = 138.935485

# OPLS-AA SPC / SPC/E: share/top/oplsaa.ff/ffnonbonded.itp:127-128, spc.itp:7-9, spce.itp:7-9
SIGMA_OW, EPS_OW = 0.316557, 0.650194
Q_SPC = (-0.82, 0.41, 0.41)
Q_SPCE = (-0.8476, 0.4238, 0.4238)


@dataclass
class System:
    """Topology + coordinates in caller (molecule) order."""
    x: np.ndarray            # [n,3] float32, inside the box
    box: np.ndarray          # [3] rectangular, or the [3, 3] GROMACS matrix (rows are box vectors) of a triclinic cell
    type: np.ndarray         # [n] int32, 0..ntype-1 (without the dummy type)
    q: np.ndarray            # [n] float32
    ntype: int               # number of real types; the list format adds one dummy type
    c6: np.ndarray           # [ntype,ntype] C6
    c12: np.ndarray          # [ntype,ntype] C12
    excl_index: np.ndarray   # CSR exclusions, caller numbering (t_blocka)
    excl_a: np.ndarray
    name: str = ""

    @property
    def natoms(self):
        return len(self.x)

    @property
    def box_diag(self):
        """the brick the atoms live in: put_atoms_in_box leaves them in [0, box[d][d]) (gmxlib/pbc.c:1562)"""
        b = np.asarray(self.box, dtype=np.float32)
        return np.ascontiguousarray(np.diag(b)) if b.shape == (3, 3) else b

    def nbfp(self):
        """6*C6 / 12*C12 table including the zero dummy type: nbnxn_atomdata.c:561,611-617,636-638."""
        nt = self.ntype + 1
        t = np.zeros((nt, nt, 2), dtype=np.float32)
        t[:self.ntype, :self.ntype, 0] = 6.0 * self.c6
        t[:self.ntype, :self.ntype, 1] = 12.0 * self.c12
        return t


@dataclass
class InteractionParams:
    """The scalars of interaction_const_t this path reads (types/interaction_const.h:68-130)."""
    el_type: int
    rcoulomb: float
    rvdw: float
    rlist: float
    epsfac: float = ONE_4PI_EPS0
    epsilon_r: float = 1.0
    epsilon_rf: float = 0.0
    k_rf: float = 0.0
    c_rf: float = 0.0
    ewaldcoeff_q: float = 0.0
    sh_ewald: float = 0.0
    disp_cpot: float = 0.0
    rep_cpot: float = 0.0
    # LJ modifier (types/enums.h:92-93) and its constants (forcerec.c:1942-1975)
    vdw_modifier: int = abi.MOD_POTSHIFT
    rvdw_switch: float = 0.0
    disp_c2: float = 0.0
    disp_c3: float = 0.0
    rep_c2: float = 0.0
    rep_c3: float = 0.0
    sw_c3: float = 0.0
    sw_c4: float = 0.0
    sw_c5: float = 0.0
    # LJ-PME (ic->vdwtype == evdwPME): combination rule of the mesh part, its Ewald coefficient
    # and the potential shift of forcerec.c:2012-2018
    vdwtype: int = abi.VDW_CUT
    ljpme_comb_rule: int = abi.LJCR_GEOM
    ewaldcoeff_lj: float = 0.0
    sh_lj_ewald: float = 0.0

    def to_struct(self):
        ic = abi.InteractionConst()
        ic.vdwtype = self.vdwtype
        ic.ljpme_comb_rule = self.ljpme_comb_rule
        ic.ewaldcoeff_lj = self.ewaldcoeff_lj
        ic.sh_lj_ewald = self.sh_lj_ewald
        ic.vdw_modifier = self.vdw_modifier
        ic.rvdw = self.rvdw
        ic.rvdw_switch = self.rvdw_switch
        ic.dispersion_shift.c2, ic.dispersion_shift.c3 = self.disp_c2, self.disp_c3
        ic.repulsion_shift.c2, ic.repulsion_shift.c3 = self.rep_c2, self.rep_c3
        ic.vdw_switch.c3, ic.vdw_switch.c4, ic.vdw_switch.c5 = self.sw_c3, self.sw_c4, self.sw_c5
        ic.dispersion_shift.cpot = self.disp_cpot
        ic.repulsion_shift.cpot = self.rep_cpot
        ic.sh_invrc6 = -self.disp_cpot
        ic.eeltype = abi.EEL_RF if self.el_type == abi.EL_RF else abi.EEL_PME
        ic.coulomb_modifier = abi.MOD_POTSHIFT
        ic.rcoulomb = self.rcoulomb
        ic.rlist = self.rlist
        ic.rlistlong = self.rlist
        ic.ewaldcoeff_q = self.ewaldcoeff_q
        ic.sh_ewald = self.sh_ewald
        ic.epsilon_r = self.epsilon_r
        ic.epsfac = self.epsfac
        ic.epsilon_rf = self.epsilon_rf
        ic.k_rf = self.k_rf
        ic.c_rf = self.c_rf
        return ic


def _force_switch_constants(p, rsw, rc):
    """force_switch_constants, forcerec.c:1942-1957."""
    c2 = ((p + 1) * rsw - (p + 4) * rc) / (rc ** (p + 2) * (rc - rsw) ** 2)
    c3 = -((p + 1) * rsw - (p + 3) * rc) / (rc ** (p + 2) * (rc - rsw) ** 3)
    cpot = -rc ** -p + p * c2 / 3 * (rc - rsw) ** 3 + p * c3 / 4 * (rc - rsw) ** 4
    return c2, c3, cpot


def with_force_switch(ip, rvdw_switch):
    """vdw-modifier = force-switch (eintmodFORCESWITCH, forcerec.c:2020-2026)."""
    import dataclasses
    d2, d3, dp = _force_switch_constants(6.0, rvdw_switch, ip.rvdw)
    r2, r3, rp = _force_switch_constants(12.0, rvdw_switch, ip.rvdw)
    return dataclasses.replace(ip, vdw_modifier=abi.MOD_FORCESWITCH, rvdw_switch=rvdw_switch,
                               disp_c2=d2, disp_c3=d3, disp_cpot=dp, rep_c2=r2, rep_c3=r3, rep_cpot=rp)


def with_potential_switch(ip, rvdw_switch):
    """vdw-modifier = potential-switch (eintmodPOTSWITCH, forcerec.c:1959-1975, 2027-2031): no shift."""
    import dataclasses
    d = ip.rvdw - rvdw_switch
    return dataclasses.replace(ip, vdw_modifier=abi.MOD_POTSWITCH, rvdw_switch=rvdw_switch, disp_cpot=0.0, rep_cpot=0.0,
                               sw_c3=-10 * d ** -3, sw_c4=15 * d ** -4, sw_c5=-6 * d ** -5)


def calc_ewaldcoeff_lj(rc, rtol):
    """gmxlib/ewald_util.c:86-129: bisection on exp(-x^2)(1 + x^2 + x^4/2) = rtol, x = beta_lj*rc."""
    fn = lambda x: math.exp(-(x * rc) ** 2) * (1 + (x * rc) ** 2 + (x * rc) ** 4 / 2.0)
    x, i = 5.0, 0
    while True:
        i += 1
        x *= 2
        if fn(x) <= rtol:
            break
    lo, hi = 0.0, x
    for _ in range(i + 60):
        x = 0.5 * (lo + hi)
        if fn(x) > rtol:
            lo = x
        else:
            hi = x
    return x


def with_ljpme(ip, comb_rule=abi.LJCR_GEOM, rtol_lj=1e-3):
    """vdwtype = PME with potential shift (forcerec.c:2000-2018): ewaldcoeff_lj from ewald_rtol_lj
    (forcerec.c:2803), sh_lj_ewald = (exp(-x)(1 + x + x^2/2) - 1) / rvdw^6 with x = (beta_lj rvdw)^2."""
    import dataclasses
    b = calc_ewaldcoeff_lj(ip.rvdw, rtol_lj)
    crc2 = (b * ip.rvdw) ** 2
    sh = (math.exp(-crc2) * (1 + crc2 + 0.5 * crc2 * crc2) - 1) * ip.rvdw ** -6
    return dataclasses.replace(ip, vdwtype=abi.VDW_PME, ljpme_comb_rule=comb_rule, ewaldcoeff_lj=b, sh_lj_ewald=sh)


def nbfp_comb(nbfp, ntype, comb_rule):
    """Per-type combination parameters from the diagonal of the 6*C6 / 12*C12 table, as
    nbnxn_atomdata.c:397-429 fills nbat->nbfp_comb (float results of double arithmetic)."""
    t = np.asarray(nbfp, dtype=np.float32).reshape(ntype, ntype, 2)
    out = np.zeros((ntype, 2), dtype=np.float32)
    for i in range(ntype):
        c6, c12 = float(t[i, i, 0]), float(t[i, i, 1])
        if comb_rule == abi.LJCR_GEOM:
            out[i] = math.sqrt(c6), math.sqrt(c12)
        elif c6 > 0 and c12 > 0:
            out[i] = 0.5 * (c12 / c6) ** (1.0 / 6.0), math.sqrt(c6 * c6 / c12)
    return out


def with_tabulated_ewald(ip):
    """The reference's tabulated-force Ewald flavour (eelCuEWALD_TAB, GMX_CUDA_NB_TAB_EWALD)."""
    import dataclasses
    assert ip.el_type != abi.EL_RF
    return dataclasses.replace(ip, el_type=abi.EL_EWALD_TAB)


def calc_ewaldcoeff_q(rc, rtol):
    """gmxlib/ewald_util.c:55-84: bisection on erfc(beta*rc) = rtol."""
    x, i = 5.0, 0
    while True:
        i += 1
        x *= 2
        if math.erfc(x * rc) <= rtol:
            break
    lo, hi = 0.0, x
    for _ in range(i + 60):
        x = 0.5 * (lo + hi)
        if math.erfc(x * rc) > rtol:
            lo = x
        else:
            hi = x
    return x


def reaction_field(rc, rlist=None, epsilon_rf=78.0, epsilon_r=1.0):
    """calc_rffac (mdlib/rf_util.c:205-245, kappa = 0) + potential-shift constants
    (forcerec.c:2005-2011: cpot = -rc^-6 / -rc^-12)."""
    if epsilon_rf == 0:
        k_rf = 1.0 / (2 * rc ** 3)
    else:
        k_rf = (epsilon_rf - epsilon_r) / ((2 * epsilon_rf + epsilon_r) * rc ** 3)
    c_rf = 1.0 / rc + k_rf * rc * rc
    return InteractionParams(el_type=abi.EL_RF, rcoulomb=rc, rvdw=rc, rlist=rlist or rc,
                             epsfac=ONE_4PI_EPS0 / epsilon_r, epsilon_r=epsilon_r, epsilon_rf=epsilon_rf,
                             k_rf=k_rf, c_rf=c_rf, disp_cpot=-rc ** -6, rep_cpot=-rc ** -12)


def ewald(rc, rlist=None, rtol=1e-5, epsilon_r=1.0):
    """forcerec.c:2048-2053: beta from ewald_rtol; sh_ewald = erfc(beta*rc) exactly as the
    reference computes it (no 1/rc factor in 5.0.4; identical at rc = 1 nm)."""
    beta = calc_ewaldcoeff_q(rc, rtol)
    return InteractionParams(el_type=abi.EL_EWALD_ANA, rcoulomb=rc, rvdw=rc, rlist=rlist or rc,
                             epsfac=ONE_4PI_EPS0 / epsilon_r, epsilon_r=epsilon_r,
                             ewaldcoeff_q=beta, sh_ewald=math.erfc(beta * rc),
                             disp_cpot=-rc ** -6, rep_cpot=-rc ** -12)


def _spc216():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "spc216.npz"))
    return d["x"].astype(np.float64), d["box"].astype(np.float64)


def water_box(nx, ny, nz, model="spc"):
    """spc216 tiled nx*ny*nz times (the reference's `gmx genconf -nbox`): 648*nx*ny*nz atoms.
    5x1x1 = 3,240 (config 1), 5x5x6 = 97,200 (configs 2/3), 13x13x14 = 1,533,168 (config 4)."""
    x0, b0 = _spc216()
    shifts = np.array([[i, j, k] for i in range(nx) for j in range(ny) for k in range(nz)], dtype=np.float64) * b0
    x = (x0[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    box = b0 * np.array([nx, ny, nz])
    x = x - np.floor(x / box) * box          # atom-wise put_atoms_in_box
    x = x.astype(np.float32)
    boxf = box.astype(np.float32)
    x = np.where(x >= boxf, x - boxf, x)     # float rounding at the upper edge
    n = len(x)
    nmol = n // 3
    qs = Q_SPC if model == "spc" else Q_SPCE
    type_ = np.tile(np.array([0, 1, 1], dtype=np.int32), nmol)
    q = np.tile(np.array(qs, dtype=np.float32), nmol)
    c6 = np.zeros((2, 2))
    c12 = np.zeros((2, 2))
    c6[0, 0] = 4 * EPS_OW * SIGMA_OW ** 6
    c12[0, 0] = 4 * EPS_OW * SIGMA_OW ** 12
    # each atom excludes itself and the two others of its molecule (nrexcl 2 / [exclusions])
    excl_index = np.arange(0, 3 * n + 1, 3, dtype=np.int32)
    base = (np.arange(n, dtype=np.int32) // 3) * 3
    excl_a = (base[:, None] + np.arange(3, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
    return System(x=x, box=boxf, type=type_, q=q, ntype=2, c6=c6, c12=c12,
                  excl_index=excl_index, excl_a=excl_a, name=f"{model}216x{nx}x{ny}x{nz}")


def water_film(nx, ny, nz, liquid_fraction=0.4, vapour_keep=8, model="spc"):
    """A liquid film under thin vapour: water_box with, above z = liquid_fraction * box_z, only every
    `vapour_keep`-th molecule kept.  A density step along z: what an equal-thickness cut balances badly
    (the workload of the load-balancing tests and of bench.py --dd-balance)."""
    w = water_box(nx, ny, nz, model=model)
    nmol = w.natoms // 3
    z_o = w.x[0::3, 2]
    keep_mol = (z_o < liquid_fraction * float(w.box[2])) | (np.arange(nmol) % int(vapour_keep) == 0)
    keep = np.repeat(keep_mol, 3)
    n = int(keep.sum())
    excl_index = np.arange(0, 3 * n + 1, 3, dtype=np.int32)
    base = (np.arange(n, dtype=np.int32) // 3) * 3
    excl_a = (base[:, None] + np.arange(3, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
    return System(x=np.ascontiguousarray(w.x[keep]), box=w.box, type=np.ascontiguousarray(w.type[keep]), q=np.ascontiguousarray(w.q[keep]),
                  ntype=w.ntype, c6=w.c6, c12=w.c12, excl_index=excl_index, excl_a=excl_a, name=f"{w.name} film {liquid_fraction:g}/{vapour_keep}")


def put_atoms_in_box(x, box):
    """put_atoms_in_box (gmxlib/pbc.c:1562-1625) in float32: every atom shifted by whole box vectors, from z down
    to x, into the brick [0, box[d][d]).  box: [3] or the [3, 3] matrix (rows are box vectors)."""
    x = np.array(x, dtype=np.float32)
    b = np.asarray(box, dtype=np.float32)
    if b.shape == (3,):
        b = np.diag(b)
    for m in (2, 1, 0):
        while True:
            lo = x[:, m] < 0
            hi = x[:, m] >= b[m, m]
            if not (lo.any() or hi.any()):
                break
            x[lo, :m + 1] += b[m, :m + 1]
            x[hi, :m + 1] -= b[m, :m + 1]
    return x


def lj_chain_mixture(natoms, ntype=8, seed=2026, density=100.0, chain_len=(12, 40), skew=None):
    """Config-5-like synthetic system: an ntype-type LJ mixture of chain molecules with
    1-2 and 1-3 exclusions and partial charges, on a jittered lattice (no reference fixture
    exists for it).  Exclusion-dense, mixed LJ types.
    skew = (kyx, kzx, kzy): a triclinic cell whose off-diagonal elements box[YY][XX], box[ZZ][XX],
    box[ZZ][YY] are these multiples of the lattice spacing, so the lattice stays commensurate with its
    periodic images and no two atoms come closer than in the rectangular cell."""
    rng = np.random.default_rng(seed)
    vol = natoms / density
    L = vol ** (1.0 / 3.0)
    m = int(math.ceil(natoms ** (1.0 / 3.0)))
    a = L / m
    idx = np.arange(m ** 3)[:natoms]
    # serpentine order so that consecutive atoms (chain neighbours) are lattice neighbours
    iz = idx // (m * m)
    iy = (idx // m) % m
    ix = idx % m
    iy = np.where(iz % 2 == 1, m - 1 - iy, iy)
    ix = np.where((iy + iz * m) % 2 == 1, m - 1 - ix, ix)
    x = (np.stack([ix, iy, iz], axis=1) + 0.5) * a
    x = x + rng.uniform(-0.12 * a, 0.12 * a, size=x.shape)
    box = np.array([L, L, L], dtype=np.float32)
    x = (x - np.floor(x / L) * L).astype(np.float32)
    x = np.where(x >= box, x - box, x)
    type_ = rng.integers(0, ntype, size=natoms).astype(np.int32)
    sig = np.linspace(0.25, 0.36, ntype)
    eps = np.linspace(0.2, 0.9, ntype)
    s = 0.5 * (sig[:, None] + sig[None, :])
    e = np.sqrt(eps[:, None] * eps[None, :])
    c6 = 4 * e * s ** 6
    c12 = 4 * e * s ** 12
    q = rng.uniform(-0.4, 0.4, size=natoms)
    # chains: neutralise each chain
    lens = []
    tot = 0
    while tot < natoms:
        l = int(rng.integers(chain_len[0], chain_len[1] + 1))
        l = min(l, natoms - tot)
        lens.append(l)
        tot += l
    starts = np.concatenate([[0], np.cumsum(lens)])
    ex_lists = []
    for s0, l in zip(starts[:-1], lens):
        q[s0:s0 + l] -= q[s0:s0 + l].mean()
        for k in range(l):
            lo, hi = max(0, k - 2), min(l - 1, k + 2)
            ex_lists.append(np.arange(s0 + lo, s0 + hi + 1, dtype=np.int32))
    excl_index = np.concatenate([[0], np.cumsum([len(e_) for e_ in ex_lists])]).astype(np.int32)
    excl_a = np.concatenate(ex_lists).astype(np.int32)
    if skew is not None:
        box = np.diag(box).astype(np.float32)
        box[1, 0], box[2, 0], box[2, 1] = skew[0] * a, skew[1] * a, skew[2] * a
    return System(x=x, box=box, type=type_, q=q.astype(np.float32), ntype=ntype, c6=c6, c12=c12,
                  excl_index=excl_index, excl_a=excl_a, name=f"ljchain{natoms}" + ("" if skew is None else "_tric"))
