"""Generates the golden fixtures under tests/golden/ by running the REFERENCE's own kernels
(oracle/_ref = nbnxn_kernel_ref.c / nbnxn_kernel_gpu_ref.c compiled from /root/reference,
see oracle/Makefile) on inputs made by this repo's host producers.

Run in the build container (the reference tree is needed):  python tests/golden/make_golden.py
A fixture is data only: inputs (atom data, GPU-format list, parameters) and the reference's
outputs (forces, shift forces, energies).  The reference tree cannot travel to the GPU box,
these small files can.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gromacs_cosmo_amd as g  # noqa: E402
import oracle as O             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PFIELDS = ["el_type", "rcoulomb", "rvdw", "rlist", "epsfac", "k_rf", "c_rf", "ewaldcoeff_q", "sh_ewald",
           "disp_cpot", "rep_cpot"]


def params_array(ip):
    return np.array([getattr(ip, k) for k in PFIELDS], dtype=np.float64)


def small_case(name, system, ip, mcb=0):
    p = g.workload.prepare(system, ip, mcb)
    r_ref = O.RefKernelRef(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip).run(calc_ener=True)
    r_gpu = O.ref_kernel_gpu_ref(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip, calc_ener=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        params=params_array(ip), xq=p.xq, type=p.type, ntype=p.ntype, nbfp=p.nbfp, shift_vec=p.shift_vec,
        sci=p.pl.sci, cj4=p.pl.cj4, excl=p.pl.excl, nci_tot=p.pl.nci_tot,
        x=system.x, box=system.box, q=system.q, atype=system.type, excl_index=system.excl_index, excl_a=system.excl_a,
        ref_f=r_ref["f"], ref_fshift=r_ref["fshift"], ref_Vc=r_ref["Vc"], ref_Vvdw=r_ref["Vvdw"],
        gpuref_f=r_gpu["f"], gpuref_fshift=r_gpu["fshift"], gpuref_Vc=r_gpu["Vc"], gpuref_Vvdw=r_gpu["Vvdw"])
    print(f"{name}: natoms {len(p.xq)} nsci {p.pl.nsci} ncj4 {p.pl.ncj4} nexcl {p.pl.nexcl} "
          f"ref Vvdw {r_ref['Vvdw']:.6e} Vc {r_ref['Vc']:.6e}  gpu_ref Vvdw {r_gpu['Vvdw']:.6e} Vc {r_gpu['Vc']:.6e}")


XFIELDS = ["vdw_modifier", "rvdw_switch", "disp_c2", "disp_c3", "rep_c2", "rep_c3", "sw_c3", "sw_c4", "sw_c5",
           "vdwtype", "ljpme_comb_rule", "ewaldcoeff_lj", "sh_lj_ewald"]


def switch_case(name, system, ip, mcb=0):
    """LJ force / potential switch and LJ-PME: only nbnxn_kernel_ref has these flavours
    (nbnxn_kernel_gpu_ref is plain cut-off LJ), so the fixture holds its outputs alone."""
    p = g.workload.prepare(system, ip, mcb)
    r_ref = O.RefKernelRef(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip).run(calc_ener=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        params=params_array(ip), params_ext=np.array([getattr(ip, k) for k in XFIELDS], dtype=np.float64),
        xq=p.xq, type=p.type, ntype=p.ntype, nbfp=p.nbfp, shift_vec=p.shift_vec,
        sci=p.pl.sci, cj4=p.pl.cj4, excl=p.pl.excl, nci_tot=p.pl.nci_tot,
        ref_f=r_ref["f"], ref_fshift=r_ref["fshift"], ref_Vc=r_ref["Vc"], ref_Vvdw=r_ref["Vvdw"])
    print(f"{name}: natoms {len(p.xq)} nsci {p.pl.nsci} ncj4 {p.pl.ncj4} ref Vvdw {r_ref['Vvdw']:.6e} Vc {r_ref['Vc']:.6e}")


def large_case(name, tiles, model, ip, stride=97):
    """97k atoms: the inputs are regenerated from the seedless tiling recipe by the test; only
    the reference's outputs (energies, shift forces, every `stride`-th atom's force in grid
    order, force norm) are stored."""
    system = g.system.water_box(*tiles, model=model)
    p = g.workload.prepare(system, ip, 0)
    rk = O.RefKernelRef(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip, nlist=8)
    r = rk.run(calc_ener=True, nthreads=8)
    r_gpu = O.ref_kernel_gpu_ref(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip, calc_ener=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        params=params_array(ip), tiles=np.array(tiles), model=model, stride=stride,
        natoms_padded=len(p.xq), nsci=p.pl.nsci, ncj4=p.pl.ncj4, nexcl=p.pl.nexcl, nci_tot=p.pl.nci_tot,
        ref_f_sample=r["f"][::stride], ref_f_norm=float(np.sqrt((r["f"].astype(np.float64) ** 2).sum())),
        ref_fshift=r["fshift"], ref_Vc=r["Vc"], ref_Vvdw=r["Vvdw"],
        gpuref_f_sample=r_gpu["f"][::stride], gpuref_fshift=r_gpu["fshift"], gpuref_Vc=r_gpu["Vc"], gpuref_Vvdw=r_gpu["Vvdw"])
    print(f"{name}: natoms {len(p.xq)} nsci {p.pl.nsci} ncj4 {p.pl.ncj4} nci_tot {p.pl.nci_tot} "
          f"ref Vvdw {r['Vvdw']:.6e} Vc {r['Vc']:.6e} gpu_ref Vvdw {r_gpu['Vvdw']:.6e} Vc {r_gpu['Vc']:.6e}")


if __name__ == "__main__":
    assert O.have_ref(), "oracle/_ref is not built: make -C oracle"
    small_case("spc216_rf09", g.system.water_box(1, 1, 1), g.system.reaction_field(0.9))
    small_case("spc3240_rf09", g.system.water_box(5, 1, 1), g.system.reaction_field(0.9))
    small_case("spce3240_ewald09", g.system.water_box(5, 1, 1, model="spce"), g.system.ewald(0.9))
    small_case("ljchain2000_rf", g.system.lj_chain_mixture(2000, ntype=8, seed=2026), g.system.reaction_field(0.9, epsilon_rf=0.0))
    small_case("ljchain2000_ewald", g.system.lj_chain_mixture(2000, ntype=8, seed=2026), g.system.ewald(0.9), mcb=400)
    switch_case("spc648_rf09_fswitch07", g.system.water_box(1, 1, 1),
                g.system.with_force_switch(g.system.reaction_field(0.9), 0.7))
    switch_case("ljchain1000_ewald09_pswitch07", g.system.lj_chain_mixture(1000, ntype=8, seed=2026),
                g.system.with_potential_switch(g.system.ewald(0.9), 0.7))
    switch_case("ljchain1000_ewald09_fswitch07", g.system.lj_chain_mixture(1000, ntype=8, seed=2026),
                g.system.with_force_switch(g.system.ewald(0.9), 0.7), mcb=100)
    switch_case("ljchain1000_ewald09_ljpme_geom", g.system.lj_chain_mixture(1000, ntype=8, seed=2026),
                g.system.with_ljpme(g.system.ewald(0.9), g.abi.LJCR_GEOM))
    switch_case("ljchain1000_ewald09_ljpme_lb", g.system.lj_chain_mixture(1000, ntype=8, seed=2026),
                g.system.with_ljpme(g.system.ewald(0.9), g.abi.LJCR_LB), mcb=100)
    switch_case("spc648_rf09_ljpme_geom", g.system.water_box(1, 1, 1),
                g.system.with_ljpme(g.system.reaction_field(0.9), g.abi.LJCR_GEOM))
    # two distinct, mutually excluded atoms on one point (r = 0): the reference's r^2 increment at work
    from util import coincident_excluded_system
    co, _ = coincident_excluded_system()
    small_case("ljchain1000_coincident_rf09", co, g.system.reaction_field(0.9), mcb=100)
    small_case("ljchain1000_coincident_ewald09", co, g.system.ewald(0.9), mcb=100)
    switch_case("ljchain1000_coincident_ewald09_fswitch07", co, g.system.with_force_switch(g.system.ewald(0.9), 0.7), mcb=100)
    switch_case("ljchain1000_coincident_ewald09_pswitch07", co, g.system.with_potential_switch(g.system.ewald(0.9), 0.7), mcb=100)
    switch_case("ljchain1000_coincident_ewald09_ljpme_geom", co, g.system.with_ljpme(g.system.ewald(0.9), g.abi.LJCR_GEOM), mcb=100)
    switch_case("ljchain1000_coincident_rf09_ljpme_lb", co, g.system.with_ljpme(g.system.reaction_field(0.9), g.abi.LJCR_LB), mcb=100)
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        sys.exit(0)
    large_case("spc97k_rf10", (5, 5, 6), "spc", g.system.reaction_field(1.0))
    large_case("spce97k_ewald10", (5, 5, 6), "spce", g.system.ewald(1.0))
