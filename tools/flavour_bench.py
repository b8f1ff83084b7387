"""Kernel time of every flavour axis on the 97k-atom water box (event timing of the module,
resident coordinates): electrostatics x LJ treatment, F and V+F.  usage: python tools/flavour_bench.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi

nb = g.gpu.NbnxnHip(0)
mcb = nb.min_ci_balanced()
S = g.system
rows = []
for el_name, base, model in (("rf", S.reaction_field(1.0), "spc"), ("ewald-ana", S.ewald(1.0), "spce"),
                             ("ewald-tab", S.with_tabulated_ewald(S.ewald(1.0)), "spce")):
    s = S.water_box(5, 5, 6, model=model)
    p = g.workload.prepare(s, base, mcb)
    for lj_name, ip in (("cut", base), ("fswitch", S.with_force_switch(base, 0.8)), ("pswitch", S.with_potential_switch(base, 0.8)),
                        ("ljpme", S.with_ljpme(base, abi.LJCR_GEOM))):
        g.workload.upload(nb, p, ip)
        nb.compute(p.xq, abi.FORCE_VIRIAL)     # prune step
        out = []
        for flags in (0, abi.FORCE_VIRIAL):
            ts = []
            for _ in range(25):
                nb.clear_outputs(flags); nb.launch_kernel_resident(flags); nb.wait_gpu(flags)
                ts.append(nb.last_kernel_ms() * 1e3)
            out.append(float(np.median(ts[5:])))
        print(f"{el_name:10s} {lj_name:8s} F {out[0]:6.1f} us  V+F {out[1]:6.1f} us   ({p.pair_slots / out[0] / 1e3:.0f} / {p.pair_slots / out[1] / 1e3:.0f} Gpair/s)", flush=True)
