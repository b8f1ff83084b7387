"""Kernel A/B bench: times the resident kernel for RF-F and Ewald-VF on the 97k box and
checks parity against the oracle (which is why it lives under tests/: only tests, smoke() and
bench.py's CPU baseline touch oracle/).  Not collected by pytest.  usage: python tests/dev_kbench.py [variant ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
import oracle as O

variants = [0]
mcb = int(os.environ.get("MCB", "-1"))
nb = g.gpu.NbnxnHip(0)
if mcb < 0:
    mcb = nb.min_ci_balanced()
tag = os.path.basename(os.environ.get("NBNXN_HIP_LIB", "default"))
for name, ip, model, flags in (("rf-F", g.system.reaction_field(1.0), "spc", 0), ("ewald-VF", g.system.ewald(1.0), "spce", abi.FORCE_VIRIAL),
                               ("ewald-F", g.system.ewald(1.0), "spce", 0)):
    s = g.system.water_box(5, 5, 6, model=model)
    p = g.workload.prepare(s, ip, mcb)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    for v in variants:
        nb.set_kernel_variant(v)
        g.workload.upload(nb, p, ip)
        f, e = nb.compute(p.xq, flags)     # prune step
        err = float(np.sqrt(((f - o["f"]) ** 2).sum() / (o["f"] ** 2).sum()))
        ts = []
        for _ in range(30):
            nb.clear_outputs(flags); nb.launch_kernel_resident(flags); nb.wait_gpu(flags)
            ts.append(nb.last_kernel_ms() * 1e3)
        ts = np.array(ts[5:])
        de = abs(e[1] - o["Vc"]) / abs(o["Vc"]) if flags else 0.0
        print(f"{tag} v{v} {name} nsci {p.pl.nsci}: relrms {err:.2e} dEel {de:.1e}  kernel us min {ts.min():.1f} med {np.median(ts):.1f}"
              f"  -> {p.pair_slots / (np.median(ts) * 1e-6) / 1e9:.0f} Gpair/s", flush=True)
