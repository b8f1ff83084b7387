"""First GPU bring-up check: HIP path vs oracle on small and 97k water, RF and Ewald, F and VF, prune."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
import oracle as O

def rr(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))

nb = g.gpu.NbnxnHip(0)
for tiles, mcb in (((1, 1, 1), 0), ((5, 1, 1), 0), ((5, 5, 6), 0), ((5, 5, 6), 10240)):
    for name, ip in (("rf", g.system.reaction_field(0.9 if min(tiles) == 1 else 1.0)), ("ew", g.system.ewald(0.9 if min(tiles) == 1 else 1.0))):
        s = g.system.water_box(*tiles)
        p = g.workload.prepare(s, ip, mcb)
        g.workload.upload(nb, p, ip)
        assert nb.verify_pairlist(p.pl) == 0
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
        # energy + prune step
        f, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
        t_prune = nb.last_kernel_ms()
        cj4_dev = nb.download_cj4(p.pl)
        cj4_ref = O.prune_list(p.xq, p.shift_vec, p.pl, ip.rlist)
        nd = int((cj4_dev["imei"]["imask"] != cj4_ref["imei"]["imask"]).sum())
        fsh_o = o["fshift"].copy(); fsh_o[abi.CENTRAL] = 0
        print(f"{s.name} {name} mcb {mcb} nsci {p.pl.nsci} VF+prune: relrms {rr(f, o['f']):.2e} "
              f"Elj {elj:.6e}/{o['Vvdw']:.6e} Eel {eel:.6e}/{o['Vc']:.6e} fshift {rr(fsh, fsh_o):.2e} prune-diff {nd} t {t_prune*1e3:.1f} us")
        # force-only on the pruned list
        f2, _ = nb.compute(p.xq, 0)
        t_f = nb.last_kernel_ms()
        f3, (elj3, eel3, _) = nb.compute(p.xq, abi.FORCE_VIRIAL)
        t_vf = nb.last_kernel_ms()
        ts = []
        for _ in range(5):
            nb.compute(p.xq, 0); ts.append(nb.last_kernel_ms())
        print(f"    F: relrms {rr(f2, o['f']):.2e} t {t_f*1e3:.1f} us ; VF noprune relrms {rr(f3, o['f']):.2e} dE {abs(elj3-o['Vvdw'])/abs(o['Vvdw']):.1e} {abs(eel3-o['Vc'])/abs(o['Vc']):.1e} t {t_vf*1e3:.1f} us; F x5 {[round(t*1e3,1) for t in ts]}"
              f"  -> {p.pair_slots/ (min(ts)*1e-3)/1e9:.1f} Gpair/s")
