"""Dynamic pruning at 97 k atoms: force-kernel time on the outer list (rlist 1.1, pruned once by the kernel,
the 5.0.4 way) against the inner list (re-pruned to 1.02 by the prune-only kernel), and the cost of that
kernel (whole list / one of 10 slices).  usage: python tools/prune_bench.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi

nb = g.gpu.NbnxnHip(0)
s = g.system.water_box(5, 5, 6, model="spce")
ip = g.system.ewald(1.0, rlist=1.1)
p = g.workload.prepare(s, ip, nb.min_ci_balanced())
g.workload.upload(nb, p, ip)


def ktime(flags, n=25):
    ts = []
    for _ in range(n):
        nb.clear_outputs(flags); nb.launch_kernel_resident(flags); nb.wait_gpu(flags)
        ts.append(nb.last_kernel_ms() * 1e3)
    return float(np.median(ts[5:]))


nb.compute(p.xq, 0)                     # prunes at 1.1
t_outer = ktime(0)
nb.prune_only(1.02); torch.cuda.synchronize()      # first call on the list: allocates the outer masks
t0 = time.perf_counter()
for _ in range(20):
    nb.prune_only(1.02, 1)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / 20 * 1e6
t_inner = ktime(0)
t0 = time.perf_counter()
for _ in range(50):
    nb.prune_only(1.02, 10)
torch.cuda.synchronize()
t_part = (time.perf_counter() - t0) / 50 * 1e6
cj4 = nb.download_cj4(p.pl)["imei"]["imask"][:, 0]
print(f"cluster pairs: as built {p.pl.nci_tot}, inner list {int(sum(bin(int(m)).count('1') for m in cj4))}")
print(f"force kernel (Ewald F): outer list {t_outer:.1f} us, inner list {t_inner:.1f} us; prune-only whole list {t_full:.1f} us/call, 1/10 slice {t_part:.1f} us/call (host-timed, back to back)")
