"""Pair search: device (nbnxn_hip_put_on_grid + nbnxn_hip_make_pairlist) against the host-side
builder on this machine's cores, 97 k and 1.53 M atoms.  usage: python tools/search_bench.py [tiles...]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi

nb = g.gpu.NbnxnHip(0)
mcb = nb.min_ci_balanced()
cases = [(5, 5, 6), (13, 13, 14)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for tiles in cases:
    s = g.system.water_box(*tiles)
    ip = g.system.reaction_field(1.0)
    nb.init_const(ip, s.ntype + 1, s.nbfp())
    t0 = time.perf_counter()
    p = g.workload.prepare(s, ip, mcb)
    t_host = time.perf_counter() - t0
    xd = torch.from_numpy(s.x).cuda()
    tg, tl = [], []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb.lib.nbnxn_hip_put_on_grid(nb.h, abi.ptr(nb._box_matrix(s.box)), None, s.natoms,
                                     xd.data_ptr(), 1, abi.ptr(s.type) if it == 0 else None, abi.ptr(s.q) if it == 0 else None, 0.0)
        t1 = time.perf_counter()
        if it == 0:
            nb.make_pairlist(ip.rlist, s.excl_index, s.excl_a, mcb)
        else:
            nb.make_pairlist(ip.rlist, None, None, mcb)
        t2 = time.perf_counter()
        tg.append(t1 - t0); tl.append(t2 - t1)
    nb.natoms, nb.natoms_real = p.natoms, s.natoms
    dev = nb.download_pairlist()
    same = (dev.nsci, dev.ncj4, dev.nexcl, dev.nci_tot) == (p.pl.nsci, p.pl.ncj4, p.pl.nexcl, p.pl.nci_tot) and np.array_equal(dev.sci, p.pl.sci)
    print(f"{s.natoms} atoms: host grid+list {t_host * 1e3:.1f} ms | device grid {np.median(tg[1:]) * 1e3:.2f} ms (x, types, charges resident) "
          f"+ list {np.median(tl[1:]) * 1e3:.2f} ms (exclusions resident) = {(np.median(tg[1:]) + np.median(tl[1:])) * 1e3:.2f} ms"
          f" | nsci {dev.nsci} ncj4 {dev.ncj4} nexcl {dev.nexcl} same-as-host {same}", flush=True)
