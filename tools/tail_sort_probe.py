"""What the order of the entries WITHIN a size class is worth: the list of the bench workload as the builder leaves it
(descending group count, stable) against the same entries with a second sort key -- the live cluster pairs of an entry,
descending -- taken from the unpruned masks (what a builder could use) or from the pruned ones (the ceiling), and with
the whole list sorted by live pairs alone.  Same process, same handle, blocks of launches alternating.
usage: python tools/tail_sort_probe.py [ewald-vf|rf-f ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
from gromacs_cosmo_amd.host import PairList

POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int64)


def live_pairs(pl, cj4):
    im = (cj4["imei"]["imask"][:, 0] | cj4["imei"]["imask"][:, 1]).astype(np.uint32)
    per_group = POP[im & 0xff] + POP[(im >> 8) & 0xff] + POP[(im >> 16) & 0xff] + POP[(im >> 24) & 0xff]
    cs = np.concatenate([[0], np.cumsum(per_group)])
    return cs[pl.sci["cj4_ind_end"].astype(np.int64)] - cs[pl.sci["cj4_ind_start"].astype(np.int64)]


def reordered(pl, order):
    return PairList(np.ascontiguousarray(pl.sci[order]), pl.cj4, pl.excl, pl.nci_tot, pl.rlist)


for case in sys.argv[1:] or ["ewald-vf", "rf-f"]:
    el, fl = case.split("-")
    ip = g.system.ewald(1.0) if el == "ewald" else g.system.reaction_field(1.0)
    flags = abi.FORCE_VIRIAL if fl == "vf" else 0
    s = g.system.water_box(5, 5, 6, model="spce" if el == "ewald" else "spc")
    nb = g.gpu.NbnxnHip(0)
    nb.set_timing(False); nb.set_clear_in_kernel(True)
    p = g.workload.prepare(s, ip, nb.min_ci_balanced())
    g.workload.upload(nb, p, ip)
    f0, e0 = nb.compute(p.xq, abi.FORCE_VIRIAL)           # prunes
    pruned = nb.download_cj4(p.pl)
    ng = (p.pl.sci["cj4_ind_end"] - p.pl.sci["cj4_ind_start"]).astype(np.int64)
    m = 2 * p.pl.ncj4 // p.pl.nsci
    key = np.minimum(ng, m)
    lp_raw, lp_pruned = live_pairs(p.pl, p.pl.cj4), live_pairs(p.pl, pruned)
    n = p.pl.nsci
    im = (p.pl.cj4["imei"]["imask"][:, 0] | p.pl.cj4["imei"]["imask"][:, 1]).astype(np.uint32)
    jl = ((im & 0xff) != 0).astype(np.int64) + ((im & 0xff00) != 0) + ((im & 0xff0000) != 0) + ((im & 0xff000000) != 0)
    cs = np.concatenate([[0], np.cumsum(jl)])
    jcl = cs[p.pl.sci["cj4_ind_end"].astype(np.int64)] - cs[p.pl.sci["cj4_ind_start"].astype(np.int64)]
    variants = {
        "builder order": np.arange(n),
        "groups, then unpruned pairs": np.lexsort((-lp_raw, -key)),
        "groups, then pairs + j-clusters": np.lexsort((-(lp_raw + jcl), -key)),
        "groups, then pairs + 2 j-clusters": np.lexsort((-(lp_raw + 2 * jcl), -key)),
        "groups, then j-clusters": np.lexsort((-jcl, -key)),
    }
    for thr in (3, 4, 6):
        variants[f"groups, then pairs (entries <= {thr} groups only)"] = np.lexsort((-np.where(ng <= thr, lp_raw, 0), -key))
    res = {k: [] for k in variants}
    for rep in range(6):
        for name, order in variants.items():
            nb.init_pairlist(reordered(p.pl, order), abi.LOCAL)
            f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)      # prunes this copy
            assert np.abs(f - f0).max() < 1e-3 * np.abs(f0).max()
            for _ in range(20):
                nb.clear_outputs(flags); nb.launch_kernel_resident(flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                nb.clear_outputs(flags); nb.launch_kernel_resident(flags)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 300 * 1e6)
    base = np.median(res["builder order"][1:])
    for name, v in res.items():
        print(f"{case}: {name:48s} {np.median(v[1:]):7.2f} us  ({np.median(v[1:]) / base:.4f})", flush=True)
    nb.close()
