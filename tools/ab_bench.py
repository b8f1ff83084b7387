"""Same-process A/B of two builds of libnbnxn_hip.so on the 97,200-atom box: the two libraries are loaded side by
side, the same list is installed in a handle of each, and blocks of back-to-back launches alternate between them, so
that clock and box differences cancel.  usage: python tools/ab_bench.py libA.so libB.so [ewald-vf|ewald-f|rf-f|rf-vf ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi, gpu


def handle(path):
    os.environ["NBNXN_HIP_LIB"] = os.path.abspath(path)
    gpu._lib = None
    nb = gpu.NbnxnHip(0)
    nb.set_timing(False)
    nb.set_clear_in_kernel(True)
    return nb


libs = sys.argv[1:3]
cases = sys.argv[3:] or ["ewald-vf", "rf-f"]
import torch
hs = [handle(p) for p in libs]
for case in cases:
    el, fl = case.split("-")
    ip = g.system.ewald(1.0) if el == "ewald" else g.system.reaction_field(1.0)
    flags = abi.FORCE_VIRIAL if fl == "vf" else 0
    tiles = tuple(int(v) for v in os.environ.get("TILES", "5,5,6").split(","))
    s = g.system.water_box(*tiles, model="spce" if el == "ewald" else "spc")
    p = g.workload.prepare(s, ip, hs[0].min_ci_balanced())
    for nb in hs:
        g.workload.upload(nb, p, ip)
        nb.compute(p.xq, abi.FORCE_VIRIAL)
    res = [[], []]
    for rep in range(12):
        for k, nb in enumerate(hs):
            for _ in range(10):
                nb.clear_outputs(flags); nb.launch_kernel_resident(flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(int(os.environ.get("AB_STEPS", "200"))):
                nb.clear_outputs(flags); nb.launch_kernel_resident(flags)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / int(os.environ.get("AB_STEPS", "200")) * 1e6)
    a, b = np.array(res[0][2:]), np.array(res[1][2:])
    print(f"{case}: A {os.path.basename(libs[0])} {np.median(a):.2f} us (min {a.min():.2f})   B {os.path.basename(libs[1])} {np.median(b):.2f} us (min {b.min():.2f})"
          f"   B/A {np.median(b) / np.median(a):.4f}", flush=True)
