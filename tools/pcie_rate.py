"""PCIe-inclusive rate of the reference's calling convention (host xq in, f and energies out every
step: clear_outputs -> launch_kernel -> launch_cpyback -> wait_gpu) on the bench workload."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
ip = g.system.ewald(1.0); s = g.system.water_box(5, 5, 6, model="spce")
nb = g.gpu.NbnxnHip(0); p = g.workload.prepare(s, ip, nb.min_ci_balanced()); g.workload.upload(nb, p, ip)
fl = abi.FORCE_VIRIAL
for timing in (True, False):
    nb.set_timing(timing)
    for _ in range(5):
        nb.compute(p.xq, fl)
    n, t0 = 100, time.perf_counter()
    for _ in range(n):
        nb.compute(p.xq, fl)
    dt = (time.perf_counter() - t0) / n
    print(f"module event timing {timing}: {dt * 1e3:.4f} ms per step incl. H2D xq + D2H f (pageable numpy buffers), "
          f"{p.pair_slots / dt / 1e9:.1f} Gpair/s", flush=True)
