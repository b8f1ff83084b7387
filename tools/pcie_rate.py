"""PCIe-inclusive rate of the reference's calling convention (host xq in, f and energies out every
step: clear_outputs -> launch_kernel -> launch_cpyback -> wait_gpu) on the bench workload, with the
host arrays pageable (numpy) and pinned (nbnxn_hip_pmalloc, what the reference's nbat->alloc hook gives
nbat->x and out[0].f: forcerec.c:2247-2262)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi

ip = g.system.ewald(1.0)
s = g.system.water_box(5, 5, 6, model="spce")
nb = g.gpu.NbnxnHip(0)
p = g.workload.prepare(s, ip, nb.min_ci_balanced())
g.workload.upload(nb, p, ip)
fl = abi.FORCE_VIRIAL
nb.set_clear_in_kernel(os.environ.get("PCIE_CLEAR_IN_KERNEL", "1") != "0")


def pinned(shape, dtype=np.float32):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = C.c_void_p()
    nb.lib.nbnxn_hip_pmalloc(C.byref(ptr), n)
    assert ptr.value
    return np.frombuffer((C.c_char * n).from_address(ptr.value), dtype=dtype).reshape(shape)


def step(xq, f, e_lj, e_el, fsh):
    nb.clear_outputs(fl)
    nb.launch_kernel(xq, fl)
    nb.launch_cpyback(f, fl)
    nb.wait_gpu(fl, abi.LOCAL, e_lj, e_el, fsh)


for kind in ("pageable", "pinned"):
    mk = (lambda sh: np.zeros(sh, np.float32)) if kind == "pageable" else pinned
    xq = mk(p.xq.shape); xq[:] = p.xq
    f, e_lj, e_el, fsh = mk((p.natoms, 3)), mk((1,)), mk((1,)), mk((45, 3))
    for timing in (True, False):
        nb.set_timing(timing)
        for _ in range(5):
            step(xq, f, e_lj, e_el, fsh)
        n, t0 = 200, time.perf_counter()
        for _ in range(n):
            step(xq, f, e_lj, e_el, fsh)
        dt = (time.perf_counter() - t0) / n
        print(f"{kind:8s} host arrays, module event timing {str(timing):5s}: {dt * 1e3:.4f} ms per step incl. H2D xq + D2H f, "
              f"{p.pair_slots / dt / 1e9:.1f} Gpair/s", flush=True)
        if timing:
            t = nb.timings()
            c = max(t.nb_c, 1)
            print(f"         device-side event times per step: H2D {t.nb_h2d_t / c * 1e3:.1f} us, kernel {sum(t.ktime[a][b].t for a in range(2) for b in range(2)) / c * 1e3:.1f} us, "
                  f"D2H {t.nb_d2h_t / c * 1e3:.1f} us", flush=True)
            nb.reset_timings()
    # host time of each call (pinned arrays, timing off): where the step's wall time goes
    if kind == "pinned":
        import collections
        acc = collections.Counter()
        for _ in range(200):
            t0 = time.perf_counter(); nb.clear_outputs(fl)
            t1 = time.perf_counter(); nb.launch_kernel(xq, fl)
            t2 = time.perf_counter(); nb.launch_cpyback(f, fl)
            t3 = time.perf_counter(); nb.wait_gpu(fl, abi.LOCAL, e_lj, e_el, fsh)
            t4 = time.perf_counter()
            acc["clear_outputs"] += t1 - t0; acc["launch_kernel"] += t2 - t1; acc["launch_cpyback"] += t3 - t2; acc["wait_gpu"] += t4 - t3
        print("         host time per call (us): " + ", ".join(f"{k} {v / 200 * 1e6:.1f}" for k, v in acc.items()), flush=True)
