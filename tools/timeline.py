"""Wave timeline of the force kernel (debug build: make -C gromacs-cosmo_amd/csrc timeline; run with
NBNXN_HIP_LIB=gromacs-cosmo_amd/lib/dbg/libnbnxn_hip_tl.so).  Prints how many waves are resident over the launch,
the ramp and the tail.  usage: NBNXN_HIP_LIB=... python tools/timeline.py [mcb]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi

nb = g.gpu.NbnxnHip(0)
mcb = int(sys.argv[1]) if len(sys.argv) > 1 else nb.min_ci_balanced()
ip = g.system.ewald(1.0)
s = g.system.water_box(5, 5, 6, model="spce")
p = g.workload.prepare(s, ip, mcb)
g.workload.upload(nb, p, ip)
flags = abi.FORCE_VIRIAL
nb.compute(p.xq, flags)
for _ in range(5):
    nb.clear_outputs(flags); nb.launch_kernel_resident(flags); nb.wait_gpu(flags)
n = p.pl.nsci
tl = np.zeros((n, 3), dtype=np.uint64)
nb.lib.nbnxn_hip_debug_timeline.argtypes = [C.c_void_p, C.c_int]
assert nb.lib.nbnxn_hip_debug_timeline(tl.ctypes.data_as(C.c_void_p), n) == 0
t0 = tl[:, 0].astype(np.int64); t1 = tl[:, 1].astype(np.int64)
ncj4 = (tl[:, 2] >> np.uint64(32)).astype(np.int64)
base = t0.min()
t0 = (t0 - base) * 0.01; t1 = (t1 - base) * 0.01      # us
dur = t1 - t0
print(f"nsci {n}  kernel span {t1.max():.1f} us (event timing {nb.last_kernel_ms() * 1e3:.1f} us)  sum of wave times {dur.sum():.0f} us"
      f"  = {dur.sum() / t1.max():.0f} waves resident on average of 4096 slots")
print(f"wave duration us: min {dur.min():.2f} median {np.median(dur):.2f} max {dur.max():.2f};  per cj4 group median {np.median(dur / np.maximum(ncj4, 1)):.2f}")
edges = np.linspace(0, t1.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = int(((t0 <= mid) & (t1 > mid)).sum())
    print(f"  t = {mid:6.1f} us: {res:5d} waves resident, {int(((t0 >= a) & (t0 < b)).sum()):5d} started")
last = np.sort(t1)[::-1]
print("last waves end at", np.round(last[:5], 1), " 99%/95%/90% of waves ended by", np.round(np.percentile(t1, [99, 95, 90]), 1))
print("first wave of the final 10% starts at", np.round(np.sort(t0)[int(0.9 * n)], 1))
