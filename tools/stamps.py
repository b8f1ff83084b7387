"""Diagnostic: wave lifetimes / placement of the v2 F kernel (needs the NBX_STAMPS build)."""
import sys, os, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
nb = g.gpu.NbnxnHip(0)
ip = g.system.reaction_field(1.0)
s = g.system.water_box(5, 5, 6)
p = g.workload.prepare(s, ip, int(os.environ.get("MCB", nb.min_ci_balanced())))
g.workload.upload(nb, p, ip)
nb.compute(p.xq, 0)
for rep in range(3):
    nb.clear_outputs(0); nb.launch_kernel_resident(0); nb.wait_gpu(0)
n = p.pl.nsci
raw = np.zeros(n * 10, dtype=np.uint32)
nb.lib.nbnxn_hip_debug_read_partial.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
assert nb.lib.nbnxn_hip_debug_read_partial(nb.h, 0, abi.ptr(raw), raw.nbytes) == 0
buf = raw[:n * 8].view(np.uint64).reshape(n, 4).copy()
extra = raw[n * 8:].reshape(n, 2)
w3 = buf[:, 3].copy()
buf[:, 3] = w3 & 0xffff
pro = ((w3 >> 16) & 0xfffff).astype(np.int64); loop = ((w3 >> 36) & 0xffffff).astype(np.int64)
wait = extra[:, 0].astype(np.int64); epi = extra[:, 1].astype(np.int64)
print("cycles per wave (s_memtime): prologue mean %.0f, loop mean %.0f of which exposed memory wait %.0f, epilogue %.0f" % (pro.mean(), loop.mean(), wait.mean(), epi.mean()))
nc = (w3 & 0xffff).astype(int)
for k in sorted(set(nc)):
    m = nc == k
    if m.sum() > 100:
        print(f"   ncj4 {k}: prologue {pro[m].mean():.0f} loop {loop[m].mean():.0f} wait {wait[m].mean():.0f} epilogue {epi[m].mean():.0f}")
t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64)
base = t0.min()
t0 = (t0 - base) * 0.01; t1 = (t1 - base) * 0.01   # us (100 MHz)
dur = t1 - t0
hw = buf[:, 2] & 0xffffffff; xcc = buf[:, 2] >> 32
ncj4 = buf[:, 3].astype(int)
print("kernel ms", nb.last_kernel_ms(), "waves", len(dur), "span us", t1.max())
print("wave duration us: mean %.1f med %.1f p90 %.1f max %.1f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
for n in sorted(set(ncj4)):
    m = ncj4 == n
    print(f"  ncj4 {n:3d}: {m.sum():6d} waves, dur mean {dur[m].mean():6.1f} us, start mean {t0[m].mean():6.1f}")
# concurrency over time
ts = np.linspace(0, t1.max(), 21)
for t in ts:
    print("t %.1f us: resident waves %d" % (t, ((t0 <= t) & (t1 > t)).sum()))
# per (xcc, cu, simd) counts: HW_ID bits: wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
u, c = np.unique(key, return_counts=True)
print("distinct SIMDs used", len(u), "waves per SIMD min/mean/max", c.min(), c.mean(), c.max())
busy = np.array([dur[key == k].sum() for k in u])
print("sum of wave durations per SIMD: min %.0f mean %.0f max %.0f us" % (busy.min(), busy.mean(), busy.max()))
last = np.array([t1[key == k].max() for k in u])
print("last wave end per SIMD: min %.1f mean %.1f max %.1f" % (last.min(), last.mean(), last.max()))
