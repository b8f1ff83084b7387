"""Wave timeline of the force kernel (debug build: make -C gromacs-cosmo_amd/csrc timeline; run with
NBNXN_HIP_LIB=gromacs-cosmo_amd/lib/dbg/libnbnxn_hip_tl.so).  Prints how many waves are resident over the launch,
the ramp and the tail, and where a wave's time goes: before its first pair (prologue), in the pair loop, after it
(i-force reduction, energies).  usage: NBNXN_HIP_LIB=... python tools/timeline.py [ewald-vf|ewald-f|rf-f|rf-vf] [mcb]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi

case = sys.argv[1] if len(sys.argv) > 1 else "ewald-vf"
el, fl = case.split("-")
nb = g.gpu.NbnxnHip(0)
mcb = int(sys.argv[2]) if len(sys.argv) > 2 else nb.min_ci_balanced()
ip = g.system.ewald(1.0) if el == "ewald" else g.system.reaction_field(1.0)
s = g.system.water_box(5, 5, 6, model="spce" if el == "ewald" else "spc")
p = g.workload.prepare(s, ip, mcb)
g.workload.upload(nb, p, ip)
flags = abi.FORCE_VIRIAL if fl == "vf" else 0
nb.compute(p.xq, abi.FORCE_VIRIAL)
nb.set_timing(True)
for _ in range(5):
    nb.clear_outputs(flags); nb.launch_kernel_resident(flags); nb.wait_gpu(flags)
n = p.pl.nsci
W = 6
tl = np.zeros((n, W), dtype=np.uint64)
fn = nb.lib.nbnxn_hip_debug_timeline if el == "ewald" else nb.lib.nbnxn_hip_debug_timeline_rf
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(tl.ctypes.data_as(C.c_void_p), n) == 0
t = tl[:, :4].astype(np.int64)
ncj4 = ((tl[:, 4] >> np.uint64(32)) & np.uint64(0xffffff)).astype(np.int64)
xcc = (tl[:, 4] >> np.uint64(56)).astype(np.int64)
hw = (tl[:, 4] & np.uint64(0xffffffff)).astype(np.int64)
base = t[:, 0].min()
t = (t - base) * 0.01      # us
t0, t1, t2, t3 = t.T
dur = t3 - t0
span = t3.max()
cyc = tl[:, 5].astype(np.float64)
big = dur > 5.0
print(f"shader clock while the kernel runs (cycles of a wave / its wall time, waves longer than 5 us): median {np.median(cyc[big] / dur[big]) / 1e3:.3f} GHz,"
      f" 5th / 95th percentile {np.percentile(cyc[big] / dur[big], 5) / 1e3:.3f} / {np.percentile(cyc[big] / dur[big], 95) / 1e3:.3f}")
slots = len(set(hw.tolist()))
print(f"{case}: nsci {n}  kernel span {span:.1f} us (event timing {nb.last_kernel_ms() * 1e3:.1f} us)  sum of wave times {dur.sum():.0f} us"
      f"  = {dur.sum() / span:.0f} waves resident on average; {slots} distinct hardware wave slots seen")
print(f"wave duration us: min {dur.min():.2f} median {np.median(dur):.2f} max {dur.max():.2f};  per cj4 group median {np.median(dur / np.maximum(ncj4, 1)):.2f}")
pro, loop, epi = t1 - t0, t2 - t1, t3 - t2
print(f"share of all wave time: prologue {pro.sum() / dur.sum():.3f}  pair loop {loop.sum() / dur.sum():.3f}  epilogue {epi.sum() / dur.sum():.3f}")
print(f"per entry, median us: prologue {np.median(pro):.2f}  pair loop {np.median(loop):.2f}  epilogue {np.median(epi):.2f};"
      f"  pair loop per cj4 group: median {np.median(loop / np.maximum(ncj4, 1)):.2f} us")
q = np.argsort(t0)
for name, sel in (("first 25% of entries", q[: n // 4]), ("middle 50%", q[n // 4: 3 * n // 4]), ("last 25%", q[3 * n // 4:])):
    print(f"  {name}: groups per entry {ncj4[sel].mean():.1f}, prologue {pro[sel].mean():.2f}  loop per group {(loop[sel] / np.maximum(ncj4[sel], 1)).mean():.2f}"
          f"  epilogue {epi[sel].mean():.2f} us")
edges = np.linspace(0, span, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = int(((t0 <= mid) & (t3 > mid)).sum())
    inloop = int(((t1 <= mid) & (t2 > mid)).sum())
    print(f"  t = {mid:6.1f} us: {res:5d} waves resident ({inloop:5d} in the pair loop), {int(((t0 >= a) & (t0 < b)).sum()):5d} started")
last = np.sort(t3)[::-1]
print("last waves end at", np.round(last[:5], 1), " 99%/95%/90% of waves ended by", np.round(np.percentile(t3, [99, 95, 90]), 1))
print("first wave of the final 10% starts at", np.round(np.sort(t0)[int(0.9 * n)], 1))

# the tail: which waves are still running at the end, and how evenly the SIMDs finish (a SIMD = XCC id + HW_ID without its
# wave-slot bits [3:0])
simd = xcc * (1 << 20) + (hw >> 4)
ids = np.unique(simd)
last_end = np.array([t3[simd == s].max() for s in ids])
busy = np.array([dur[simd == s].sum() for s in ids])
print(f"{len(ids)} SIMDs seen; a SIMD's last wave ends at: min {last_end.min():.1f}  5% {np.percentile(last_end, 5):.1f}  median {np.median(last_end):.1f}"
      f"  95% {np.percentile(last_end, 95):.1f}  max {last_end.max():.1f} us;  sum of wave times per SIMD: min {busy.min():.0f} median {np.median(busy):.0f} max {busy.max():.0f} us")
for frac in (0.85, 0.9, 0.95):
    m = t3 > frac * span
    print(f"  waves ending after {frac * span:5.1f} us: {int(m.sum()):5d}; groups per entry {ncj4[m].mean():.2f}; started: median {np.median(t0[m]):.1f}, earliest {t0[m].min():.1f} us;"
          f" of them started in the first 2 us: {int((t0[m] < 2).sum())}; duration median {np.median(dur[m]):.1f} us")
for a, b in ((0.0, 2.0), (2.0, 0.5 * span), (0.5 * span, 0.8 * span), (0.8 * span, span)):
    m = (t0 >= a) & (t0 < b)
    if m.sum():
        print(f"  entries started in [{a:5.1f}, {b:5.1f}) us: {int(m.sum()):5d}, groups {ncj4[m].mean():.2f}, end: median {np.median(t3[m]):.1f}  95% {np.percentile(t3[m], 95):.1f}  max {t3[m].max():.1f} us")

# where the time per unit of work goes over the launch: entries binned by their start time; work = live cluster pairs
# (set bits of the pruned interaction masks) and j-clusters with at least one live pair
pl = nb.download_pairlist()
im = (pl.cj4["imei"]["imask"][:, 0] | pl.cj4["imei"]["imask"][:, 1]).astype(np.uint32)
live = np.array([bin(int(v)).count("1") for v in im], dtype=np.int64)
jlive = np.array([sum(1 for k in range(4) if (int(v) >> (8 * k)) & 0xff) for v in im], dtype=np.int64)
cs_live, cs_j = np.concatenate([[0], np.cumsum(live)]), np.concatenate([[0], np.cumsum(jlive)])
a0, a1 = pl.sci["cj4_ind_start"].astype(np.int64), pl.sci["cj4_ind_end"].astype(np.int64)
e_live, e_j = cs_live[a1] - cs_live[a0], cs_j[a1] - cs_j[a0]
print(f"live cluster pairs {live.sum()}, per group {live.mean():.2f}, per live j-cluster {live.sum() / max(jlive.sum(), 1):.2f}")
print("entries by start time:   n  groups/entry  live pairs/group   loop us/group  loop ns/live pair   prologue  epilogue")
edges = np.percentile(t0, [0, 10, 25, 40, 55, 70, 85, 100])
edges[-1] += 1
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (t0 >= lo) & (t0 < hi)
    if m.sum() == 0:
        continue
    print(f"  start {lo:5.1f}-{hi:5.1f} us: {int(m.sum()):5d}  {ncj4[m].mean():6.1f}  {e_live[m].sum() / max(ncj4[m].sum(), 1):8.2f}"
          f"  {loop[m].sum() / max(ncj4[m].sum(), 1):10.2f}  {1e3 * loop[m].sum() / max(e_live[m].sum(), 1):10.1f}    {pro[m].mean():6.2f}  {epi[m].mean():6.2f}")
if os.environ.get("TIMELINE_DUMP"):
    np.savez_compressed(os.environ["TIMELINE_DUMP"], t=t, ncj4=ncj4, hw=hw, cyc=cyc, e_live=e_live, e_j=e_j)
