"""Work throughput over the launch, from the dumps of tools/timeline.py (TIMELINE_DUMP=file.npz): live cluster pairs
completed per microsecond in 2 us bins (every wave is taken to work through its entry's live pairs at an even pace
between the start and the end of its pair loop), the gap between a wave leaving a hardware slot and the next one
starting in it (slots of one HW_ID value matched first-freed-first-used; the XCC is not part of HW_ID, so eight slots
share a value), and cycles per live cluster pair of the waves by the time they started.
usage: python tools/timeline_rate.py file.npz [...]"""
import sys
import numpy as np

for path in sys.argv[1:]:
    d = np.load(path)
    t, live, ncj4, ej, cyc, hw = d["t"], d["e_live"].astype(float), d["ncj4"].astype(float), d["e_j"].astype(float), d["cyc"], d["hw"]
    t0, t1, t2, t3 = t.T
    dur, span = t3 - t0, t3.max()
    print(f"== {path}: {len(t0)} entries, span {span:.1f} us, {int(live.sum())} live cluster pairs, {len(np.unique(hw)) * 8} wave slots")
    edges = np.arange(0.0, span + 2.0, 2.0)
    rate = []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(t2, b) - np.maximum(t1, a), 0, None)
        rate.append((live * ov / np.maximum(t2 - t1, 1e-3)).sum() / (b - a))
    rate = np.array(rate)
    print("  k live pairs per us, 2 us bins:", " ".join(f"{r / 1e3:.1f}" for r in rate))
    full = rate[2:8].mean()
    print(f"  all slots in the pair loop (4-16 us): {full / 1e3:.1f} k pairs/us; the whole launch at that rate: {live.sum() / full:.1f} us"
          f" = {live.sum() / full / span:.2f} of the span")
    gaps = []
    for h in np.unique(hw):
        m = np.nonzero(hw == h)[0]
        ev = sorted([(t0[i], 1) for i in m] + [(t3[i], 0) for i in m])
        free = []
        for tt, kind in ev:
            if kind == 0:
                free.append(tt)
            elif tt > 1.5 and free:
                gaps.append(tt - free.pop(0))
    g = np.array(gaps)
    print(f"  {len(g)} slot refills: gap between a wave's end and the next wave's first instruction: median {np.median(g):.2f} us, mean {g.mean():.2f},"
          f" 10th / 90th percentile {np.percentile(g, 10):.2f} / {np.percentile(g, 90):.2f}")
    pro, epi = t1 - t0, t3 - t2
    refill = t0 > 1.5
    print(f"  slot time outside the pair loop per refilled entry: gap {g.mean():.2f} + prologue {pro[refill].mean():.2f} + epilogue {epi[refill].mean():.2f} us;"
          f" its pair loop {(t2 - t1)[refill].mean():.2f} us for {ncj4[refill].mean():.1f} groups")
    for lo, hi in ((0, 1.2), (1.2, 30), (30, 40), (40, 50), (50, 60)):
        m = (t0 >= lo) & (t0 < hi) & (dur > 3)
        if m.sum():
            print(f"  waves started {lo:4.1f}-{hi:4.1f} us: {int(m.sum()):5d}, groups per entry {ncj4[m].mean():4.1f}, live pairs per j-cluster {live[m].sum() / ej[m].sum():.2f},"
                  f" shader cycles per live pair in the loop {(cyc[m] * (t2 - t1)[m] / dur[m]).sum() / live[m].sum():.0f}")
