"""Gaps of the PCIe-inclusive step from a rocprofv3 --kernel-trace of tools/pcie_rate.py: per step, the start and end of the
copy-in kernel, the force kernel and the copy-out kernel.  usage: python tools/pcie_gaps.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
import numpy as np

(f,) = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
steps = []
i = 0
while i + 2 < len(ev):
    a, b, c = ev[i], ev[i + 1], ev[i + 2]
    if "nbx_copy16" in a[2] and "nbnxn_kernel" in b[2] and "nbx_copy16" in c[2]:
        steps.append((a, b, c))
        i += 3
    else:
        i += 1
steps = steps[len(steps) // 2:]          # the second half: pinned arrays, timing off
d = np.array([[a[1] - a[0], b[0] - a[1], b[1] - b[0], c[0] - b[1], c[1] - c[0]] for a, b, c in steps]) / 1e3
per = np.array([steps[k + 1][0][0] - steps[k][0][0] for k in range(len(steps) - 1)]) / 1e3
idle = np.array([steps[k + 1][0][0] - steps[k][2][1] for k in range(len(steps) - 1)]) / 1e3
print(f"{len(steps)} steps: copy-in {np.median(d[:, 0]):.1f} us, gap {np.median(d[:, 1]):.1f}, force kernel {np.median(d[:, 2]):.1f}, gap {np.median(d[:, 3]):.1f}, "
      f"copy-out {np.median(d[:, 4]):.1f}; step period {np.median(per):.1f} us, GPU idle between steps (end of copy-out to next copy-in) {np.median(idle):.1f} us")
