"""Kernel timeline of a few steps of `bench.py --dd-self` from a rocprofv3 kernel trace.
usage: python tools/dd_trace.py <dir with *_kernel_trace.csv> [nsteps]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last local force kernels: the long ones on the local queue
names = [r["Kernel_Name"] for r in rows]
big = [i for i, r in enumerate(rows) if "nbnxn_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0 = big[-nsteps - 50]          # well inside the timed region
i1 = big[-50]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"# {nsteps} steps from the rocprofv3 kernel trace (us from the first local kernel's start: start, end; queue; kernel)")
for r in rows[i0 - 8:i1 + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
