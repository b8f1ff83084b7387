"""Copies the summaries of one `tools/profile_round.sh <tag>` pass from gpurun_out/<tag>/ into profiles/ (the tracked
directory): kernel stats of both workloads, the counter summary, the counted-flops file and the HBM traffic file that
bench.py quotes.  usage: python tools/collect_profiles.py <tag> "<one-line description of the kernel>" """
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, what = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for name, out in (("stats_ewald", f"{tag}_kernel_stats.csv"), ("stats_rf", f"{tag}_rf_f_kernel_stats.csv")):
    (f,) = glob.glob(os.path.join(src, name, "*", "*_kernel_stats.csv"))
    shutil.copy(f, os.path.join(dst, out))
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, f"{tag}_pmc_summary.txt"))
shutil.copy(os.path.join(src, "pmc_flops.json"), os.path.join(dst, f"{tag}_pmc_flops.json"))
summ = open(os.path.join(src, "pmc_summary.txt")).read()
kern = "nbnxn_kernel<1, true, false, false, 0, true>"


def counter(section, name):
    block = summ[summ.index(section + "\n"):]
    line = [l for l in block.split("\n") if l.startswith(kern)][0]
    return int(re.search(r"'%s': (\d+)" % name, line).group(1))


fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
stats = open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read().split("\n")
avg_ns = float([l for l in stats if kern in l][0].split('",')[1].split(",")[2])
total = (2 * fetch + write) * 1000
json.dump({
    "kernel": f"{kern} (analytical Ewald, energies+shift forces, no prune, plain LJ), bench.py default workload, {what}",
    "commands": ["rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu --steps 20 --warmup 5 --precondition 0",
                 "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu --steps 20 --warmup 5 --precondition 0"],
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
    "correction": "MI355X_MICROARCH.md, HBM section: counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane reads, so it is doubled; WRITE_SIZE is exact for float atomics",
    "hbm_bytes_per_launch": total,
    "kernel_avg_us_rocprofv3": avg_ns / 1e3,
    "GBps_at_kernel_avg": total / avg_ns,
    "source": f"profiles/{tag}_pmc_summary.txt (pmc_fetch / pmc_write passes of tools/profile_round.sh {tag})",
}, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(f"{tag}: Ewald V+F {avg_ns / 1e3:.2f} us avg, HBM {total / 1e6:.1f} MB per launch")
