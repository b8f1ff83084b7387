"""Averages rocprofv3 --pmc counters per kernel name.  usage: pmc_summary.py <dir> [substr]"""
import csv, glob, collections, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "nbnxn_kernel"
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    if sub in k:
        print(k, {c: round(x / cnt[k][c]) for c, x in v.items()})
