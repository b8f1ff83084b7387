"""Hardware-counted FP32 operations of the force kernels from a rocprofv3 --pmc pass with the per-type VALU counters
(tools/profile_round.sh: SQ_INSTS_VALU_FMA_F32, _MUL_F32, _ADD_F32, _TRANS_F32, SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU).
Writes profiles/<tag>_pmc_flops.json, which bench.py quotes as roofline.achieved_counted while the kernel source is the
one the counters were taken on (its SHA-256 is stored with them).

Counting rule: the SQ counters count WAVE instructions, whatever the EXEC mask; flops = (2 FMA + MUL + ADD + TRANS) x 64
lanes, i.e. issued lane slots (the cut-off test switches lanes off inside a pair: SQ_THREAD_CYCLES_VALU / (64 x
SQ_INSTS_VALU) is the share of lanes that were on, over all vector instructions).
usage: pmc_flops.py <gpurun_out/tag dir> <profiles/out.json>"""
import csv, glob, collections, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    return {k: {c: x / cnt[k][c] for c, x in v.items()} for k, v in acc.items()}


def flops(c):
    return 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_TRANS_F32"])


def main():
    src, out = sys.argv[1], sys.argv[2]
    res = {"kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, "gromacs-cosmo_amd/csrc/nbnxn_kernel.hip"), "rb").read()).hexdigest(),
           "counting_rule": "(2 x FMA_F32 + MUL_F32 + ADD_F32 + TRANS_F32) x 64 lanes per wave instruction, EXEC mask not considered",
           "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 "
                      "SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -- python3 bench.py --no-cpu --steps 20 --warmup 5 --precondition 0 [--workload rf --force-only]"}
    for key, d, pat in (("ewald_vf", "pmc_flops_ewald", "nbnxn_kernel<1, true, false, false, 0, true>"), ("rf_f", "pmc_flops_rf", "nbnxn_kernel<0, false, false, false, 0, true>")):
        ks = per_kernel(os.path.join(src, d))
        k = [n for n in ks if n.startswith(pat)]
        if not k:
            raise SystemExit(f"no kernel {pat} in {d}: {list(ks)[:4]}")
        c = ks[k[0]]
        res[key] = {"kernel": k[0][:60], "counters_per_launch": {n: round(v) for n, v in c.items()}, "flop_per_launch": flops(c),
                    "lanes_on_share": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else None}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
