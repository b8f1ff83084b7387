#!/bin/bash
# Tail split of the force kernel's launch (GMX_HIP_NB_TAIL_GROUPS / GMX_HIP_NB_TAIL_SPLIT): kernel time of the bench workloads
# for several settings, separate bench.py processes on one box.  usage: bash tools/tail_split_scan.sh > out.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {   # groups, split, extra args
    out=$(GMX_HIP_NB_TAIL_GROUPS=$1 GMX_HIP_NB_TAIL_SPLIT=$2 timeout -k 10 120 python3 $R/bench.py --no-cpu --steps 600 ${@:3} 2>/dev/null | grep '^{') || { echo "$1 $2 FAILED"; return 1; }
    python3 - "$1" "$2" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
par = d.get("parity", {})
print(f"tail groups {sys.argv[1]} split {sys.argv[2]}: nsci {d['config']['nsci']:6d} kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  step {d['ms_per_step']*1e3:6.2f} us  frac {d['roofline']['frac']:.3f}"
      f"  force rel rms vs oracle {par.get('force_rel_rms_vs_oracle', float('nan')):.2e}", flush=True)
PY
}
for wl in "" "--workload rf --force-only"; do
    echo "== workload: ${wl:-ewald V+F}"
    for s in "0 1" "1 2" "1 4" "2 2" "2 4" "3 2" "3 4" "4 4" "0 1"; do run $s $wl || exit 1; done
done
