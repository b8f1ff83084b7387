#!/bin/bash
# Tuning aid: kernel time of the bench workload for several size-class tables of the progressive list split
# (NBNXN_PROG_FACTORS, host builder only) and split targets (--mcb = entries aimed for).  usage: bash tools/ab_prog.sh > out.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {   # label, factors, extra args
    out=$(NBNXN_PROG_FACTORS="$2" timeout -k 10 120 python3 $R/bench.py --no-cpu ${@:3} 2>/dev/null | grep '^{') || { echo "$1 FAILED"; return 1; }
    python3 - "$1" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:34s} nsci {d['config']['nsci']:6d} kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  step {d['ms_per_step']*1e3:6.2f} us  frac {d['roofline']['frac']:.3f}", flush=True)
PY
}
for wl in "" "--workload rf --force-only"; do
    echo "== workload: ${wl:-ewald V+F}"
    run "built-in (1.5x6,.75,.4)" "" $wl || exit 1
    run "1.5x5,1,.5,.25" "1.5,1.5,1.5,1.5,1.5,1.0,0.5,0.25" $wl || exit 1
    run "1.6x5,.8,.4,.2" "1.6,1.6,1.6,1.6,1.6,0.8,0.4,0.2" $wl || exit 1
    run "1.5x4,1,.7,.4,.2" "1.5,1.5,1.5,1.5,1.0,0.7,0.4,0.2" $wl || exit 1
    run "2x4,1,.5,.3,.15" "2,2,2,2,1.0,0.5,0.3,0.15" $wl || exit 1
    run "1.3x6,.6,.3" "1.3,1.3,1.3,1.3,1.3,1.3,0.6,0.3" $wl || exit 1
    run "1.5x4,1,.7,.4,.2 mcb 30/CU" "1.5,1.5,1.5,1.5,1.0,0.7,0.4,0.2" --mcb 7680 $wl || exit 1
    run "1.5x4,1,.7,.4,.2 mcb 50/CU" "1.5,1.5,1.5,1.5,1.0,0.7,0.4,0.2" --mcb 12800 $wl || exit 1
    run "uniform" "1,1,1,1,1,1,1,1" $wl || exit 1
done
