#!/bin/bash
# Round 3, after the second sort key: size-class tables of the progressive split and split targets re-scanned
# (NBNXN_PROG_FACTORS, host builder only; --mcb = entries aimed for).  usage: bash tools/ab_prog2.sh > out.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {   # label, factors, extra args
    out=$(NBNXN_PROG_FACTORS="$2" timeout -k 10 120 python3 $R/bench.py --no-cpu --steps 600 ${@:3} 2>/dev/null | grep '^{') || { echo "$1 FAILED"; return 1; }
    python3 - "$1" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:34s} nsci {d['config']['nsci']:6d} kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  step {d['ms_per_step']*1e3:6.2f} us  frac {d['roofline']['frac']:.3f}", flush=True)
PY
}
for wl in "" "--workload rf --force-only"; do
    echo "== workload: ${wl:-ewald V+F}"
    run "built-in (1.5x6,.75,.4) 40/CU" "" $wl || exit 1
    run "built-in 30/CU" "" --mcb 7680 $wl || exit 1
    run "built-in 35/CU" "" --mcb 8960 $wl || exit 1
    run "built-in 50/CU" "" --mcb 12800 $wl || exit 1
    run "uniform 40/CU" "1,1,1,1,1,1,1,1" $wl || exit 1
    run "uniform 30/CU" "1,1,1,1,1,1,1,1" --mcb 7680 $wl || exit 1
    run "1.3x6,.6,.3" "1.3,1.3,1.3,1.3,1.3,1.3,0.6,0.3" $wl || exit 1
    run "1.25x6,1,.5" "1.25,1.25,1.25,1.25,1.25,1.25,1.0,0.5" $wl || exit 1
    run "1.7x6,.6,.3" "1.7,1.7,1.7,1.7,1.7,1.7,0.6,0.3" $wl || exit 1
    run "1.5x5,1,.5,.25" "1.5,1.5,1.5,1.5,1.5,1.0,0.5,0.25" $wl || exit 1
    run "built-in again" "" $wl || exit 1
done
