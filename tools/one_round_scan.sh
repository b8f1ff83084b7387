#!/bin/bash
# Tuning aid: does a list of about as many entries as there are wave slots (one dispatch round, no refills) beat the
# built-in split?  Kernel time of the bench workload for split targets around 5120 entries, uniform and progressive classes.
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
    run "built-in" "" $wl || exit 1
    for m in 3500 4200 4800 5120 5600; do
        run "uniform mcb $m" "1,1,1,1,1,1,1,1" --mcb $m $wl || exit 1
        run "built-in classes mcb $m" "" --mcb $m $wl || exit 1
    done
done
