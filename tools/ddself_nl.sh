#!/bin/bash
# the decomposed step on one rank (its own neighbour) for several pair budgets of the non-local list entries
R=${GRAFT_REPO_ROOT:-$(pwd)}
port=29800
for sc in ${NL_SCALES:-1 2 4 8 1000}; do
    port=$((port + 1))
    out=$(NBNXN_NL_SPLIT_SCALE=$sc MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 120 python3 $R/bench.py --dd-self --no-cpu $@ 2>/dev/null | grep '^{') || { echo "scale $sc FAILED"; exit 1; }
    python3 - "$sc" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"nl split scale {sys.argv[1]:5s} step {d['ms_per_step']*1e3:6.2f} us  local kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  nsci local/non-local {d['config']['nsci_local']}/{d['config']['nsci_nonlocal']}  {d['value']/1e9:.0f} Gpair/s", flush=True)
PY
done
