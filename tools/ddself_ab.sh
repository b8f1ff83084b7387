#!/bin/bash
# the decomposed step on one rank (its own neighbour) with and without the CU reservation of the local stream
R=${GRAFT_REPO_ROOT:-$(pwd)}
port=29700
for cus in default 0 16 64; do
    port=$((port + 1))
    if [ "$cus" = default ]; then unset GMX_HIP_NB_RESERVE_CUS; else export GMX_HIP_NB_RESERVE_CUS=$cus; fi
    out=$(MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 120 python3 $R/bench.py --dd-self --no-cpu $@ 2>/dev/null | grep '^{') || { echo "reserve $cus FAILED"; exit 1; }
    python3 - "$cus" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"reserve {sys.argv[1]:8s} step {d['ms_per_step']*1e3:6.2f} us  local kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  halo {d['config']['halo_atoms_rank0']}", flush=True)
PY
done
