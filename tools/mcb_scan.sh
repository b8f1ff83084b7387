#!/bin/bash
# kernel time against the list-split target (entries per CU) for the headline kernel and the RF force-only kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for per_cu in ${1:-30 40 50 60 80}; do
    mcb=$((per_cu * 256))
    e=$(timeout -k 10 120 python3 bench.py --no-cpu --steps 400 --mcb $mcb 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us kernel, %.4f ms/step, nsci %d' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step'], d['config']['nsci']))") || exit 1
    r=$(timeout -k 10 120 python3 bench.py --no-cpu --steps 400 --mcb $mcb --workload rf --force-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us kernel, %.4f ms/step' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']))") || exit 1
    echo "$per_cu entries per CU: Ewald V+F $e | RF F $r"
done
