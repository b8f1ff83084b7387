#!/bin/bash
# Round 3: the multi-process GPU tests once, then the decomposed step on one rank with the non-local stream at normal
# and at high priority (what the priority costs or buys where there is one process per GPU), then the occupancy census.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 10 700 python3 -u -m pytest tests/test_gpu_dd_ranks_one_gpu.py -x -v --timeout 280 --timeout-method thread > gpurun_out/r3_dd_ranks.txt 2>&1
rc=$?; tail -12 gpurun_out/r3_dd_ranks.txt
[ $rc -eq 0 ] || exit 0
for p in normal high normal high; do
    GMX_HIP_NB_NONLOCAL_PRIORITY=$p timeout -k 10 200 python3 bench.py --dd-self --steps 1000 --no-cpu >> gpurun_out/r3_ddself_$p.json 2>> gpurun_out/r3_ddself_$p.err || exit 0
done
for p in normal high; do
    GMX_HIP_NB_NONLOCAL_PRIORITY=$p timeout -k 10 200 python3 bench.py --dd-self --dd-self-dims 0,1,2 --steps 1000 --no-cpu >> gpurun_out/r3_ddself3_$p.json 2>> gpurun_out/r3_ddself3_$p.err || exit 0
done
timeout -k 10 120 tools/ubench/occupancy_census > gpurun_out/r3_occupancy_census.txt 2>&1
cat gpurun_out/r3_occupancy_census.txt
