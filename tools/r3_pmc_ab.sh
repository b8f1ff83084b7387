#!/bin/bash
# SQ counters of the RF F workload for two builds of the library (NBNXN_HIP_LIB), a rocprofv3 --pmc run each
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r3_pmc_ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_BRANCH"
for v in "$@"; do
    export NBNXN_HIP_LIB=$R/gromacs-cosmo_amd/lib/$v
    timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $out/$(basename $v .so) -- python3 $R/bench.py --no-cpu --steps 20 --warmup 5 --precondition 0 --workload rf --force-only > $out/$(basename $v .so).txt 2>&1 || exit 1
    echo $v; python3 $R/tools/pmc_summary.py $out/$(basename $v .so)
done
