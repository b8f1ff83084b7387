#!/bin/bash
# Round 3 diagnosis: phase timeline of the RF F and Ewald V+F kernels, and two counter passes each (where a wave's
# cycles go; the per-type VALU counters).  usage: bash tools/r3_probe.sh <tag>
tag=${1:-r3_probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for c in rf-f ewald-vf; do
    NBNXN_HIP_LIB=$R/gromacs-cosmo_amd/lib/dbg/libnbnxn_hip_tl.so timeout -k 10 120 python3 tools/timeline.py $c > $out/timeline_$c.txt 2>&1 || exit 1
done
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --steps 20 --warmup 5 --precondition 0"
WAVE="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
FLOP="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"
timeout -s KILL 120 rocprofv3 --pmc $WAVE --output-format csv -d $out/pmc_wave_ewald -- $B > $out/pmc_wave_ewald.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $WAVE --output-format csv -d $out/pmc_wave_rf -- $B --workload rf --force-only > $out/pmc_wave_rf.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $FLOP --output-format csv -d $out/pmc_flops_ewald -- $B > $out/pmc_flops_ewald.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $FLOP --output-format csv -d $out/pmc_flops_rf -- $B --workload rf --force-only > $out/pmc_flops_rf.txt 2>&1 || exit 1
for d in pmc_wave_ewald pmc_wave_rf pmc_flops_ewald pmc_flops_rf; do echo $d; python3 $R/tools/pmc_summary.py $out/$d; done > $out/pmc_summary.txt
cat $out/timeline_rf-f.txt $out/timeline_ewald-vf.txt $out/pmc_summary.txt
