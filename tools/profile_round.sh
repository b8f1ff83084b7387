#!/bin/bash
# Profiles of one round, written under gpurun_out/<tag>/ on the GPU box: kernel stats of the default bench command,
# SQ counters and HBM traffic counters in passes of their own (rocprofv3 --pmc never together with a trace),
# for the headline workload and for BASELINE configs[1] (reaction field, forces only).
# usage: bash tools/profile_round.sh <tag>
tag=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_ewald -- $B > $out/stats_ewald.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_rf -- $B --workload rf --force-only > $out/stats_rf.txt 2>&1 || exit 1
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"
S="--steps 20 --warmup 5 --precondition 0"
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $out/pmc_sq_ewald -- $B $S > $out/pmc_sq_ewald.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $out/pmc_sq_rf -- $B $S --workload rf --force-only > $out/pmc_sq_rf.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B $S > $out/pmc_fetch.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B $S > $out/pmc_write.txt 2>&1 || exit 1
# where a wave's cycles go, and the per-type VALU counters (hardware-counted FP32 operations): passes of their own
WAVE="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
FLOP="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"
timeout -s KILL 120 rocprofv3 --pmc $WAVE --output-format csv -d $out/pmc_wave_ewald -- $B $S > $out/pmc_wave_ewald.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $WAVE --output-format csv -d $out/pmc_wave_rf -- $B $S --workload rf --force-only > $out/pmc_wave_rf.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $FLOP --output-format csv -d $out/pmc_flops_ewald -- $B $S > $out/pmc_flops_ewald.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $FLOP --output-format csv -d $out/pmc_flops_rf -- $B $S --workload rf --force-only > $out/pmc_flops_rf.txt 2>&1 || exit 1
for d in pmc_sq_ewald pmc_sq_rf pmc_wave_ewald pmc_wave_rf pmc_flops_ewald pmc_flops_rf pmc_fetch pmc_write; do echo $d; python3 $R/tools/pmc_summary.py $out/$d; done > $out/pmc_summary.txt
python3 $R/tools/pmc_flops.py $out $out/pmc_flops.json > /dev/null
