#!/bin/bash
# A/B runs of kernel variant libraries (tools/build_variant.sh): pairs "A:B" as arguments, cases in $CASES
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=gromacs-cosmo_amd/lib/dbg
for pair in "$@"; do
    a=${pair%%:*}; b=${pair##*:}
    timeout -k 10 300 python3 tools/ab_bench.py $D/libnbnxn_hip_$a.so $D/libnbnxn_hip_$b.so ${CASES:-rf-f ewald-vf} 2>&1 | grep -v amdgpu.ids || exit 1
done
