#!/bin/bash
# builds gromacs-cosmo_amd/lib/dbg/libnbnxn_hip_<name>.so with extra flags on the three kernel units (A/B experiments
# with tools/ab_bench.py).  usage: bash tools/build_variant.sh <name> [-DFLAG ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../gromacs-cosmo_amd/csrc"
make -s all
mkdir -p ../lib/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -I../../include"
for el in 0 1 2; do
    X="$*"
    if [ $el -eq 2 ]; then X="${X//-DNBX_TIMELINE/}"; fi      # the timeline symbols exist once per electrostatics family
    if [ $el -ne 0 ]; then X="$X -mllvm -amdgpu-sched-strategy=max-ilp"; fi      # as the Makefile builds the Ewald units
    hipcc $F $X -DNBX_UNIT_EL=$el -c nbnxn_kernel.hip -o ../lib/dbg/k_${name}_el$el.o &
done
wait
O=../lib/obj
hipcc --offload-arch=gfx950 -shared ../lib/dbg/k_${name}_el0.o ../lib/dbg/k_${name}_el1.o ../lib/dbg/k_${name}_el2.o $O/nbnxn_atomdata.o $O/nbnxn_hip_api.o $O/nbnxn_search.o $O/nbnxn_search_prims.o $O/nbnxn_gpu_utils.o $O/nbnxn_prune.o -o ../lib/dbg/libnbnxn_hip_$name.so -ldl
rm -f ../lib/dbg/k_${name}_el*.o
echo built ../lib/dbg/libnbnxn_hip_$name.so
