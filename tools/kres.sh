#!/bin/bash
# per-kernel register / spill / occupancy table of nbnxn_kernel.hip (extra hipcc flags as arguments)
cd "$(dirname "$0")/../gromacs-cosmo_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -I../../include "$@" -c nbnxn_kernel.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re,sys
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        n=m.group(1); k=re.search(r"nbnxn_kernel(_v\d)?ILi(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)",n)
        cur="k el%s ener%s prune%s vdwc%s vmod%s lds%s"%k.groups()[1:] if k else n[:40]; d={}
        continue
    for key in ("VGPRs:","VGPRs Spill:","Occupancy [waves/SIMD]:","LDS Size [bytes/block]:","ScratchSize [bytes/lane]:"):
        if key in l:
            d[key]=l.split(key)[1].split()[0]
            if key.startswith("LDS"):
                print(cur, " ".join("%s%s"%(k.split()[0],v) for k,v in d.items()))
'
