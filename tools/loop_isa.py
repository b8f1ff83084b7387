"""Register use and the instruction mix of the pair loop of one kernel flavour, from a device assembly file
(hipcc ... --cuda-device-only -S).  usage: python tools/loop_isa.py file.s <mangled-name fragment> [...]"""
import collections
import re
import sys

f = sys.argv[1]
s = open(f).read()
for pat in sys.argv[2:]:
    name = re.findall(r'^(_ZN\S*nbnxn_kernel' + pat + r'\S*):', s, flags=re.M)[0]
    i = s.index(name + ":")
    lines = s[i:s.index("s_endpgm", i)].split("\n")
    labels = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for k, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > 600:
            loops.append((labels[m.group(1)], k))
    a, b = min(loops, key=lambda x: x[1] - x[0])
    c = collections.Counter(m.group(1) for l in lines[a:b] for m in [re.match(r'\s+(\w+)', l)] if m)
    meta = s[s.index(".amdhsa_kernel " + name):]
    tot = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    vg, sg = re.search(r'next_free_vgpr (\d+)', meta).group(1), re.search(r'next_free_sgpr (\d+)', meta).group(1)
    print(f"{pat}: vgpr {vg} sgpr {sg};"
          f" pair loop {sum(c.values())} instr: valu {tot('v_')} salu {tot('s_')} readlane {c['v_readlane_b32']} writelane {c['v_writelane_b32']}"
          f" scratch {tot('scratch')} s_load {tot('s_load')} atomics {tot('global_atomic')};"
          f" whole kernel: readlane {sum('v_readlane' in l for l in lines)} scratch {sum('scratch_' in l for l in lines)}")
