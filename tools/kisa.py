"""Instruction histogram of the kernels in an assembly listing made with
   hipcc ... -S --cuda-device-only nbnxn_kernel.hip -o k.s   usage: kisa.py k.s [name-substring ...]"""
import sys, collections
s = open(sys.argv[1]).read().split('\n')
tags = sys.argv[2:] or ['nbnxn_kernelILi0ELb0ELb0ELb1E', 'nbnxn_kernelILi1ELb1ELb0ELb1E']
for tag in tags:
    st = [i for i, l in enumerate(s) if l.startswith('_Z') and tag in l and ':' in l]
    if not st:
        print(tag, 'not found'); continue
    st = st[0]
    en = [i for i in range(st, len(s)) if 's_endpgm' in s[i]][0]
    c = collections.Counter()
    for l in s[st:en]:
        l = l.strip()
        if not l or l.startswith(('.', ';')):
            continue
        c[l.split()[0]] += 1
    print(tag, 'instr', sum(c.values()), 'VALU', sum(v for k, v in c.items() if k.startswith('v_')),
          'SALU', sum(v for k, v in c.items() if k.startswith('s_')))
    print('  ' + ' '.join(f'{k}:{v}' for k, v in c.most_common(36)))
