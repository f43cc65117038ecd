#!/usr/bin/env python3
"""Static instruction mix of the kernels in an assembly listing (hipcc -S --cuda-device-only).  DEV TOOL, container only.
   usage: isa_mix.py file.s [kernel-name-substring]"""
import re, sys, collections
t = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", t, flags=re.S | re.M):
    n, body = m.group(1), m.group(2)
    if want not in n or '.amdhsa' in body[:200]: continue
    ins = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for i in ins:
        op = i.split()[0]
        k = ('s_waitcnt' if op.startswith('s_waitcnt') else 's_nop' if op.startswith('s_nop') else 's_branch' if op.startswith(('s_cbranch', 's_branch')) else
             's_barrier' if op.startswith('s_barrier') else 'smem' if op.startswith(('s_load', 's_buffer')) else 'salu' if op.startswith('s_') else
             'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'scratch' if op.startswith('scratch_') else 'vmem')
        c[k] += 1
    print(n[:70], len(ins), dict(c))
    print('  scalar ops:', collections.Counter(i.split()[0] for i in ins if i.startswith('s_')).most_common(16))
