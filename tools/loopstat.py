#!/usr/bin/env python3
"""Instruction mix of the largest basic block (the frame loop) of one kernel in a device assembly file.
  python tools/loopstat.py file.s 'k_welch_carry<4096, true, 8, true>'"""
import re, subprocess, sys
from collections import Counter
s = open(sys.argv[1]).read()
want = sys.argv[2]
for nm in re.findall(r'^(_Z\S+):', s, re.M):
    d = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout
    if want not in d:
        continue
    a = s.index('\n' + nm + ':')
    b = s.index('.Lfunc_end', a)
    blocks = re.split(r'\n(\.LBB\d+_\d+):', s[a:b])
    print(d.strip().split('(')[0])
    for i in range(2, len(blocks), 2):
        ins = [l.strip().split()[0] for l in blocks[i].split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//'))]
        if len(ins) < 60:
            continue
        c = Counter(ins)
        print(' block', blocks[i - 1], 'instrs', len(ins), 'VALU', sum(v for k, v in c.items() if k.startswith('v_')))
        print('   ' + ', '.join('%s %d' % kv for kv in c.most_common(24)))
