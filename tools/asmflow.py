#!/usr/bin/env python3
"""Run-length view of a kernel's instruction stream (valu / ds_read / ds_write / loads / waits / barriers).
  python tools/asmflow.py file.s 'k_welch_pipe<4096, true, 8, true>' [first-label]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
want = sys.argv[2]
for nm in re.findall(r'^(_Z\S+):', s, re.M):
    d = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout
    if want not in d:
        continue
    a = s.index('\n' + nm + ':')
    body = s[a:s.index('.Lfunc_end', a)]
    if len(sys.argv) > 3:
        body = body[body.index(sys.argv[3]):]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
    def cat(l):
        op = l.split()[0]
        if op.startswith('.LBB'): return 'LABEL ' + op
        if op.startswith('.') or op.endswith(':'): return None
        if op.startswith('v_'): return 'valu'
        if op.startswith('ds_write'): return 'ds_write'
        if op.startswith('ds_read'): return 'ds_read'
        if op.startswith(('global_load', 'scratch_load')): return op
        if op.startswith(('global_store', 'scratch_store')): return op
        if op in ('s_waitcnt',) or op.startswith(('s_cbranch', 's_branch')): return l
        if op == 's_barrier': return 'BARRIER'
        return 'salu'
    out, prev, cnt = [], None, 0
    for l in lines:
        c = cat(l)
        if c is None: continue
        if c == prev: cnt += 1
        else:
            if prev: out.append('%s x%d' % (prev, cnt) if cnt > 1 else prev)
            prev, cnt = c, 1
    out.append('%s x%d' % (prev, cnt))
    print('\n'.join(out))
    break
