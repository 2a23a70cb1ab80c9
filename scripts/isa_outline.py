#!/usr/bin/env python3
"""Outline of one kernel in a hipcc -S / -save-temps .s file: labels, waits, barriers, branches, LDS / global memory
instructions in order, runs of MFMAs collapsed to a count.   isa_outline.py file.s <substring of the mangled or demangled name> [from] [to]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
names = re.findall(r'^(_Z\S+):', s, flags=re.M)
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    if want not in d and want not in n:
        continue
    i = s.index('\n' + n + ':')
    body = s[i:s.index('.Lfunc_end', i)].splitlines()
    out, m = [], 0
    for l in body:
        t = l.strip()
        if not t or t.startswith(';'):
            continue
        if t.startswith('v_mfma'):
            m += 1
            continue
        if m:
            out.append('    ... %d mfma' % m)
            m = 0
        if re.match(r'^(\.LBB|s_waitcnt|s_barrier|s_cbranch|s_branch|ds_|global_|buffer_|scratch_|v_exp|s_endpgm|s_setprio|s_sleep)', t):
            out.append(t[:110])
    a = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    b = int(sys.argv[4]) if len(sys.argv) > 4 else len(out)
    print('==', d[:150], len(out), 'outline lines')
    print('\n'.join(out[a:b]))
    break
