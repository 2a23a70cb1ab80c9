#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a `hipcc -save-temps` .s file: isa_loops.py file.s <substring of the demangled name> [min_mfma]
Prints, for every block with >= min_mfma MFMAs, the counts of MFMA / LDS / DMA / scratch / wait instructions and the waits in order."""
import re
import subprocess
import sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 6
names = re.findall(r'^(_Z\S+):', s, flags=re.M)
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    if want not in d:
        continue
    i = s.index('\n' + n + ':')
    body = s[i:s.index('.Lfunc_end', i)].splitlines()
    print('==', d[:160], len(body), 'lines')
    labels = [k for k, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)] + [len(body)]
    for a in range(len(labels) - 1):
        seg = body[labels[a]:labels[a + 1]]
        ops = [x.split()[0] for x in seg if x.strip() and not x.strip().startswith((';', '.'))]
        c = Counter(ops)
        nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
        if nm < min_mfma:
            continue
        keep = {k: v for k, v in c.items() if k.startswith(('s_waitcnt', 's_barrier', 'ds_', 'global_load', 'global_store', 'scratch_', 'buffer_', 'v_mfma', 's_nop', 'v_exp', 's_cbranch'))}
        print(seg[0].split(':')[0], 'instrs', len(ops), keep)
        print('   ', [x.strip() for x in seg if 's_waitcnt' in x or 's_barrier' in x])
