#!/usr/bin/env python3
"""Register / LDS / spill table of the kernels in a `hipcc -save-temps` assembly file (gfx950 .s): kernel_regs.py file.s [filter ...]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2:]
md = s[s.find('amdhsa.kernels'):]
names, rows = [], []
for it in re.split(r'\n  - ', md)[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, it) or [None, '?'])[1]
    name = re.search(r'\.name:\s+(\S+)', it)
    if name:
        names.append(name.group(1))
        rows.append((g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size')))
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
print('vgpr agpr sgpr spill scratch  kernel')
for r, n in zip(rows, dem):
    if not flt or any(f in n for f in flt):
        print('%4s %4s %4s %5s %7s  %s' % (*r, n[:170]))
