#!/usr/bin/env python3
"""Per-kernel static ISA statistics from a `hipcc -save-temps` assembly file:
   python tools/isa_stats.py <file.s> [substring ...]"""
import re
import sys

s = open(sys.argv[1]).read()
pats = sys.argv[2:]
funcs = re.split(r'\n(_Z[\w]+):[^\n]*\n', s)
for i in range(1, len(funcs), 2):
    name = funcs[i]
    if pats and not any(p in name for p in pats):
        continue
    rest = funcs[i + 1]
    body = rest.split('.Lfunc_end')[0]
    ins = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    def cnt(pre):
        return sum(1 for l in ins if l.startswith(pre))
    def meta(k):
        m = re.search(r'; %s: (\d+)' % k, rest)
        return m.group(1) if m else '?'
    print('%-90s valu %4d (pk %3d) salu %4d s_load %3d s_nop %3d ds %3d vmem %3d flat %d | vgpr %s sgpr %s scratch %s occ %s' % (
        name[:90], cnt('v_'), cnt('v_pk_'), cnt('s_') - cnt('s_load') - cnt('s_nop') - cnt('s_waitcnt'), cnt('s_load'), cnt('s_nop'), cnt('ds_'),
        cnt('global_') + cnt('buffer_'), cnt('flat_'), meta('NumVgprs'), meta('NumSgprs'), meta('ScratchSize'), meta('Occupancy')))
