"""Compress a kernel's gfx950 ISA (hipcc -save-temps .s) into a one-letter-per-instruction stream:
M mfma, t transcendental (exp/rcp/log/...), a accvgpr move, v other VALU, n s_nop, w s_waitcnt,
s other scalar, G global/scratch memory, L LDS, B branch.  Usage: isa_stream.py file.s kernel_regex"""
import re, sys, textwrap
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(%s\S*):[^\n]*\n(.*?)s_endpgm' % sys.argv[2], s, re.S | re.M)
ins = []
for l in m.group(2).split('\n'):
    t = l.strip()
    if not l.startswith('\t') or not t or t[0] in '.;':
        continue
    ins.append(t.split()[0])
print(m.group(1)[:80], len(ins), "instructions")
print(Counter(ins).most_common(30))
def code(i):
    if 'mfma' in i: return 'M'
    if re.match(r'v_(exp|rcp|log|sqrt|rsq|sin|cos)', i): return 't'
    if i.startswith('v_accvgpr'): return 'a'
    if i.startswith('v_'): return 'v'
    if i.startswith('s_nop'): return 'n'
    if i.startswith('s_waitcnt'): return 'w'
    if i.startswith('s_cbranch') or i.startswith('s_branch'): return 'B'
    if i.startswith('s_'): return 's'
    if i.startswith(('global', 'scratch', 'buffer', 'flat')): return 'G'
    if i.startswith('ds_'): return 'L'
    return '?'
print('\n'.join(textwrap.wrap(''.join(code(i) for i in ins), 160)))
