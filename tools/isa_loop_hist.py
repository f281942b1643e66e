"""Opcode histogram of the largest loop of one kernel in a gfx950 .s file (hipcc -save-temps).
Usage: isa_loop_hist.py file.s kernel_symbol_regex [top]"""
import collections, re, sys
text = open(sys.argv[1]).read()
m = re.search(r'^(%s\S*):[^\n]*\n(.*?)s_endpgm' % sys.argv[2], text, re.S | re.M)
lines = m.group(2).split('\n')
labels = {}
for i, l in enumerate(lines):
    k = re.match(r'^(\.LBB\d+_\d+):', l)
    if k: labels[k.group(1)] = i
loops = []
for i, l in enumerate(lines):
    k = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if k and labels.get(k.group(1), 1 << 30) < i: loops.append((i - labels[k.group(1)], labels[k.group(1)], i))
_, a, b = max(loops)
ops = [l.split()[0] for l in lines[a:b + 1] if l.startswith('\t') and l.strip() and l.strip()[0] not in '.;']
valu = sum(1 for o in ops if o.startswith('v_') and 'mfma' not in o)
print(m.group(1)[:100])
print("loop instructions %d, VALU (non-MFMA) %d, MFMA %d, accvgpr moves %d" % (
    len(ops), valu, sum('mfma' in o for o in ops), sum(o.startswith('v_accvgpr') for o in ops)))
for k in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy'):
    r = re.search(r'^(%s\S*):.*?; %s: (\d+)' % (sys.argv[2], k), text, re.S | re.M)
    if r: print(k, r.group(2), end='  ')
print()
for k, v in collections.Counter(ops).most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25): print("%5d %s" % (v, k))
