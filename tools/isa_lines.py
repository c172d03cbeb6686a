#!/usr/bin/env python3
"""Static VALU / LDS / VMEM / SALU instruction count per SOURCE LINE of one kernel (hipcc -S -gline-tables-only output):
tools/isa_lines.py engine.s <kernel-substring> [min_count].  Inlined code is charged to the line of the innermost callee."""
import re, sys, collections
path, sub = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lines = open(path).read().split('\n')
a = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and sub in l)
b = next(i for i in range(a + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
loc = (0, 0)
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]*)"', l)
    if m: files[int(m.group(1))] = m.group(2).split('/')[-1]
cnt = collections.defaultdict(collections.Counter)
for l in lines[a:b]:
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)\s+(\d+)', l)
    if m: loc = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'^\s+([a-z_0-9]+)\s', l)
    if not m or l.lstrip().startswith(('.', ';')): continue
    op = m.group(1)
    k = 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')) else 'salu' if op.startswith('s_') else 'other'
    cnt[loc][k] += 1
tot = collections.Counter()
for c in cnt.values(): tot.update(c)
print('total', dict(tot))
for loc in sorted(cnt):
    c = cnt[loc]
    if sum(c.values()) >= minc: print(f"{files.get(loc[0], loc[0])}:{loc[1]:<5d}  valu {c['valu']:4d}  salu {c['salu']:4d}  lds {c['lds']:3d}  vmem {c['vmem']:3d}")
