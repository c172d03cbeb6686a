#!/usr/bin/env python3
"""Static instruction mix per kernel from hipcc -S output (tools/isa_mix.py engine.s [name-substring ...]).
Counts are static (every branch of the kernel once), a guide to where the instruction stream goes, not a profile."""
import re, sys, collections

def classify(op):
    if op.startswith('v_pk_'): return 'v_pk'
    if op.startswith('v_perm'): return 'v_perm'
    if op.startswith('v_cndmask'): return 'v_cndmask'
    if op.startswith('v_cmp'): return 'v_cmp'
    if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): return 'v_lane'
    if op.startswith('v_mov'): return 'v_mov'
    if op.startswith('v_'): return 'v_other'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith(('s_cbranch', 's_branch')): return 's_branch'
    if op.startswith('s_'): return 's_other'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'vmem'
    return 'other'

def main():
    path = sys.argv[1]
    subs = sys.argv[2:]
    lines = open(path).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^[A-Za-z_][A-Za-z0-9_]*:', l)]
    starts.append((len(lines), None))
    for (a, name), (b, _) in zip(starts, starts[1:]):
        if subs and not any(s in name for s in subs): continue
        c = collections.Counter()
        for l in lines[a:b]:
            m = re.match(r'^\s+([a-z_0-9]+)\s', l)
            if m and not l.lstrip().startswith(('.', ';')): c[classify(m.group(1))] += 1
        tot = sum(c.values())
        if tot < 20: continue
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print(f"{name[:70]:70s} total {tot:5d} valu {valu:5d}  " + ' '.join(f"{k}={v}" for k, v in sorted(c.items())))

if __name__ == '__main__':
    main()
