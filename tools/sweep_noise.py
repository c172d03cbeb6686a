#!/usr/bin/env python3
"""Parity sweep with damage ANYWHERE in the stream (parameter sets, NAL headers, start codes, slice data): random streams
from the bitstream writer are hit by byte noise, truncation, a garbage burst, bit flips in the first 120 bytes, bit flips
anywhere, or a deleted span (seed % 6), then decoded by the compiled reference (zeroed allocations, tests/synth.py) and by
the host parser + oracle; call traces and output pictures must be identical.  TEST TOOL (uses oracle/).
usage: sweep_noise.py <first seed> <count> [nor]      (nor: no_output_reordering = seed & 1)"""
import sys, os, time, random
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import h264writer, synth
os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
first, count = int(sys.argv[1]), int(sys.argv[2]); nor_mode = len(sys.argv) > 3
bad=[]; t0=time.time(); npics=0
for seed in range(first, first+count):
    rng=random.Random(seed)
    cfg=h264writer.random_config(seed)
    d=bytearray(h264writer.StreamWriter(**cfg).build()); n=len(d)
    kind=seed%6
    if kind==0:
        for _ in range(1+rng.randrange(20)): d[rng.randrange(n)]=rng.randrange(256)
    elif kind==1:
        d=d[:10+rng.randrange(n-10)]
    elif kind==2 and n>300:
        s=rng.randrange(n-200)
        for i in range(rng.randrange(8,200)): d[s+i]=rng.randrange(256)
    elif kind==3:
        for _ in range(8): d[rng.randrange(min(n,120))]^=1<<rng.randrange(8)      # headers: SPS / PPS / first slice header
    elif kind==4:
        for _ in range(1+rng.randrange(60)): d[rng.randrange(n)]^=1<<rng.randrange(8)
    else:
        a=rng.randrange(n); b=min(n,a+rng.randrange(1,400)); del d[a:b]
    data=bytes(d)
    nor = seed & 1 if nor_mode else 0
    ref=synth.decode_reference(data,nor); ours=synth.decode_ours(data,'oracle',nor); npics+=len(ref[1])
    if ref!=ours:
        bad.append(seed); print('MISMATCH', seed, 'kind', kind, 'trace equal', ref[0]==ours[0], len(ref[1]), len(ours[1]), flush=True)
    if (seed-first)%200==199: print('...', seed-first+1, len(bad), flush=True)
print(f'noise sweep {first}..{first+count-1}: {count-len(bad)} identical, {len(bad)} not {bad[:20]}; {npics} pictures, {time.time()-t0:.0f} s')
