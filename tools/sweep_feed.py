#!/usr/bin/env python3
"""Parity sweep over the CALL PROTOCOL: the same (intact or damaged) stream is fed to h264bsdDecode in windows of random
length (1 byte .. 40 KB, advancing by readBytes) instead of "everything that is left", to the compiled reference and to
this library's parser; a window that ends inside a NAL unit makes the decoder see a truncated unit, and both must make
the same calls, return the same codes and readBytes, and output the same pictures.  TEST TOOL (uses oracle/).
usage: sweep_feed.py <first seed> <count> [flush]      (flush: whole-buffer calls, but h264bsdFlushBuffer + draining the
output queue after 6 % of the calls, as an application does when it seeks)"""
import sys, os, time, random, ctypes, hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import h264writer, synth, damage as dmg
import numpy as np
from oracle import pyoracle
from h264bsd_amd import capi
os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
libc=ctypes.CDLL(None)
FLUSH = len(sys.argv) > 3 and sys.argv[3] == 'flush'
def windows(seed, n):
    rng=random.Random(seed*7+1); 
    while FLUSH: yield 1 << 30

    while True: yield rng.choice([rng.randrange(1,40), rng.randrange(40,400), rng.randrange(400,4000), rng.randrange(400,4000), rng.randrange(4000,40000), rng.randrange(4000,40000), rng.randrange(4000,40000), rng.randrange(4000,40000), rng.randrange(4000,40000), rng.randrange(4000,40000)])
def run_ref(data, seed):
    libc.mallopt(-6, 0xFF)
    try:
        ref=pyoracle.RefDecoder(); lib=ref.lib
        buf=ctypes.create_string_buffer(data,len(data)); base=ctypes.addressof(buf)
        dec=lib.h264bsdAlloc(); lib.h264bsdInit(dec,0)
        off=0; rb=ctypes.c_uint32(0); trace=[]; pics=[]; a,b,c=ctypes.c_uint32(),ctypes.c_uint32(),ctypes.c_uint32(); pid=0; stall=0
        def drain():
            w,h=lib.h264bsdPicWidth(dec), lib.h264bsdPicHeight(dec)
            while True:
                p=lib.h264bsdNextOutputPicture(dec, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
                if not p: break
                pics.append((hashlib.sha1(ctypes.string_at(p,w*h*384)).hexdigest(), a.value,b.value,c.value))
        win=windows(seed,len(data)); frng=random.Random(seed*13+5)
        while off<len(data):
            ln=min(len(data)-off, next(win))
            r=lib.h264bsdDecode(dec, base+off, ln, pid, ctypes.byref(rb)); trace.append((int(r),int(rb.value))); off+=rb.value
            if r==1: pid+=1; drain()
            if FLUSH and frng.random() < 0.06: lib.h264bsdFlushBuffer(dec); drain()      # the application empties the output queue in mid-stream (a seek)
            stall = stall+1 if rb.value==0 else 0
            if stall>12: break
        lib.h264bsdFlushBuffer(dec); drain(); lib.h264bsdShutdown(dec); lib.h264bsdFree(dec)
        return trace,pics
    finally: libc.mallopt(-6,0)
def run_ours(data, seed):
    pics=[]; trace=[]; state={"dpb":None}
    def on_job(blob):
        if state["dpb"] is None or pyoracle.blob_header(blob)["n_slots"]!=len(state["dpb"].slots) or state["dpb"].frame_bytes!=pyoracle.blob_header(blob)["n_mbs"]*384:
            state["dpb"]=pyoracle.OracleDpb(blob)
        state["dpb"].decode(blob)
    dec=capi.Decoder(0, capture=on_job)
    buf=ctypes.create_string_buffer(data,len(data)); base=ctypes.addressof(buf); off=0; pid=0; stall=0
    def drain():
        while True:
            o=dec.next_output_info()
            if o is None: break
            slot,p,idr,nerr=o
            frame=state["dpb"].slots[slot][:state["dpb"].frame_bytes]
            pics.append((hashlib.sha1(np.ascontiguousarray(frame).tobytes()).hexdigest(), p,idr,nerr))
    win=windows(seed,len(data)); frng=random.Random(seed*13+5)
    while off<len(data):
        ln=min(len(data)-off, next(win))
        r,rb=dec.decode(base+off, ln, pid); trace.append((r,rb)); off+=rb
        if r==1: pid+=1; drain()
        elif r==2: state["dpb"]=None
        if FLUSH and frng.random() < 0.06: dec.flush_buffer(); drain()
        stall = stall+1 if rb==0 else 0
        if stall>12: break
    dec.flush_buffer(); drain(); dec.close()
    return trace,pics
first,count=int(sys.argv[1]),int(sys.argv[2]); bad=[]; t0=time.time(); npics=0
for seed in range(first, first+count):
    cfg=h264writer.random_config(seed)
    data=h264writer.StreamWriter(**cfg).build()
    if seed%3: data=dmg.damage(data, seed, p_drop=0.1, p_flip=0.2, p_trunc=0.1)
    r=run_ref(data,seed); o=run_ours(data,seed); npics+=len(r[1])
    if r!=o:
        bad.append(seed); print('MISMATCH',seed,'trace equal',r[0]==o[0],len(r[1]),len(o[1]),flush=True)
        if r[0]!=o[0]:
            for i,(x,y) in enumerate(zip(r[0],o[0])):
                if x!=y: print('  first trace diff at call',i,x,y); break
    if (seed-first)%200==199: print('...',seed-first+1,len(bad),flush=True)
print(f'chunk sweep {first}..{first+count-1}: {count-len(bad)} identical, {len(bad)} not {bad[:20]}; {npics} pictures, {time.time()-t0:.0f} s')
