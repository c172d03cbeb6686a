#!/usr/bin/env python3
"""Debugging aid (GPU box): capture the frame jobs of a fixture of tests/test_damaged_streams.py on the host, run them
one by one through the kernels (replay set, one stream) and through the CPU oracle, and report the first job after
which the two frames differ, with the macroblocks.  usage: replay_vs_oracle.py <fixture name>"""
import ctypes, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import h264bsd_amd as h
from h264bsd_amd import capi
from oracle import pyoracle
from test_damaged_streams import stream_of
data = stream_of(sys.argv[1])
jobs = []
dec = capi.Decoder(0, capture=lambda b: jobs.append(bytes(b)))
buf = ctypes.create_string_buffer(data, len(data)); base = ctypes.addressof(buf); off = pid = stall = 0
while off < len(data) and stall <= 3:
    r, rb = dec.decode(base + off, len(data) - off, pid); off += rb; pid += r == 1
    stall = stall + 1 if rb == 0 else 0
dec.close()
rep = h.Replay(jobs, n_streams=1)
dpb = pyoracle.OracleDpb(jobs[0])
prev = None
for i, j in enumerate(jobs):
    hd = h.job_header(j)
    want = np.array(dpb.decode(j))
    rep.run(i, 1); rep.sync()
    got = rep.fetch(0, hd["cur_slot"])
    if not np.array_equal(want, got[: len(want)]):
        if prev is not None and prev[0] == hd['cur_slot']:
            print('  GPU frame equals the frame before this job (nothing ran):', np.array_equal(prev[1], got[: len(want)]), '; bytes changed by the GPU', int((prev[1] != got[:len(want)]).sum()), 'by the oracle', int((prev[1] != want).sum()))
        w = hd["width_mbs"]; n = hd["n_mbs"]
        Y = (want[: n * 256] != got[: n * 256]).reshape(-1, w * 16)
        mbs = sorted(set(int((y // 16) * w + x // 16) for y, x in zip(*np.nonzero(Y))))
        print(f"job {i} (pic_seq {hd['pic_seq']} ghost {hd['ghost']} dbk_only {hd['dbk_only']}): frames differ, luma MBs {mbs}, chroma bytes {int((want[n*256:] != got[n*256:len(want)]).sum())}")
        a = mbs[0]; x0 = (a % w) * 16; y0 = (a // w) * 16
        W = want[: n * 256].reshape(-1, w * 16)[y0:y0 + 16, x0:x0 + 16].astype(int); G = got[: n * 256].reshape(-1, w * 16)[y0:y0 + 16, x0:x0 + 16].astype(int)
        print('  first differing MB', a, ': oracle - GPU'); print(W - G)
        if prev is not None: print('  oracle - before:'); print(W - prev[1][: n * 256].reshape(-1, w * 16)[y0:y0 + 16, x0:x0 + 16].astype(int))
        print('  kinds', [j[hd['rec_off'] + 32 * k] for k in range(n)]); print('  dbk  ', [j[hd['rec_off'] + 32 * k + 5] for k in range(n)])
        break
    prev = (hd['cur_slot'], want.copy())
else:
    print("all", len(jobs), "jobs identical")
rep.close()
