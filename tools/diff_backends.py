#!/usr/bin/env python3
"""Debugging aid (GPU box): decode one fixture of tests/test_damaged_streams.py with the CPU oracle and with the product
and list the macroblocks in which the output pictures differ.  usage: diff_backends.py <fixture name>"""
import hashlib, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth
from test_damaged_streams import stream_of, ALL
name = sys.argv[1]
data = stream_of(name)
frames = {"oracle": [], "gpu": []}
orig = hashlib.sha1
for be in ("oracle", "gpu"):
    class H:
        def __init__(s, b): s.b = bytes(b); frames[be].append(s.b)
        def hexdigest(s): return orig(s.b).hexdigest()
    synth.hashlib = type("x", (), {"sha1": staticmethod(H)})
    res = synth.decode_ours(data, be)
    frames[be + "_pics"] = res[1]
w = ALL[name][0]["wmb"]
for i, (a, b) in enumerate(zip(frames["oracle"], frames["gpu"])):
    if a != b:
        fa, fb = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
        n = len(fa) // 384; h = n // w
        Y = (fa[:n * 256] != fb[:n * 256]).reshape(h * 16, w * 16)
        mbs = sorted(set(int((y // 16) * w + x // 16) for y, x in zip(*np.nonzero(Y))))
        print("picture", i, frames["oracle_pics"][i][1:], "differs: luma MBs", mbs, "chroma bytes", int((fa[n * 256:] != fb[n * 256:]).sum()))
print("compared", len(frames["oracle"]), "pictures")
