#!/usr/bin/env python3
"""Run only tick `TICK` (default 0, an IDR picture of every stream) of the lock-step replay N times: a workload for
counter passes over the per-picture kernels on one kind of picture.  usage: itick.py [tick] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd
tick = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
jobs, _, _ = h264bsd_amd.capture_stream(data)
rep = h264bsd_amd.Replay(jobs, n_streams=256)
rep.run(0, tick + 1); rep.sync()          # the pictures before it, once (references)
for _ in range(reps):
    rep.run(tick, 1)
rep.sync()
t = rep.timings()
print({k: t[k] for k in h264bsd_amd.Replay.KERNELS})
rep.close()
