#!/usr/bin/env python3
"""Desynchronised streams (every stream at its own picture index): lap time of the common-tick schedule and of the
heavy-lane schedules, verified against the golden checksums at the end of a lap.
usage: desync_probe.py [streams] [K,D[,G] ...]   K heavy lanes, D rejoin delay in ticks, G stream groups
OFFSETS=staggered: the staggered set of bench.py (odd streams start at the second IDR) instead of every stream at its own index"""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = os.environ.get("STREAM", "test_1920x1080")
jobs, _, _ = h.capture_stream(open(os.path.join(root, "tests", "golden", name + ".h264"), "rb").read())
golden = json.load(open(os.path.join(root, "tests", "golden", "golden.json")))[name]["frame_checksum64"]
heads = [h.job_header(j) for j in jobs]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = len(jobs)
offsets = [(s * P) // S for s in range(S)]
if os.environ.get("OFFSETS") == "staggered":
    idr = [i for i, hd in enumerate(heads) if hd["is_idr"] and i > 0][0]
    offsets = [idr if s & 1 else 0 for s in range(S)]
cfgs = ([] if os.environ.get("NOBASE") else [(0, 0, 1)]) + [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]]
for cfg in cfgs:
    K, D = cfg[0], cfg[1]
    G = cfg[2] if len(cfg) > 2 else 1
    rep = h.Replay(jobs, n_streams=S, offsets=offsets, heavy_lanes=K, heavy_delay=D, groups=G)
    def verify():
        bad = 0
        sums = {slot: rep.checksums(slot) for slot in set(hd["cur_slot"] for hd in heads)}
        for s in range(S):
            last = (offsets[s] - 1) % P
            bad += int(sums[heads[last]["cur_slot"]][s]) != golden[last]
        return bad
    rep.run(); rep.sync(); b1 = verify()
    rep.run(); rep.sync(); b2 = verify()
    t, th = [], []
    for _ in range(3):
        h0 = time.perf_counter(); rep.run(); th.append((time.perf_counter() - h0) * 1e3); t.append(rep.timings()["total_ms"])
    b3 = verify()
    ms = sum(t) / len(t)
    print(f"lanes {K} delay {D} groups {G}: host enqueue {sum(th) / len(th):.1f} ms, {ms:.1f} ms per lap = {S * P * heads[0]['n_mbs'] / ms / 1e3:.1f} M MB/s; mismatching streams after lap 1/2/5: {b1}/{b2}/{b3}")
    rep.close()
