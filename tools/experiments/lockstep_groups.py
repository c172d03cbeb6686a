#!/usr/bin/env python3
"""Lock-step set (every stream on the same picture index) under the lane scheduler: G stream groups on lanes of their own, each with its
side streams for k_copy / k_dbk when G <= 2 (engine.hip replay_schedule), against the plain lock-step schedule.  Verified after every lap.
usage: lockstep_groups.py [G ...]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import h264bsd_amd as h
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = "test_1920x1080"
jobs, _, _ = h.capture_stream(open(os.path.join(root, "tests", "golden", name + ".h264"), "rb").read(), copy_elision=True)
golden = json.load(open(os.path.join(root, "tests", "golden", "golden.json")))[name]["frame_checksum64"]
heads = [h.job_header(j) for j in jobs]
S, P = 256, len(jobs)
last_in_slot = {}
for i in range(P): last_in_slot[heads[i]["cur_slot"]] = i
rep = h.Replay(jobs, n_streams=S)
def check():
    return sum(int((rep.checksums(sl) != np.uint64(golden[i])).sum()) for sl, i in last_in_slot.items())
def timed(label, laps=12):
    t_end = time.time() + 3
    while time.time() < t_end: rep.run(); rep.sync()
    t0 = time.perf_counter()
    for _ in range(laps): rep.run()
    rep.sync()
    dt = (time.perf_counter() - t0) / laps
    print(f"{label}: {dt * 1e3:.1f} ms per lap = {S * P * heads[0]['n_mbs'] / dt / 1e6:.1f} M MB/s; mismatching streams {check()}", flush=True)
timed("lock-step")
for a in sys.argv[1:]:
    G = int(a)
    rep.reschedule(offsets=[0] * S, heavy_lanes=0, groups=G)
    timed(f"{G} groups on lanes")
rep.reschedule()
timed("lock-step again")
