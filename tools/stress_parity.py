#!/usr/bin/env python3
"""Soak test of the per-picture schedulers (hand-over inside a workgroup without a wait for the stores): the
256-stream 1080p replay, every picture of every stream checked against the reference's checksum after EVERY tick, lap
after lap, in the lock-step schedule and in the banded ones; then whole laps without a host synchronisation between ticks.  usage: stress_parity.py [laps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import h264bsd_amd as h
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
name = "test_1920x1080"
g = json.load(open(os.path.join(root, "golden.json")))[name]["frame_checksum64"]
jobs, _, _ = h.capture_stream(open(os.path.join(root, name + ".h264"), "rb").read(), copy_elision=True)
heads = [h.job_header(j) for j in jobs]
laps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S, n = 256, len(jobs)
rep = h.Replay(jobs, n_streams=S)
t0 = time.time(); ticks = bad = 0
for lap in range(laps):
    for i in range(n):
        rep.run(i, 1)
        sums = rep.checksums(heads[i]["cur_slot"])
        ticks += 1
        if not (sums == np.uint64(g[i])).all():
            bad += 1; print(f"lap {lap} picture {i}: {(sums != np.uint64(g[i])).sum()} streams differ")
print(f"lock-step: {ticks} ticks x {S} streams verified, {bad} bad, {time.time() - t0:.0f} s")
# few streams: the pictures are split into row bands (hand-over between workgroups), also soaked
for S2 in (4, 16):
    rep.close()
    rep = h.Replay(jobs, n_streams=S2)
    t0 = time.time(); ticks = 0
    for lap in range(laps):
        for i in range(n):
            rep.run(i, 1)
            sums = rep.checksums(heads[i]["cur_slot"])
            ticks += 1
            if not (sums == np.uint64(g[i])).all():
                bad += 1; print(f"{S2} streams, lap {lap} picture {i}: differ")
    print(f"{S2} streams (row bands): {ticks} ticks verified, {bad} bad in total, {time.time() - t0:.0f} s")
rep.close()
# whole laps at full speed (no host synchronisation between the ticks: k_copy and k_dbk of tick i + 1 run on their own HIP streams
# right behind tick i's k_frame_dbk): after every lap, the picture every frame-buffer slot was written with last
rep = h.Replay(jobs, n_streams=S)
last_in_slot = {}
for i in range(n):
    last_in_slot[heads[i]["cur_slot"]] = i
t0 = time.time(); checked = 0
for lap in range(laps):
    rep.run(); rep.sync()
    for slot, i in last_in_slot.items():
        sums = rep.checksums(slot)
        checked += 1
        if not (sums == np.uint64(g[i])).all():
            bad += 1; print(f"full-speed lap {lap}: slot {slot} (picture {i}): {(sums != np.uint64(g[i])).sum()} streams differ")
print(f"full-speed laps: {laps} laps of {n} ticks x {S} streams, {checked} slot checks ({len(last_in_slot)} slots per lap), {bad} bad in total, {time.time() - t0:.0f} s")
rep.close()
print("device errors", h.device_errors())
sys.exit(1 if bad else 0)
