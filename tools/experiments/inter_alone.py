#!/usr/bin/env python3
"""How much do k_copy / k_recon_inter lose to k_dbk running next to them?  The lock-step replay with and without the
deblocking stage (stages 3 = reconstruction only: the pictures are then not the reference's, only the timings matter)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import h264bsd_amd as h
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
jobs, _, _ = h.capture_stream(data, copy_elision=True)
rep = h.Replay(jobs, n_streams=256)
for stages in (7, 3, 7, 3):
    rep.set_stages(stages)
    rep.run(); rep.sync()
    acc = {}
    for _ in range(3):
        rep.run()
        t = rep.timings()
        for k in h.Replay.KERNELS:
            acc[k] = acc.get(k, 0.0) + (t[k][0] or 0.0) / 3
    print("stages", stages, {k: round(v, 1) for k, v in acc.items()}, "total", round(t["total_ms"], 1))
rep.close()
