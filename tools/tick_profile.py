#!/usr/bin/env python3
"""Per-tick kernel times of the lock-step replay (256 x test_1920x1080): which ticks are expensive in which kernel.
usage (GPU box): python tools/tick_profile.py [streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
jobs, _, _ = h264bsd_amd.capture_stream(data)
heads = [h264bsd_amd.job_header(j) for j in jobs]
rep = h264bsd_amd.Replay(jobs, n_streams=n)
for _ in range(2):
    rep.run(); rep.sync()
K = h264bsd_amd.Replay.KERNELS
tot = {k: 0.0 for k in K}
rows = []
for lap in range(2):
    for i in range(len(jobs)):
        rep.run(i, 1)
        t = rep.timings()
        if lap:
            rows.append((i, heads[i]["n_intra"], heads[i]["n_dbk"], heads[i]["n_copy"], heads[i]["n_gen"], [t[k][0] or 0.0 for k in K], t["total_ms"]))
            for k in K:
                tot[k] += t[k][0] or 0.0
print("tick n_intra n_dbk n_copy n_gen | " + " ".join(K) + " | total")
for r in rows:
    if os.environ.get("ALL") or r[0] < 6 or r[0] in (39, 40, 41, 42) or r[0] % 10 == 0:
        print(r[0], r[1], r[2], r[3], r[4], "|", " ".join(f"{v:.3f}" for v in r[5]), "|", f"{r[6]:.3f}")
print("sum", {k: round(v, 1) for k, v in tot.items()})
rep.close()
