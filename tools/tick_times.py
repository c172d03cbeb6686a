import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import h264bsd_amd as h
jobs, _, _ = h.capture_stream(open("/root/repo/tests/golden/test_1920x1080.h264", "rb").read())
rep = h.Replay(jobs, n_streams=256)
rep.run(); rep.sync()
acc = {}
for lap in range(2):
    for tick in range(len(jobs)):
        rep.run(tick, 1); rep.sync()
        t = rep.timings()
        for k, v in t.items():
            if isinstance(v, tuple): acc.setdefault(k, np.zeros(len(jobs)))[tick] += v[0] / 2
for k in ("k_frame_dbk", "k_frame_intra", "k_recon_inter", "k_copy"):
    a = acc[k]
    print(k, "sum %.1f ms; I ticks (0, %d): %s; P ticks mean %.3f min %.3f max %.3f" % (a.sum(), int(np.argsort(a)[-2]), np.round(np.sort(a)[-2:], 3), np.sort(a)[:-2].mean(), a.min(), np.sort(a)[-3]))
print("dbk per tick:", np.round(acc["k_frame_dbk"], 2).tolist())
