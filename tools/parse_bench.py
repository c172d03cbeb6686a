#!/usr/bin/env python3
"""Host parser throughput on its own (no GPU): N capture-mode instances of the bundled 1080p stream advanced
picture by picture by h264bsdmiDecodePictureBatch on T parser threads; frame jobs are built and dropped.
usage: parse_bench.py [--streams 256] [--threads 1,8,32,64,128]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--threads", default="1,8,32,64,128")
ap.add_argument("--stream", default="test_1920x1080")
args = ap.parse_args()
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", args.stream + ".h264"), "rb").read()
L = h.lib()
for t in [int(x) for x in args.threads.split(",")]:
    n = args.streams if t > 1 else max(4, args.streams // 32)
    T = L.h264bsdmiSetParserThreads(t)
    decs = [h.Decoder(capture="discard") for _ in range(n)]
    drv = h.BatchDriver(decs, [data] * n)
    t0 = time.perf_counter()
    rounds = 0
    while drv.step():
        rounds += 1
    dt = time.perf_counter() - t0
    print(f"threads {t:3d} (pool has {T}): {n} streams x {rounds} pictures in {dt:.2f} s = {n * rounds / dt:8.0f} pictures/s "
          f"({dt / rounds * 1e3:.1f} ms per round, {dt * min(t, n) / (n * rounds) * 1e3:.2f} ms per picture and thread)", flush=True)
    for d in decs:
        d.close()
