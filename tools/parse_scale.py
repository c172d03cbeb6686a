#!/usr/bin/env python3
"""The host parser at scale, without a device: N capture-mode decoders (frame jobs dropped) advanced picture by picture on the
library's parser threads (h264bsdmiDecodePictureBatch), like the end-to-end leg of bench.py does with device-bound decoders.
Prints milliseconds per round of N pictures, the share of it spent inside the batch call, and what the single-thread rate
(tools/parse_bench.py) would give on the same number of CPUs.  usage: parse_scale.py [threads] [streams] [laps] [elide]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h
L = h.lib()
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
threads = L.h264bsdmiSetParserThreads(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 256
laps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
elide = len(sys.argv) > 4 and sys.argv[4] == "elide"
decs = [h.Decoder(capture="discard", copy_elision=elide) for _ in range(streams)]
drv = h.BatchDriver(decs, [data * (laps + 1)] * streams)
t_call = 0.0
real = L.h264bsdmiDecodePictureBatch
def timed(*a):
    global t_call
    t = time.perf_counter(); r = real(*a); t_call += time.perf_counter() - t
    return r
drv.L = type("Shim", (), {"__getattr__": lambda self, k: timed if k == "h264bsdmiDecodePictureBatch" else getattr(L, k)})()
for pic in range(73 * (laps + 1)):
    if pic == 73:
        t0 = time.perf_counter(); t_call = 0.0
    drv.step()
dt = time.perf_counter() - t0
n = 73 * laps
try: quota = len(os.sched_getaffinity(0))
except Exception: quota = os.cpu_count()
print(f"threads {threads}, {streams} streams: {dt / n * 1e3:.2f} ms per round ({t_call / n * 1e3:.2f} inside the batch call), "
      f"{streams * n / dt:.0f} pictures/s, {dt / n * 1e3 * min(threads, quota) / streams:.3f} thread-ms per picture")
