#!/usr/bin/env python3
"""Host parser alone (capture mode, frame jobs dropped): milliseconds per 1080p picture on ONE thread, best of N passes over
tests/golden/test_1920x1080.h264.  usage: parse_bench.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
times = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 9):
    d = h.Decoder(capture="discard")
    t0 = time.perf_counter()
    tr = d.decode_stream(data, drain=False)
    times.append((time.perf_counter() - t0) * 1e3 / 73)
    d.close()
times.sort()
print(f"parser, one thread: best {times[0]:.3f} ms per picture, median {times[len(times) // 2]:.3f} ({len(times)} passes of 73 pictures)")
