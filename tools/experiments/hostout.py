#!/usr/bin/env python3
"""GPU box: the end-to-end leg with host output, one mode per process (H264BSD_VARIANT picks the library).
usage: hostout.py barrier|overlap|halves|quarters|none [laps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
data = open(os.path.join(bench.ROOT, "tests", "golden", "test_1920x1080.h264"), "rb").read()


def timeline():
    """where does a round of the combined call go?  batch call (pull + parse) / FlushAsync call, averaged over the timed lap"""
    import time
    import h264bsd_amd as h
    L = h.lib()
    decs = [h.Decoder() for _ in range(256)]
    threads = L.h264bsdmiSetParserThreads(0)
    drv = h.BatchDriver(decs, [data * 2] * 256)
    t_batch = t_flush = 0.0
    for pic in range(146):
        if pic == 73:
            assert L.h264bsdmiFlush() == 0
            t_batch = t_flush = 0.0
            t0 = time.perf_counter()
        a = time.perf_counter()
        drv.step(pull=True)
        b = time.perf_counter()
        assert L.h264bsdmiFlushAsync() == 0
        c = time.perf_counter()
        t_batch += b - a; t_flush += c - b
    dt = time.perf_counter() - t0
    print(os.environ.get("H264BSD_VARIANT", "default"), f"timeline: {256 * 73 / dt:.0f} fps, per round {dt / 73 * 1e3:.1f} ms = batch call {t_batch / 73 * 1e3:.1f} + FlushAsync call {t_flush / 73 * 1e3:.1f} ms, {threads} threads")
    h.pull_batch(decs)
    L.h264bsdmiFlush()
    for d in decs:
        d.close()


mode = sys.argv[1]
if mode == "timeline":
    timeline()
    sys.exit(0)
laps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
leg = bench.end_to_end(data, 256, 0, laps=laps, pull={"barrier": "barrier", "overlap": "whole", "none": False, "halves": "halves", "quarters": "quarters"}[mode])
print(os.environ.get("H264BSD_VARIANT", "default"), mode, f"{leg['fps']:.0f} fps, {leg['seconds']:.2f} s, {leg['parser_threads']} threads, device errors {leg['device_errors']}")
