#!/usr/bin/env python3
"""Where a round of the end-to-end leg goes: Python preparation of the batch call, the batch call itself (parser threads),
h264bsdmiFlushAsync (enqueue of the H2D copies and launches)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import h264bsd_amd as h
L = h.lib()
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
streams, laps = (int(sys.argv[2]) if len(sys.argv) > 2 else 256), (int(sys.argv[3]) if len(sys.argv) > 3 else 2)
threads = L.h264bsdmiSetParserThreads(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
decs = [h.Decoder() for _ in range(streams)]
drv = h.BatchDriver(decs, [data * (laps + 1)] * streams)
t_step = t_flush = t_call = 0.0
import threading
overlap = len(sys.argv) > 4 and sys.argv[4] == "overlap"
flusher = None
orig = L.h264bsdmiDecodePictureBatch
for pic in range(73 * (laps + 1)):
    if pic == 73:
        L.h264bsdmiFlush(); t0 = time.perf_counter(); c0 = time.process_time(); t_step = t_flush = 0.0
    a = time.perf_counter(); drv.step(); b = time.perf_counter()
    if overlap:
        # the enqueueing of round k on a thread of its own, next to the parse of round k + 1 (ctypes releases the GIL)
        if flusher is not None: flusher.join()
        flusher = threading.Thread(target=L.h264bsdmiFlushAsync); flusher.start()
    else:
        L.h264bsdmiFlushAsync()
    c = time.perf_counter()
    t_step += b - a; t_flush += c - b
if flusher is not None: flusher.join()
L.h264bsdmiFlush()
dt = time.perf_counter() - t0
cpu = time.process_time() - c0              # user + system time of every thread of the process
n = 73 * laps
print(f"threads {threads}: {streams * n / dt:.0f} fps; per round {dt / n * 1e3:.2f} ms = step {t_step / n * 1e3:.2f} + FlushAsync {t_flush / n * 1e3:.2f} + rest {(dt - t_step - t_flush) / n * 1e3:.2f}; CPU time {cpu / n * 1e3:.0f} ms per round = {cpu / dt:.1f} CPUs busy, {cpu / (streams * n) * 1e3:.3f} ms per picture")
