#!/usr/bin/env python3
"""End-to-end throughput THROUGH THE DROP-IN C API (host parse + H2D of the frame jobs + kernels [+ D2H]):
N decoder instances driven round-robin by T host threads; every round each instance parses one picture
(h264bsdDecode until PIC_RDY), then ONE h264bsdmiFlush() reconstructs the N queued pictures as a single tick.
This is the PCIe-inclusive figure quoted in DESIGN.md; bench.py's `value` is the HBM-resident replay.

usage: e2e_bench.py [--streams 256] [--threads 64] [--pull K] [--pictures 73]
  --pull K : additionally fetch the finished picture of K instances per round to host memory (D2H 3.1 MB each)"""
import argparse, ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--threads", type=int, default=64)
ap.add_argument("--pull", type=int, default=0)
ap.add_argument("--pictures", type=int, default=73)
ap.add_argument("--native", action="store_true",
                help="drive the instances with h264bsdmiDecodePictureBatch (parser threads inside the library) and "
                     "h264bsdmiFlushAsync, so that parsing round k+1 overlaps the reconstruction of round k")
args = ap.parse_args()

data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
L = h.lib()
N, T = args.streams, args.threads
decs = [h.Decoder() for _ in range(N)]
if args.native:
    # the stream starts with an IDR, so it can be fed again and again to the same instances; the first pass warms up
    # (pinned staging buffers are allocated on first use) and is not timed
    T = L.h264bsdmiSetParserThreads(T)
    loops = 1 + max(1, (args.pictures + 72) // 73)
    drv = h.BatchDriver(decs, [data * loops] * N)
    t_parse = t_gpu = t_pull = 0.0
    timed = 0
    for pic in range(73 * loops):
        if pic == 73:
            assert L.h264bsdmiFlush() == 0
            t0 = time.perf_counter()
        a = time.perf_counter()
        ready = drv.step()
        assert len(ready) == N
        b = time.perf_counter()
        assert L.h264bsdmiFlushAsync() == 0
        c = time.perf_counter()
        for k in range(args.pull):
            assert decs[k].next_output_picture() is not None
        d = time.perf_counter()
        if pic >= 73:
            t_parse += b - a; t_gpu += c - b; t_pull += d - c; timed += 1
    a = time.perf_counter()
    assert L.h264bsdmiFlush() == 0
    t_gpu += time.perf_counter() - a
    elapsed = time.perf_counter() - t0
    pics = N * timed
    print(f"native: streams {N} parser threads {T} (of {os.cpu_count()} CPUs) pull {args.pull}: {pics / elapsed:.0f} fps = "
          f"{pics * 8160 / elapsed / 1e6:.1f} M MB/s end to end over {timed} rounds (parse {t_parse:.2f} s, enqueue+final wait "
          f"{t_gpu:.2f} s, D2H {t_pull:.2f} s of {elapsed:.2f} s)")
    for d_ in decs:
        d_.close()
    sys.exit(0)
bufs = [ctypes.create_string_buffer(data, len(data)) for _ in range(N)]
offs = [0] * N
barrier = threading.Barrier(T + 1)
stop = False

def worker(t):
    mine = range(t, N, T)
    while True:
        barrier.wait()
        if stop:
            return
        for k in mine:
            while offs[k] < len(data):
                r, rb = decs[k].decode(ctypes.addressof(bufs[k]) + offs[k], len(data) - offs[k])
                offs[k] += rb
                if r == h.H264BSD_PIC_RDY:
                    break
                assert r < h.H264BSD_ERROR
        barrier.wait()

threads = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(T)]
for th in threads:
    th.start()
t_parse = t_gpu = t_pull = 0.0
t0 = time.perf_counter()
for pic in range(args.pictures):
    a = time.perf_counter()
    barrier.wait(); barrier.wait()                      # all instances parse one picture
    b = time.perf_counter()
    assert L.h264bsdmiFlush() == 0                      # one batched tick
    c = time.perf_counter()
    for k in range(args.pull):
        assert decs[k].next_output_picture() is not None
    d = time.perf_counter()
    t_parse += b - a; t_gpu += c - b; t_pull += d - c
elapsed = time.perf_counter() - t0
stop = True
barrier.wait()
pics = N * args.pictures
print(f"streams {N} threads {T} pull {args.pull}: {pics / elapsed:.0f} fps = {pics * 8160 / elapsed / 1e6:.1f} M MB/s end to end "
      f"(parse {t_parse:.2f} s, H2D+kernels {t_gpu:.2f} s, D2H {t_pull:.2f} s of {elapsed:.2f} s)")
for d_ in decs:
    d_.close()
