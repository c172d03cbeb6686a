#!/usr/bin/env python3
"""N decoder instances on N threads (the reference's harness loop: one picture pulled after every PIC_RDY), repeated until
a picture differs from the reference's sha256; reports which picture, where (macroblocks), and the library settings."""
import sys, os, json, hashlib, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import h264bsd_amd as h
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "test_640x360"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))[name]
data = open(os.path.join(ROOT, "tests", "golden", name + ".h264"), "rb").read()
wmb, hmb = gold["width_mbs"], gold["height_mbs"]
good = {}            # picture index -> frame (first correct copy seen)
bad = []
lock = threading.Lock()

def worker(tid, rnd):
    d = h.Decoder()
    k = [0]
    def on_pic(frame, pid, idr, nerr):
        i = k[0]; k[0] += 1
        ok = hashlib.sha256(frame.tobytes()).hexdigest() == gold["frame_sha256"][i]
        with lock:
            if ok:
                good.setdefault(i, frame.copy())
            else:
                bad.append((rnd, tid, i, frame.copy()))
    d.decode_stream(data, on_picture=on_pic)
    d.close()

t0 = time.time()
for rnd in range(rounds):
    th = [threading.Thread(target=worker, args=(t, rnd)) for t in range(T)]
    for t in th: t.start()
    for t in th: t.join()
    if bad:
        break
print(f"{name} x {T} threads: {rnd + 1} rounds in {time.time() - t0:.1f} s, {len(bad)} wrong pictures; env",
      {k: v for k, v in os.environ.items() if k.startswith("H264BSDMI")})
W, H = 16 * wmb, 16 * hmb
first = {}
for rnd, tid, i, fr in bad:
    first.setdefault((rnd, tid), i)
for rnd, tid, i, fr in bad[:12]:
    ref = good.get(i)
    if ref is None:
        print("round", rnd, "thread", tid, "picture", i, "(no good copy to compare with)"); continue
    dy = (fr[:W * H] != ref[:W * H]).reshape(hmb, 16, wmb, 16).any(axis=(1, 3))
    mbs = np.argwhere(dy)
    nb = int((fr != ref).sum())
    print("round", rnd, "thread", tid, "picture", i, "first wrong picture of this decoder:", first[(rnd, tid)], "| luma MBs differing:", len(mbs),
          "bytes differing:", nb, "| first MBs (y,x):", mbs[:10].tolist())
    if i == first[(rnd, tid)] and len(mbs):
        y, x = mbs[0]
        a = fr[:W * H].reshape(H, W)[16 * y:16 * y + 16, 16 * x:16 * x + 16].astype(int)
        b = ref[:W * H].reshape(H, W)[16 * y:16 * y + 16, 16 * x:16 * x + 16].astype(int)
        print("  differing samples in that MB (rows x cols):"); print((a != b).astype(int))
        print("  got row:", a[np.argwhere(a != b)[0][0]].tolist()); print("  want   :", b[np.argwhere(a != b)[0][0]].tolist())
