import sys, os, ctypes, numpy as np
sys.path.insert(0, "/root/repo")
import h264bsd_amd as h
L = h.lib()
jobs, _, _ = h.capture_stream(open("/root/repo/tests/golden/test_1920x1080.h264", "rb").read())
rep = h.Replay(jobs, n_streams=256)
done = 0
for tick in (9, 22, 41):
    if tick > done:
        rep.run(done, tick - done); rep.sync()
    L.h264bsdmiDebugTailProfile(1, None)
    rep.run(tick, 1); rep.sync()
    done = tick + 1
    buf = np.zeros(16 * 16 + 16 * 8, dtype=np.uint64); out = buf[:256].reshape(16, 16)
    L.h264bsdmiDebugTailProfile(0, ctypes.c_void_p(buf.ctypes.data))
    t = rep.timings()
    print(f"tick {tick}: k_frame_dbk {t['k_frame_dbk'][0]:.3f} ms")
    for w in range(16):
        o = out[w].astype(float)
        if o[5] == 0: continue
        print(f"  wave {w:2d}: total {o[4]:.0f} cyc; idle {o[0]/o[4]:.0%} work {o[1]/o[4]:.0%} release {o[2]/o[4]:.0%}; steps {o[5]:.0f} MBs {o[3]:.0f} ({o[3]/o[5]:.2f}/step); per step: work {o[1]/o[5]:.0f} = loadwait {o[8]/o[5]:.0f} V {o[9]/o[5]:.0f} H {o[10]/o[5]:.0f} store {o[11]/o[5]:.0f} other {(o[1]-o[8:12].sum())/o[5]:.0f} (queue read at {o[12]/o[5]:.0f}, loads issued at {o[13]/o[5]:.0f}); release {o[2]/o[5]:.0f}")
