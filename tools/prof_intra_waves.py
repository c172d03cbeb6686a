import sys, os, ctypes, numpy as np
sys.path.insert(0, "/root/repo")
import h264bsd_amd as h
L = h.lib()
jobs, _, _ = h.capture_stream(open("/root/repo/tests/golden/test_1920x1080.h264", "rb").read())
rep = h.Replay(jobs, n_streams=256)
done = 0
for tick in (0, 9, 40):
    if tick > done:
        rep.run(done, tick - done); rep.sync()
    L.h264bsdmiDebugTailProfile(1, None)
    rep.run(tick, 1); rep.sync()
    done = tick + 1
    buf = np.zeros(16 * 16 + 16 * 8, dtype=np.uint64); intra = buf[256:].reshape(16, 8)
    L.h264bsdmiDebugTailProfile(0, ctypes.c_void_p(buf.ctypes.data))
    t = rep.timings()
    print(f"tick {tick}: k_frame_intra {t['k_frame_intra'][0]:.3f} ms")
    for w in range(16):
        o = intra[w].astype(float)
        if o[3] == 0: continue
        tot = o[0] + o[1] + o[2]
        print(f"  wave {w:2d}: MBs {o[3]:.0f}; idle {o[0]/tot:.0%} work {o[1]/tot:.0%} release {o[2]/tot:.0%}; total {tot:.0f} cyc; per MB: work {o[1]/o[3]:.0f} (record wait {o[4]/o[3]:.0f}, residual+staging {o[5]/o[3]:.0f}, luma pred {o[6]/o[3]:.0f}, chroma+stores {o[7]/o[3]:.0f}) release {o[2]/o[3]:.0f}")
