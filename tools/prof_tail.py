#!/usr/bin/env python3
"""(needs a library built with -DH264K_TAIL_PROFILE: tools/experiments/build_variant.sh prof -DH264K_TAIL_PROFILE, then H264BSD_VARIANT=prof)
Cycle accounting of k_frame_dbk's dataflow loop (workgroup 0 = stream 0, per wavefront) for a few ticks of the
256-stream 1080p replay: how much of a wave's life is spent with nothing ready / filtering / waiting for stores."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h
L = h.lib()
jobs, _, _ = h.capture_stream(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read())
rep = h.Replay(jobs, n_streams=256)
done = 0
for tick in (0, 9, 22, 28, 41, 60):
    if tick > done:
        rep.run(done, tick - done); rep.sync()
    L.h264bsdmiDebugTailProfile(1, None)
    rep.run(tick, 1); rep.sync()
    done = tick + 1
    buf = np.zeros(16 * 16 + 16 * 8, dtype=np.uint64); out = buf[:256].reshape(16, 16); intra = buf[256:].reshape(16, 8)
    L.h264bsdmiDebugTailProfile(0, ctypes.c_void_p(buf.ctypes.data))
    t = rep.timings()
    tot = out[:, 4].astype(float).sum()
    it = float(intra[:, :3].sum()) or 1.0
    print(f"tick {tick}: k_frame_intra {t['k_frame_intra'][0]:.3f} ms; MBs by WG0: {int(intra[:,3].sum())}; idle {intra[:,0].sum()/it:.0%} work {intra[:,1].sum()/it:.0%} "
          f"store-wait+release {intra[:,2].sum()/it:.0%}; cycles per MB: work {intra[:,1].sum()/max(1,intra[:,3].sum()):.0f} (of which the record round trip {intra[:,4].sum()/max(1,intra[:,3].sum()):.0f}) release {intra[:,2].sum()/max(1,intra[:,3].sum()):.0f}; inside intra_mb per MB: residual + staging {intra[:,5].sum()/max(1,intra[:,3].sum()):.0f}, luma prediction {intra[:,6].sum()/max(1,intra[:,3].sum()):.0f}, chroma + stores {intra[:,7].sum()/max(1,intra[:,3].sum()):.0f}")
    print(f"tick {tick}: k_frame_dbk {t['k_frame_dbk'][0]:.3f} ms; MBs filtered by WG0: {int(out[:,3].sum())}; "
          f"idle {out[:,0].sum()/tot:.0%}  filter {out[:,1].sum()/tot:.0%}  store-wait+release {out[:,2].sum()/tot:.0%}; "
          f"active waves {int((out[:,5] > 0).sum())}; wave cycles {tot/max(1,(out[:,5] > 0).sum()):.0f} avg; steps/wave {out[:,5].sum()/max(1,(out[:,5] > 0).sum()):.0f}, MBs per step {out[:,3].sum()/max(1,out[:,5].sum()):.2f}, "
          f"cycles per step {out[:,1].sum()/max(1,out[:,5].sum()):.0f} = load {out[:,8].sum()/max(1,out[:,5].sum()):.0f} + V {out[:,9].sum()/max(1,out[:,5].sum()):.0f} "
          f"+ H {out[:,10].sum()/max(1,out[:,5].sum()):.0f} + store {out[:,11].sum()/max(1,out[:,5].sum()):.0f} (LDS reads done at {out[:,12].sum()/max(1,out[:,5].sum()):.0f}, own tile stored at {out[:,13].sum()/max(1,out[:,5].sum()):.0f}, neighbours at {out[:,14].sum()/max(1,out[:,5].sum()):.0f}); "
          f"per step outside deblock_mb: claim + load issue {(out[:,1].sum()-out[:,8:12].sum())/max(1,out[:,5].sum()):.0f}, release {out[:,2].sum()/max(1,out[:,5].sum()):.0f}")
