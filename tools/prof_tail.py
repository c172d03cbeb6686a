import sys, ctypes, numpy as np
sys.path.insert(0,'/root/repo')
import h264bsd_amd as h
L=h.lib()
jobs,_,_=h.capture_stream(open('/root/repo/tests/golden/test_1920x1080.h264','rb').read())
rep=h.Replay(jobs, n_streams=256)
rep.run(0,10); rep.sync()
for tick in (10, 20, 40):
    L.h264bsdmiDebugTailProfile(1, None)
    rep.run(tick,1); rep.sync()
    out=np.zeros((16,8),dtype=np.uint64)
    L.h264bsdmiDebugTailProfile(0, ctypes.c_void_p(out.ctypes.data))
    print('tick',tick,'timings',{k:v for k,v in rep.timings().items()})
    print(' cols: setup filter barrier nfilt maxlevel | load V H store (cycles, per wave)')
    for w in (0,7,15): print(' wave',w, out[w,:5], out[w,5], out[w,6], out[w,7]&0xffffffff, out[w,7]>>32)
    if tick==10: rep.run(11,9); rep.sync()
    if tick==20: rep.run(21,19); rep.sync()
