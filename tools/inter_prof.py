#!/usr/bin/env python3
"""(needs a library built with -DH264K_INTER_PROFILE: tools/experiments/build_variant.sh iprof -DH264K_INTER_PROFILE, then H264BSD_VARIANT=iprof)
Where a wavefront of k_recon_inter<0> spends its life: cycles from its first instruction to the list entry's arrival, to the staged
windows, through prediction, residual and store (every 64th workgroup of every picture reports)."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h
L = h.lib()
jobs, _, _ = h.capture_stream(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read())
rep = h.Replay(jobs, n_streams=256)
rep.run(); rep.sync()
buf = np.zeros(24, dtype=np.uint64)
L.h264bsdmiDebugReadCounters(ctypes.c_void_p(buf.ctypes.data))
rep.run(); rep.sync()
L.h264bsdmiDebugReadCounters(ctypes.c_void_p(buf.ctypes.data))
n = float(buf[5]) or 1.0
names = ["start -> list entry here", "-> windows + coefficients staged", "-> prediction done", "-> residual added", "-> stored"]
print(f"{int(n)} sampled wavefronts ({100 * float(buf[6]) / n:.0f} % with coefficients); cycles per wavefront: " + "; ".join(f"{nm} {float(buf[k]) / n:.0f}" for k, nm in enumerate(names)) + f"; total {float(buf[:5].sum()) / n:.0f}")
