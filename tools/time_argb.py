"""GPU box: kernel times of the lock-step replay with the config-3 conversion inside the run (hosted by k_frame_dbk / launches only).
tools/time_argb.py [laps [conversion wavefronts (0 = default) [hosting 0/1]]]"""
import sys, os
sys.path.insert(0, os.getcwd())
import h264bsd_amd
laps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
conv_waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hosting = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
data = open("tests/golden/test_1920x1080.h264", "rb").read()
jobs, _, _ = h264bsd_amd.capture_stream(data)
rep = h264bsd_amd.Replay(jobs, n_streams=256)
rep.run(); rep.sync()
for mode in ("off", "on"):
    if mode == "on":
        rep.set_convert(h264bsd_amd.FMT_BGRA, hosting=hosting, conv_waves=conv_waves)
    else:
        rep.set_convert(-1)
    rep.run(); rep.sync()
    tot, cms, cn = {}, 0.0, 0
    for _ in range(laps):
        rep.run(); t = rep.timings()
        for k, v in t.items():
            tot[k] = tot.get(k, 0) + (v[0] if isinstance(v, tuple) else v) / laps
        if mode == "on":
            ms, n = rep.convert_timings(); cms += ms / laps; cn += n / laps
    print(f"convert {mode}:", {k: round(v, 1) for k, v in tot.items()}, f"conversion launches per lap {cn:.0f}, {cms:.1f} ms" if mode == "on" else "")
