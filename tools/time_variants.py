import sys, json, os
sys.path.insert(0, os.getcwd())
import h264bsd_amd
data = open("tests/golden/test_1920x1080.h264","rb").read()
jobs,_,_ = h264bsd_amd.capture_stream(data)
rep = h264bsd_amd.Replay(jobs, n_streams=256)
rep.run(); rep.sync()
tot = {}
for _ in range(2):
    rep.run(); t = rep.timings()
    for k,v in t.items():
        tot[k] = tot.get(k,0) + (v[0] if isinstance(v,tuple) else v)/2
print({k: round(v,1) for k,v in tot.items()})
