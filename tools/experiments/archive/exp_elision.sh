#!/bin/bash
# GPU box: copy elision — parity (elided replay, product API path), then lock-step timings with and without
set -u
out=gpurun_out/elision; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests/test_copy_elision.py tests/test_gpu_api.py tests/test_c_caller.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
for v in on off on off; do
extra=""; [ $v = off ] && extra="--no-copy-elision"
timeout 300 python bench.py --steps 5 --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant $extra 2> $out/err_$v.log | tail -1 > $out/b_$v.json
python - $v <<'P'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/elision/b_{sys.argv[1]}.json").read())
    r = d["roofline"]["device_ms_per_step"]
    print(sys.argv[1], round(d["value"]/1e6,1), round(d["ms_per_step"],1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"), d["config"]["copy_elision"]["elided"])
except Exception as e:
    print("failed", e)
P
done
timeout 600 python bench.py --steps 20 --no-cpu-baseline 2> $out/err_full.log | tail -1 > $out/full.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/elision/full.json").read())
print("full", round(d["value"]/1e6,1), "stag", round(d["staggered"]["value"]/1e6,1), "argb", round(d["argb"]["value"]/1e6,1),
      {k: round(v["value"]/1e6,1) for k, v in d["desynchronised"].items() if isinstance(v, dict)}, "e2e", round(d["end_to_end"]["fps"]), "err", d["device_errors"])
P
grep -v amdgpu.ids $out/err_on.log | tail -3
