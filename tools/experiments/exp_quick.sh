#!/bin/bash
# GPU box: kernel parity (random jobs + bundled streams), then one lock-step timing
set -u
out=gpurun_out/quick; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_jobs.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
for i in 1 2 3; do
if [ $i = 3 ]; then export H264BSDMI_NO_AHEAD=1; echo "no ahead:"; fi
timeout 300 python bench.py --steps 3 --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant ${BENCH_EXTRA:-} 2> $out/err.log | tail -1 > $out/b.json
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/quick/b.json").read())
    r = d["roofline"]["device_ms_per_step"]
    print(round(d["value"]/1e6,1), round(d["ms_per_step"],1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"))
except Exception as e:
    print("failed", e)
P
done
grep -v amdgpu.ids $out/err.log | tail -3
