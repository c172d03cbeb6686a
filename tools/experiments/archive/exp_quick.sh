#!/bin/bash
# GPU box: kernel parity (TESTS, default: random jobs + bundled + synthetic streams), then lock-step timings
set -u
out=gpurun_out/quick; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_gpu_random_jobs.py tests/test_synth_streams.py} -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant ${BENCH_EXTRA:-} 2> $out/err.log | tail -1 > $out/b.json
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
