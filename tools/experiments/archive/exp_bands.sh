#!/bin/bash
# GPU box: parity of the row-band kernels, then lock-step timings for a few band / wavefront configurations
# (H264BSDMI_TAIL = dbk rows light, heavy, waves, intra rows light, heavy, waves).
set -u
out=gpurun_out/bands; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_random_jobs.py tests/test_gpu_parity.py -x -q -m gpu > $out/tests.log 2>&1
tail -5 $out/tests.log
for cfg in "0,0,12,0,0,12" "17,9,4,0,9,4" "0,0,4,0,0,4" "17,17,4,0,17,4" "9,9,4,0,9,4" "17,9,6,0,9,6" "34,17,4,0,17,4" "17,9,3,0,9,3"; do
  echo "== $cfg"
  H264BSDMI_TAIL=$cfg timeout 300 python bench.py --steps 3 --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end 2> $out/err_$cfg.log | tail -1 > $out/b_$cfg.json
  python - "$out/b_$cfg.json" <<'P'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d["roofline"]["device_ms_per_step"]
    print(round(d["value"]/1e6,1), "groups4", round(d.get("lock_step_4_stream_groups",{}).get("value",0)/1e6,1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"))
except Exception as e:
    print("failed", e)
P
  tail -2 "$out/err_$cfg.log"
done
