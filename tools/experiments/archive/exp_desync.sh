#!/bin/bash
set -u
out=gpurun_out/desync; rm -rf $out; mkdir -p $out
run() {
  name=$1; shift
  echo -n "$name: "
  env "$@" timeout 600 python bench.py --steps 10 --warmup 1 --ramp-seconds 1 --no-cpu-baseline --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant 2> $out/err_$name.log | tail -1 > $out/b_$name.json
  python - "$out/b_$name.json" <<'P'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read())
    print(round(d["value"]/1e6,1), "stag", round(d["staggered"]["value"]/1e6,1), {k: round(v["value"]/1e6,1) for k,v in d["desynchronised"].items() if isinstance(v, dict)}, "err", d.get("device_errors"))
except Exception as e:
    print("failed", e)
P
}
run hb64 A=1
run hb128 H264BSDMI_HEAVY_BUDGET=128
run hb256_r4 H264BSDMI_HEAVY_BUDGET=256 H264BSDMI_TAIL=17,4,12,0,4,12
run hb256_r6 H264BSDMI_HEAVY_BUDGET=256 H264BSDMI_TAIL=17,6,12,0,6,12
run hb512_r2 H264BSDMI_HEAVY_BUDGET=512 H264BSDMI_TAIL=17,2,12,0,2,12
