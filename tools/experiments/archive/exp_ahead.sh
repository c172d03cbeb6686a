#!/bin/bash
set -u
out=gpurun_out/ahead; rm -rf $out; mkdir -p $out
run() {
  name=$1; shift
  echo -n "$name: "
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant 2> $out/err_$name.log | tail -1 > $out/b_$name.json
  python - "$out/b_$name.json" <<'P'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d["roofline"]["device_ms_per_step"]
    print(round(d["value"]/1e6,1), round(d["ms_per_step"],1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"))
except Exception as e:
    print("failed", e)
P
}
run ahead_w12 A=1
run ahead_w8 H264BSDMI_TAIL=0,0,8,0,0,8
run ahead_w8_i12 H264BSDMI_TAIL=0,0,8,0,0,12
run ahead_w10 H264BSDMI_TAIL=0,0,10,0,0,10
run noahead_w12 H264BSDMI_NO_AHEAD=1
run noahead_w8 H264BSDMI_NO_AHEAD=1 H264BSDMI_TAIL=0,0,8,0,0,8
