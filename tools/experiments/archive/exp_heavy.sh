#!/bin/bash
set -u
out=gpurun_out/heavy; rm -rf $out; mkdir -p $out
one() {
  name=$1; shift
  env "$@" timeout 400 python bench.py --steps 5 --side-steps 5 --warmup 1 --ramp-seconds 2 --no-cpu-baseline ${DESYNC:---no-desync} --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant 2> $out/err_$name.log | tail -1 > $out/b.json
  python - "$name" <<'P'
import json, sys
try:
    d = json.loads(open("gpurun_out/heavy/b.json").read())
    st = d["staggered"]; ds = d.get("desynchronised")
    print(sys.argv[1], "lock-step", round(d["value"]/1e6,1), "staggered", round(st["value"]/1e6,1), round(st["vs_lock_step"],3), {k: round(v,1) for k,v in st["device_ms_per_step"].items()},
          {k: round(v["value"]/1e6,1) for k, v in ds.items() if isinstance(v, dict)} if ds else "", "err", d.get("device_errors"))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
one hb64 X=1
one hb128 H264BSDMI_HEAVY_BUDGET=128
one hb256 H264BSDMI_HEAVY_BUDGET=256
one hb64 X=1
