#!/bin/bash
# GPU box: what row bands buy when few streams share the GPU (bands on / off), and the full default line
set -u
out=gpurun_out/small; rm -rf $out; mkdir -p $out
for n in 4 16 32 64 128; do
  for b in 320 0; do
    echo -n "streams $n band budget $b: "
    H264BSDMI_BAND_BUDGET=$b timeout 300 python bench.py --streams $n --steps 5 --warmup 1 --ramp-seconds 1 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant 2> $out/err.log | tail -1 > $out/b.json
    python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/small/b.json").read())
    r = d["roofline"]["device_ms_per_step"]
    print(round(d["value"]/1e6,1), "M MB/s", round(d["ms_per_step"],2), "ms/step", {k: round(v,2) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"))
except Exception as e:
    print("failed", e)
P
  done
done
echo "== full default"
timeout 900 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2> $out/err_full.log | tail -1 > $out/full.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/small/full.json").read())
print(round(d["value"]/1e6,1), "stag", round(d["staggered"]["value"]/1e6,1), "argb", round(d["argb"]["value"]/1e6,1), {k: round(v["value"]/1e6,1) for k,v in d["desynchronised"].items() if isinstance(v, dict)}, "e2e", round(d["end_to_end"]["fps"]), "err", d["device_errors"])
P
