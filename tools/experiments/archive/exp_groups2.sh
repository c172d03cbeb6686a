#!/bin/bash
# GPU box: lock-step timings, stream groups x wavefronts per per-picture workgroup x row bands
set -u
out=gpurun_out/groups2; rm -rf $out; mkdir -p $out
run() {  # name, env..., -- bench args
  name=$1; shift
  echo "== $name"
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant $EXTRA 2> $out/err_$name.log | tail -1 > $out/b_$name.json
  python - "$out/b_$name.json" <<'P'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d["roofline"]["device_ms_per_step"]
    print(round(d["value"]/1e6,1), round(d["ms_per_step"],1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"))
except Exception as e:
    print("failed", e)
P
  grep -v amdgpu.ids "$out/err_$name.log" | tail -2
}
EXTRA="" run base A=1
EXTRA="--groups 2" run g2_w12_nb H264BSDMI_BAND_BUDGET=0
EXTRA="--groups 2" run g2_w8_nb H264BSDMI_BAND_BUDGET=0 H264BSDMI_TAIL=0,0,8,0,0,8
EXTRA="--groups 4" run g4_w8_nb H264BSDMI_BAND_BUDGET=0 H264BSDMI_TAIL=0,0,8,0,0,8
EXTRA="--groups 4" run g4_w6_nb H264BSDMI_BAND_BUDGET=0 H264BSDMI_TAIL=0,0,6,0,0,6
EXTRA="--groups 4" run g4_w8_i12_nb H264BSDMI_BAND_BUDGET=0 H264BSDMI_TAIL=0,0,8,0,0,12
EXTRA="--groups 8" run g8_w8_nb H264BSDMI_BAND_BUDGET=0 H264BSDMI_TAIL=0,0,8,0,0,8
EXTRA="--groups 4" run g4_w12_bands A=1
EXTRA="--groups 4" run g4_w8_bands H264BSDMI_TAIL=17,9,8,0,9,8
EXTRA="--groups 4" run g4_w4_bands H264BSDMI_TAIL=17,9,4,0,9,4
EXTRA="--groups 8" run g8_w6_bands H264BSDMI_TAIL=17,9,6,0,9,6
echo "== full default (desync legs)"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-argb 2> $out/err_full.log | tail -1 > $out/full.json
echo "== full, bands off"
H264BSDMI_BAND_BUDGET=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-argb 2> $out/err_full_nb.log | tail -1 > $out/full_nb.json
python - <<'P'
import json
for n in ("full", "full_nb"):
    try:
        d = json.loads(open(f"gpurun_out/groups2/{n}.json").read())
        print(n, round(d["value"]/1e6,1), "stag", round(d["staggered"]["value"]/1e6,1), {k: round(v["value"]/1e6,1) for k,v in d["desynchronised"].items() if isinstance(v, dict)}, "err", d["device_errors"])
    except Exception as e:
        print(n, "failed", e)
P
