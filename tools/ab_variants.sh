#!/bin/bash
# A/B timing of prebuilt library variants (gpurun_variants/lib_<name>.so) on ONE box: usage ab_variants.sh A B A B
for v in "$@"; do
  cp gpurun_variants/lib_$v.so h264bsd_amd/lib/libh264bsd_mi355x_bench.so
  echo -n "$v: "
  timeout 400 python bench.py --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --ramp-seconds 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['device_ms_per_step']; print(round(d['value']/1e6,1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))})"
done
