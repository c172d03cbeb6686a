#!/bin/bash
# A/B of prebuilt library variants on the ARGB leg of the bench (k_convert_tiles)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "$@"; do
  cp gpurun_variants/lib_$v.so h264bsd_amd/lib/libh264bsd_mi355x_bench.so
  echo -n "$v: "
  timeout 400 python bench.py --no-cpu-baseline --no-staggered --no-desync --no-end-to-end --no-groups-variant --no-full-copies-variant --ramp-seconds 1 --steps 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d['argb']; print(round(d['value']/1e6,1), 'argb', round(a['value']/1e6,1), 'k_convert us', round(a['k_convert']['avg_launch_us'],1), 'TB/s', round(a['k_convert']['achieved_GBs']/1e3,2))"
done
