#!/bin/bash
# GPU box: lock-step + config-3 (BGRA conversion inside the timed region) legs of library variants; ARGB_FLAGS=--argb-no-hosting for launches only
for v in "$@"; do
  H264BSD_VARIANT=$v timeout 600 python bench.py --no-cpu-baseline --no-staggered --no-desync --no-end-to-end --no-groups-variant --no-full-copies-variant --steps ${STEPS:-20} --ramp-seconds 2 $ARGB_FLAGS 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); a=d['argb']
print('${v:-default}', round(d['value']/1e6,1), 'M MB/s; argb', round(a['value']/1e6,1), 'M MB/s', round(a['ms_per_step'],1), 'ms per step, k_convert', round(a['k_convert']['avg_launch_us']), 'us per launch,', a['conversion'])"
done
