#!/bin/bash
# quick_groups.sh [label] [groups...]: lock-step value with the streams split into N stream groups (bench.py --groups N)
label=${1:-run}; shift
for g in "$@"; do
timeout 600 python bench.py --groups $g --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant --steps 20 --ramp-seconds 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$label groups=$g', round(d['value']/1e6,1), 'M MB/s', round(d['ms_per_step'],1), 'ms/step')"
done
