#!/bin/bash
# GPU box: lock-step + staggered legs of bench.py under different heavy-picture budgets: staggered.sh "<H264BSDMI_HEAVY_BUDGET>[:<H264BSDMI_TAIL>]" ...
for spec in "$@"; do IFS=: read hb tail <<< "$spec"; echo -n "heavy_budget=${hb:-default} tail=${tail:-default}: "
  env ${hb:+H264BSDMI_HEAVY_BUDGET=$hb} ${tail:+H264BSDMI_TAIL=$tail} timeout 600 python bench.py --no-cpu-baseline --no-argb --no-desync --no-end-to-end --no-groups-variant --no-full-copies-variant --steps ${STEPS:-20} --ramp-seconds 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['staggered']
print(round(d['value']/1e6,1), 'M MB/s; staggered', round(s['value']/1e6,1), 'M MB/s =', round(s['value']/d['value'],3), 'of lock-step;', {k: round(v,1) for k,v in s['device_ms_per_step'].items()})"
done
