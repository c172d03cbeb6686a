#!/bin/bash
# GPU box: lock-step + desynchronised legs of library variants
for v in "$@"; do
  H264BSD_VARIANT=$v timeout 900 python bench.py --no-cpu-baseline --no-staggered --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant --steps ${STEPS:-10} --ramp-seconds 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ds=d['desynchronised']
print('${v:-default}', round(d['value']/1e6,1), 'M MB/s;', {k: round(v['value']/1e6,1) for k,v in ds.items() if isinstance(v, dict)}, 'errors', d['device_errors'])"
done
