#!/bin/bash
# GPU box: the quick lock-step bench line (verified, copy elision on) of library variants, in the order given: ab.sh "" old "" old
for v in "$@"; do
  H264BSD_VARIANT=$v timeout 600 python bench.py --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant --steps ${STEPS:-20} --ramp-seconds 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['device_ms_per_step']
print('${v:-default}', round(d['value']/1e6,1), 'M MB/s', {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))})"
done
