#!/bin/bash
# quick_bench.sh [label]: the lock-step bench line only (no side legs), printed as MB/s + device ms per step per kernel
label=${1:-run}
timeout 600 python bench.py --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant --steps 20 --ramp-seconds 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['device_ms_per_step']
print('$label', round(d['value']/1e6,1), 'M MB/s', {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))})"
