#!/bin/bash
# GPU box: the quick lock-step bench line with the two per-picture kernels fused (k_frame_tail) and as two launches, alternating:
# fuse_ab.sh [H264BSDMI_TAIL_INTRA_WAVES values ...]
run() {
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant --steps ${STEPS:-20} --ramp-seconds 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['device_ms_per_step']
print('$*', round(d['value']/1e6,1), 'M MB/s', {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, 'errors', d['device_errors'])"
}
for rep in 1 2; do
  run H264BSDMI_FUSE_TAIL=0
  for w in "${@:-4}"; do run H264BSDMI_FUSE_TAIL=1 H264BSDMI_TAIL_INTRA_WAVES=$w; done
done
