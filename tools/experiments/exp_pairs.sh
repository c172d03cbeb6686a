#!/bin/bash
# experiment: two pictures per deblocking workgroup (k_frame_dbk2), lock-step / 4 groups / desynchronised
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export NOBASE=1
H264BSDMI_DBK_PAIRS=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for pairs in "" 1 "" 1; do
  echo "== pairs '$pairs'"
  for g in 1 4; do
    echo -n "lock-step groups $g: "
    env ${pairs:+H264BSDMI_DBK_PAIRS=1} timeout 300 python bench.py --groups $g --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --ramp-seconds 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['device_ms_per_step']; print(round(d['value']/1e6,1), round(r['k_frame_dbk'],1), round(r['total'],1))"
  done
  H264BSDMI_DBK_PAIRS=$pairs timeout 300 python tools/desync_probe.py 256 3,4,9 4,4,1 2>&1 | grep lanes | cut -c1-110
done
