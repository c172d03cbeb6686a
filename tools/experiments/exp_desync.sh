#!/bin/bash
# experiment: prebuilt library variants under lane scheduling (desynchronised streams)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export NOBASE=1
for v in "$@"; do
  cp gpurun_variants/lib_$v.so h264bsd_amd/lib/libh264bsd_mi355x_bench.so
  echo "== $v"
  timeout 300 python tools/desync_probe.py 256 ${CFGS:-4,4,8 3,4,9} 2>&1 | grep lanes
done
