#!/bin/bash
# experiment: HW queue limits with / without scratch, then desynchronised streams with a scratch-free build
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export NOBASE=1
export GPU_MAX_HW_QUEUES=${HWQ:-32}

cp gpurun_variants/lib_nosc.so h264bsd_amd/lib/libh264bsd_mi355x_bench.so
echo "== hwq $GPU_MAX_HW_QUEUES, no side lanes, scratch-free kernels"
H264BSDMI_NO_SIDE_LANES=1 timeout 600 python tools/desync_probe.py 256 4,4,8 6,6,8 4,4,16 4,4,24 6,6,32 2>&1 | tail -14
