#!/bin/bash
# GPU box, round 5 start: lock-step line of the current build, cycle accounting of the per-picture kernels (variant "prof"),
# and the lane-utilisation probe under the SQ counters that tools/sq_counters.py divides.
set -u
out=gpurun_out/r5a; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/experiments/quick_bench.sh base > $out/base.txt 2>&1; cat $out/base.txt
H264BSD_VARIANT=prof timeout 300 python tools/prof_tail.py > $out/prof_tail.txt 2>&1; tail -12 $out/prof_tail.txt
./tools/probes/lane_util_probe > $out/lane_probe.txt 2>&1; cat $out/lane_probe.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -d $out/lp -- ./tools/probes/lane_util_probe > $out/lp.log 2>&1
python tools/pmc_dump.py $out/lp > $out/lane_probe_pmc.txt; cat $out/lane_probe_pmc.txt
rm -rf $out/lp
