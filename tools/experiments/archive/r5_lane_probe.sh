#!/bin/bash
# GPU box: the lane-utilisation probe, self-timed and under the SQ counters tools/sq_counters.py divides
set -u
out=gpurun_out/r5lane; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
./tools/probes/lane_util_probe > $out/lane_probe.txt 2>&1; cat $out/lane_probe.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -d $out/lp -- ./tools/probes/lane_util_probe > $out/lp.log 2>&1
python tools/pmc_dump.py $out/lp | grep -v rocclr > $out/lane_probe_pmc.txt; cat $out/lane_probe_pmc.txt
rm -rf $out/lp
