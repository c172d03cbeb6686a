#!/bin/bash
# GPU box: the end-of-round evidence pass of round 6 in one call (outputs under gpurun_out/r6/)
set -u
out=gpurun_out/r6; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x -n 1 > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
timeout 900 python tools/stress_parity.py ${LAPS:-3} > $out/stress_parity.txt 2>&1; tail -4 $out/stress_parity.txt
timeout 300 tools/probes/vmem_issue_probe > $out/vmem_issue_probe.txt 2>&1; tail -2 $out/vmem_issue_probe.txt
H264BSD_VARIANT=prof timeout 300 python tools/prof_dbk_waves.py > $out/dbk_cycle_accounting.txt 2>&1; tail -3 $out/dbk_cycle_accounting.txt
bash tools/sq_profile.sh > $out/sq_profile.log 2>&1; tail -3 $out/sq_profile.log
