#!/bin/bash
# GPU box: the evidence pass of the round's final build in one call (outputs under gpurun_out/r6/): GPU suite, stress parity, fresh-seed
# sweeps through the product against the compiled reference, SQ counters
set -u
out=gpurun_out/r6; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x -n 1 > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
timeout 1200 python tools/stress_parity.py ${LAPS:-200} > $out/stress_parity.txt 2>&1; tail -6 $out/stress_parity.txt
{
timeout 500 python tools/sweep.py 9800000 1500 --backend gpu 2>&1 | tail -1
timeout 500 python tools/sweep.py 9810000 1200 --damage --flip 0.3 --keep-redundant --keep-gaps --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9830000 600 --huge-mv 0.3 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9850000 600 --sizes 11-18,1-4 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9860000 600 --still 0.95 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9870000 300 --long --backend gpu 2>&1 | tail -1
} > $out/sweeps.txt 2>&1
cat $out/sweeps.txt
bash tools/sq_profile.sh > $out/sq_profile.log 2>&1; tail -3 $out/sq_profile.log
