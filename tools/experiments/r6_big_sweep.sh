#!/bin/bash
# GPU box: a larger differential sweep of the round's last build through the product against the compiled reference (fresh seeds 9,900,000+)
set -u
mkdir -p gpurun_out/r6
{
timeout 900 python tools/sweep.py 9900000 6000 --backend gpu 2>&1 | tail -1
timeout 900 python tools/sweep.py 9910000 5000 --damage --flip 0.3 --keep-redundant --keep-gaps --backend gpu 2>&1 | tail -1
timeout 600 python tools/sweep.py 9930000 2400 --huge-mv 0.3 --backend gpu 2>&1 | tail -1
timeout 600 python tools/sweep.py 9950000 2400 --sizes 11-18,1-4 --backend gpu 2>&1 | tail -1
timeout 600 python tools/sweep.py 9960000 2400 --still 0.95 --backend gpu 2>&1 | tail -1
timeout 900 python tools/sweep.py 9970000 1200 --long --backend gpu 2>&1 | tail -1
} > gpurun_out/r6/big_sweep.txt 2>&1
cat gpurun_out/r6/big_sweep.txt
