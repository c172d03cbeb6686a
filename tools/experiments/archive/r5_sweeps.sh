#!/bin/bash
# GPU box, round 5 after the kernel changes: differential sweeps through the product against the compiled reference + the soak test
set -u
mkdir -p gpurun_out
{
timeout 600 python tools/sweep.py 5000000 1500 --backend gpu 2>&1 | tail -2
timeout 600 python tools/sweep.py 5100000 1500 --damage --flip 0.3 --keep-redundant --keep-gaps --backend gpu 2>&1 | tail -2
timeout 600 python tools/sweep.py 5200000 1000 --overflow 0.05 --backend gpu 2>&1 | tail -2
timeout 600 python tools/sweep.py 5300000 600 --huge-mv 0.3 --backend gpu 2>&1 | tail -2
timeout 600 python tools/sweep.py 5400000 600 --still 0.95 --backend gpu 2>&1 | tail -2
timeout 600 python tools/sweep.py 5500000 400 --sizes 11-18,1-4 --damage --flip 0.2 --backend gpu 2>&1 | tail -2
timeout 900 python tools/stress_parity.py 20 2>&1 | tail -5
} > gpurun_out/r5_sweeps.txt 2>&1
cat gpurun_out/r5_sweeps.txt
