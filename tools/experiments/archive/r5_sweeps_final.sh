#!/bin/bash
# GPU box, final build of round 5 (XCD-aware list positions in k_recon_inter): differential sweeps with fresh seeds + the soak test
set -u
mkdir -p gpurun_out
{
timeout 400 python tools/sweep.py 7000000 1200 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 7100000 1000 --damage --flip 0.3 --keep-redundant --keep-gaps --backend gpu 2>&1 | tail -1
timeout 300 python tools/sweep.py 7300000 400 --huge-mv 0.3 --backend gpu 2>&1 | tail -1
timeout 300 python tools/sweep.py 7500000 400 --sizes 11-18,1-4 --backend gpu 2>&1 | tail -1
timeout 600 python tools/stress_parity.py 10 2>&1 | tail -4
} > gpurun_out/r5_sweeps_final.txt 2>&1
cat gpurun_out/r5_sweeps_final.txt
