#!/bin/bash
# GPU box, final kernels of round 6 (rebuilt k_frame_dbk, packed chroma prediction in k_recon_inter): differential sweeps against
# the compiled reference through the product, fresh seeds
set -u
mkdir -p gpurun_out
{
timeout 500 python tools/sweep.py 9000000 1500 --backend gpu 2>&1 | tail -1
timeout 500 python tools/sweep.py 9100000 1200 --damage --flip 0.3 --keep-redundant --keep-gaps --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9300000 600 --huge-mv 0.3 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9500000 600 --sizes 11-18,1-4 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9600000 600 --still 0.95 --backend gpu 2>&1 | tail -1
timeout 400 python tools/sweep.py 9700000 300 --long --backend gpu 2>&1 | tail -1
} > gpurun_out/r6_sweeps.txt 2>&1
cat gpurun_out/r6_sweeps.txt
