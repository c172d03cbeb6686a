#!/bin/bash
# GPU box: the whole GPU suite + the quick lock-step line
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x -n 1 > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
bash tools/experiments/quick_bench.sh ${1:-run} 2>&1 | tail -1 | tee gpurun_out/quick.txt
