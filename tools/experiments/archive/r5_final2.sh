#!/bin/bash
# GPU box, end of round 5 after the host-output work (engine.hip / api.c only: the kernels and their counter passes stand): the GPU suite,
# the bench line + rocprofv3 stats + FETCH / WRITE passes (tools/refresh_profiles.sh), smoke()
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; tail -3 gpurun_out/full_gpu_tests.log
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -2 gpurun_out/refresh.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
