#!/bin/bash
# GPU box: the host-output leg with look-ahead pulls in the batch workers: where a round goes, with 20 / 28 / 40 threads; the tests of the pull paths
mkdir -p gpurun_out
{
for t in 8 12 16 20; do H264BSDMI_THREADS=$t timeout 300 python tools/experiments/hostout.py timeline 2>&1 | tail -1; done

timeout 300 python tools/experiments/hostout.py barrier 2>&1 | tail -1
timeout 300 python tools/experiments/hostout.py overlap 2>&1 | tail -1
timeout 900 python -m pytest tests/test_synth_streams.py tests/test_gpu_api.py tests/test_c_caller.py -q -m gpu -x 2>&1 | grep -v "^Extension modules" | tail -15
} > gpurun_out/hostout.txt 2>&1
cat gpurun_out/hostout.txt
