#!/bin/bash
# GPU box: batches that pull on (CPUs of quota) threads (the default now) against all 20 of the pool; the bench's two end-to-end legs; the pull tests
mkdir -p gpurun_out
{
for i in 1 2; do
timeout 300 python tools/experiments/hostout.py timeline 2>&1 | tail -1
H264BSDMI_THREADS=20 timeout 300 python tools/experiments/hostout.py timeline 2>&1 | tail -1
done
timeout 300 python bench.py --steps 2 --ramp-seconds 1 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-groups-variant --no-full-copies-variant 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k in ('end_to_end', 'end_to_end_host_output'): print(k, round(d[k]['fps']), 'fps', d[k]['parser_threads'], 'threads')"
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_synth_streams.py -q -m gpu -x -k "pull or batch or lifecycle" 2>&1 | grep -v "^Extension modules" | tail -5
} > gpurun_out/hostout.txt 2>&1
cat gpurun_out/hostout.txt
