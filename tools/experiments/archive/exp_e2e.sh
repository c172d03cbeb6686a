#!/bin/bash
set -u
for i in 1 2; do
for v in 1 0; do echo "async enqueue thread $v:"; H264BSDMI_ASYNC_ENQUEUE=$v timeout 200 python tools/experiments/e2e_split.py 16 2>&1 | grep -v amdgpu; done
done
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_parse_pool.py -x -q -m gpu 2>&1 | tail -2
