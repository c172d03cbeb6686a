cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
H264BSDMI_LANES=1,0 timeout 600 python -m pytest tests/test_gpu_api.py tests/test_c_caller.py -m gpu -x -q 2>&1 | tail -2
for l in 1,0 4,2 4,2 1,0; do echo -n "e2e lanes $l: "; H264BSDMI_LANES=$l timeout 300 python tools/e2e_bench.py --native 2>&1 | tail -1; done
NOBASE=1 timeout 300 python tools/desync_probe.py 256 4,4,8 2>&1 | tail -1
