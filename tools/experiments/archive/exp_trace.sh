cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export NOBASE=1 GPU_MAX_HW_QUEUES=${HWQ:-16}
rm -rf gpurun_out/trace; mkdir -p gpurun_out/trace
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python tools/desync_probe.py 256 ${CFG:-4,4,8} 2>&1 | grep lanes
find gpurun_out/trace -name '*kernel_trace.csv' | head -3; du -sh gpurun_out/trace
