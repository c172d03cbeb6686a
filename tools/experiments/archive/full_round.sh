set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; tail -3 gpurun_out/full_gpu_tests.log
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -3 gpurun_out/refresh.log
bash tools/sq_profile.sh > gpurun_out/sq.log 2>&1; tail -3 gpurun_out/sq.log
