set -u
mkdir -p gpurun_out
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -2 gpurun_out/refresh.log
bash tools/sq_profile.sh > gpurun_out/sq.log 2>&1; tail -2 gpurun_out/sq.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
