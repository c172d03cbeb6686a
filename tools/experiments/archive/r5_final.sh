#!/bin/bash
# GPU box, end of round 5: the GPU suite, the bench line + rocprofv3 stats + FETCH / WRITE passes (tools/refresh_profiles.sh), the SQ
# counter passes (tools/sq_profile.sh), the three instruction-cost probes, the inter kernel's cycle accounting, smoke()
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; tail -3 gpurun_out/full_gpu_tests.log
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -2 gpurun_out/refresh.log
bash tools/sq_profile.sh > gpurun_out/sq.log 2>&1; tail -2 gpurun_out/sq.log
{ echo "== tools/probes/lane_util_probe (self-timed)"; ./tools/probes/lane_util_probe; echo; echo "== tools/probes/exec_mask_probe"; ./tools/probes/exec_mask_probe; echo; echo "== tools/probes/op_cost_probe"; ./tools/probes/op_cost_probe; } > gpurun_out/probes.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU -d gpurun_out/lp -- ./tools/probes/lane_util_probe > /dev/null 2>&1
{ echo; echo "== rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU -- tools/probes/lane_util_probe"; python tools/pmc_dump.py gpurun_out/lp | grep -v rocclr; } >> gpurun_out/probes.txt; rm -rf gpurun_out/lp
H264BSD_VARIANT=iprof timeout 300 python tools/inter_prof.py > gpurun_out/inter_prof.txt 2>&1; tail -1 gpurun_out/inter_prof.txt
H264BSD_VARIANT=prof timeout 300 python tools/prof_tail.py > gpurun_out/prof_tail.txt 2>&1; tail -2 gpurun_out/prof_tail.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
