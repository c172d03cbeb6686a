#!/bin/bash
# SQ counters of the kernels on ONE kind of picture: tick $1 of the lock-step replay (0 = IDR), own pass per counter group
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
t=${1:-0}
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  rm -rf gpurun_out/pmc_tick; mkdir -p gpurun_out/pmc_tick
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc_tick -- python tools/itick.py $t 10 2>&1 | tail -1
  python tools/pmc_dump.py gpurun_out/pmc_tick | grep -v rocclr
done
rm -rf gpurun_out/pmc_tick
