#!/bin/bash
# GPU box: SQ / TCC counters of the five kernels of the lock-step bench, one rocprofv3 --pmc pass per counter group (never
# together with runtime traces).  Output: gpurun_out/sq/*.txt; tools/sq_counters.py turns them into profiles/<tag>_sq_counters.txt.
set -u
out=gpurun_out/sq; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
CMD="python bench.py --steps 1 --warmup 0 --ramp-seconds 0 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -- $CMD > $out/p$i.log 2>&1
  echo "# rocprofv3 --kernel-trace --pmc $grp -- $CMD" > $out/pass$i.txt
  python tools/pmc_dump.py $out/p$i | grep -v rocclr >> $out/pass$i.txt
  rm -rf $out/p$i
  tail -2 $out/p$i.log | cut -c1-300
done
timeout 300 $CMD 2>/dev/null | tail -1 > $out/bench_line.json
ls -la $out
