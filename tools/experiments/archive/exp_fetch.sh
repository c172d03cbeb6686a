#!/bin/bash
set -u
out=gpurun_out/fetch; rm -rf $out; mkdir -p $out
bash tools/experiments/exp_quick.sh 2>&1 | head -4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$n -- python bench.py --steps 1 --warmup 0 --ramp-seconds 0 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant > $out/pmc_$n.log 2>&1
  python tools/pmc_dump.py $out/pmc_$n | grep "recon_inter\|k_copy"
  rm -rf $out/pmc_$n
done
