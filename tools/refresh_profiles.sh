#!/bin/bash
# Run on the GPU box (gpurun): bench line, rocprofv3 kernel stats and the two HBM-traffic PMC passes of the same
# command.  Outputs land in gpurun_out/refresh/; tools/refresh_profiles.py turns them into profiles/rNN_*.
set -u
out=gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -- python bench.py --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant > $out/stats.log 2>&1
python tools/rocprof_summary.py $out/stats $out/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant, run right after the bench line on the warm box (256 x 1080p streams, steps 3, warmup 1, + 73-tick verification pass); durations in microseconds" > /dev/null
for c in FETCH_SIZE WRITE_SIZE ${EXTRA_PMC:-}; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -- python bench.py --steps 1 --warmup 0 --ramp-seconds 0 --no-cpu-baseline --no-staggered --no-desync --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant > $out/pmc_$c.log 2>&1
  python tools/pmc_dump.py $out/pmc_$c > $out/pmc_$c.txt
done
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
