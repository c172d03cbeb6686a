#!/bin/bash
# end-to-end throughput against parser threads and pinning policy (host-side experiment, GPU box)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
lscpu | grep -E 'NUMA|Socket|Model name' | head -8
for pin in 2 1 0; do for t in 16 32 64; do
  echo -n "pin $pin threads $t: "; H264BSDMI_PIN=$pin timeout 120 python tools/e2e_bench.py --native --threads $t 2>&1 | tail -n 1 | cut -c1-200
done; done
