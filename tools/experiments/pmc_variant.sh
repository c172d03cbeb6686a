#!/bin/bash
# GPU box: SQ counters of the lock-step kernels for library variants: pmc_variant.sh "" a b   ("" = default build)
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
CMD="python tools/time_variants.py"
for v in "$@"; do
  out=gpurun_out/pmcv_${v:-default}; rm -rf $out; mkdir -p $out
  H264BSD_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $out/p -- $CMD > $out/log.txt 2>&1
  echo "== ${v:-default}"; python tools/pmc_dump.py $out/p | grep -v "rocclr\|k_checksum\|k_spin" | grep "${FILTER:-.}"
  rm -rf $out/p
done
