#!/bin/bash
# GPU box: the config-3 leg of bench.py (verified) with the conversion hosted by k_frame_dbk (default) and with launches only, alternating
for f in "" --argb-no-hosting "" --argb-no-hosting; do echo -n "${f:-hosted}: "; ARGB_FLAGS=$f bash tools/experiments/argb.sh "" 2>&1 | tail -1; done
