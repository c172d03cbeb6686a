#!/bin/bash
# GPU box: kernel times of the lock-step replay with the conversion inside the run (tools/time_argb.py, no verification):
# conv_ab.sh "<variant>:<conversion wavefronts of k_frame_dbk>:<host 0/1>" ...   (variants: build_variant.sh, e.g. -DCONV_BATCH_N=4 -DCONV_PRIO=3)
for spec in "$@"; do IFS=: read v w h <<< "$spec"; echo -n "variant=${v:-default} conv_waves=$w host=$h: "; H264BSD_VARIANT=$v timeout 300 python tools/time_argb.py 3 $w $h 2>&1 | tail -1; done
