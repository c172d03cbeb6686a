for cfg in 8,4 8,3 10,4 10,5 12,4 12,5 12,6; do
  w=${cfg%,*}; c=${cfg#*,}
  H264BSDMI_TAIL=17,9,$w,0,9,12,$c bash tools/experiments/quick_bench.sh w${w}c${c} 2>&1 | tail -1
done
for pr in 0 2 3; do H264BSD_VARIANT=cp$pr H264BSDMI_TAIL=17,9,12,0,9,12,5 bash tools/experiments/quick_bench.sh prio${pr}_w12c5 2>&1 | tail -1; done
