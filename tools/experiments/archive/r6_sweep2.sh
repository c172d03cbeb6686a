for cfg in 12,4 12,5 12,6 10,4 10,5 9,3 8,3 8,4; do
  w=${cfg%,*}; c=${cfg#*,}
  echo -n "w$w c$c: "; H264BSDMI_TAIL=17,9,$w,0,9,12,$c timeout 300 python tools/time_variants.py 2>&1 | tail -1
done
