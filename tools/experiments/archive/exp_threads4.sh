#!/bin/bash
# GPU box: failure rate of the 8-thread C caller (with frame dumps, which widen the timing) under band settings
set -u
out=gpurun_out/threads; rm -rf $out; mkdir -p $out /tmp/dh
gcc -O2 -std=gnu11 -Iinclude tests/c_caller/decode_hash.c -o $out/dh -Lh264bsd_amd/lib -lh264bsd_mi355x -lpthread -Wl,-rpath,$PWD/h264bsd_amd/lib || exit 1
want=$(python -c "import json;print(json.load(open('tests/golden/golden.json'))['test_640x360']['sha256_all'])")
run() {
  name=$1; shift; bad=0
  for i in $(seq 1 ${N:-25}); do
    env DH_DUMP=/tmp/dh "$@" $out/dh -t 8 tests/golden/test_640x360.h264 2>$out/err_$name.log | grep '^decoder' | awk '{print $NF}' > $out/o.txt
    n=$(grep -vc "$want" $out/o.txt); [ "$n" != 0 ] && bad=$((bad+1))
  done
  echo "$name: $bad bad runs of ${N:-25}"
}
run default X=1
run no_light_dbk_bands H264BSDMI_TAIL=0,9,12,0,9,12
run no_heavy_bands H264BSDMI_TAIL=17,0,12,0,0,12
run no_intra_bands H264BSDMI_TAIL=17,9,12,0,0,12
run no_dbk_bands H264BSDMI_TAIL=0,0,12,0,9,12
run nobands H264BSDMI_BAND_BUDGET=0 H264BSDMI_HEAVY_BUDGET=0
run noelision H264BSDMI_COPY_ELISION=0
run onelane H264BSDMI_LANES=1,0
