#!/bin/bash
# Host parser profile (gprof flat profile) + plain timing of the -O3 build, on whatever machine this runs on.
set -e
root=$(cd "$(dirname "$0")/.." && pwd); C=$root/h264bsd_amd/csrc; out=${TMPDIR:-/tmp}/h264bsd_parse_prof; mkdir -p $out
SRCS="$root/tools/probes/parse_harness.c $root/tests/fuzz_asan/stub_engine.c $C/hd_nal.c $C/hd_params.c $C/hd_slice.c $C/hd_dpb.c $C/hd_cavlc.c $C/hd_resid.c $C/hd_mb.c $C/hd_core.c $C/api.c"
gcc -O3 -std=gnu11 -I$C -DH264BSD_BUILD $SRCS -lpthread -o $out/fast 2>/dev/null
gcc -O2 -pg -fno-inline-functions -std=gnu11 -I$C -DH264BSD_BUILD $SRCS -lpthread -o $out/prof 2>/dev/null
for i in 1 2 3; do $out/fast $root/tests/golden/test_1920x1080.h264 10; done
cd $out && ./prof $root/tests/golden/test_1920x1080.h264 20 > /dev/null && gprof -b -p ./prof gmon.out 2>/dev/null | head -24
