#!/bin/bash
# usage: tests/fuzz_asan/run.sh <stream.h264> [cases=500] [seed=1]
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd); C=$root/h264bsd_amd/csrc
out=${TMPDIR:-/tmp}/h264bsd_fuzz_asan
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=gnu11 -I$C -DH264BSD_BUILD $here/fuzz.c $here/stub_engine.c \
    $C/hd_nal.c $C/hd_params.c $C/hd_slice.c $C/hd_dpb.c $C/hd_cavlc.c $C/hd_resid.c $C/hd_mb.c $C/hd_core.c $C/api.c $root/oracle/pixel_oracle.c \
    -lpthread -o $out
ASAN_OPTIONS=detect_leaks=0 $out "$1" "${2:-500}" "${3:-1}" 2>&1 | grep -v "left shift of negative" | tail -5
