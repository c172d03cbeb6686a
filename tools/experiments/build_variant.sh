#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: the two libraries with other kernel flags into h264bsd_amd/lib_<name>/
# (H264BSD_VARIANT=<name> makes the Python mirror, and with it bench.py, load that one)
set -e
name=$1; shift
case " $* " in *tgsplit*) echo "refusing -mtgsplit: release_stores() relies on a workgroup living on one compute unit" >&2; exit 1;; esac
cd "$(dirname "$0")/../../h264bsd_amd/csrc"
mkdir -p build_$name ../lib_$name
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value "$@" -c engine.hip -o build_$name/engine.o
objs=$(ls build/hd_*.o build/api.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,--version-script=exports.map -o ../lib_$name/libh264bsd_mi355x.so $objs build_$name/engine.o -lpthread
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,--version-script=exports_bench.map -o ../lib_$name/libh264bsd_mi355x_bench.so $objs build_$name/engine.o -lpthread
echo built ../lib_$name
