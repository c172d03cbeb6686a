#!/bin/bash
# GPU box: lock-step timing of library variants (tools/experiments/build_variant.sh), alternating with the default build
set -u
out=gpurun_out/variants; rm -rf $out; mkdir -p $out
one() {
  H264BSD_VARIANT=$1 timeout 300 python bench.py --steps ${STEPS:-5} --warmup 1 --ramp-seconds 2 --no-cpu-baseline --no-staggered ${EXTRA:---no-desync} --no-argb --no-end-to-end --no-groups-variant --no-full-copies-variant 2> $out/err_$1.log | tail -1 > $out/b.json
  python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open("gpurun_out/variants/b.json").read())
    r = d["roofline"]["device_ms_per_step"]
    ds = d.get("desynchronised")
    print(sys.argv[1] or "default", round(d["value"]/1e6,1), round(d["ms_per_step"],1), {k: round(v,1) for k,v in r.items() if isinstance(v,(int,float))}, "err", d.get("device_errors"),
          {k: round(v["value"]/1e6,1) for k, v in ds.items() if isinstance(v, dict)} if ds else "")
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
for rep in 1 2; do for v in "" $VARIANTS; do one "$v"; done; done
