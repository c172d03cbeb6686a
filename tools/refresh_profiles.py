#!/usr/bin/env python3
"""Turn gpurun_out/refresh/ (tools/refresh_profiles.sh) into the committed artefacts profiles/<tag>_*:
bench line, kernel stats, PMC summary and the per-kernel HBM traffic table bench.py reads.
usage: refresh_profiles.py <tag, e.g. r01>"""
import json, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "refresh"), os.path.join(root, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line)
open(os.path.join(dst, f"{tag}_kernel_stats.txt"), "w").write(open(os.path.join(src, "kernel_stats.txt")).read())
head = ("# rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline\n"
        "# separate passes (FETCH_SIZE | WRITE_SIZE), 256 x 1080p streams; values in KB per dispatch (= per tick);\n"
        "# MI355X_MICROARCH.md: FETCH_SIZE on gfx950 reads 1/2 of a wide (16 B/lane) coalesced stream - the accesses here are\n"
        "# 4-16 B wide and were NOT rescaled (uncalibrated, reported as collected).\n")
body = open(os.path.join(src, "pmc_FETCH_SIZE.txt")).read() + open(os.path.join(src, "pmc_WRITE_SIZE.txt")).read()
open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w").write(head + body)
kern = {}
for l in body.splitlines():
    m = re.match(r"h264k::(\w+)\(.*?\s+(FETCH_SIZE|WRITE_SIZE)\s+total=\S+ per_dispatch=(\S+)", l)
    if m:
        kern.setdefault(m.group(1), {})["fetch_bytes_per_launch" if m.group(2) == "FETCH_SIZE" else "write_bytes_per_launch"] = float(m.group(3)) * 1024
json.dump({"source": f"profiles/{tag}_pmc_summary.txt (separate rocprofv3 --pmc passes, uncalibrated)", "kernels": kern},
          open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(line[:300]); print(json.dumps(kern)[:600])
