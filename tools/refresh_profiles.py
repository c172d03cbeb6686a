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
        "# 4-16 B wide; raw values below are as collected.  Calibration factors from k_copy's known byte count are in\n"
        "# the traffic table next to this file (tools/refresh_profiles.py).\n")
body = open(os.path.join(src, "pmc_FETCH_SIZE.txt")).read() + open(os.path.join(src, "pmc_WRITE_SIZE.txt")).read()
open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w").write(head + body)
kern = {}
for l in body.splitlines():
    m = re.match(r"(?:void )?h264k::(\w+)(<\w+>)?\(.*?\s+(FETCH_SIZE|WRITE_SIZE)\s+total=\S+ per_dispatch=(\S+)", l)
    if m:
        # k_recon_inter<0> and <1> are the two halves of one tick's inter reconstruction: their bytes add up (the banded and the
        # single-band instantiations of the per-picture kernels never run in the same tick of this command)
        k = kern.setdefault(m.group(1), {})
        key = "fetch_bytes_per_launch" if m.group(3) == "FETCH_SIZE" else "write_bytes_per_launch"
        k[key] = k.get(key, 0.0) + float(m.group(4)) * 1024
# Calibration on a known byte count in our own access pattern (MI355X_MICROARCH.md, HBM section): k_copy moves exactly
# 384 bytes in and 384 bytes out per copied macroblock, as contiguous 16-byte-per-lane pieces of macroblock tiles.
sys.path.insert(0, root)
import h264bsd_amd
# (the same capture as the bench line's: with copy elision the jobs list fewer copies)
elide = bool(json.loads(line).get("config", {}).get("copy_elision", {}).get("on", False))
jobs, _, _ = h264bsd_amd.capture_stream(open(os.path.join(root, "tests", "golden", "test_1920x1080.h264"), "rb").read(), copy_elision=elide)
heads = [h264bsd_amd.job_header(j) for j in jobs]
ticks_with_copies = sum(1 for h in heads if h["n_copy"])
copy_alg = sum(h["n_copy_mbs"] for h in heads) * 384 * 256 / ticks_with_copies          # bytes per dispatch, each direction
cal = {"fetch": copy_alg / kern["k_copy"]["fetch_bytes_per_launch"], "write": copy_alg / kern["k_copy"]["write_bytes_per_launch"],
       "k_copy_algorithmic_bytes_per_launch_each_way": copy_alg,
       "method": "factor = algorithmic bytes of k_copy / counter value of k_copy; applied to every kernel"}
for k in kern.values():
    k["fetch_bytes_per_launch_calibrated"] = k["fetch_bytes_per_launch"] * cal["fetch"]
    k["write_bytes_per_launch_calibrated"] = k["write_bytes_per_launch"] * cal["write"]
from h264bsd_amd.srchash import kernel_source_sha256
src_sha = kernel_source_sha256(root)
json.dump({"source": f"profiles/{tag}_pmc_summary.txt (separate rocprofv3 --pmc passes)", "kernel_source_sha256": src_sha, "calibration": cal, "kernels": kern},
          open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
# the bench line was printed before the PMC passes of this session existed: fill its traffic field from them
bl = json.loads(line)
dom = bl["roofline"]["kernel"]
if bl["roofline"].get("traffic") is None and dom in kern:
    bl["roofline"]["traffic"] = kern[dom]["fetch_bytes_per_launch_calibrated"] + kern[dom]["write_bytes_per_launch_calibrated"]
    bl["roofline"]["traffic_source"] = (f"profiles/{tag}_traffic.json: FETCH_SIZE + WRITE_SIZE of this kernel, separate rocprofv3 --pmc passes run "
                                        "right after this line on the same box (filled in by tools/refresh_profiles.py), calibrated on k_copy's known byte count")
    names = ("k_copy", "k_recon_inter", "k_dbk", "k_frame_intra", "k_frame_dbk")
    per = {k: kern[k]["fetch_bytes_per_launch_calibrated"] + kern[k]["write_bytes_per_launch_calibrated"] for k in names if k in kern}
    bl["roofline"]["traffic_per_kernel"] = per
    bl["roofline"]["traffic_whole_path"] = sum(per.values())
    bl["roofline"]["traffic_ratio"] = sum(per.values()) / bl["roofline"]["alg_bytes_per_launch"]
    open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(json.dumps(bl) + "\n")
print(line[:300]); print(json.dumps(kern)[:600])
