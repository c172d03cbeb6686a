#!/usr/bin/env python3
"""The legs of a bench.py JSON line, one per line (tools/show_bench.py line.json)"""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"] / 1e6, 1), "M MB/s,", round(d["ms_per_step"], 1), "ms/step")
def walk(prefix, v):
    if isinstance(v, dict):
        if "value" in v and isinstance(v["value"], (int, float)):
            extra = {k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.items() if k in ("fraction_of_lock_step", "fps", "ms_per_step", "threads", "unit")}
            print(prefix, round(v["value"] / 1e6, 1), extra)
        else:
            for k, x in v.items(): walk(prefix + "." + k if prefix else k, x)
for k, v in d.items():
    if k not in ("roofline", "config", "cpu_baseline"): walk(k, v)
print("device ms per step", {k: round(v, 1) for k, v in d["roofline"]["device_ms_per_step"].items() if isinstance(v, (int, float))})
print("roofline", {k: v for k, v in d["roofline"].items() if k in ("achieved", "frac", "whole_path_frac", "whole_path_GBs")})
print("cpu_baseline", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"})
