#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc results (results.db) per kernel: usage pmc_dump.py <dir>"""
import glob, sqlite3, sys, collections
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if not view:
        print("no counters view; tables:", [t for t in tabs if "pmc" in t or "counter" in t]); continue
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in c.execute(f"select kernel_name, counter_name, value from {view}"):
        a = agg[(r[0][:48], r[1])]; a[0] += r[2]; a[1] += 1
    for (k, n), (v, cnt) in sorted(agg.items()):
        print(f"{k:50s} {n:24s} total={v:.4g} per_dispatch={v/cnt:.4g} n={cnt}")
