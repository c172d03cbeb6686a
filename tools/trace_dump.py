#!/usr/bin/env python3
"""Per-dispatch kernel durations from a rocprofv3 results.db: usage trace_dump.py <dir> [kernel-substr]"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = list(c.execute("select name, start, end from kernels order by start"))
sel = [(n, s, e) for (n, s, e) in rows if sub in n]
print(len(rows), "dispatches,", len(sel), "selected")
for i, (n, s, e) in enumerate(sel):
    print(i, n[:40], (e - s) / 1000.0)
