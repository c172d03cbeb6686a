#!/usr/bin/env python3
"""Turn a rocprofv3 output directory (results.db from `rocprofv3 --kernel-trace --stats`) into the small
text summary that is committed under profiles/.  usage: rocprof_summary.py <dir> <out.txt> [note]"""
import glob
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    dbs = glob.glob(src + "/**/*.db", recursive=True)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({dbs[0] if dbs else 'no db'})", f"# {note}",
             "name,total_calls,total_duration_us,average_us,percentage"]
    for db in dbs:
        c = sqlite3.connect(db)
        for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(",".join(str(x) for x in r))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
