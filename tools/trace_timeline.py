#!/usr/bin/env python3
"""A stretch of a rocprofv3 --kernel-trace CSV as a timeline: queue, kernel, start and end relative to the first kernel shown
(µs), duration, workgroups.  tools/trace_timeline.py <kernel_trace.csv> [skip_fraction=0.5] [n=60]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'h264k' in r['Kernel_Name'] and 'checksum' not in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['k'] = r['Kernel_Name'].split('(')[0].split('::')[-1].replace('void ', '')
rows.sort(key=lambda r: r['s'])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
i0 = int(len(rows) * skip)
t0 = rows[i0]['s']
qs = sorted({int(r['Queue_Id']) for r in rows})
for r in rows[i0:i0 + n]:
    q = qs.index(int(r['Queue_Id']))
    wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y'])
    print(f"q{q} {'      ' * q}{r['k']:<22s} {(r['s'] - t0) / 1e3:9.1f} -> {(r['e'] - t0) / 1e3:9.1f}  ({(r['e'] - r['s']) / 1e3:7.1f} us, {wgs} WGs)")
