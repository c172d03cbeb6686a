#!/usr/bin/env python3
"""Per-queue busy time, kernel overlap and a CU-demand estimate from a rocprofv3 --kernel-trace CSV of a lane schedule.
usage: trace_lanes.py <kernel_trace.csv> [window_ms]"""
import csv, collections, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'h264k' in r['Kernel_Name'] and 'checksum' not in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['k'] = r['Kernel_Name'].split('(')[0].split('::')[-1]
    r['wgs'] = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y'])
t_end = max(r['e'] for r in rows)
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 230.0
win = [r for r in rows if r['s'] > t_end - win_ms * 1e6]
span = (t_end - min(r['s'] for r in win)) / 1e6
print(len(win), 'kernels in the last %.1f ms' % span)
byq = collections.defaultdict(list)
for r in win: byq[int(r['Queue_Id'])].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r['e'] - r['s'] for r in rs) / 1e6
    print('queue %2d: %4d kernels, busy %6.1f ms (%.0f %%)' % (q, len(rs), busy, 100 * busy / span))
byk = collections.defaultdict(list)
for r in win: byk[(r['k'], 'heavy' if r['k'].startswith('k_frame') and r['wgs'] < 16 else 'light')].append(r)
for k, rs in sorted(byk.items()):
    d = sorted((r['e'] - r['s']) / 1e3 for r in rs)
    print('%-18s %-5s n %4d  avg %7.1f us  median %7.1f  p90 %7.1f  max %7.1f  sum %6.1f ms  avg WGs %6.0f' % (k[0], k[1], len(rs), sum(d) / len(d), d[len(d) // 2], d[int(len(d) * .9)], d[-1], sum(d) / 1e3, sum(r['wgs'] for r in rs) / len(rs)))
ev = []
for r in win: ev.append((r['s'], 1)); ev.append((r['e'], -1))
ev.sort(); cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print('kernels running at once:', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
