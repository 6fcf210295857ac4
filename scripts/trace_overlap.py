"""Reads a rocprofv3 kernel trace (CSV) and prints, for the search kernels, the queues used, the share of time during
which two or more of them overlap, and a window of consecutive launches.  usage: trace_overlap.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if 'k_net_level_rows' in r['Kernel_Name'] or 'k_leaf_backup' in r['Kernel_Name']]
ks.sort(key=lambda r: int(r['Start_Timestamp']))
byq = collections.Counter((r['Queue_Id'], r.get('Stream_Id')) for r in ks)
print('launches per (queue, stream):', dict(byq))
ev = []
for r in ks:
    ev.append((int(r['Start_Timestamp']), 1))
    ev.append((int(r['End_Timestamp']), -1))
ev.sort()
depth, last, busy = 0, ev[0][0], collections.Counter()
for t, d in ev:
    busy[depth] += t - last
    last, depth = t, depth + d
total = sum(busy.values())
print('share of the span with n search kernels running:', {k: round(v / total, 3) for k, v in sorted(busy.items())})
mid = ks[len(ks) // 2: len(ks) // 2 + 14]
t0 = int(mid[0]['Start_Timestamp'])
for r in mid:
    print(r['Queue_Id'], r.get('Stream_Id'), r['Kernel_Name'][9:30], int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0,
          r['Grid_Size_X'], r['Grid_Size_Y'])
