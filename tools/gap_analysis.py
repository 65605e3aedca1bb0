#!/usr/bin/env python
"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: how much of the wall time has NO kernel running (union over all
queues), and which kernels are followed by the longest gaps on their queue.  usage: gap_analysis.py kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get('Queue_Id', '0')))
rows.sort()
# keep the steady part: skip the first 30 % (init, warm-up)
t_lo = rows[0][0] + int(0.3 * (rows[-1][1] - rows[0][0]))
rows = [r for r in rows if r[0] >= t_lo]
wall = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _, _ in rows)
print(f'kernels {len(rows)}  wall {wall / 1e6:.1f} ms  busy(union) {busy / 1e6:.1f} ms = {100 * busy / wall:.1f} %  idle {100 - 100 * busy / wall:.1f} %  sum of durations {ksum / 1e6:.1f} ms')
by_q = defaultdict(list)
for r in rows:
    by_q[r[3]].append(r)
gap_after = defaultdict(lambda: [0, 0])
hist = defaultdict(int)
for q, lst in by_q.items():
    for a, b in zip(lst, lst[1:]):
        g = b[0] - a[1]
        if g > 0:
            gap_after[a[2]][0] += g; gap_after[a[2]][1] += 1
            hist[min(int(g / 1000), 50)] += 1
print('queues:', {q: len(v) for q, v in by_q.items()})
print('gap histogram (us: count):', dict(sorted(hist.items())[:12]), '...')
tot = sum(v[0] for v in gap_after.values())
print(f'sum of same-queue gaps {tot / 1e6:.1f} ms')
for name, (g, n) in sorted(gap_after.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f'  {g / 1e6:7.2f} ms over {n:5d} gaps (avg {g / n / 1e3:5.1f} us) after {name}')
