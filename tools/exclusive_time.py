#!/usr/bin/env python
"""Concurrency profile of a rocprofv3 --kernel-trace CSV (steady part): wall time by the number of kernels in flight, and the time
during which exactly ONE kernel runs, by kernel -- the launches nothing overlaps with (phase boundaries, optimizer, tails).
usage: exclusive_time.py kernel_trace.csv [skip_fraction=0.4]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:56]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t_lo = rows[0][0] + int(skip * (rows[-1][1] - rows[0][0]))
rows = [r for r in rows if r[0] >= t_lo]
ev = []
for i, (s, e, n) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
live = set()
by_level = defaultdict(int)
alone = defaultdict(int)
alone_n = defaultdict(int)
pair = defaultdict(int)
last_t = ev[0][0]
for t, d, i in ev:
    dt = t - last_t
    if dt > 0:
        k = len(live)
        by_level[min(k, 4)] += dt
        if k == 1:
            n = rows[next(iter(live))][2]
            alone[n] += dt
        elif k == 2:
            a, b = sorted(rows[j][2] for j in live)
            pair[(a, b)] += dt
    if d == 1:
        live.add(i)
    else:
        live.discard(i)
    last_t = t
wall = rows[-1][1] - rows[0][0]
print(f'kernels {len(rows)}  wall {wall / 1e6:.1f} ms')
for k in sorted(by_level):
    print(f'  {k}{"+" if k == 4 else " "} kernels in flight: {by_level[k] / 1e6:8.1f} ms  {100.0 * by_level[k] / wall:5.1f} %')
print('time with exactly one kernel in flight, by kernel:')
for n, t in sorted(alone.items(), key=lambda x: -x[1])[:25]:
    print(f'  {t / 1e6:8.2f} ms  {100.0 * t / wall:5.2f} %  {n}')
print('most common pairs:')
for (a, b), t in sorted(pair.items(), key=lambda x: -x[1])[:12]:
    print(f'  {t / 1e6:8.2f} ms  {100.0 * t / wall:5.2f} %  {a} | {b}')
