#!/usr/bin/env python
"""Where do the optimizer kernels sit in a rocprofv3 --kernel-trace CSV of bench.py?  For every adam_ema_kernel dispatch:
start (ms since the first of them), duration, queue, and how much OTHER kernel time (any queue) overlaps its interval.
usage: adam_overlap.py kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40], r.get('Queue_Id', '?')))
rows.sort()
adam = [r for r in rows if 'adam_ema' in r[2]]
t0 = adam[0][0]
for s, e, n, q in adam[-12:]:
    ov = 0
    names = {}
    for s2, e2, n2, q2 in rows:
        if e2 <= s or s2 >= e or 'adam_ema' in n2:
            continue
        d = min(e, e2) - max(s, s2)
        ov += d
        names[n2] = names.get(n2, 0) + d
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print(f'start {(s - t0) / 1e6:9.2f} ms  dur {(e - s) / 1e3:8.1f} us  queue {q}  other-kernel time inside {ov / 1e3:8.1f} us  '
          + ', '.join(f'{k} {v / 1e3:.0f}' for k, v in top))
