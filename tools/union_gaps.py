#!/usr/bin/env python
"""Union-of-queues idle analysis of a rocprofv3 --kernel-trace CSV (steady part): busy fraction, histogram of the idle gaps
between busy intervals, idle time by the kernel that ENDS before the gap and by the kernel that STARTS after it.
usage: union_gaps.py kernel_trace.csv [skip_fraction=0.4]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t_lo = rows[0][0] + int(skip * (rows[-1][1] - rows[0][0]))
rows = [r for r in rows if r[0] >= t_lo]
wall = rows[-1][1] - rows[0][0]
gaps = []           # (length, kernel before, kernel after)
cur_e, last = rows[0][1], rows[0][2]
busy = 0
cur_s = rows[0][0]
for s, e, n in rows[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, last, n))
        busy += cur_e - cur_s
        cur_s, cur_e, last = s, e, n
    elif e > cur_e:
        cur_e, last = e, n
busy += cur_e - cur_s
idle = sum(g[0] for g in gaps)
print(f'kernels {len(rows)}  wall {wall / 1e6:.1f} ms  busy(union) {busy / 1e6:.1f} ms ({100 * busy / wall:.1f} %)  idle {idle / 1e6:.1f} ms in {len(gaps)} gaps  '
      f'sum of kernel durations {sum(e - s for s, e, _ in rows) / 1e6:.1f} ms')
edges = [2, 5, 10, 20, 50, 100, 500, 1e9]
h = defaultdict(lambda: [0, 0])
for g, _, _ in gaps:
    for ed in edges:
        if g / 1e3 <= ed:
            h[ed][0] += 1; h[ed][1] += g
            break
print('gap length (us) : count, total ms')
for ed in edges:
    print(f'   <= {ed:g}: {h[ed][0]:6d}  {h[ed][1] / 1e6:8.2f}')
for title, idx in (('before the gap', 1), ('after the gap', 2)):
    agg = defaultdict(lambda: [0, 0])
    for g in gaps:
        agg[g[idx]][0] += g[0]; agg[g[idx]][1] += 1
    print(f'idle by kernel {title}:')
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f'   {t / 1e6:8.2f} ms  {c:6d} gaps  avg {t / c / 1e3:7.1f} us  {k}')
