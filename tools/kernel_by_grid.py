#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> time per (kernel, grid, block) for kernels whose name contains a pattern."""
import csv, sys
from collections import defaultdict
pat = sys.argv[2].split(',')
acc = defaultdict(lambda: [0, 0])
tot = 0
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        tot += d
        n = r['Kernel_Name']
        if any(p in n for p in pat):
            k = (n.split('(')[0][-40:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
            acc[k][0] += d; acc[k][1] += 1
print(f'total kernel time {tot / 1e6:.1f} ms')
for k, (d, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{d / 1e6:8.2f} ms {100 * d / tot:5.2f} %  n={n:5d} avg {d / n / 1e3:7.1f} us  {k}')
