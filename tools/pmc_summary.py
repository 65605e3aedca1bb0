#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: python tools/pmc_summary.py file.csv [name-filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    if flt and flt not in k:
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k].add(r['Dispatch_Id'])
for k, cs in agg.items():
    print(f'{k[:90]}  dispatches={len(cnt[k])}')
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} {v:16.6g}  per-dispatch {v / len(cnt[k]):14.6g}')
