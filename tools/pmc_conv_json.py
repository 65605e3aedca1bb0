#!/usr/bin/env python
"""Build profiles/<name>.json (HBM-side traffic per conv launch) from two rocprofv3 counter CSVs.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- python tools/bench_kernels.py conv
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -- python tools/bench_kernels.py conv
    python tools/pmc_conv_json.py out_f/.../*counter_collection.csv out_w/.../*counter_collection.csv profiles/r01_conv_pmc.json

Units/corrections follow MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE counts 128-byte requests as 64 B on
gfx950 (x2); separate passes, no trace domains besides the kernel trace.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        acc[k][0] += float(r['Counter_Value'])
        acc[k][1].add(r['Dispatch_Id'])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def main():
    fpath, wpath, out = sys.argv[1:4]
    what = sys.argv[4] if len(sys.argv) > 4 else 'conv'          # 'gemm': the dense-GEMM micro-benchmark (bench_kernels.py gemm)
    f, w = per_kernel(fpath, 'FETCH_SIZE'), per_kernel(wpath, 'WRITE_SIZE')
    conv = [k for k in f if ('gemm' in k and 'GemmParams' in k and 'finish' not in k)]
    res = {'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python tools/bench_kernels.py {what}` '
                     + ('(SD1.5 conv3x3 shapes, batch 16)' if what == 'conv' else '(SD1.5 dense GEMM shapes: Linear / 1x1 conv over tokens, batch 16)'),
           'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (FETCH_SIZE x2: gfx950 counts 128-B requests as 64 B)',
           'per_kernel': {}}
    tot_b, tot_n = 0.0, 0
    for k in sorted(set(f) | set(w)):
        if 'gemm' not in k:
            continue
        fk, nf = f.get(k, (0.0, 0))
        wk, nw = w.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        res['per_kernel'][k] = {'launches': n, 'FETCH_SIZE_KB_avg': fk / n, 'WRITE_SIZE_KB_avg': wk / n}
        if k in conv:
            tot_b += (2 * fk + wk) * 1024.0
            tot_n += n
    res[what + '_launches'] = tot_n
    res['avg_hbm_side_bytes_per_launch'] = tot_b / max(tot_n, 1)
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in (what + '_launches', 'avg_hbm_side_bytes_per_launch')}))


if __name__ == '__main__':
    main()
