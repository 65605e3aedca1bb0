#!/usr/bin/env python
"""HBM-side traffic per CALL of a kernel family from two rocprofv3 counter CSVs of `tools/bench_kernels.py <mode>`:

    python tools/pmc_traffic_json.py FETCH.csv WRITE.csv OUT.json MODE FAMILY=primary_pattern[+helper_pattern...] ...
    (pattern syntax: substrings of the kernel name; `a|b` alternatives, `a&b` both)

A call of a family is one launch of its PRIMARY kernel (name contains primary_pattern); helper kernels (split reductions, finish
kernels, statistics passes) are added to the family's bytes but not to its call count.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024:
both counters are in KiB and gfx950 counts a 128-byte fetch request as 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section); the two
counters come from separate --pmc passes with no trace domain besides the kernel trace.  The kernels run alone (micro-benchmark at
the step's layer shapes, batch 16): PMC passes serialise kernels, so this is each kernel's own traffic, not the step's."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        acc[r['Kernel_Name']][0] += float(r['Counter_Value'])
        acc[r['Kernel_Name']][1].add(r['Dispatch_Id'])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def main():
    fpath, wpath, out, mode = sys.argv[1:5]
    alg = {}
    specs = list(sys.argv[5:])
    if specs and specs[0].endswith('.json'):        # algorithmic bytes per call of the same micro-benchmark (bench_kernels.py --trace-json)
        alg = json.load(open(specs.pop(0)))
    f, w = per_kernel(fpath, 'FETCH_SIZE'), per_kernel(wpath, 'WRITE_SIZE')
    res = {'source': f'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python tools/bench_kernels.py {mode}` '
                     '(the step\'s layer shapes, batch 16; kernels run alone)',
           'correction': 'bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (KiB counters; gfx950 counts 128-B fetch requests as 64 B)', 'families': {}}
    for spec in specs:
        fam, pats = spec.split('=')
        pats = pats.split('+')          # first = primary; inside a pattern `|` separates alternatives, `&` parts that must all occur

        def match(pat, name):
            return any(all(part in name for part in alt.split('&')) for alt in pat.split('|'))
        tot, calls, kern = 0.0, 0, {}
        for k in sorted(set(f) | set(w)):
            if not any(match(p, k) for p in pats):
                continue
            fk, nf = f.get(k, (0.0, 0))
            wk, nw = w.get(k, (0.0, 0))
            n = max(nf, nw, 1)
            kern[k[:120]] = {'launches': n, 'fetch_bytes_avg': 2 * fk / n * 1024, 'write_bytes_avg': wk / n * 1024}
            tot += (2 * fk + wk) * 1024.0
            if match(pats[0], k):
                calls += n
        res['families'][fam] = {'calls': calls, 'avg_hbm_side_bytes_per_call': tot / max(calls, 1), 'kernels': kern}
        if fam in alg:
            a = alg[fam]['algorithmic_bytes_per_call']
            res['families'][fam].update(algorithmic_bytes_per_call=a, traffic_over_algorithmic=tot / max(calls, 1) / a if a else None,
                                        ms_per_call_alone=alg[fam]['ms_per_call'])
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: (v['calls'], round(v['avg_hbm_side_bytes_per_call'] / 1e6, 1), round(v.get('traffic_over_algorithmic') or 0, 2)) for k, v in res['families'].items()}))


if __name__ == '__main__':
    main()
