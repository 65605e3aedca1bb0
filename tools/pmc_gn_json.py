#!/usr/bin/env python
"""Build profiles/<name>.json (HBM-side traffic of the GroupNorm kernels) from two rocprofv3 counter CSVs.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python tools/bench_kernels.py norm
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python tools/bench_kernels.py norm
    python tools/pmc_gn_json.py out_f/.../*counter_collection.csv out_w/.../*counter_collection.csv profiles/r02_gn_pmc.json

Units / corrections as in tools/pmc_conv_json.py (KiB counters, FETCH_SIZE x2 on gfx950, separate passes)."""
import json
import sys

from pmc_conv_json import per_kernel


def main():
    fpath, wpath, out = sys.argv[1:4]
    f, w = per_kernel(fpath, 'FETCH_SIZE'), per_kernel(wpath, 'WRITE_SIZE')
    res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python tools/bench_kernels.py norm` '
                     '(GroupNorm shapes of the SD1.5 UNet at batch 16: (HW,C) = (4096,320) (4096,640) (1024,640) (1024,1920) (256,1280) (64,2560); '
                     'fwd + bwd, 13 launches each)',
           'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md section HBM)',
           'algorithmic_bytes_note': 'forward = read x + write y = 4 B/element; the two-kernel split reads x twice (stats + apply): the second '
                                     'read is served by L2 / Infinity Cache for tensors <= ~100 MB',
           'per_kernel': {}}
    fwd = 0.0
    for k in sorted(set(f) | set(w)):
        if 'gn_' not in k:
            continue
        fk, nf = f.get(k, (0.0, 0))
        wk, nw = w.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        b = (2 * fk + wk) * 1024.0 / n
        res['per_kernel'][k] = {'launches': n, 'FETCH_SIZE_KB_avg': fk / n, 'WRITE_SIZE_KB_avg': wk / n, 'hbm_side_bytes_avg': b}
        if 'gn_stats_kernel' in k or 'gn_apply_kernel' in k:
            fwd += b
    shapes = ((4096, 320), (4096, 640), (1024, 640), (1024, 1920), (256, 1280), (64, 2560))
    res['avg_algorithmic_bytes_fwd_per_launch'] = sum(16 * hw * c * 4.0 for hw, c in shapes) / len(shapes)
    res['avg_hbm_side_bytes_fwd_per_launch'] = fwd          # stats + apply kernels of one forward call
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in ('avg_algorithmic_bytes_fwd_per_launch', 'avg_hbm_side_bytes_fwd_per_launch')}))


if __name__ == '__main__':
    sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
    main()
