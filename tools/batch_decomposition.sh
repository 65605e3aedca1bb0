#!/bin/bash
# tools/batch_decomposition.sh OUTDIR : what part of the step does not scale with the batch?  bench.py as one HIP graph per iteration (no host floor)
# at batch_gpu 1 ... 16 plus the eager loop at the small batches; one JSON line per run, per-family est. ms/step in `fracs` (third entry).
out=${1:-gpurun_out/decomp}
mkdir -p $out
for b in 1 2 4 8 12 16; do
  python bench.py --graph --batch-gpu $b --steps 10 --warmup 3 --no-cpu-baseline > $out/graph_b$b.json 2> $out/graph_b$b.err
done
for b in 1 2 4; do
  python bench.py --batch-gpu $b --steps 10 --warmup 3 --no-cpu-baseline > $out/eager_b$b.json 2> $out/eager_b$b.err
done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, '*_b*.json'))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    mode, b = os.path.basename(f)[:-5].split('_b')
    rows.append((mode, int(b), d['ms_per_step'], d['value'], {k: v[2] for k, v in d.get('fracs', {}).items()}))
rows.sort()
fams = sorted({k for r in rows for k in r[4]})
with open(os.path.join(out, 'table.txt'), 'w') as fh:
    hdr = f"{'mode':6s} {'b':>3s} {'ms/step':>9s} {'img/s':>7s} " + ' '.join(f'{k:>10s}' for k in fams)
    print(hdr); fh.write(hdr + '\n')
    for mode, b, ms, v, fr in rows:
        line = f'{mode:6s} {b:3d} {ms:9.2f} {v:7.2f} ' + ' '.join(f'{fr.get(k, float("nan")):10.2f}' for k in fams)
        print(line); fh.write(line + '\n')
    g = {b: ms for mode, b, ms, v, fr in rows if mode == 'graph'}
    if 4 in g and 8 in g and 16 in g:
        for lo, hi in ((4, 8), (8, 16)):
            slope = (g[hi] - g[lo]) / (hi - lo)
            line = f'graph: slope {lo}->{hi}: {slope:.2f} ms/image, intercept {g[lo] - slope * lo:.1f} ms'
            print(line); fh.write(line + '\n')
PY
