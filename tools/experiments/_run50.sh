set -x
mkdir -p gpurun_out; rm -f gpurun_out/r50_ab.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])" >> gpurun_out/r50_ab.log; }
for i in 1 2 3 4 5 6 7 8; do run SIDLSG_GEGLU_FUSE_MIN_K=640; run SIDLSG_GEGLU_FUSE_MIN_K=320; done
python - <<'P'
import collections, statistics
d = collections.defaultdict(list)
for l in open('gpurun_out/r50_ab.log'):
    k, v = l.split(); d[k].append(float(v))
for k, v in d.items(): print(k, 'median', round(statistics.median(v), 2), 'mean', round(statistics.mean(v), 2), 'min', round(min(v), 2), [round(x, 1) for x in v])
P
