# per-(kernel, grid) ranking of one iteration with the side streams folded into the main one (clean stand-alone durations)
set -x
root=$(pwd); out=$root/gpurun_out/r38; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SIDLSG_TEACHER_STREAM=0 SIDLSG_WGRAD_STREAM=0 SIDLSG_EARLY_GFWD=0 timeout 900 rocprofv3 --kernel-trace -d $out/prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 2 > $out/bench.json 2> $out/bench.err
cd $root
tail -n 1 $out/bench.json | cut -c1-300
python tools/kernel_by_grid.py $out/prof/b_kernel_trace.csv "" > $out/by_grid_all.txt
python - <<'P'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0]); tot = 0
with open('gpurun_out/r38/prof/b_kernel_trace.csv') as f:
    for r in csv.DictReader(f):
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d
        k = (r['Kernel_Name'].split('(')[0][-48:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
        acc[k][0] += d; acc[k][1] += 1
with open('gpurun_out/r38/by_grid_top.txt', 'w') as o:
    o.write(f'total kernel time {tot/1e6:.1f} ms\n')
    for k, (d, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:150]:
        o.write(f'{d/1e6:8.2f} ms {100*d/tot:5.2f} %  n={n:5d} avg {d/n/1e3:7.1f} us  {k}\n')
P
rm -rf $out/prof
head -80 $out/by_grid_top.txt
