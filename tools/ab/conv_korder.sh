#!/bin/bash
# In-session A/B of the conv K-tile visiting order: timing of the conv micro-benchmark for each library variant + FETCH_SIZE
# (L2 -> fabric reads) for two of them.   bash tools/ab/conv_korder.sh OUTDIR lib1 lib2 ...
root=$(pwd); out=$root/$1; shift
mkdir -p $out
for rep in 1 2; do
  for l in "$@"; do
    SIDLSG_LIB=$root/tools/ab/lib$l.so python tools/bench_kernels.py conv > $out/time_${l}_$rep.log 2>&1
  done
done
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  SIDLSG_LIB=$root/tools/ab/lib$l.so rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/F_$l -o f --output-format csv -- python $root/tools/bench_kernels.py conv > $out/F_$l.log 2>&1
  python - $out/F_$l/f_counter_collection.csv > $out/fetch_$l.txt <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'gemm_v3_kernel' in r['Kernel_Name']:
        k = (r['Kernel_Name'][:40], r['Grid_Size'])
        acc[k][0] += float(r['Counter_Value']) * 1024 * 2; acc[k][1] += 1
for k, (b, n) in sorted(acc.items()):
    print(k, n, f'{b / n / 1e6:9.1f} MB per launch')
PY
  rm -rf $out/F_$l
done
cd $root
