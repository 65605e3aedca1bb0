set -x
mkdir -p gpurun_out
for v in 0 1; do echo "== SIDLSG_GN_ONEPASS=$v"; SIDLSG_GN_ONEPASS=$v python tools/bench_kernels.py norm 2>/dev/null | grep GN; done > gpurun_out/r7_gn_micro.log
for v in 0 1; do echo "== SIDLSG_GN_ONEPASS=$v batch 32"; SIDLSG_GN_ONEPASS=$v python tools/bench_kernels.py norm --batch 32 2>/dev/null | grep GN; done >> gpurun_out/r7_gn_micro.log
for v in 0 1; do echo "== SIDLSG_GN_ONEPASS=$v batch 8"; SIDLSG_GN_ONEPASS=$v python tools/bench_kernels.py norm --batch 8 2>/dev/null | grep GN; done >> gpurun_out/r7_gn_micro.log
cat gpurun_out/r7_gn_micro.log
bash tools/collect_traffic.sh gpurun_out/r7_traffic > gpurun_out/r7_traffic.log 2>&1
tail -n 12 gpurun_out/r7_traffic.log
