set -x
mkdir -p gpurun_out
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    s=torch.cuda.Stream(priority=p); print(p, s.priority)
"
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r36_ab.log; }
rm -f gpurun_out/r36_ab.log
run "X=1"
run "SIDLSG_WGRAD_PRIO=1"
run "SIDLSG_WGRAD_PRIO=-1"
run "SIDLSG_SIDE_PRIO=-1"
run "SIDLSG_SIDE_PRIO=1"
run "X=1"
run "SIDLSG_WGRAD_PRIO=1 SIDLSG_SIDE_PRIO=-1"
cat gpurun_out/r36_ab.log
