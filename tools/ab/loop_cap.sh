#!/bin/bash
# Repeats the RCCL-in-captured-iteration worker (tests/mp_ddp_worker.py graph) under torchrun until it fails: how the
# hipErrorCapturedEvent abort of the process group's watchdog was caught (1-2 in 10 runs before SiDStep.iteration_graphed
# paused for the watchdog ahead of the capture; 0 in 24 after).   bash tools/ab/loop_cap.sh
cd /root/repo/tests
for i in $(seq 1 12); do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600+i)) mp_ddp_worker.py graph /tmp/gg$i > /tmp/cap_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v "^\s*$" /tmp/cap_$i.log | grep -v "Traceback\|File \"/usr" | head -60; break; fi
done
