#!/bin/bash
cd /root/repo/tests
for i in $(seq 1 12); do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600+i)) mp_ddp_worker.py graph /tmp/gg$i > /tmp/cap_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v "^\s*$" /tmp/cap_$i.log | grep -v "Traceback\|File \"/usr" | head -60; break; fi
done
