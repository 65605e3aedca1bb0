#!/usr/bin/env python
"""What does the process group's flight recorder say about retired collectives (world 1 on RCCL)?  Probe for a deterministic drain in front
of the graph capture (sid_step.SiDStep.iteration_graphed) instead of a fixed pause."""
import os
import pickle
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29571')
import torch
import torch.distributed as dist

dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.ones(1 << 20, device='cuda')
for _ in range(4):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.time()
from torch._C._distributed_c10d import _dump_nccl_trace
for i in range(8):
    tr = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    ent = tr.get('entries', [])
    act = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=True)).get('entries', [])
    print(f'{(time.time() - t0) * 1e3:7.1f} ms: {len(ent)} entries, states {[e.get("state") for e in ent]}, retired {[e.get("retired") for e in ent]}, '
          f'discovered {[e.get("time_discovered_completed_ns") is not None for e in ent]}, active-only {len(act)}')
    time.sleep(0.05)
print('keys of an entry:', sorted(ent[0].keys()) if ent else None)
dist.destroy_process_group()
