#!/usr/bin/env python
"""Dense weight-gradient shapes of one SD1.5 backward pass (CFG batch 16 and generator batch 8) under the current dispatch:
SIDLSG_WGRAD_SQ160=0/1 python tools/ab/wgrad_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
ops.ensure_workspace(dev)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot = {}
for B in (16, 8):
    for hw, c in ((4096, 320), (1024, 640), (256, 1280)):
        M = B * hw
        for N, K in ((c, c), (3 * c, c), (8 * c, c), (c, 4 * c)):
            dy, a = torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16()
            dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
            t = timeit(lambda: lib.sidlsg_wgrad_bf16(dy.data_ptr(), N, a.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K, ops._s()))
            print(f'M {M:6d} N {N:5d} K {K:5d}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s')
            tot[B] = tot.get(B, 0) + t
print('sum', {k: round(v, 1) for k, v in tot.items()})
