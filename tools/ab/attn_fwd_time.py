#!/usr/bin/env python
"""Forward attention time only (pre-scaled-query entry point), for ablation builds: python tools/ab/attn_fwd_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
out = [os.path.basename(os.environ.get('SIDLSG_LIB', 'default'))]
for N, heads, D, B in ((4096, 8, 40, 16), (4096, 5, 64, 16), (1024, 8, 80, 16)):
    C = heads * D
    qkv = torch.randn(B, N, 3 * C, device=dev)
    qkv[:, :, :C] *= 1.4426950408889634 / D ** 0.5            # pre-scaled queries, as the projection delivers them
    qkv = qkv.to(torch.bfloat16)
    o = torch.empty(B, N, C, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, heads, N, device=dev, dtype=torch.float32)
    s = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.sidlsg_attn_fwd_ps(qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C, o.data_ptr(), lse.data_ptr(), B, heads, N, N, D,
                                        3 * C, 3 * C, 3 * C, C, N * 3 * C, N * 3 * C, N * 3 * C, N * C, s)
    for _ in range(40):                # long warm-up: the clocks need a few ms of load to ramp (3 launches read 20 % slow)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out.append(f'N{N} d{D}: {e0.elapsed_time(e1) * 50:7.1f} us')
print('  '.join(out), flush=True)
