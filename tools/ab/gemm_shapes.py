#!/usr/bin/env python
"""Time a few dense GEMM shapes through the library selected by SIDLSG_LIB: python tools/ab/gemm_shapes.py M,N,K [M,N,K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
tag = os.path.basename(os.environ.get('SIDLSG_LIB', 'default')) + (' AS=0' if os.environ.get('SIDLSG_GEMM_AS') == '0' else '')
out = [f'{tag:22s}']
for spec in sys.argv[1:]:
    M, N, K = map(int, spec.split(','))
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out=c, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, w, out=c, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    out.append(f'{M}x{N}x{K}: {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF/s')
print('  '.join(out), flush=True)
