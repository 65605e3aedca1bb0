#!/usr/bin/env python
"""Workload for the attention PMC passes (tools/attn_pmc.sh): self-attention forward + backward through the C ABI,
N=4096 h8 d40 and N=4096 h5 d64 at B=16, plain and pre-scaled (`_ps`) entry points, 3 calls each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ab'))
from attn_ps import BF16, F32, call, dev  # noqa: E402

torch.manual_seed(0)
B = 16
for N, heads, D in ((4096, 8, 40), (4096, 5, 64)):
    C = heads * D
    qkv = torch.randn(B, N, 3 * C, device=dev).to(BF16)
    do = torch.randn(B, N, C, device=dev).to(BF16)
    o = torch.empty(B, N, C, device=dev, dtype=BF16)
    lse = torch.empty(B, heads, N, device=dev, dtype=F32)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B, heads, N, device=dev, dtype=F32)
    for ps in (False, True):
        for _ in range(3):
            call(ps, qkv, o, lse, heads, D)
            call(ps, qkv, o, lse, heads, D, do, dqkv, delta)
torch.cuda.synchronize()
