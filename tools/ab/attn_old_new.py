#!/usr/bin/env python
"""Forward / backward time of the self-attention shapes through the library selected by SIDLSG_LIB (plain entry points, and
the pre-scaled ones when the library has them).  One line per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_ps import BF16, F32, call, dev, timeit  # noqa: E402

B = 16
tag = os.path.basename(os.environ.get('SIDLSG_LIB', 'default'))
for N, heads, D in ((4096, 8, 40), (1024, 8, 80), (4096, 5, 64)):
    C = heads * D
    qkv = torch.randn(B, N, 3 * C, device=dev).to(BF16)
    do = torch.randn(B, N, C, device=dev).to(BF16)
    o = torch.empty(B, N, C, device=dev, dtype=BF16)
    lse = torch.empty(B, heads, N, device=dev, dtype=F32)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B, heads, N, device=dev, dtype=F32)
    line = f'{tag:18s} N{N} d{D}:'
    for ps in ((False,) if 'old' in tag else (False, True)):
        tf = timeit(lambda: call(ps, qkv, o, lse, heads, D), iters=20)
        tb = timeit(lambda: call(ps, qkv, o, lse, heads, D, do, dqkv, delta), iters=10)
        line += f"  {'ps' if ps else 'plain'} fwd {tf * 1e6:7.1f} bwd {tb * 1e6:7.1f} us"
    print(line, flush=True)
