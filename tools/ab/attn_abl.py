#!/usr/bin/env python
"""Time the pre-scaled forward (N=4096 h8 d40 / h5 d64, B=16) of the library selected by SIDLSG_LIB; one line per run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_ps import BF16, F32, call, dev, timeit  # noqa: E402

B = 16
out = [os.path.basename(os.environ.get('SIDLSG_LIB', 'default'))]
for N, heads, D in ((4096, 8, 40), (4096, 5, 64)):
    C = heads * D
    qkv = torch.randn(B, N, 3 * C, device=dev).to(BF16)
    o = torch.empty(B, N, C, device=dev, dtype=BF16)
    lse = torch.empty(B, heads, N, device=dev, dtype=F32)
    t = timeit(lambda: call(True, qkv, o, lse, heads, D), iters=20)
    out.append(f'd{D}: {t * 1e6:7.1f} us')
print('  '.join(out), flush=True)
