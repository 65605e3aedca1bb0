#!/usr/bin/env python
"""GEGLU fusions: v3 kernels (SIDLSG_P8_GEGLU=0) against the 256 x 320 kernels (default), stand-alone, per case: forward without h, forward with h,
backward; single-set entry points at the FeedForward shapes of SD1.5 (batch 16 and the grouped batch 32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402
from sid_lsg_amd.ops import _p, _s  # noqa: E402

dev = torch.device('cuda')
ops.ensure_workspace(dev)
BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('SIDLSG_P8_GEGLU =', os.environ.get('SIDLSG_P8_GEGLU', '1'))
for M, C in [(65536, 320), (131072, 320), (16384, 640), (32768, 640), (4096, 1280), (8192, 1280)]:
    N2, F = 8 * C, 4 * C
    x = torch.randn(M, C, device=dev).to(BF)
    w1 = (torch.randn(N2, C, device=dev) * C ** -0.5).to(BF)
    b1 = torch.randn(N2, device=dev)
    w2t = (torch.randn(F, C, device=dev) * F ** -0.5).to(BF)
    h = torch.empty(M, N2, device=dev, dtype=BF)
    y = torch.empty(M, F, device=dev, dtype=BF)
    dout = torch.randn(M, C, device=dev).to(BF)
    dh = torch.empty(M, N2, device=dev, dtype=BF)
    t0 = timeit(lambda: lib.sidlsg_gemm_geglu_bf16(_p(x), C, _p(w1), None, N2, _p(y), F, _p(b1), M, N2, C, _s()))
    t1 = timeit(lambda: lib.sidlsg_gemm_geglu_bf16(_p(x), C, _p(w1), _p(h), N2, _p(y), F, _p(b1), M, N2, C, _s()))
    t2 = timeit(lambda: lib.sidlsg_gemm_geglu_bwd_bf16(_p(dout), C, _p(w2t), _p(h), _p(dh), N2, M, F, C, _s()))
    print(f'M {M:6d} C {C:4d}: fwd no h {t0:7.1f} us   fwd + h {t1:7.1f} us   bwd {t2:7.1f} us')
