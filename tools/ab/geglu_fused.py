#!/usr/bin/env python
"""FF-in projection + GEGLU: the fused kernel against linear + geglu, forward only, at the step's FF shapes (batch 16 and 8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def timeit(fn, iters=20, warm=5):
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


for B in (16, 8):
    for hw, c in ((4096, 320), (1024, 640), (256, 1280)):
        M, K, F = B * hw, c, 4 * c
        x = torch.randn(M, K, device=dev).to(BF16)
        w = torch.nn.Parameter((torch.randn(2 * F, K, device=dev) * K ** -0.5), requires_grad=False)
        b = torch.nn.Parameter(torch.randn(2 * F, device=dev), requires_grad=False)
        w16 = w.detach().to(BF16).contiguous()
        w16t = w16.t().contiguous()
        h = torch.empty(M, 2 * F, device=dev, dtype=BF16)
        y = torch.empty(M, F, device=dev, dtype=BF16)
        t_lin = timeit(lambda: ops.gemm(x, w16, out=h, bias=b))
        t_geg = timeit(lambda: lib.sidlsg_geglu_fwd(h.data_ptr(), y.data_ptr(), M, F, ops._s()))
        t_fus = timeit(lambda: lib.sidlsg_gemm_geglu_bf16(x.data_ptr(), K, w16.data_ptr(), h.data_ptr(), 2 * F, y.data_ptr(), F, b.data_ptr(), M, 2 * F, K, ops._s()))
        t_fnh = timeit(lambda: lib.sidlsg_gemm_geglu_bf16(x.data_ptr(), K, w16.data_ptr(), None, 2 * F, y.data_ptr(), F, b.data_ptr(), M, 2 * F, K, ops._s()))
        os.environ['X'] = '1'
        print(f'M {M} F {F} K {K}: linear {t_lin:7.1f} us + geglu {t_geg:7.1f} us = {t_lin + t_geg:7.1f} | fused (h kept) {t_fus:7.1f} us | fused (no h) {t_fnh:7.1f} us')
