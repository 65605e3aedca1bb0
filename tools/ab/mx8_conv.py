#!/usr/bin/env python
"""sidlsg_conv3x3_mx8 next to the bf16 conv on the SD1.5 shapes (batch 16).   python tools/ab/mx8_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def timeit(fn, iters=20, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


B = 16
for H, cin, cout in ((64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280)):
    x = torch.randn(B, H, H, cin, device=dev).to(BF16)
    w16 = (torch.randn(cout, 9 * cin, device=dev) * 0.02).to(BF16)
    bias = torch.randn(cout, device=dev)
    x8, w8 = ops.cast_fp8(x), ops.Fp8Weight(w16)
    t16 = timeit(lambda: ops.conv3x3(x, w16, bias=bias))
    t8 = timeit(lambda: ops.conv3x3_mx8(x8, w8, bias=bias))
    fl = 2.0 * B * H * H * cout * 9 * cin
    print(f'B{B} {H}x{H} {cin}->{cout}: mx8 {t8 * 1e6:7.1f} us {fl / t8 / 1e12:6.0f} TF/s | bf16 {t16 * 1e6:7.1f} us {fl / t16 / 1e12:6.0f} TF/s | x{t16 / t8:4.2f}', flush=True)
