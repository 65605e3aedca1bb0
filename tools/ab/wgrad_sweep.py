#!/usr/bin/env python
"""Dense weight-gradient shapes of one SD1.5 backward pass (CFG batch 16 and generator batch 8) under the current dispatch:
SIDLSG_WGRAD_SQ160=0/1 python tools/ab/wgrad_sweep.py            (dense layers)
SIDLSG_WGRAD_CONV160=0/1/2 python tools/ab/wgrad_sweep.py conv   (3x3 convs, weighted by their count in the UNet)"""
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


# (output H, Cin, Cout, stride, fused 2x upsample, count in the SD1.5 UNet)
CONVS = [(64, 320, 320, 1, 0, 7), (32, 320, 320, 2, 0, 1), (32, 320, 640, 1, 0, 1), (32, 640, 640, 1, 0, 6), (16, 640, 640, 2, 0, 1),
         (16, 640, 1280, 1, 0, 1), (16, 1280, 1280, 1, 0, 6), (8, 1280, 1280, 2, 0, 1), (8, 1280, 1280, 1, 0, 11), (8, 2560, 1280, 1, 0, 3),
         (16, 1280, 1280, 1, 1, 1), (16, 2560, 1280, 1, 0, 2), (16, 1920, 1280, 1, 0, 1), (32, 1280, 1280, 1, 1, 1), (32, 1920, 640, 1, 0, 1),
         (32, 1280, 640, 1, 0, 1), (32, 960, 640, 1, 0, 1), (64, 640, 640, 1, 1, 1), (64, 960, 320, 1, 0, 1), (64, 640, 320, 1, 0, 2)]
if 'conv' in sys.argv[1:]:
    tot = {}
    for B in (16, 8):
        for H, cin, cout, stride, ups, cnt in CONVS:
            Hi = H * stride // (2 if ups else 1)           # stored input extent
            x, dy = torch.randn(B, Hi, Hi, cin, device=dev).bfloat16(), torch.randn(B, H, H, cout, device=dev).bfloat16()
            dw, db = torch.zeros(cout, 9 * cin, device=dev), torch.zeros(cout, device=dev)
            Hl = Hi * (2 if ups else 1)                    # logical input extent
            t = timeit(lambda: lib.sidlsg_conv3x3_wgrad_bf16(dy.data_ptr(), cout, x.data_ptr(), cin, dw.data_ptr(), db.data_ptr(), B, Hl, Hl, cin, cout, stride, ups, ops._s()))
            print(f'B {B:2d} out {H:2d}x{H:2d} {cin:5d}->{cout:5d} s{stride} u{ups} x{cnt:2d}: {t:8.1f} us  {2.0 * B * H * H * cout * 9 * cin / t / 1e6:7.1f} TF/s')
            tot[B] = tot.get(B, 0) + t * cnt
    print('weighted sum', {k: round(v, 1) for k, v in tot.items()})
    sys.exit(0)

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
