#!/usr/bin/env python
"""The step's SMALL dense GEMMs (the dispatcher's 64 x 64-tile class), for `rocprofv3 --kernel-trace --stats` A/B runs of
SIDLSG_GEMM_S64 (gemm_s64_kernel: 3 K-tiles in flight) against gemm_bf16_kernel<64, 64, 0> (one process per setting)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
SHAPES = [(1232, 640, 768), (1232, 1280, 768), (1232, 2560, 768), (616, 640, 768), (616, 2560, 768), (1024, 1280, 1280), (1024, 3840, 1280),
          (512, 1280, 1280), (2048, 1280, 1280), (16, 1280, 320), (16, 1280, 1280), (1024, 1280, 5120)]
for M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    ref = a.float() @ w.float().t() + bias
    for _ in range(8):
        y = ops.gemm(a, w, bias=bias)
    torch.cuda.synchronize()
    e = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    print(f'{M}x{N}x{K}: rel err {e:.2e}')
    assert e < 1.5e-2
print('ok')
