#!/usr/bin/env python
"""Does the FF block (FF-in GEMM -> GEGLU -> FF-out GEMM + residual) run faster when the token rows are processed in chunks whose
intermediate h / y fit the 256 MiB Infinity Cache?  (Round 5: deferring the weight-gradient slab reductions was SLOWER because the slabs
left the cache before they were read back; the same effect should make a 672 MB h of the grouped pass at the 64x64 stage expensive.)
Forward only, no autograd: gemm + geglu + gemm(+res) on row slices, events around 10 repetitions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def run(M, C, chunks, reps=10, bwd=False):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, C, device=dev, generator=g).to(BF16)
    w1 = (torch.randn(8 * C, C, device=dev, generator=g) * C ** -0.5).to(BF16)
    b1 = torch.randn(8 * C, device=dev, generator=g)
    w2 = (torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5).to(BF16)
    b2 = torch.randn(C, device=dev, generator=g)
    h = torch.empty(M, 8 * C, device=dev, dtype=BF16)
    y = torch.empty(M, 4 * C, device=dev, dtype=BF16)
    out = torch.empty(M, C, device=dev, dtype=BF16)
    step = M // chunks

    def once():
        for c in range(chunks):
            r0, r1 = c * step, (c + 1) * step
            ops.gemm(x[r0:r1], w1, out=h[r0:r1], bias=b1)
            lib.sidlsg_geglu_fwd(h[r0:r1].data_ptr(), y[r0:r1].data_ptr(), step, 4 * C, ops._s())
            ops.gemm(y[r0:r1], w2, out=out[r0:r1], bias=b2, res=x[r0:r1])
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


for M, C in ((131072, 320), (65536, 320), (32768, 640), (16384, 640)):
    base = None
    for chunks in (1, 2, 4, 8):
        t, out = run(M, C, chunks)
        if base is None:
            base, ref = t, out.clone()
        assert torch.equal(out, ref)
        print(f'M {M} C {C} h = {M * 8 * C * 2 / 1e6:.0f} MB: {chunks} chunk(s) {t:8.1f} us  ({t / base:.3f})')
