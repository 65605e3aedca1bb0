#!/usr/bin/env python
"""Do two builds of the library (SIDLSG_LIB) give bit-identical LayerNorm / GroupNorm results?   norm_variant_compare.py dump OUT.pt
writes the outputs (forward and backward) of seeded inputs over the shapes of the tiny and full-size networks;
norm_variant_compare.py cmp A.pt B.pt compares two dumps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(path):
    from sid_lsg_amd import ops
    dev = torch.device('cuda')
    out = {}
    g = torch.Generator().manual_seed(0)
    for rows, C in [(64, 80), (128, 160), (37, 320), (4096, 320), (1000, 640), (256, 1280), (65536, 320), (16384, 640), (77, 96)]:
        x = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
        ga, be = torch.nn.Parameter(torch.randn(C, generator=g).to(dev)), torch.nn.Parameter(torch.randn(C, generator=g).to(dev))
        ga.grad, be.grad = torch.zeros_like(ga), torch.zeros_like(be)
        y = ops.layer_norm(x, ga, be)
        dy = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev)
        y.backward(dy)
        ops.flush_deferred() if hasattr(ops, 'flush_deferred') else None
        out[f'ln_{rows}_{C}'] = (y.detach().cpu(), x.grad.cpu(), ga.grad.cpu(), be.grad.cpu())
    for B, H, C, G, silu in [(1, 8, 80, 8, True), (2, 8, 160, 8, False), (1, 4, 320, 8, True), (2, 16, 80, 8, True), (16, 64, 320, 32, True), (16, 32, 640, 32, False),
                             (3, 64, 320, 32, True), (16, 16, 1280, 32, True), (5, 32, 960, 32, True), (2, 64, 640, 32, False)]:
        x = torch.randn(B, H, H, C, generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
        ga, be = torch.nn.Parameter(torch.randn(C, generator=g).to(dev)), torch.nn.Parameter(torch.randn(C, generator=g).to(dev))
        ga.grad, be.grad = torch.zeros_like(ga), torch.zeros_like(be)
        y = ops.group_norm(x, ga, be, G, 1e-5, silu)
        dy = torch.randn(B, H, H, C, generator=g).to(torch.bfloat16).to(dev)
        y.backward(dy)
        ops.flush_deferred() if hasattr(ops, 'flush_deferred') else None
        out[f'gn_{B}_{H}_{C}_{G}_{int(silu)}'] = (y.detach().cpu(), x.grad.cpu(), ga.grad.cpu(), be.grad.cpu())
    torch.cuda.synchronize()
    torch.save(out, path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    for k in A:
        msgs = []
        for name, u, v in zip(('y', 'dx', 'dgamma', 'dbeta'), A[k], B[k]):
            if torch.equal(u, v):
                msgs.append(f'{name} equal')
            else:
                d = (u.float() - v.float()).abs()
                msgs.append(f'{name} DIFF n={int((d > 0).sum())} max {float(d.max()):.3g} (ref max {float(v.float().abs().max()):.3g})')
        print(f'{k:24s} ' + '; '.join(msgs))


if __name__ == '__main__':
    dump(sys.argv[2]) if sys.argv[1] == 'dump' else cmp(sys.argv[2], sys.argv[3])
