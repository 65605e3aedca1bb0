#!/usr/bin/env python
"""How much does running the generator's forward (phase B) beside the fake-score network's forward+backward (phase A) save?
Times each alone and both together on two streams (bench configuration: 16-sample CFG batch for psi, 8 samples for G)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sid_lsg_amd import ops
from sid_lsg_amd._lib import lib
from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
lib.load()
dev = torch.device('cuda:0')
BF = torch.bfloat16
psi = HipUNet2DCondition(CONFIGS['sd15']).materialize(dev, seed=0).requires_grad_(True)
G = HipUNet2DCondition(CONFIGS['sd15']).materialize(dev, seed=1).requires_grad_(True)
def inputs(b):
    x = torch.zeros(b, 64, 64, 8, device=dev); x[..., :4] = torch.randn(b, 64, 64, 4, device=dev)
    return x.to(BF), torch.randint(20, 980, (b,), device=dev), torch.randn(b, 77, 768, device=dev).to(BF)
x16, t16, c16 = inputs(16)
x8, t8, c8 = inputs(8)
side = torch.cuda.Stream()
ops.ensure_stream_workspace(side)
def psi_fb():
    eps = psi.forward_nhwc(x16, t16, c16)
    eps.float().square().mean().backward()
def g_fwd():
    return G.forward_nhwc(x8, t8, c8)
def timed(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def both():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        y = g_fwd()
    psi_fb()
    cur.wait_stream(side)
    return y
t1 = timed(psi_fb); t2 = timed(g_fwd); t12 = timed(both)
print(f'psi fwd+bwd (16): {t1:.1f} ms | G fwd (8): {t2:.1f} ms | sum {t1 + t2:.1f} | concurrent {t12:.1f} ms -> saves {t1 + t2 - t12:.1f} ms')
