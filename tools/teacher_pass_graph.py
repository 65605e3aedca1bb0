#!/usr/bin/env python
"""Is the CFG teacher pass (phi forward on 2b = 16 samples + guidance + x0, no grad) device-bound when it is launched eagerly?
Eager (events around 5 back-to-back passes, host enqueue time beside it) against the same pass replayed as ONE HIP graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sid_lsg_amd.scheduler import DDPMScheduler  # noqa: E402
from sid_lsg_amd.sd_util import hip_denoise, hip_prepare_denoise  # noqa: E402
from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition  # noqa: E402

dev = torch.device('cuda')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lat = 64
phi = HipUNet2DCondition(CONFIGS['sd15']).materialize(dev, seed=0, with_grad_buffers=False)
phi.requires_grad_(False)
sched = DDPMScheduler().to(dev)
g = torch.Generator(device=dev).manual_seed(0)
ctx = torch.randn(b, 77, 768, device=dev, generator=g).to(torch.bfloat16)


def timed(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    h0 = time.time()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.time() - h0) / n * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, host


with torch.no_grad():
    prep = hip_prepare_denoise(torch.randn(b, 4, lat, lat, device=dev, generator=g), torch.randn(b, 4, lat, lat, device=dev, generator=g),
                               torch.randint(20, 980, (b,), device=dev, generator=g), ctx, ctx.clone(), sched, True)
    run = lambda: hip_denoise(phi, prep, 1.5, predict_x0=True)  # noqa: E731
    for _ in range(3):
        run()
    dev_ms, host_ms = timed(run)
    print(f'eager : {dev_ms:7.3f} ms per pass on the device, {host_ms:7.3f} ms of host time to enqueue it')
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        run()
    for _ in range(2):
        graph.replay()
    gdev, ghost = timed(graph.replay)
    print(f'graph : {gdev:7.3f} ms per pass on the device, {ghost:7.3f} ms of host time per replay')
