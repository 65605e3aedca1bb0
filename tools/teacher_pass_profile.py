#!/usr/bin/env python
"""Run only the CFG teacher forward (phi on [uncond; cond], 2b = 16 samples) a few times: for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sid_lsg_amd.scheduler import DDPMScheduler  # noqa: E402
from sid_lsg_amd.sd_util import hip_denoise, hip_prepare_denoise  # noqa: E402
from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition  # noqa: E402

dev = torch.device('cuda')
b, lat = 8, 64
phi = HipUNet2DCondition(CONFIGS['sd15']).materialize(dev, seed=0, with_grad_buffers=False)
phi.requires_grad_(False)
sched = DDPMScheduler().to(dev)
g = torch.Generator(device=dev).manual_seed(0)
ctx = torch.randn(b, 77, 768, device=dev, generator=g).to(torch.bfloat16)
with torch.no_grad():
    prep = hip_prepare_denoise(torch.randn(b, 4, lat, lat, device=dev, generator=g), torch.randn(b, 4, lat, lat, device=dev, generator=g),
                               torch.randint(20, 980, (b,), device=dev, generator=g), ctx, ctx.clone(), sched, True)
    for _ in range(6):
        hip_denoise(phi, prep, 1.5, predict_x0=True)
torch.cuda.synchronize()
