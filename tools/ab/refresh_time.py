import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
dev = torch.device('cuda')
net = HipUNet2DCondition(CONFIGS['sd15']).materialize(dev, seed=0, with_grad_buffers=False)
for cast in (False, True):
    for _ in range(3): net.refresh_compute_weights(cast=cast)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): net.refresh_compute_weights(cast=cast)
    e1.record(); torch.cuda.synchronize()
    print(f'refresh_compute_weights(cast={cast}): {e0.elapsed_time(e1) / 10:.3f} ms')
