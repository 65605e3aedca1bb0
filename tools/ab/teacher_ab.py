"""teacher pass (phi forward on the CFG batch) timed under alternative builds of the library, one process each."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    from sid_lsg_amd.scheduler import DDPMScheduler
    from sid_lsg_amd.sd_util import hip_denoise, hip_prepare_denoise
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    dev = torch.device('cuda')
    b, lat = 8, 64
    phi = HipUNet2DCondition(CONFIGS['sd15']).materialize(dev, seed=0, with_grad_buffers=False).requires_grad_(False)
    sched = DDPMScheduler().to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    ctx = torch.randn(b, 77, 768, device=dev, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        prep = hip_prepare_denoise(torch.randn(b, 4, lat, lat, device=dev, generator=g), torch.randn(b, 4, lat, lat, device=dev, generator=g),
                                   torch.randint(20, 980, (b,), device=dev, generator=g), ctx, ctx.clone(), sched, True)
        for _ in range(3):
            hip_denoise(phi, prep, 1.5, predict_x0=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip_denoise(phi, prep, 1.5, predict_x0=True)
        e1.record(); torch.cuda.synchronize()
    print(sys.argv[2], f'teacher pass {e0.elapsed_time(e1) / 10:.2f} ms')
else:
    for name in (sys.argv[1:] or ['base']):
        env = dict(os.environ)
        if name != 'base':
            env['SIDLSG_LIB'] = os.path.join(ROOT, 'tools', 'ab', f'lib{name}.so')
        subprocess.run([sys.executable, __file__, 'child', name], env=env)
