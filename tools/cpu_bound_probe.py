#!/usr/bin/env python
"""Is the step limited by the host thread that enqueues it?  Times, per iteration at the bench configuration, (a) until
step.iteration() returns (everything enqueued) and (b) until the device is idle.  If (a) ~ (b) the GPU waits for the host."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (re-uses its helpers)
from sid_lsg_amd._lib import lib
from sid_lsg_amd.optim import FusedAdamEMA
from sid_lsg_amd.sd_util import load_sd15
from sid_lsg_amd.sid_step import SiDStep
from sid_lsg_amd.text import TextConditioner
lib.load()
dev = torch.device('cuda:0')
b, lat = int(os.environ.get('B', 8)), 64
phi, vae, sched, te, tok = load_sd15('random:sd15', None, dev, torch.bfloat16, seed=0)
psi, G, G_ema = phi.clone_network(), phi.clone_network(), phi.clone_network(with_grad_buffers=False)
te.to(torch.bfloat16)
cond = TextConditioner(tok, te)
opt_f = FusedAdamEMA(psi.parameters(), lr=1e-6, betas=(0.0, 0.999), eps=1e-8)
opt_g = FusedAdamEMA(G.parameters(), lr=1e-6, betas=(0.0, 0.999), eps=1e-8)
step = SiDStep(G, psi, phi, G_ema, sched, opt_f, opt_g, alpha=1.0, cfg_train_fake=1.5, cfg_eval_fake=1.5, cfg_eval_real=1.5,
               batch_gpu_total=b, init_timestep=625)
gen = torch.Generator(device=dev)
def one(it):
    gen.manual_seed(it)
    inputs = dict(A=[], B=[])
    for k, ph in enumerate(('A', 'B')):
        prompts = bench.synth_prompts(b, seed=it * 2 + k)
        inputs[ph].append(dict(z=torch.randn(b, 4, lat, lat, device=dev, generator=gen), noise=torch.randn(b, 4, lat, lat, device=dev, generator=gen),
                               t=torch.randint(20, 980, (b,), device=dev, generator=gen), cond=cond.encode(prompts), uncond=cond.uncond(b)))
    return step.iteration(inputs, ema_beta=0.999)
for it in range(3):
    one(it)
torch.cuda.synchronize()
ta = tb = 0.0
N = 8
for it in range(N):
    t0 = time.perf_counter()
    one(10 + it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t0
print(f'batch {b}: enqueue {ta / N * 1e3:.1f} ms, until idle {tb / N * 1e3:.1f} ms per iteration')
