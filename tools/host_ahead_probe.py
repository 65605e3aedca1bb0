#!/usr/bin/env python
"""How far does the host run ahead of the device in the bench loop?  Per iteration: host time to enqueue it (no sync), and
the device-side duration between per-iteration events.  If enqueue ~ device time in steady state the host is throttled by
the device (good: device-bound); if the device-side gaps exceed the kernels' busy time the device waits for the host."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402
from sid_lsg_amd.optim import FusedAdamEMA  # noqa: E402
from sid_lsg_amd.sd_util import load_sd15  # noqa: E402
from sid_lsg_amd.sid_step import SiDStep  # noqa: E402
from sid_lsg_amd.text import TextConditioner  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
b, lat = int(os.environ.get('B', 8)), 64
phi, vae, sched, te, tok = load_sd15('random:sd15', None, dev, torch.bfloat16, seed=0)
psi, G, G_ema = phi.clone_network(), phi.clone_network(), phi.clone_network(with_grad_buffers=False)
te.to(torch.bfloat16)
cond = TextConditioner(tok, te)
step = SiDStep(G, psi, phi, G_ema, sched, FusedAdamEMA(psi.parameters(), lr=1e-6), FusedAdamEMA(G.parameters(), lr=1e-6), alpha=1.0,
               cfg_train_fake=1.5, cfg_eval_fake=1.5, cfg_eval_real=1.5, batch_gpu_total=b, init_timestep=625)
gen = torch.Generator(device=dev)


def one(it):
    gen.manual_seed(it)
    inputs = dict(A=[], B=[])
    t0 = time.time()
    for k, ph in enumerate(('A', 'B')):
        prompts = bench.synth_prompts(b, seed=it * 2 + k)
        inputs[ph].append(dict(z=torch.randn(b, 4, lat, lat, device=dev, generator=gen), noise=torch.randn(b, 4, lat, lat, device=dev, generator=gen),
                               t=torch.randint(20, 980, (b,), device=dev, generator=gen), cond=cond.encode(prompts), uncond=cond.uncond(b)))
    t1 = time.time()
    step.iteration(inputs, ema_beta=0.99)
    return t1 - t0, time.time() - t1


for it in range(3):
    one(it)
torch.cuda.synchronize()
N = 12
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
evs[0].record()
host = []
T0 = time.time()
for it in range(N):
    host.append(one(3 + it))
    evs[it + 1].record()
t_enq = time.time() - T0
torch.cuda.synchronize()
t_all = time.time() - T0
print(f'{N} iterations: host finished enqueueing after {t_enq * 1e3:.0f} ms, device finished after {t_all * 1e3:.0f} ms ({t_all / N * 1e3:.1f} ms/iteration)')
print('host per iteration (input prep + text encode, step.iteration) ms:', ' '.join(f'{a * 1e3:.0f}+{c * 1e3:.0f}' for a, c in host))
print('device per iteration ms:', ' '.join(f'{evs[i].elapsed_time(evs[i + 1]):.0f}' for i in range(N)))
