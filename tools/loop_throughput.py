#!/usr/bin/env python
"""Throughput of the PRODUCT loop (sid_lsg_amd.training_loop.training_loop) at the bench configuration: SD1.5 (random init),
kappa 1.5, 512^2, batch 8 on one GPU -- what `Timing/images_per_sec` reports, next to bench.py's figure for the bare step.
    python tools/loop_throughput.py [iterations=40]        ($SIDLSG_PREFETCH_INPUTS=0: inputs prepared serially)"""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SIDLSG_ALLOW_RANDOM_INIT', '1')
from sid_lsg_amd import training_loop as tl  # noqa: E402
from sid_lsg_amd.dnnlib_util import EasyDict  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sync_every = '--sync' in sys.argv
tmp = tempfile.mkdtemp()
pdir = os.path.join(tmp, 'prompts')
os.makedirs(pdir)
with open(os.path.join(pdir, 'aesthetics_65_plus.txt'), 'w') as f:
    for i in range(4096):
        f.write(f'a photo of object number {i} on a table, studio light, {i % 7} colours\n')
bs = 8
marks = []


def on_it(it, lf, lg):
    marks.append(time.time())


kw = dict(run_dir=os.path.join(tmp, 'run'), network_kwargs=EasyDict(use_fp16=True, compute_dtype='bf16'),
          dataset_prompt_text_kwargs=EasyDict(class_name='sid_lsg_amd.data.PromptDataset', path=pdir, resolution=512, prompt_only=True),
          fake_score_optimizer_kwargs=EasyDict(class_name='torch.optim.Adam', lr=1e-6, betas=[0.0, 0.999], eps=1e-8),
          g_optimizer_kwargs=EasyDict(class_name='torch.optim.Adam', lr=1e-6, betas=[0.0, 0.999], eps=1e-8),
          seed=0, batch_size=bs, batch_gpu=bs, total_kimg=iters * bs / 1000.0, ema_halflife_kimg=50, kimg_per_tick=(iters - 8) * bs / 1000.0,
          snapshot_ticks=None, state_dump_ticks=None, alpha=1.0, tmax=980, tmin=20, device=torch.device('cuda:0'), metrics=None,
          init_timestep=625, cfg_train_fake=1.5, cfg_eval_fake=1.5, cfg_eval_real=1.5, resolution=512, enable_xformers=False,
          pretrained_model_name_or_path='random:sd15', pretrained_vae_model_name_or_path='random:sd15',
          on_iteration=on_it if sync_every else None)
os.makedirs(kw['run_dir'])
tl.training_loop(**kw)
torch.cuda.synchronize()
for ln in open(os.path.join(kw['run_dir'], 'stats_1.000000.jsonl')):
    d = json.loads(ln)
    print({k: round(v['mean'], 3) for k, v in d.items() if isinstance(v, dict) and k in ('Progress/tick', 'Timing/images_per_sec', 'Timing/sec_per_kimg', 'Resources/peak_gpu_mem_gb')})
if sync_every and len(marks) > 12:
    dt = (marks[-1] - marks[8]) / (len(marks) - 9)
    print(f'per-iteration observer (a host sync every iteration): {dt * 1e3:.1f} ms per iteration = {bs / dt:.2f} images/s')
