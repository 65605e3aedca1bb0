#!/usr/bin/env python
"""Attention passes of the step's long shapes, for A/B runs of the block numbering (SIDLSG_ATTN_XCD) and of the rotated tile walk
(SIDLSG_ATTN_ROT) under `rocprofv3 --kernel-trace --stats` (the env knobs are read once per process: one process per setting).
Also checks the outputs against the default setting's (the first call of the pair writes, the second compares): a rotated walk
changes the order of the sums over tiles, nothing else."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
B = 16
PS = os.environ.get('ATTN_PS', '1') == '1'      # the `_ps` entry points the networks use (pre-scaled queries)
ref_path = sys.argv[1] if len(sys.argv) > 1 else None
out = {}
for N, heads, D in ((4096, 8, 40), (4096, 5, 64), (1024, 8, 80)):
    C = heads * D
    g = torch.Generator(device=dev).manual_seed(N + D)
    qkv = (torch.randn(B, N, 3 * C, device=dev, generator=g)).to(torch.bfloat16).requires_grad_()
    do = torch.randn(B, N, C, device=dev, generator=g).to(torch.bfloat16)
    for _ in range(6):
        y = ops.self_attention(qkv, heads, PS)
        qkv.grad = None
        y.backward(do)
    torch.cuda.synchronize()
    out[f'y_{N}_{D}'] = y.detach()[::5, ::97].float().cpu()          # (a sample: the file stays small)
    out[f'g_{N}_{D}'] = qkv.grad.detach()[::5, ::97].float().cpu()
if ref_path:
    if os.path.isfile(ref_path):
        ref = torch.load(ref_path)
        for k, v in out.items():
            e = (v - ref[k]).abs().max().item() / ref[k].abs().max().item()
            print(f'{k}: max diff vs reference setting {e:.2e}')
            assert e < 2e-2, k
    else:
        torch.save(out, ref_path)
print('ok')
