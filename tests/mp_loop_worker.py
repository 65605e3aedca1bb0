"""One rank of the 2-rank data-parallel parity run (launched by tests/test_gpu_unet.py through torch.distributed.run).

Both ranks share cuda:0 (the test box has one GPU) and exchange gradients over gloo; everything else is the production
multi-GPU path: `training_loop` with world_size 2 -> rank-strided prompt stream, per-rank seeds, rank-0 weight broadcast,
FlatGradReducer (psi exchange overlapped with the generator / teacher forwards, G exchange by gradient segments during its
backward), mean folded into the optimizer kernel.  Writes this rank's loss curve + final weights to <out>.rank<r>.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main(golden, out, precision):
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    torch.distributed.init_process_group('gloo')
    rank = torch.distributed.get_rank()
    from oracle import fixtures
    from sid_lsg_amd import training_loop as tl
    from sid_lsg_amd.scheduler import DDPMScheduler
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    from test_gpu_unet import _loop_kwargs_from_golden
    g = np.load(golden)
    cfg = str(g['cfg'])
    tmp = os.path.dirname(out)
    pdir = os.path.join(tmp, f'prompts{rank}')
    os.makedirs(pdir, exist_ok=True)
    with open(os.path.join(pdir, 'aesthetics_6_plus.txt'), 'wt') as f:
        f.write('\n'.join(str(p) for p in g['prompts']) + '\n')
    run = os.path.join(tmp, f'run{rank}')
    os.makedirs(run, exist_ok=True)

    def factory(**kw):
        ref, vae, _, te, tok = fixtures.factory(cfg)
        unet = HipUNet2DCondition(CONFIGS[cfg], compute_dtype=kw['compute_dtype']).materialize(dev, source=ref.state_dict())
        return unet, vae, DDPMScheduler().to(dev), te.to(dev), tok
    kw = _loop_kwargs_from_golden(g, run, pdir, dev)
    kw['network_kwargs']['compute_dtype'] = precision
    losses = []
    tl.load_sd15 = factory
    res = tl.training_loop(on_iteration=lambda it, lf, lg: losses.extend([lf, lg]), **kw)
    P = lambda net, n: dict(net.named_parameters())[n].detach().cpu().numpy()   # noqa: E731
    np.savez(f'{out}.rank{rank}.npz', losses=np.array(losses), G_conv_in_w=P(res['G'], 'conv_in.weight'),
             fake_conv_in_w=P(res['fake_score'], 'conv_in.weight'), G_last_b=P(res['G'], 'conv_out.bias'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(*sys.argv[1:4])
