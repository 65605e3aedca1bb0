"""One rank of the multi-process exchange tests (launched by tests/test_gpu_dist.py through torch.distributed.run).

    mp_ddp_worker.py ddp  <out>        INTEGRATION.md mode 2: fake_score wrapped in torch's DistributedDataParallel and handed
                                       to sid_sd_denoise exactly as the reference does (sid_training_loop.py:316-323,
                                       416-421; torch_utils/misc.py:168-175 no_sync rounds).  2 ranks share cuda:0, gloo.
    mp_ddp_worker.py nccl <out>        FlatGradReducer (whole-buffer start(), segment-wise start_range() from the backward
                                       markers, bf16 exchange) on backend 'nccl' (= RCCL) at whatever WORLD_SIZE the launcher
                                       gives; world 1 forces the collectives to run (min_world=1).
    mp_ddp_worker.py messages <out>    the 4 x 0.86 GB all-reduce messages of one full-size network on RCCL (world 1, forced)
                                       with FlatGradReducer's timing on.
Each rank writes <out>.rank<r>.npz; the launcher asserts on the contents."""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BF16, F32 = torch.bfloat16, torch.float32


def _inputs(cfg, dev, rank, b=2, lat=8, rounds=2):
    g = torch.Generator().manual_seed(100 + rank)
    out = []
    for _ in range(rounds):
        out.append(dict(images=torch.randn(b, 4, lat, lat, generator=g).to(dev), noise=torch.randn(b, 4, lat, lat, generator=g).to(dev),
                        t=torch.randint(20, 980, (b,), generator=g).to(dev),
                        cond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=g).to(dev),
                        uncond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=g).to(dev)))
    return out


def run_ddp(out):
    from sid_lsg_amd import ops
    from sid_lsg_amd.scheduler import DDPMScheduler
    from sid_lsg_amd.sd_util import hip_denoise, hip_prepare_denoise
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    torch.distributed.init_process_group('gloo')
    rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    cfg = CONFIGS['tiny']
    sched = DDPMScheduler().to(dev)
    psi = HipUNet2DCondition(cfg).materialize(dev, seed=5 + rank).train().requires_grad_(True)     # ranks start DIFFERENT
    w_before = psi.flat_params.clone()
    ddp = torch.nn.parallel.DistributedDataParallel(psi, device_ids=[dev], broadcast_buffers=False, find_unused_parameters=False)
    # DDP's constructor broadcast rank 0's weights through the parameter views of the flat buffer
    w0 = psi.flat_params.clone()
    gathered = [torch.empty_like(w0) for _ in range(world)]
    torch.distributed.all_gather(gathered, w0)
    same_weights = all(torch.equal(gathered[0], g) for g in gathered)
    rounds = _inputs(cfg, dev, rank)

    def backward_rounds(net):
        """the reference's accumulation loop: every round but the last under no_sync (misc.ddp_sync)"""
        losses = []
        for i, r in enumerate(rounds):
            sync = i == len(rounds) - 1
            ctx = contextlib.nullcontext() if (sync or not isinstance(net, torch.nn.parallel.DistributedDataParallel)) else net.no_sync()
            with ctx:
                prep = hip_prepare_denoise(r['images'], r['noise'], r['t'], r['cond'].to(BF16), r['uncond'].to(BF16), sched, True)
                eps = hip_denoise(net, prep, 1.5, predict_x0=False)
                loss = ops.sid_fake_score_loss(eps, r['noise'], 0.25)
                loss.backward()
                losses.append(float(loss))
        return losses

    # local gradients (no wrapper -> no exchange), then the same rounds through the DDP wrapper
    psi.flat_grads.zero_()
    backward_rounds(psi)
    torch.cuda.synchronize()
    local = psi.flat_grads.clone()
    psi.flat_grads.zero_()
    losses = backward_rounds(ddp)
    torch.cuda.synchronize()
    got = psi.flat_grads.clone()
    allg = [torch.empty_like(local) for _ in range(world)]
    torch.distributed.all_gather(allg, local)
    mean = sum(allg) / world
    scale = float(mean.abs().max())
    np.savez(f'{out}.rank{rank}.npz', same_weights=same_weights, moved=float((w0 - w_before).abs().max()),
             err=float((got - mean).abs().max()) / scale, local_vs_mean=float((local - mean).abs().max()) / scale,
             scale=scale, losses=np.array(losses))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def run_nccl(out):
    from sid_lsg_amd.distributed import FlatGradReducer
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.distributed.init_process_group('nccl', device_id=dev)
    rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    cfg = CONFIGS['tiny40']
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.zeros(4, 16, 16, 8)
    x[..., :4] = torch.randn(4, 16, 16, 4, generator=g)
    x = x.to(dev).to(BF16)
    ctx = torch.randn(4, cfg.text_len, cfg.cross_attention_dim, generator=g).to(dev).to(BF16)
    t = torch.tensor([999, 500, 250, 20], device=dev)
    dy = torch.randn(4, 256, 8, generator=g).to(dev)
    res = {}
    for name, kw in (('fp32', {}), ('bf16', dict(exchange_dtype=BF16))):
        net = HipUNet2DCondition(cfg).materialize(dev, seed=9).requires_grad_(True)
        # reference result: plain backward, then one blocking all_reduce on the default stream
        net.forward_nhwc(x, t, ctx).backward(dy)
        torch.cuda.synchronize()
        local = net.flat_grads.clone()
        want = local.clone()
        torch.distributed.all_reduce(want)
        torch.cuda.synchronize()
        # (a) whole buffer, 4 messages on the communication stream
        red = FlatGradReducer(min_world=1, **kw)
        net.flat_grads.copy_(local)
        red.start(net.flat_grads)
        red.wait()
        torch.cuda.synchronize()
        scale = float(want.abs().max())
        res[f'{name}_whole'] = float((net.flat_grads - want).abs().max()) / scale
        # (b) segment-wise, started by the backward's markers while earlier layers are still running
        net.flat_grads.zero_()
        segs = net.grad_segments()
        fired = []

        def cb(k):
            fired.append(k)
            red.start_segment(net.flat_grads, segs[k])
        net.set_grad_ready_callback(cb)
        net.forward_nhwc(x, t, ctx).backward(dy)
        net.set_grad_ready_callback(None)
        red.start_segment(net.flat_grads, segs[-1])
        red.wait()
        torch.cuda.synchronize()
        res[f'{name}_segments'] = float((net.flat_grads - want).abs().max()) / scale
        res[f'{name}_fired'] = np.array(fired)
        res[f'{name}_scale'] = scale
    np.savez(f'{out}.rank{rank}.npz', world=world, backend=torch.distributed.get_backend(), **res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def run_graph(out):
    """The gradient exchange INSIDE the captured iteration: FlatGradReducer(min_world=1) on backend 'nccl', so RCCL's
    all-reduce kernels are captured on the communication stream (psi exchange overlapped with the generator forward,
    segment-wise G exchange from the backward markers) and replayed."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from sid_lsg_amd.distributed import FlatGradReducer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.distributed.init_process_group('nccl', device_id=dev)
    from sid_lsg_amd._lib import lib
    lib.load()
    from test_gpu_unet import _graph_vs_eager
    # warm RCCL up outside any capture (communicator creation is lazy)
    w = torch.ones(8, device=dev)
    torch.distributed.all_reduce(w)
    torch.cuda.synchronize()
    results, lr, iters = _graph_vs_eager(dev, reducer_factory=lambda: FlatGradReducer(min_world=1))
    a, g = results['eager'], results['graph']
    np.savez(f'{out}.rank0.npz', eager=a['losses'], graph=g['losses'], dG=float((a['G'] - g['G']).abs().max()),
             dpsi=float((a['psi'] - g['psi']).abs().max()), lr=lr, iters=iters, ngraphs=g['ngraphs'])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def run_messages(out):
    """The gradient exchange of ONE full-size network (859.5 M fp32 values -> 4 all-reduce messages of ~0.86 GB) on RCCL,
    collectives forced at world 1, with FlatGradReducer's timing on: what `bench.py --gpus N` reports as `comm`."""
    from sid_lsg_amd.distributed import FlatGradReducer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.distributed.init_process_group('nccl', device_id=dev)
    world = torch.distributed.get_world_size()
    n = 859_520_964 + 60          # SD1.5 parameter total (the flat buffer pads every tensor to 64 elements: a little more)
    g = torch.randn(n, device=dev)
    want = g.clone() * world
    warm = torch.ones(1 << 20, device=dev)
    torch.distributed.all_reduce(warm)
    red = FlatGradReducer(min_world=1)
    red.enable_timing()
    for _ in range(2):
        g.copy_(want / world)
        red.start(g)
        # a compute-stream kernel the exchange overlaps with, then the wait whose duration is the exposed time
        (warm * 2).sum()
        red.wait(tag='G')
        torch.cuda.synchronize()
    rep = red.timing_report(world)
    np.savez(f'{out}.rank0.npz', messages=rep['messages'] // 2, bytes=rep['bytes'] // 2, comm_ms=rep['comm_ms'] / 2,
             exposed_ms=rep['exposed_ms']['G'] / 2, algbw_GBps=rep['algbw_GBps'], unchanged=bool(torch.equal(g, want)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    {'ddp': run_ddp, 'nccl': run_nccl, 'graph': run_graph, 'messages': run_messages}[sys.argv[1]](sys.argv[2])
