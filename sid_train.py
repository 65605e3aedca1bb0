#!/usr/bin/env python
"""Command line of the distillation run, option-compatible with the reference's `sid_train.py` (boundary B5).

    torchrun --standalone --nproc_per_node=8 sid_train.py --outdir runs --data_prompt_text /data/aesthetics_6_plus \\
        --sd_model /models/stable-diffusion-v1-5 --cfg_train_fake 1.5 --cfg_eval_fake 1.5 --cfg_eval_real 1.5 \\
        --batch 64 --batch-gpu 8 --duration 10 --ema 0.05 --tick 2 --snap 50 --dump 100

Same option names, defaults and derived quantities as the reference (`sid_train.py:88-157` options, `:196-330` config):
optimizer kwargs (Adam, betas (0, 0.999), eps 1e-8 / 1e-6 under --fp16; AdamW + wd 0.01), total_kimg = duration*1000,
ema_halflife_kimg = ema*1000, run-directory numbering and description string, `--resume training-state-<kimg>.pt`.
Differences (documented, not silent):
  * `--sd_model` must be a local diffusers-layout directory or `random:<arch>` (no network in this environment);
  * `--fp16` only selects the optimizer eps; masters are fp32 and compute is bf16 MFMA, or -- with the added option
    `--precision fp32` -- the fp32-accurate mode (the reference's own default precision, ~20x slower);
  * `--metrics` run through sid_lsg_amd/metrics.py at the snapshot ticks and need LOCAL detector / statistics files
    (--metric_pt_path, --data_stat); `--train_mode 0 --network_pkl <snapshot>` evaluates a snapshot (1 / 2 / 4 steps);
  * `--data` is optional (it is the COCO image set used only by the metrics).
"""
import json
import os
import re

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL on this platform: before the HIP runtime starts

import click  # noqa: E402
import torch  # noqa: E402

from sid_lsg_amd import distributed as dist  # noqa: E402
from sid_lsg_amd.dnnlib_util import EasyDict, construct_class_by_name  # noqa: E402
from sid_lsg_amd.training_loop import training_loop  # noqa: E402


def _csv(_ctx, _param, value):
    if value is None or value == '' or value.lower() == 'none':
        return None
    return value.split(',')


# (flag(s), kwargs) -- kept as a table so that the option surface can be diffed against the reference at a glance
OPTIONS = [
    (('--outdir',), dict(type=str, required=True, metavar='DIR', help='Where to save the results')),
    (('--data',), dict(type=str, default=None, metavar='ZIP|DIR', help='Image dataset (only used by --metrics)')),
    (('--data_stat',), dict(type=str, default=None, metavar='ZIP|DIR', help='Dataset statistics (only used by --metrics)')),
    (('--data_prompt_text',), dict(type=str, required=True, metavar='DIR|TXT', help='Training prompts')),
    (('--duration',), dict(type=click.FloatRange(min=0, min_open=True), default=200, show_default=True, metavar='MIMG', help='Training duration')),
    (('--batch',), dict(type=click.IntRange(min=1), default=512, show_default=True, metavar='INT', help='Total batch size')),
    (('--batch-gpu',), dict(type=click.IntRange(min=1), default=None, metavar='INT', help='Limit batch size per GPU')),
    (('--ema',), dict(type=click.FloatRange(min=0), default=0.5, show_default=True, metavar='MIMG', help='EMA half-life')),
    (('--xflip',), dict(type=float, default=0.0, show_default=True, help='Dataset x-flips (unused by the prompt stream)')),
    (('--bench',), dict(type=bool, default=True, show_default=True, help='Accepted for compatibility (no cuDNN here)')),
    (('--cache',), dict(type=bool, default=True, show_default=True, help='Accepted for compatibility')),
    (('--workers',), dict(type=click.IntRange(min=1), default=1, show_default=True, help='Accepted for compatibility')),
    (('--desc',), dict(type=str, default=None, metavar='STR', help='String to include in result dir name')),
    (('--nosubdir',), dict(is_flag=True, help='Do not create a subdirectory for results')),
    (('--tick',), dict(type=click.IntRange(min=1), default=2, show_default=True, metavar='KIMG', help='How often to print progress')),
    (('--snap',), dict(type=click.IntRange(min=1), default=50, show_default=True, metavar='TICKS', help='How often to save snapshots')),
    (('--dump',), dict(type=click.IntRange(min=1), default=100, show_default=True, metavar='TICKS', help='How often to dump state')),
    (('--seed',), dict(type=int, default=None, metavar='INT', help='Random seed  [default: random, shared by all ranks]')),
    (('--transfer',), dict(type=str, default=None, metavar='PKL', help='Initialise G / G_ema from a network snapshot')),
    (('--resume',), dict(type=str, default=None, metavar='PT', help='Resume from training-state-*.pt')),
    (('-n', '--dry-run'), dict(is_flag=True, help='Print training options and exit')),
    (('--metrics',), dict(callback=_csv, default=None, help='Comma-separated list or "none"')),
    (('--sd_model',), dict(type=str, default='runwayml/stable-diffusion-v1-5', show_default=True, help='Local diffusers directory or random:<arch>')),
    (('--resolution',), dict(type=int, default=512, show_default=True, metavar='INT', help='Image resolution')),
    (('--init_timestep',), dict(type=int, default=625, show_default=True, metavar='INT', help='t_init, in [0,999]')),
    (('--fp16',), dict(type=bool, default=False, show_default=True, metavar='BOOL', help='Reference fp16 recipe (optimizer eps 1e-6)')),
    (('--precision',), dict(type=click.Choice(['bf16', 'fp32']), default='bf16', show_default=True, help='Compute dtype of the HIP path (not a reference option)')),
    (('--teacher-weights', 'teacher_weights'), dict(type=click.Choice(['bf16', 'fp8', 'fp8-frozen']), default='bf16', show_default=True, help='fp8: frozen teacher forward weights as e4m3 + per-channel scales; fp8-frozen: every pass without weight gradients (not a reference option)')),
    (('--ls',), dict(type=click.FloatRange(min=0, min_open=True), default=1, show_default=True, help='Loss scaling')),
    (('--lsg',), dict(type=click.FloatRange(min=0, min_open=True), default=1, show_default=True, help='Loss scaling G')),
    (('--alpha',), dict(type=click.FloatRange(min=-1000, min_open=True), default=1, show_default=True, help='L2-alpha*L1')),
    (('--tmax',), dict(type=click.IntRange(min=0), default=980, show_default=True, help='Largest teacher time step')),
    (('--tmin',), dict(type=click.IntRange(min=0), default=20, show_default=True, help='Smallest teacher time step')),
    (('--lr',), dict(type=click.FloatRange(min=0, min_open=True), default=1e-6, show_default=True, help='Fake-score learning rate')),
    (('--glr',), dict(type=click.FloatRange(min=0, min_open=True), default=1e-6, show_default=True, help='Generator learning rate')),
    (('--train_mode',), dict(type=bool, default=True, show_default=True, help='Distill (True) or evaluate the metrics of --network_pkl with 1 / 2 / 4 generation steps (False)')),
    (('--network_pkl',), dict(type=str, default=None, help='network-snapshot-*.pkl to evaluate with --train_mode 0')),
    (('--cfg_train_fake',), dict(type=float, default=1, show_default=True, help='kappa1: guidance scale when training the fake score')),
    (('--cfg_eval_fake',), dict(type=float, default=1, show_default=True, help='kappa2 = kappa3: guidance scale when evaluating the fake score')),
    (('--cfg_eval_real',), dict(type=float, default=1, show_default=True, help='kappa4: guidance scale when evaluating the teacher')),
    (('--metric_pt_path',), dict(type=str, default=None, help='Accepted for compatibility')),
    (('--metric_clip_path',), dict(type=str, default=None, help='Accepted for compatibility')),
    (('--metric_open_clip_path',), dict(type=str, default=None, help='Accepted for compatibility')),
    (('--enable_xformers',), dict(type=bool, default=True, show_default=True, help='Accepted for compatibility (attention is the fused HIP kernel)')),
    (('--gradient_checkpointing',), dict(type=bool, default=False, show_default=True, help='Accepted for compatibility')),
    (('--optimizer',), dict(type=click.Choice(['adam', 'adamw']), default='adam', show_default=True, help='Optimizer')),
    (('--num_steps',), dict(type=int, default=1, show_default=True, help='Number of generation steps')),
    (('--fake_score_use_lora',), dict(type=bool, default=False, show_default=True, help='Unsupported (must be False)')),
]


def _with_options(fn):
    for flags, kw in reversed(OPTIONS):
        fn = click.option(*flags, **kw)(fn)
    return fn


def build_config(o):
    """Options -> training_loop kwargs (the reference's `c`, sid_train.py:196-330)."""
    c = EasyDict()
    if o.metrics is not None:
        # FID / CLIP scores at the snapshot ticks (sid_lsg_amd/metrics.py): the feature detector (--metric_pt_path: the
        # TorchScript Inception the reference downloads) and the real-set statistics (--data_stat: a .npz with mu / sigma, or
        # the reference's cached FeatureStats pickle) must be LOCAL files -- there is no network to fetch them from
        from sid_lsg_amd import metrics as _metrics
        bad = [m for m in o.metrics if not _metrics.is_valid_metric(m)]
        if bad:
            raise click.ClickException(f'--metrics: unknown {bad}; valid: {_metrics.list_valid_metrics()}')
        for flag, path in (('--metric_pt_path', o.metric_pt_path), ('--data_stat', o.data_stat)):
            if not path or not os.path.isfile(path):
                raise click.ClickException(f'--metrics needs {flag} to be a local file (got {path!r})')
    if o.fake_score_use_lora:
        raise click.ClickException('--fake_score_use_lora is not supported')
    if not o.train_mode:
        if o.metrics is None:
            raise click.ClickException('--train_mode 0 evaluates a network snapshot: --metrics is required')
        if not o.network_pkl or not os.path.isfile(o.network_pkl):
            raise click.ClickException(f'--train_mode 0 needs --network_pkl to be a local network-snapshot-*.pkl (got {o.network_pkl!r})')
    c.metrics, c.resolution = o.metrics, o.resolution
    c.metric_real_stats = o.data_stat
    if o.metrics is not None and o.data:
        # the evaluation caption set (reference: training.mscoco_dataset.ImageDataset on --data, sid_train.py:232-235)
        c.dataset_kwargs = EasyDict(class_name='sid_lsg_amd.data.CaptionDataset', path=o.data, resolution=o.resolution, random_flip=o.xflip)
        # fail at start-up, not at the first metrics tick hours into the run: the set is only constructed there
        try:
            from sid_lsg_amd.data import CaptionDataset
            CaptionDataset(o.data, resolution=o.resolution)
        except OSError as e:
            raise click.ClickException(f'--metrics with --data {o.data!r}: {e} (an image + .txt caption directory, or a text file with one caption per line)')
    c.data_loader_kwargs = EasyDict(pin_memory=True, num_workers=o.workers, prefetch_factor=2)
    c.dataset_prompt_text_kwargs = EasyDict(class_name='sid_lsg_amd.data.PromptDataset', path=o.data_prompt_text,
                                            resolution=o.resolution, random_flip=o.xflip, prompt_only=True)
    eps = 1e-6 if o.fp16 else 1e-8
    cls = 'torch.optim.Adam' if o.optimizer == 'adam' else 'torch.optim.AdamW'      # mapped to the fused HIP optimizer by the loop
    extra = {} if o.optimizer == 'adam' else dict(weight_decay=0.01)
    c.fake_score_optimizer_kwargs = EasyDict(class_name=cls, lr=o.lr, betas=[0.0, 0.999], eps=eps, **extra)
    c.g_optimizer_kwargs = EasyDict(class_name=cls, lr=o.glr, betas=[0.0, 0.999], eps=eps, **extra)
    c.network_kwargs = EasyDict(use_fp16=o.fp16, compute_dtype=o.get('precision', 'bf16'), teacher_weights=o.get('teacher_weights', 'bf16'))
    c.loss_kwargs = EasyDict()
    c.init_timestep = o.init_timestep
    c.total_kimg = max(int(o.duration * 1000), 1)
    c.ema_halflife_kimg = int(o.ema * 1000)
    c.update(batch_size=o.batch, batch_gpu=o.batch_gpu, loss_scaling=o.ls, loss_scaling_G=o.lsg, cudnn_benchmark=o.bench,
             kimg_per_tick=o.tick, snapshot_ticks=o.snap, state_dump_ticks=o.dump, alpha=o.alpha, tmax=o.tmax, tmin=o.tmin)
    c.update(cfg_train_fake=o.cfg_train_fake, cfg_eval_fake=o.cfg_eval_fake, cfg_eval_real=o.cfg_eval_real, num_steps=o.num_steps,
             train_mode=bool(o.train_mode), network_pkl=o.network_pkl, fake_score_use_lora=False, enable_xformers=o.enable_xformers,
             gradient_checkpointing=o.gradient_checkpointing, pretrained_model_name_or_path=o.sd_model,
             pretrained_vae_model_name_or_path=o.sd_model, metric_pt_path=o.metric_pt_path,
             metric_open_clip_path=o.metric_open_clip_path, metric_clip_path=o.metric_clip_path)
    if o.transfer is not None:
        c.resume_pkl = o.transfer
    if o.resume is not None:
        m = re.fullmatch(r'training-state-(\d+).pt', os.path.basename(o.resume))
        if not m or not os.path.isfile(o.resume):
            raise click.ClickException('--resume must point to training-state-*.pt from a previous training run')
        c.resume_training, c.resume_kimg = o.resume, int(m.group(1))
    return c


def pick_seed(seed):
    if seed is not None:
        return seed
    s = torch.randint(1 << 31, size=[], device=torch.device('cuda'))
    if dist.get_world_size() > 1:
        torch.distributed.broadcast(s, src=0)
    return int(s)


def pick_run_dir(outdir, desc, nosubdir):
    if dist.get_rank() != 0:
        return None
    if nosubdir:
        return outdir
    ids = []
    if os.path.isdir(outdir):
        for x in os.listdir(outdir):
            m = re.match(r'^\d+', x)
            if m and os.path.isdir(os.path.join(outdir, x)):
                ids.append(int(m.group()))
    run_dir = os.path.join(outdir, f'{max(ids, default=-1) + 1:05d}-{desc}')
    assert not os.path.exists(run_dir)
    return run_dir


@click.command()
@_with_options
def main(**kwargs):
    o = EasyDict(kwargs)
    dist.init()
    c = build_config(o)
    try:
        ds = construct_class_by_name(**c.dataset_prompt_text_kwargs)
    except (IOError, OSError) as err:
        raise click.ClickException(f'--data_prompt_text: {err}')
    c.seed = pick_seed(o.seed)
    dtype_str = 'fp16' if o.fp16 else 'fp32'
    desc = (f'{ds.name}-text_cond-glr{o.glr}-lr{o.lr}-initsigma{o.init_timestep}-gpus{dist.get_world_size():d}-alpha{c.alpha}'
            f'-batch{c.batch_size:d}-tmax{c.tmax:d}-{dtype_str}')
    if o.desc is not None:
        desc += f'-{o.desc}'
    c.run_dir = pick_run_dir(o.outdir, desc, o.nosubdir)

    dist.print0()
    dist.print0('Training options:')
    dist.print0(json.dumps(c, indent=2))
    dist.print0()
    dist.print0(f'Output directory:        {c.run_dir}')
    dist.print0(f'Prompts:                 {o.data_prompt_text} ({len(ds)} lines)')
    dist.print0(f'Number of GPUs:          {dist.get_world_size()}')
    dist.print0(f'Batch size:              {c.batch_size}')
    dist.print0(f'Compute:                 bf16 MFMA, fp32 masters (reference recipe: {dtype_str})')
    dist.print0()
    if o.dry_run:
        dist.print0('Dry run; exiting.')
        return
    if dist.get_rank() == 0:
        os.makedirs(c.run_dir, exist_ok=True)
        with open(os.path.join(c.run_dir, 'training_options.json'), 'wt') as f:
            json.dump(c, f, indent=2)
    training_loop(**c)


if __name__ == '__main__':
    main()
