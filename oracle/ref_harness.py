"""Import the REAL reference (/root/reference) as an oracle.  TEST INFRASTRUCTURE ONLY.

Only usable where /root/reference exists (the authoring container).  Nothing here is
copied from the reference: it is imported in place, under a small compatibility
harness:
  * stub modules for packages that are absent offline (diffusers, blobfile) and for the
    reference's own `metrics` package (needs torchvision/open_clip);
  * `torch.utils.data.Sampler.__init__` accepting the legacy positional argument
    (torch_utils/misc.py:116 vs torch 2.10);
  * a DistributedDataParallel subclass that drops `device_ids` so DDP runs on CPU/gloo
    (it stays a subclass so misc.ddp_sync's isinstance/no_sync logic is exercised);
  * no-op torch.cuda memory statistics (sid_training_loop.py:583-587).
The UNet / scheduler / tokenizer / text-encoder handed to the reference are duck-typed
objects supplied by the caller (our oracle restatements), exactly as the reference's
`load_sd15` would hand over diffusers objects (training/sid_sd_util.py:118).
"""
import contextlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get('SIDLSG_REFERENCE_ROOT', '/root/reference')


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'training', 'sid_training_loop.py'))


def _stub(name, **attrs):
    if name in sys.modules and not getattr(sys.modules[name], '_sidlsg_stub', False):
        return sys.modules[name]
    m = types.ModuleType(name)
    m._sidlsg_stub = True
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Missing:
    """Placeholder for a class the reference imports by name but never touches on our path."""

    def __init__(self, *a, **k):
        raise RuntimeError('stubbed third-party class instantiated')


def install_stubs():
    names = ['AutoencoderKL', 'DDPMScheduler', 'DiffusionPipeline', 'UNet2DConditionModel']
    d = _stub('diffusers', **{n: _Missing for n in names}, __version__='0.27.2')
    d.loaders = _stub('diffusers.loaders', StableDiffusionXLLoraLoaderMixin=_Missing)
    d.optimization = _stub('diffusers.optimization', get_scheduler=lambda *a, **k: None)
    d.utils = _stub('diffusers.utils', check_min_version=lambda *a, **k: None,
                    convert_state_dict_to_diffusers=lambda x, *a, **k: x)
    d.utils.import_utils = _stub('diffusers.utils.import_utils', is_xformers_available=lambda: False)
    d.models = _stub('diffusers.models')
    d.models.attention_processor = _stub(
        'diffusers.models.attention_processor',
        **{n: _Missing for n in ['AttnProcessor2_0', 'XFormersAttnProcessor', 'LoRAXFormersAttnProcessor',
                                 'LoRAAttnProcessor2_0', 'FusedAttnProcessor2_0']})

    def compute_snr(sched, t):
        ac = sched.alphas_cumprod.to(t.device)[t]
        return ac / (1 - ac)
    d.training_utils = _stub('diffusers.training_utils', compute_snr=compute_snr)
    _stub('blobfile')
    mt = _stub('metrics')
    mt.sid_metric_main = _stub('metrics.sid_metric_main')


_imported = {}


def import_reference():
    """Returns namespace(sd_util, loop, misc, training_stats, dnnlib, bias_act) of reference modules."""
    if _imported:
        return types.SimpleNamespace(**_imported)
    assert reference_available(), f'{REFERENCE_ROOT} not present'
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # torch-2.10 drift: Sampler.__init__ no longer takes data_source (torch_utils/misc.py:116)
    torch.utils.data.Sampler.__init__ = lambda self, *a, **k: None
    import dnnlib  # noqa
    from torch_utils import misc, training_stats, distributed as rdist  # noqa
    from torch_utils.ops import bias_act  # noqa
    from training import sid_sd_util  # noqa
    from training import sid_training_loop  # noqa
    _imported.update(sd_util=sid_sd_util, loop=sid_training_loop, misc=misc, training_stats=training_stats,
                     dnnlib=dnnlib, bias_act=bias_act, rdist=rdist)
    return types.SimpleNamespace(**_imported)


class _CpuDDP(torch.nn.parallel.DistributedDataParallel):
    def __init__(self, module, device_ids=None, **kw):
        super().__init__(module, **kw)


@contextlib.contextmanager
def cpu_process_group(rank=0, world_size=1, port=29533):
    created = False
    if not torch.distributed.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        torch.distributed.init_process_group('gloo', rank=rank, world_size=world_size)
        created = True
    try:
        yield
    finally:
        if created:
            torch.distributed.destroy_process_group()


def run_reference_training_loop(factory, prompt_dir, run_dir, *, iterations, batch_size, batch_gpu, seed=0,
                                alpha=1.0, kappa=(1.5, 1.5, 1.5), lr=1e-6, glr=1e-6, eps=1e-8, resolution=64,
                                init_timestep=625, tmin=20, tmax=980, ema_halflife_kimg=50, extra=None):
    """Runs the UNMODIFIED reference training_loop on CPU for `iterations` iterations.

    `factory()` -> (unet, vae, scheduler, text_encoder, tokenizer) replaces load_sd15
    (imported by name at sid_training_loop.py:36).  Returns dict(losses=[(name, value), ...],
    G=<module>, fake_score=<module>, G_ema=<module>) captured from the loop's own objects.
    """
    ref = import_reference()
    tl = ref.loop
    records = []
    captured = {}
    orig_report = ref.training_stats.report

    def report(name, value):
        if name in ('fake_score_Loss/loss', 'G_Loss/loss'):
            records.append((name, float(value)))
        return orig_report(name, value)

    # capture the nets the loop builds: the optimizers are constructed by class name with params=...
    orig_construct = ref.dnnlib.util.construct_class_by_name

    def construct(*a, **k):
        obj = orig_construct(*a, **k)
        if 'params' in k:
            captured.setdefault('optimizers', []).append(obj)
        return obj

    def report0(name, value):
        # the tick status line re-reports the last losses through report0 (sid_training_loop.py:585-586);
        # route it past the recorder so `records` holds exactly one entry per optimizer step
        orig_report(name, value if torch.distributed.get_rank() == 0 else [])
        return value

    saved_report0 = tl.training_stats.report0
    saved = (tl.load_sd15, tl.training_stats.report, torch.nn.parallel.DistributedDataParallel,
             torch.cuda.max_memory_allocated, torch.cuda.max_memory_reserved, torch.cuda.reset_peak_memory_stats,
             tl.dnnlib.util.construct_class_by_name)
    ws = torch.distributed.get_world_size()
    total_kimg_images = iterations * batch_size
    try:
        tl.load_sd15 = lambda **kw: factory()
        tl.training_stats.report = report
        tl.training_stats.report0 = report0
        torch.nn.parallel.DistributedDataParallel = _CpuDDP
        torch.cuda.max_memory_allocated = lambda *a, **k: 0
        torch.cuda.max_memory_reserved = lambda *a, **k: 0
        torch.cuda.reset_peak_memory_stats = lambda *a, **k: None
        tl.dnnlib.util.construct_class_by_name = construct
        if ws > 1:  # the reference's dist.init() hard-codes nccl+cuda (torch_utils/distributed.py:26-30)
            ref.training_stats.init_multiprocessing(rank=torch.distributed.get_rank(), sync_device=torch.device('cpu'))
        E = ref.dnnlib.EasyDict
        kw = dict(
            run_dir=run_dir, network_kwargs=E(use_fp16=False),
            dataset_prompt_text_kwargs=E(class_name='training.aesthetics_dataset.ImageDataset', path=prompt_dir,
                                         resolution=resolution, random_flip=0.0, prompt_only=True),
            data_loader_kwargs=dict(num_workers=0),
            fake_score_optimizer_kwargs=E(class_name='torch.optim.Adam', lr=lr, betas=[0.0, 0.999], eps=eps),
            g_optimizer_kwargs=E(class_name='torch.optim.Adam', lr=glr, betas=[0.0, 0.999], eps=eps),
            seed=seed, batch_size=batch_size, batch_gpu=batch_gpu,
            total_kimg=total_kimg_images / 1000.0, ema_halflife_kimg=ema_halflife_kimg,
            kimg_per_tick=10 ** 9, snapshot_ticks=None, state_dump_ticks=None, alpha=alpha, tmax=tmax, tmin=tmin,
            device=torch.device('cpu'), metrics=None, init_timestep=init_timestep,
            cfg_train_fake=kappa[0], cfg_eval_fake=kappa[1], cfg_eval_real=kappa[2], resolution=resolution,
            enable_xformers=False)
        if extra:
            kw.update(extra)
        tl.training_loop(**kw)
    finally:
        tl.training_stats.report0 = saved_report0
        (tl.load_sd15, tl.training_stats.report, torch.nn.parallel.DistributedDataParallel,
         torch.cuda.max_memory_allocated, torch.cuda.max_memory_reserved, torch.cuda.reset_peak_memory_stats,
         tl.dnnlib.util.construct_class_by_name) = saved
    opts = captured.get('optimizers', [])
    out = dict(losses=records)
    if len(opts) >= 2:
        out['fake_score_params'] = [p.detach().clone() for g in opts[0].param_groups for p in g['params']]
        out['G_params'] = [p.detach().clone() for g in opts[1].param_groups for p in g['params']]
    return out
