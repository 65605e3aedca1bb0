"""`training_loop(**c)` with the reference's keyword surface (training/sid_training_loop.py:148-194), driving the
HIP SiD step.  The per-iteration body is sid_lsg_amd.sid_step.SiDStep (= reference lines 383-571); this file is
the minimal harness around it that the reference also has: dataset + prompt stream, network/optimizer
construction by class name, accumulation rounds, EMA schedule, tick statistics (`stats_{alpha}.jsonl`,
`Timing/sec_per_kimg`), snapshots (`network-snapshot-*.pkl`) and training-state dumps.

Kept in meaning and in RNG consumption: seeding (:238-239), batch_gpu / accumulation arithmetic (:246-250), the base-seed
draw `iter(DataLoader)` makes from the default CPU generator (:275), the 16 consumed example prompt batches (:277-281),
prompt dropout 10% iff kappa1 or kappa2 != 1 (:208-211, 393-396), the draw order inside a round (dropout flags on the
CPU generator; z, noise, t on `device`), EMA beta schedule (:553-558), `cur_nimg` accounting (:567).  `PromptStream`
below is that stream; tests/test_host_logic.py pins it (prompt order, dropout flags, z / noise / t on the CPU
generator) to the restatement that reproduces the reference's golden loss curves, and tests/test_gpu_unet.py runs this
loop against tests/golden/loop_*.npz (curves produced by the UNMODIFIED reference training_loop).
Not reproduced: preview PNG grids / FID metrics (cold path), torch.cuda.empty_cache +
gc.collect every iteration (:384-385, a pure slowdown).  Rejected loudly rather than ignored: num_steps != 1 (the
reference's own multi-step training sampler is marked unfinished, sid_sd_util.py:165).
"""
import copy
import contextlib
import json
import os
import pickle
import time

import numpy as np
import psutil
import torch

from . import distributed as dist
from .data import InfiniteSampler, prompt_batches
from .dnnlib_util import EasyDict, construct_class_by_name, format_time
from .distributed import FlatGradReducer
from .sd_util import load_sd15, resolve_compute_dtype
from .sid_step import SiDStep
from .text import TextConditioner


class Stats:
    """Per-tick scalar statistics (count/mean), the slice of torch_utils/training_stats.py the loop uses."""

    def __init__(self):
        self.acc = {}

    def report(self, name, value):
        """value: a number, or a device scalar -- then the sum stays on the device and is read once per tick (as_dict): a
        float() per iteration is a host-device synchronisation per iteration."""
        n, s = self.acc.get(name, (0, 0.0))
        self.acc[name] = (n + 1, s + (value.detach().float() if torch.is_tensor(value) else float(value)))
        return value

    def as_dict(self):
        return {k: dict(num=n, mean=float(s) / max(n, 1)) for k, (n, s) in self.acc.items()}

    def reset(self):
        self.acc = {}


class PromptStream:
    """The per-rank input stream of the loop: which prompts, which of them are dropped, and z / noise / t of every
    accumulation round, with the reference's RNG consumption order (sid_training_loop.py:238-239, 274-281, 391-413,
    472-484).  `rng_device` is where z / noise / t are drawn: the training device in production (as the reference
    does, device=device at :398-399, 413), 'cpu' when a run has to reproduce a CPU-generated reference curve."""

    def __init__(self, dataset_obj, *, seed, rank, world, batch_gpu, lat, tmin, tmax, device, rng_device=None):
        np.random.seed((seed * world + rank) % (1 << 31))                                   # :238
        torch.manual_seed(np.random.randint(1 << 31))                                       # :239
        self.device = torch.device(device)
        self.rng_device = torch.device(rng_device) if rng_device is not None else self.device
        self.lat, self.tmin, self.tmax = lat, tmin, tmax
        sampler = InfiniteSampler(dataset_obj, rank=rank, num_replicas=world, seed=seed)    # :274
        # :275 iter(DataLoader(...)) draws the iterator's base seed from the default CPU generator: part of the stream
        # the dropout flags below come from
        torch.empty((), dtype=torch.int64).random_()
        self.prompts_it = prompt_batches(dataset_obj, sampler, batch_gpu)

    def next_prompts(self):
        return next(self.prompts_it)

    def round(self, dropout):
        prompts = self.next_prompts()
        if dropout:
            flags = (torch.rand(len(prompts)) < 0.1).tolist()                               # :394 (CPU generator)
            prompts = ['' if f else p for f, p in zip(flags, prompts)]
        z = torch.randn([len(prompts), 4, self.lat, self.lat], device=self.rng_device, dtype=torch.float32)   # :398 / :479
        noise = torch.randn_like(z)                                                         # :399 / :480
        t = torch.randint(self.tmin, self.tmax, (len(prompts),), device=self.rng_device, dtype=torch.long)    # :413 / :484
        return prompts, z.to(self.device), noise.to(self.device), t.to(self.device)


SNAPSHOT_EXTRA_TICKS = (2, 4, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100)


def training_loop(
    run_dir='.', dataset_kwargs={}, data_loader_kwargs={}, network_kwargs={}, loss_kwargs={},
    fake_score_optimizer_kwargs={}, g_optimizer_kwargs={}, augment_kwargs=None, seed=0, batch_size=512, batch_gpu=None,
    total_kimg=200000, ema_halflife_kimg=500, ema_rampup_ratio=0.05, loss_scaling=1, loss_scaling_G=1, kimg_per_tick=50,
    snapshot_ticks=50, state_dump_ticks=500, resume_pkl=None, resume_training=None, resume_kimg=0, alpha=1, tmax=980, tmin=20,
    cudnn_benchmark=True, device=torch.device('cuda'), metrics=None, init_timestep=None, metric_pt_path=None,
    metric_open_clip_path=None, metric_clip_path=None, pretrained_model_name_or_path='runwayml/stable-diffusion-v1-5',
    pretrained_vae_model_name_or_path='runwayml/stable-diffusion-v1-5', fake_score_use_lora=False,
    dataset_prompt_text_kwargs={}, cfg_train_fake=1, cfg_eval_fake=1, cfg_eval_real=1, num_steps=1, train_mode=True,
    network_pkl=None, enable_xformers=True, gradient_checkpointing=False, resolution=512, on_iteration=None,
    rng_device=None, metric_real_stats=None, metric_num_test=None,
):
    if not train_mode:
        return evaluate_network(run_dir=run_dir, dataset_kwargs=dataset_kwargs, network_kwargs=network_kwargs, device=device, metrics=metrics,
                                init_timestep=init_timestep, metric_pt_path=metric_pt_path, metric_open_clip_path=metric_open_clip_path,
                                pretrained_model_name_or_path=pretrained_model_name_or_path, network_pkl=network_pkl, resolution=resolution,
                                num_steps=num_steps, metric_real_stats=metric_real_stats, metric_num_test=metric_num_test,
                                dataset_prompt_text_kwargs=dataset_prompt_text_kwargs)
    if num_steps != 1:
        raise NotImplementedError(f'num_steps={num_steps}: only the one-step generator is trained (the reference marks its '
                                  'multi-step training sampler as unfinished, sid_sd_util.py:165)')
    rank, world = dist.get_rank(), dist.get_world_size()
    if dict(network_kwargs).get('use_fp16'):
        dist.print0('note: --fp16 is accepted for compatibility; this path keeps fp32 masters and computes in bf16 (no fp16 '
                    'weights / optimizer state, hence no fp16 gradient clipping)')
    if gradient_checkpointing:
        dist.print0('note: gradient_checkpointing is ignored (as in the reference loop, sid_training_loop.py:224-228)')
    dist.print0('Loading dataset...')
    dataset_obj = construct_class_by_name(**dataset_prompt_text_kwargs)
    # compute dtype of the HIP path (masters are always fp32): network_kwargs.compute_dtype = 'bf16' (production, default) or
    # 'fp32' (the reference's own default precision, :205; ~20x slower).  network_kwargs.use_fp16 is accepted and ignored.
    dtype = resolve_compute_dtype(dict(network_kwargs).get('compute_dtype'))
    use_dropout = (cfg_train_fake != 1 or cfg_eval_fake != 1)

    if world > 1 and rank != 0:
        torch.distributed.barrier()
    unet, vae, noise_scheduler, text_encoder, tokenizer = load_sd15(
        pretrained_model_name_or_path=pretrained_model_name_or_path, pretrained_vae_model_name_or_path=None, device=device,
        weight_dtype=dtype, enable_xformers=enable_xformers, lora_config=None, compute_dtype=dtype)
    if world > 1 and rank == 0:
        torch.distributed.barrier()
    dist.print0('Loading network completed')

    start_time = time.time()
    batch_gpu_total = batch_size // world
    if batch_gpu is None or batch_gpu > batch_gpu_total:
        batch_gpu = batch_gpu_total
    rounds = batch_gpu_total // batch_gpu
    assert batch_size == batch_gpu * rounds * world
    lat = resolution // (2 ** (len(vae.config.block_out_channels) - 1))

    stream = PromptStream(dataset_obj, seed=seed, rank=rank, world=world, batch_gpu=batch_gpu, lat=lat, tmin=tmin, tmax=tmax,
                          device=device, rng_device=rng_device)
    dist.print0('Example text prompts used for distillation:')
    for i in range(16):
        dist.print0(i, stream.next_prompts())

    true_score = unet.eval().requires_grad_(False)
    fake_score = copy.deepcopy(true_score).train().requires_grad_(True)
    G = copy.deepcopy(true_score).train().requires_grad_(True)
    # network_kwargs.teacher_weights = 'fp8': the frozen teacher's forward GEMM / conv weights as e4m3 + per-channel scales
    # (BASELINE.json configs[4]; HipUNet2DCondition.enable_fp8_weights).  The networks that train stay bf16.
    tw = dict(network_kwargs).get('teacher_weights', 'bf16')
    if tw not in ('bf16', 'fp8', 'fp8-frozen'):
        raise ValueError(f"network_kwargs.teacher_weights must be 'bf16', 'fp8' or 'fp8-frozen', got {tw!r}")
    if tw in ('fp8', 'fp8-frozen'):
        dist.print0(f'teacher: {true_score.enable_fp8_weights()} layers with fp8 (e4m3) weights')
    if tw == 'fp8-frozen':      # + every other pass without weight gradients (fake score in phase B, the generator's no-grad pass)
        for net in (fake_score, G):
            net.enable_fp8_weights(frozen_passes_only=True)
    dist.print0('Setting up optimizer...')
    opt_f = construct_class_by_name(params=fake_score.parameters(), **_hip_opt(fake_score_optimizer_kwargs))
    opt_g = construct_class_by_name(params=G.parameters(), **_hip_opt(g_optimizer_kwargs))
    G_ema = copy.deepcopy(G).eval().requires_grad_(False) if ema_halflife_kimg > 0 else G
    if resume_pkl is not None:          # --transfer: start G / G_ema from a network snapshot (sid_training_loop.py:296-303)
        dist.print0(f'Loading network weights from "{resume_pkl}"...')
        with open(resume_pkl, 'rb') as f:
            src = pickle.load(f)['ema']
        for net in (G, G_ema):
            net.load_state_dict(_snapshot_state(src))
            net.refresh_compute_weights()
        del src
    if resume_training is not None:
        data = torch.load(resume_training, map_location='cpu', weights_only=False)
        fake_score.load_state_dict(_sd(data['fake_score'])); G.load_state_dict(_sd(data['G']))
        if ema_halflife_kimg > 0:
            G_ema.load_state_dict(_sd(data['G_ema']))
        for opt, net in ((opt_f, fake_score), (opt_g, G)):      # the owner's name -> offset table: states of another flat layout are permuted
            if hasattr(opt, 'set_owner'):
                opt.set_owner(net)
        opt_f.load_state_dict(data['fake_score_optimizer_state']); opt_g.load_state_dict(data['g_optimizer_state'])
        for net in (fake_score, G, G_ema):
            net.refresh_compute_weights()
        del data
    if world > 1:   # what DDP's constructor does (:316-323): everybody starts from rank 0's weights
        for net in (fake_score, G, G_ema):
            torch.distributed.broadcast(net.flat_params, src=0)
            net.refresh_compute_weights()
    fake_score.eval().requires_grad_(False); G.eval().requires_grad_(False)

    cond = TextConditioner(tokenizer, text_encoder, out_dtype=true_score.compute_dtype)
    step = SiDStep(G, fake_score, true_score, G_ema, noise_scheduler, opt_f, opt_g, alpha=alpha, cfg_train_fake=cfg_train_fake,
                   cfg_eval_fake=cfg_eval_fake, cfg_eval_real=cfg_eval_real, loss_scaling=loss_scaling, loss_scaling_G=loss_scaling_G,
                   batch_gpu_total=batch_gpu_total, init_timestep=init_timestep, reducer=FlatGradReducer() if world > 1 else None,
                   world_size=world)

    def make_round(dropout):
        prompts, z, noise, t = stream.round(dropout)
        return dict(z=z, noise=noise, t=t, cond=cond.encode(prompts), uncond=cond.uncond(len(prompts)))

    dist.print0(f'Training for {total_kimg} kimg...')
    stats = Stats()
    cur_nimg, cur_tick = resume_kimg * 1000, 0
    tick_start_nimg, tick_start_time = cur_nimg, time.time()
    maintenance_time = tick_start_time - start_time
    if world > 1:
        torch.distributed.barrier()
    dist.print0('Start Running')
    # Inputs of the NEXT iteration (prompt batch, dropout flags, z / noise / t, tokenisation + text encoding: ~100 one-block
    # kernels) are prepared on their own stream while the current iteration runs -- a data-loader prefetch.  The draws keep the
    # reference's order (the step itself consumes no RNG); the one look-ahead set that is never used costs a prompt batch.
    prep_stream = torch.cuda.Stream(device) if (torch.cuda.is_available() and os.environ.get('SIDLSG_PREFETCH_INPUTS', '1') != '0') else None

    def build_inputs():
        with torch.cuda.stream(prep_stream) if prep_stream is not None else contextlib.nullcontext():
            # RNG order of the reference: all phase-A draws, then all phase-B draws (the compute in between consumes no RNG)
            inputs = dict(A=[make_round(use_dropout) for _ in range(rounds)])
            inputs['B'] = [make_round(False) for _ in range(rounds)]
            ev = None
            if prep_stream is not None:
                ev = torch.cuda.Event()
                ev.record()
        return inputs, ev

    if prep_stream is not None:
        # the text encoder's weight upload / casts and the weight broadcast were enqueued on the main stream
        prep_stream.wait_stream(torch.cuda.current_stream())
    # (the prefetch draws the prompt / dropout / noise RNG one iteration ahead: a state captured at a tick is one draw set past
    # the iteration it belongs to; the training-state files do not carry RNG state -- neither do the reference's, :656 -- so a
    # resumed run re-seeds exactly as the reference does)
    upcoming = build_inputs()
    while True:
        inputs, ev = upcoming
        if prep_stream is not None:
            upcoming = build_inputs()
            cur_stream = torch.cuda.current_stream()
            cur_stream.wait_event(ev)
            for ph in ('A', 'B'):
                for r in inputs[ph]:
                    for v in r.values():
                        v.record_stream(cur_stream)
        ema_beta = None
        if ema_halflife_kimg > 0:
            half = ema_halflife_kimg * 1000
            if ema_rampup_ratio is not None:
                half = min(half, cur_nimg * ema_rampup_ratio)
            ema_beta = 0.5 ** (batch_size / max(half, 1e-8))
        loss_f, loss_g = step.iteration(inputs, ema_beta=ema_beta)
        # (the reference reads both losses every iteration, :452 -- a host-device synchronisation; here they stay on the device
        # until the tick, unless a per-iteration observer wants them)
        stats.report('fake_score_Loss/loss', loss_f); stats.report('G_Loss/loss', loss_g)
        if on_iteration is not None:
            on_iteration(cur_nimg // batch_size, float(loss_f), float(loss_g))
        if prep_stream is None:
            upcoming = build_inputs() if cur_nimg + batch_size < total_kimg * 1000 else (None, None)
        cur_nimg += batch_size
        done = cur_nimg >= total_kimg * 1000
        if (not done) and (cur_tick != 0) and (cur_nimg < tick_start_nimg + kimg_per_tick * 1000):
            continue
        loss_f, loss_g = float(loss_f), float(loss_g)

        tick_end_time = time.time()
        sec_per_kimg = (tick_end_time - tick_start_time) / (cur_nimg - tick_start_nimg) * 1e3
        for k, v in (('Progress/tick', cur_tick), ('Progress/kimg', cur_nimg / 1e3), ('Timing/total_sec', tick_end_time - start_time),
                     ('Timing/sec_per_tick', tick_end_time - tick_start_time), ('Timing/sec_per_kimg', sec_per_kimg),
                     ('Timing/images_per_sec', 1e3 / sec_per_kimg), ('Timing/maintenance_sec', maintenance_time),
                     ('Resources/cpu_mem_gb', psutil.Process(os.getpid()).memory_info().rss / 2 ** 30),
                     ('Resources/peak_gpu_mem_gb', torch.cuda.max_memory_allocated(device) / 2 ** 30 if torch.cuda.is_available() else 0)):
            stats.report(k, v)
        dist.print0(f'tick {cur_tick:<5d} kimg {cur_nimg / 1e3:<9.1f} time {format_time(tick_end_time - start_time):<12s} '
                    f'sec/tick {tick_end_time - tick_start_time:<7.1f} sec/kimg {sec_per_kimg:<7.2f} maintenance {maintenance_time:<6.1f} '
                    f'loss_fake_score {loss_f:<6.2f} loss_G {loss_g:<6.2f}')
        if torch.cuda.is_available():
            torch.cuda.reset_peak_memory_stats()
        if (not done) and dist.should_stop():
            done = True
        # reference cadence (sid_training_loop.py:597): tick 0, every snapshot_ticks, a fixed early schedule, and the last tick
        if snapshot_ticks is not None and (done or cur_tick % snapshot_ticks == 0 or cur_tick in SNAPSHOT_EXTRA_TICKS) and rank == 0 and run_dir:
            with open(os.path.join(run_dir, f'network-snapshot-{alpha:03f}-{cur_nimg // 1000:06d}.pkl'), 'wb') as f:
                pickle.dump(dict(ema=G_ema), f)
        # metrics at the snapshot cadence, from tick 1 on (sid_training_loop.py:616-639, `if cur_tick>0`): the EMA generator
        # through the one-step sampler + VAE
        if metrics and cur_tick > 0 and snapshot_ticks is not None and (done or cur_tick % snapshot_ticks == 0 or cur_tick in SNAPSHOT_EXTRA_TICKS):
            from functools import partial

            from . import metrics as metric_main
            from .sd_util import sid_sd_sampler
            if G_ema is not G:
                # the fused Adam + EMA kernel writes G_ema's fp32 masters through raw pointers; its forward compute copies are
                # only as new as the last refresh -- without this the scores would be those of the INITIAL EMA weights
                G_ema.refresh_compute_weights()
            G_eval = partial(sid_sd_sampler, unet=G_ema, noise_scheduler=noise_scheduler, text_encoder=text_encoder, tokenizer=tokenizer,
                             resolution=resolution, dtype=torch.float32, return_images=True, vae=vae, train_sampler=False)
            # evaluation prompts: the caption set of `dataset_kwargs` (--data, the COCO-2014 validation captions the real-set
            # statistics of --data_stat were computed on; sid_training_loop.py:636, sid_metric_utils.py:419-421)
            if dataset_kwargs:
                msrc = dict(dataset_kwargs=dict(dataset_kwargs))
            else:
                dist.print0('WARNING: metrics without dataset_kwargs (--data): the evaluation prompts are the TRAINING prompts, so '
                            'fid30k_full / fid_clip_30k_full are NOT comparable with the reference\'s COCO-2014 numbers')
                msrc = dict(dataset=dataset_obj)
            for metric in metrics:
                extra = dict(num_test=metric_num_test) if metric_num_test is not None else {}
                result = metric_main.calc_metric(metric, G=G_eval, resolution=resolution, init_timestep=init_timestep, detector=metric_pt_path,
                                                 real_stats=metric_real_stats, open_clip_detector=metric_open_clip_path, device=device,
                                                 **msrc, **extra)
                metric_main.report_metric(result, run_dir=run_dir, alpha=alpha,
                                          snapshot_pkl=os.path.join(run_dir, f'network-snapshot-{alpha:03f}-{cur_nimg // 1000:06d}.pkl') if run_dir else None)
                for k, v in result.results.items():
                    stats.report(f'Metrics/{k}', v)
        if state_dump_ticks is not None and (done or cur_tick % state_dump_ticks == 0) and cur_tick != 0 and rank == 0 and run_dir:
            torch.save(dict(fake_score=fake_score.state_dict(), G=G.state_dict(), G_ema=G_ema.state_dict(),
                            fake_score_optimizer_state=opt_f.state_dict(), g_optimizer_state=opt_g.state_dict()),
                       os.path.join(run_dir, f'training-state-{cur_nimg // 1000:06d}.pt'))
        if rank == 0 and run_dir:
            with open(os.path.join(run_dir, f'stats_{alpha:03f}.jsonl'), 'at') as f:
                f.write(json.dumps(dict(stats.as_dict(), timestamp=time.time())) + '\n')
        stats.reset()
        dist.update_progress(cur_nimg // 1000, total_kimg)
        cur_tick += 1
        tick_start_nimg, tick_start_time = cur_nimg, time.time()
        maintenance_time = tick_start_time - tick_end_time
        if done:
            break
    dist.print0('\nExiting...')
    return dict(G=G, fake_score=fake_score, G_ema=G_ema)


def evaluate_network(run_dir, dataset_kwargs, network_kwargs, device, metrics, init_timestep, metric_pt_path, metric_open_clip_path,
                     pretrained_model_name_or_path, network_pkl, resolution, num_steps=1, metric_real_stats=None, metric_num_test=None,
                     dataset_prompt_text_kwargs=None):
    """`--train_mode 0` (sid_training_loop.py:680-745): load the text encoder / VAE / scheduler, un-pickle the distilled generator
    from `network_pkl` (`pickle.load(f)['ema']`, the file the training loop writes at the snapshot ticks) and evaluate every metric
    with 1, 2 and 4 generation steps; each result goes to `<dirname(run_dir)>/<metric><number>_<steps>.txt` in the reference's
    `key: value` format.  The preview PNG grids of that branch are not produced (cold path, like the training loop's)."""
    import re
    from functools import partial

    from . import metrics as metric_main
    from .sd_util import sid_sd_sampler
    if not metrics:
        raise ValueError('--train_mode 0 evaluates metrics: pass --metrics')
    if not network_pkl or not os.path.isfile(network_pkl):
        raise FileNotFoundError(f'--network_pkl {network_pkl!r}: a local network-snapshot-*.pkl is needed (there is no network to fetch one)')
    dtype = resolve_compute_dtype(dict(network_kwargs).get('compute_dtype'))
    _, vae, noise_scheduler, text_encoder, tokenizer = load_sd15(
        pretrained_model_name_or_path=pretrained_model_name_or_path, pretrained_vae_model_name_or_path=None, device=device,
        weight_dtype=dtype, lora_config=None, compute_dtype=dtype)
    dist.print0('Loading network completed')
    dist.print0(f'Loading network from "{network_pkl}"...')
    with open(network_pkl, 'rb') as f:
        G_ema = pickle.load(f)['ema'].to(device)
    G_ema.eval().requires_grad_(False)
    m = re.search(r'-(\d+)\.pkl$', network_pkl)
    number_part = m.group(1) if m else '_final'
    if dataset_kwargs:
        msrc = dict(dataset_kwargs=dict(dataset_kwargs))
    else:
        if not dataset_prompt_text_kwargs:
            raise ValueError('evaluate_network: neither dataset_kwargs (--data: the evaluation caption set) nor dataset_prompt_text_kwargs '
                             '(--data_prompt_text) was given -- there are no prompts to evaluate on')
        dist.print0('WARNING: no dataset_kwargs (--data): evaluating on the training prompts; not comparable with the reference\'s COCO numbers')
        msrc = dict(dataset_kwargs=dict(dataset_prompt_text_kwargs))
    out = {}
    for num_steps_eval in (1, 2, 4):
        for metric in metrics:
            G_eval = partial(sid_sd_sampler, unet=G_ema, noise_scheduler=noise_scheduler, text_encoder=text_encoder, tokenizer=tokenizer,
                             resolution=resolution, dtype=torch.float32, return_images=True, vae=vae, num_steps=num_steps, train_sampler=False,
                             num_steps_eval=num_steps_eval)
            extra = dict(num_test=metric_num_test) if metric_num_test is not None else {}
            result = metric_main.calc_metric(metric, G=G_eval, resolution=resolution, init_timestep=init_timestep, detector=metric_pt_path,
                                             real_stats=metric_real_stats, open_clip_detector=metric_open_clip_path, device=device, **msrc, **extra)
            out[(metric, num_steps_eval)] = result
            if dist.get_rank() == 0:
                print(result.results)
                txt = os.path.join(os.path.dirname(run_dir) if run_dir else '.', f'{metric}{number_part}_{num_steps_eval:d}.txt')
                print(txt)
                with open(txt, 'w') as f:                          # save_metric (sid_training_loop.py:134-137)
                    for k, v in result.items():
                        f.write(f'{k}: {v}\n')
    return out


def _hip_opt(kw):
    """The reference CLI names torch.optim.Adam/AdamW (sid_train.py:219-226); the HIP step needs the fused flat-buffer
    optimizer, which has the same hyper-parameter names.  Anything else is an error, never a silent substitute."""
    kw = EasyDict(kw)
    name = kw.get('class_name', 'sid_lsg_amd.optim.FusedAdamEMA')
    mapping = {'torch.optim.Adam': 'sid_lsg_amd.optim.FusedAdamEMA', 'torch.optim.AdamW': 'sid_lsg_amd.optim.FusedAdamWEMA'}
    name = mapping.get(name, name)
    if not name.startswith('sid_lsg_amd.optim.'):
        raise ValueError(f'optimizer {name} cannot drive the flat-buffer HIP step; use sid_lsg_amd.optim.FusedAdamEMA / FusedAdamWEMA')
    kw['class_name'] = name
    return kw


def _sd(obj):
    return obj.state_dict() if isinstance(obj, torch.nn.Module) else obj


def _snapshot_state(net):
    """fp32 state of a network unpickled from network-snapshot-*.pkl (not yet materialised on a GPU) or of a live one."""
    pending = getattr(net, '_pending_state', None)
    return pending if pending is not None else net.state_dict()
