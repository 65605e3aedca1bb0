#!/usr/bin/env python
"""Benchmark: SiD-LSG distillation images/sec (BASELINE.json metric) on N MI355X of one node.

One "step" = one loop iteration of training/sid_training_loop.py:383-571 = one fake-score update + one generator
update on `batch_gpu` synthetic 512x512 samples per GPU (64x64x4 latents), SD1.5 architecture, kappa=1.5 on every
branch (CFG batch [uncond ; cond] for both the fake-score net and the teacher), alpha=1, Adam(beta1=0), EMA.
Nothing is skipped inside the timed region: noise/timestep sampling, CLIP text encoding of the step's prompts
(PyTorch-ROCm, per north_star), 5 UNet forwards + 4 UNet backwards (18 F), both optimizer steps, EMA, and for N>1
the gradient all-reduces.  Weights are seeded random (no SD checkpoints offline) and prompts are synthetic
Aesthetic-style captions tokenised by the offline stand-in tokenizer: `"data": "synthetic"`.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL on this platform: must be set before the HIP runtime starts

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_GMAC = {'sd15': 401.63672064, 'sd21-base': 402.12873216}     # UNet forward GMAC / sample at 64x64 latents
PEAK_BF16_TFLOPS = 2500.0                                        # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

PROMPT_WORDS = ('a highly detailed photograph of portrait landscape castle mountain river at sunset golden hour cinematic lighting '
                'oil painting watercolor studio light bokeh sharp focus intricate elegant trending concept art dramatic sky').split()


def synth_prompts(n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        k = int(torch.randint(6, 18, (1,), generator=g))
        idx = torch.randint(0, len(PROMPT_WORDS), (k,), generator=g).tolist()
        out.append(' '.join(PROMPT_WORDS[i] for i in idx))
    return out


class DispatchTrace:
    """Per-family kernel time of the step from per-DISPATCH timestamps (include/sidlsg_hip.h "in-library kernel timing"): every
    `stride`-th call of a family launches its kernels with start / stop events bound to the kernel's own dispatch packet, so the
    durations are the begin -> end intervals `rocprofv3 --kernel-trace` reports for the same command -- not event records around a
    launch, which add barrier packets and over-state ~50 us kernels under three concurrent streams by ~1.5x (round 3)."""
    FAM = dict(gemm=0, conv=1, attn=2, attn_bwd=3, wgrad=4, conv_wgrad=5, gn=6, gn_bwd=7, ln=8, ln_bwd=9)
    # every STRIDE-th call of a family is timed (primes: coprime to the per-layer launch pattern, every shape gets sampled).  Round 5: x4 --
    # sampling every 3rd-7th call cost the timed region 2.7 ms per iteration (1.3 %; hipExtLaunchKernelGGL + two events per kernel), every
    # 11th-29th 0.9 ms, with 460 dense-GEMM launches still timed in 12 iterations (tools/experiments/r5_call19.sh)
    STRIDE = dict(gemm=29, conv=19, attn=11, attn_bwd=11, wgrad=19, conv_wgrad=11, gn=19, gn_bwd=19, ln=29, ln_bwd=29)

    def __init__(self, lib):
        self.lib = lib

    def start(self):
        self.lib.sidlsg_trace_enable(1 << 17)
        mul = int(os.environ.get('SIDLSG_BENCH_TRACE_STRIDE_MUL', '1'))      # (A/B: cost of the sampling itself)
        for k, f in self.FAM.items():
            self.lib.sidlsg_trace_set_stride(f, self.STRIDE[k] * mul)

    def stop(self):
        torch.cuda.synchronize()
        out = {}
        buf = torch.zeros(9, dtype=torch.float64)
        for k, f in self.FAM.items():
            buf[7], buf[8] = PEAK_BF16_TFLOPS * 1e12, PEAK_HBM_GBS * 1e9
            self.lib.sidlsg_trace_read(f, buf.data_ptr())
            ms, work, sampled, calls, kernels, nbytes, bound_ms = buf.tolist()[:7]
            out[k] = dict(launches=int(sampled), calls=int(calls), ms=ms, flops=work, kernels=int(kernels), bytes=nbytes, bound_ms=bound_ms)
        self.lib.sidlsg_trace_enable(0)
        return out


LINE_LIMIT = 4096          # bytes of the ONE stdout line (the driver keeps the last ~8 KB of stdout)
FAMILY_KERNEL = {'gemm': 'gemm_v3_kernel<0> (+gemm_bf16/gemm_as/gemm_finish): dense GEMM fwd+dgrad', 'conv': 'gemm_v3_kernel<1>: implicit-GEMM conv3x3 fwd+dgrad',
                 'attn': 'attn_q_kernel<*,*,0>: flash attention fwd', 'attn_bwd': 'attn_q_kernel<*,*,1> + attn_dkdv_kernel: flash attention bwd',
                 'wgrad': 'wgrad_v2_kernel<0>/wgrad_v2s_kernel (+wgrad_reduce): dense dW', 'conv_wgrad': 'wgrad_v2f/v2wf/v2<1> (+wgrad_reduce): conv dW',
                 'gn': 'gn_stats+gn_apply / gn_rows / gn_small: GroupNorm+SiLU fwd', 'gn_bwd': 'gn_bwd_stats+gn_bwd_apply / gn_small_bwd: GroupNorm+SiLU bwd',
                 'ln': 'ln_fwd_kernel', 'ln_bwd': 'ln_bwd_kernel'}


def _r(x, n=4):
    return None if x is None else (round(x, n) if isinstance(x, float) else x)


def compact_line(full):
    """The ONE JSON line of stdout (< LINE_LIMIT bytes; tests/test_host_logic.py::test_bench_line_is_compact) from the full result
    dict, which goes to bench_detail.json.  Keys: the driver's contract + `roofline` of the dominant kernel family (live,
    per-dispatch timestamps over the timed region; `traffic` is null: HBM-side PMC bytes need their own serialising rocprofv3
    passes and live under profiles/) + one fraction per other family + `cpu_baseline`."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
    out = {k: _r(full[k]) for k in keep}
    c = full['config']
    out['config'] = {'workload': c['workload'][:200], 'global_batch': c['global_batch'], 'parallelism': c['parallelism'],
                     'teacher_weights': c['teacher_weights']}
    for k in ('step_tflops', 'step_mfma_frac', 'loss_check', 'grouped_frozen_pass', 'graph', 'peak_mem_gb', 'loss_fake', 'loss_G'):
        if k in full:
            out[k] = _r(full[k])
    for k in ('teacher_pass', 'frozen_pair_pass'):
        if full.get(k):
            out[k] = {'ms': _r(full[k]['ms'], 3), 'mfma_frac': _r(full[k]['mfma_frac'])}
    if full.get('comm'):
        cm = full['comm']
        out['comm'] = {'comm_exposed_ms': _r(cm['comm_exposed_ms'], 3), 'exposed_ms_per_step': {k: _r(v, 3) for k, v in cm['exposed_ms_per_step'].items()},
                       'messages_per_step': _r(cm['messages_per_step'], 2), 'bytes_per_step': cm['bytes_per_step'],
                       'allreduce_algbw_GBps': _r(cm['allreduce_algbw_GBps'], 2), 'allreduce_busbw_GBps': _r(cm['allreduce_busbw_GBps'], 2),
                       'backend': cm['backend']}
        if cm.get('expected'):
            out['comm']['expected_allreduce_ms'] = _r(cm['expected']['allreduce_ms_per_step'], 2)
            out['comm']['expected_exposed_ms'] = _r(cm['expected']['exposed_ms_per_step'], 2)
            out['comm']['allreduce_ms_per_step'] = _r(cm['allreduce_ms_per_step'], 2)
    r = full.get('roofline')
    if r:
        iso = r.get('isolated') or {}
        out['roofline'] = {'family': r['family'], 'kernel': FAMILY_KERNEL.get(r['family'], r['family']), 'bound': r['bound'],
                           'achieved': _r(r['achieved'], 2), 'peak': r['peak'], 'unit': r['unit'], 'frac': _r(r['frac']),
                           'isolated_frac': _r(iso.get('frac')), 'frac_of_per_call_roofline': _r(r.get('frac_of_per_call_roofline')),
                           'avg_launch_ms': _r(r['avg_launch_ms'], 5), 'launches_timed': r['launches'], 'est_ms_per_step': _r(r['est_ms_per_step'], 2),
                           'traffic': None, 'timing': 'per-dispatch start/stop timestamps over the timed region (3 streams share the chip); '
                                                      'isolated_frac: same kernels, streams folded into one'}
        out['fracs'] = {k[len('roofline_'):]: [_r(v['frac']), _r((v.get('isolated') or {}).get('frac')), _r(v['est_ms_per_step'], 1)]
                        for k, v in full.items() if k.startswith('roofline_')}
        out['fracs_legend'] = '[frac of peak in the timed region, stand-alone, est ms/step]; gn/ln: of 8 TB/s, others of 2.5 PFLOP/s'
    cb = full.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = {'value': _r(cb['value'], 5), 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb['sample'][:120]}
    out['detail'] = 'bench_detail.json'
    # the line must fit the driver's tail: shed optional keys rather than end a finished run in an AssertionError with no line
    for key in ('fracs_legend', 'fracs', 'comm', 'teacher_pass', 'frozen_pair_pass'):
        if len(json.dumps(out)) < LINE_LIMIT:
            break
        out.pop(key, None)
    if len(json.dumps(out)) >= LINE_LIMIT and 'cpu_baseline' in out:
        out['cpu_baseline']['sample'] = out['cpu_baseline']['sample'][:40]
    return out


def cpu_baseline(arch, threads, kappa=1.5):
    """The reference CPU loop timed on the host cores (kind 'port': oracle/sid_ref.py::sid_iteration_ref, the restatement of
    sid_training_loop.py:383-571 that reproduces the reference's golden loss curves, on the oracle UNet): ONE complete
    iteration at BASELINE.json configs[0] -- full-size architecture, batch 1, 64x64x4 latents, fp32, kappa on every branch:
    generator forward, fake-score CFG forward + backward, Adam; generator forward + backward through the fake-score and
    teacher CFG passes, Adam, EMA.  One iteration = one image at batch 1.  Bounded: a single iteration (~1-2 min)."""
    import copy

    from oracle import sid_ref
    from oracle.scheduler_ref import DDPMSchedulerRef
    from oracle.unet_ref import CONFIGS, UNet2DConditionRef
    torch.set_num_threads(threads)
    cfg = CONFIGS[arch]
    t_setup = time.time()
    torch.manual_seed(0)
    phi = UNet2DConditionRef(cfg).eval().requires_grad_(False)
    psi, G = copy.deepcopy(phi), copy.deepcopy(phi)
    nets = dict(true_score=phi, fake_score=psi, G=G, G_ema=copy.deepcopy(G))
    st = dict(fake_score=[{} for _ in psi.parameters()], G=[{} for _ in G.parameters()])
    g = torch.Generator().manual_seed(1)
    lat, b = 64, 1

    def rnd_round():
        return dict(z=torch.randn(b, 4, lat, lat, generator=g), noise=torch.randn(b, 4, lat, lat, generator=g),
                    t=torch.randint(20, 980, (b,), generator=g), cond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=g),
                    uncond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=g))
    inputs = dict(A=[rnd_round()], B=[rnd_round()])
    hp = dict(alpha=1.0, kappa1=kappa, kappa2=kappa, kappa4=kappa, ls=1.0, lsg=1.0, batch_gpu_total=b, lr=1e-6, glr=1e-6,
              betas=(0.0, 0.999), eps=1e-8, init_t=625, batch_size=b, ema_halflife_kimg=50, ema_rampup_ratio=0.05, cur_nimg=0)
    # page-in / oneDNN primitive creation outside the timed region: one small forward
    with torch.no_grad():
        phi(torch.randn(1, 4, 16, 16), torch.tensor([625]), encoder_hidden_states=inputs['A'][0]['cond'])
    t_setup = time.time() - t_setup
    t0 = time.time()
    out = sid_ref.sid_iteration_ref(nets, st, DDPMSchedulerRef(), inputs, hp)
    dt = time.time() - t0
    return dict(value=b / dt, unit='images/s', cores=threads, kind='port',
                sample=f'1 full iteration of oracle/sid_ref.py (fp32 CPU restatement), {arch} full size, batch 1, 64x64x4, {dt:.1f} s',
                detail=f'fake-score step + generator step + Adam + EMA, kappa={kappa}, {threads} threads (setup {t_setup:.0f} s not timed); '
                       f'loss_fake {out["loss_fake"]:.1f}')


LOSS_REFERENCE = os.path.join(ROOT, 'tests', 'golden', 'bench_loss_reference.json')
LOSS_TOL = (2e-3, 1e-2)          # bf16 production mode against fp32: tests/test_gpu_unet.py::TOL_LOSS[bfloat16]


def loss_reference_key(arch, batch_gpu, resolution, kappa):
    return f'{arch}_b{batch_gpu}_{resolution}_k{kappa:g}'


def load_loss_reference(args, world):
    if args.teacher_weights != 'bf16' or not os.path.isfile(LOSS_REFERENCE):
        return None
    with open(LOSS_REFERENCE) as f:
        return json.load(f).get(loss_reference_key(args.arch, args.batch_gpu, args.resolution, args.kappa))


def check_losses(ref, it, lf, lg):
    """bf16 losses of iteration `it` against the stored fp32 references of this workload: the fp32 CPU ORACLE itself where it has
    been run (iteration 0: `oracle_fp32`, oracle/make_bench_oracle_reference.py), the HIP fp32-accurate mode (pinned to the oracle
    at iteration 0 by tests/test_gpu_bench_parity.py) for the later iterations."""
    if it >= len(ref['loss_fake']):
        return dict(iteration=it, ok=True, skipped=f'reference holds {len(ref["loss_fake"])} iterations')
    orc = ref.get('oracle_fp32')
    if orc is not None and it < len(orc['loss_fake']):
        rf, rg, src = orc['loss_fake'][it], orc['loss_G'][it], 'oracle_fp32'
    else:
        rf, rg, src = ref['loss_fake'][it], ref['loss_G'][it], 'hip_fp32'
    ef, eg = abs(lf - rf) / abs(rf), abs(lg - rg) / abs(rg)
    return dict(iteration=it, reference=src, loss_fake=lf, loss_fake_ref=rf, rel_fake=ef, loss_G=lg, loss_G_ref=rg, rel_G=eg,
                bounds=list(LOSS_TOL), ok=bool(ef <= LOSS_TOL[0] and eg <= LOSS_TOL[1]))


class _Setup:
    pass


def setup_step(arch, b, resolution, kappa, dev, rank=0, world=1, teacher_weights='bf16', reducer=None, compute_dtype=torch.bfloat16):
    """The benchmark's networks, optimizers, SiDStep and synthetic input stream (also used by tests/test_gpu_bench_parity.py and
    tools/make_bench_loss_reference.py, which run the SAME workload in the fp32-accurate mode)."""
    from sid_lsg_amd.optim import FusedAdamEMA
    from sid_lsg_amd.sd_util import load_sd15
    from sid_lsg_amd.sid_step import SiDStep
    from sid_lsg_amd.text import TextConditioner
    S = _Setup()
    lat = resolution // 8
    phi, vae, sched, text_encoder, tokenizer = load_sd15(f'random:{arch}', None, dev, compute_dtype, seed=0, compute_dtype=compute_dtype)
    psi = phi.clone_network()
    G = phi.clone_network()
    G_ema = phi.clone_network(with_grad_buffers=False)
    if teacher_weights in ('fp8', 'fp8-frozen'):
        phi.enable_fp8_weights()
    if teacher_weights == 'fp8-frozen':       # + the fake-score network's phase-B evaluation and the generator's no-grad pass
        psi.enable_fp8_weights(frozen_passes_only=True)
        G.enable_fp8_weights(frozen_passes_only=True)
    text_encoder.to(torch.bfloat16)          # the text states are bf16 in both compute modes (PyTorch-ROCm CLIP, north_star)
    cond = TextConditioner(tokenizer, text_encoder, out_dtype=compute_dtype)
    opt_f = FusedAdamEMA(psi.parameters(), lr=1e-6, betas=(0.0, 0.999), eps=1e-8)
    opt_g = FusedAdamEMA(G.parameters(), lr=1e-6, betas=(0.0, 0.999), eps=1e-8)
    step = SiDStep(G, psi, phi, G_ema, sched, opt_f, opt_g, alpha=1.0, cfg_train_fake=kappa, cfg_eval_fake=kappa,
                   cfg_eval_real=kappa, batch_gpu_total=b, init_timestep=625, reducer=reducer, world_size=world)
    batch_size = b * world
    gen = torch.Generator(device=dev)

    # The inputs of iteration it+1 (noise, timesteps, tokenisation + CLIP text encoding of both phases' prompts: ~100 one-block
    # kernels, 2.6 ms of an otherwise serial stream) are prepared on their own stream while iteration it runs -- a data-loader
    # prefetch: every iteration still prepares exactly one set of inputs inside the timed region.  $SIDLSG_BENCH_PREFETCH=0: serial.
    prefetch = os.environ.get('SIDLSG_BENCH_PREFETCH', '1') != '0'
    prep_stream = torch.cuda.Stream(dev) if prefetch else None
    if prefetch:
        prep_stream.wait_stream(torch.cuda.current_stream())
    pending = {}

    def prepare(it):
        with torch.cuda.stream(prep_stream) if prefetch else contextlib.nullcontext():
            gen.manual_seed(1000 * rank + it)
            inputs = dict(A=[], B=[])
            for k, ph in enumerate(('A', 'B')):
                prompts = synth_prompts(b, seed=(it * 2 + k) * 1000 + rank)      # rank 0's stream does not depend on the world size
                z = torch.randn(b, 4, lat, lat, device=dev, generator=gen)
                noise = torch.randn(b, 4, lat, lat, device=dev, generator=gen)
                t = torch.randint(20, 980, (b,), device=dev, generator=gen)
                inputs[ph].append(dict(z=z, noise=noise, t=t, cond=cond.encode(prompts), uncond=cond.uncond(b)))
            ev = None
            if prefetch:
                ev = torch.cuda.Event()
                ev.record()
        return inputs, ev

    def one_iteration(it):
        inputs, ev = pending.pop(it, None) or prepare(it)
        if prefetch:
            pending[it + 1] = prepare(it + 1)
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for ph in ('A', 'B'):
                for r in inputs[ph]:
                    for v in r.values():
                        v.record_stream(cur)
        half = min(50 * 1000, it * batch_size * 0.05)
        beta = 0.5 ** (batch_size / max(half, 1e-8))
        return (step.iteration_graphed if S.use_graph else step.iteration)(inputs, ema_beta=beta)

    S.use_graph = False
    S.step, S.phi, S.psi, S.G, S.G_ema, S.cond, S.sched, S.gen = step, phi, psi, G, G_ema, cond, sched, gen
    S.batch_size, S.prepare, S.one_iteration, S.lat = batch_size, prepare, one_iteration, lat
    return S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch-gpu', type=int, default=8)
    ap.add_argument('--arch', default='sd15')
    ap.add_argument('--kappa', type=float, default=1.5)
    ap.add_argument('--resolution', type=int, default=512, help='image resolution (latents are resolution/8); 768 for BASELINE config #4')
    ap.add_argument('--teacher-weights', default='bf16', choices=['bf16', 'fp8', 'fp8-frozen'],
                    help="fp8: the frozen teacher's forward weights as e4m3 + per-channel scales (BASELINE configs[4] precision; "
                         "not the headline configuration); fp8-frozen: every pass without weight gradients -- teacher, the fake-score "
                         "network's phase-B evaluation, the generator's no-grad pass of phase A")
    ap.add_argument('--graph', action='store_true',
                    help='run the timed region through SiDStep.iteration_graphed (one HIP graph per iteration); per-kernel event '
                         'timing then happens on eager iterations after the timed region')
    ap.add_argument('--force-exchange', action='store_true',
                    help='single rank only: run the data-parallel gradient exchange anyway (RCCL all-reduces over a world of 1, '
                         'FlatGradReducer(min_world=1)) -- the schedule, stream waits and launch path of a multi-GPU run on one GPU; '
                         '`comm` then reports how long the compute stream stood still for it (no xGMI traffic: a lower bound)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    # test hook (1-GPU boxes): SIDLSG_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for the exchange, so the
    # whole multi-rank control flow (reducer, barriers, max-over-ranks timing) can be exercised without a second GPU
    share = os.environ.get('SIDLSG_BENCH_SHARE_GPU', '0') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.distributed.init_process_group('gloo' if share else 'nccl', init_method='env://', **({} if share else dict(device_id=dev)))
    elif args.force_exchange and os.environ.get('SIDLSG_BENCH_NO_PG', '0') != '1':      # (NO_PG: diagnosis, with SIDLSG_EXCHANGE_DRYRUN=1)
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0, device_id=dev)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if torch.distributed.is_initialized():
        # RCCL creates its communicator at the first collective: do that HERE (main thread, default stream) -- the step's first
        # collective is issued from inside the backward pass (autograd's device thread, communication stream)
        warm = torch.ones(1 << 16, device=dev)
        torch.distributed.all_reduce(warm)
        torch.cuda.synchronize()
        del warm

    from sid_lsg_amd._lib import lib
    from sid_lsg_amd.distributed import FlatGradReducer
    lib.load()

    b = args.batch_gpu
    lat = args.resolution // 8
    reducer = FlatGradReducer() if world > 1 else (FlatGradReducer(min_world=1) if args.force_exchange else None)
    if os.environ.get('SIDLSG_BENCH_PG_ONLY', '0') == '1':      # diagnosis: process group alive, no exchange
        reducer = None
    if reducer is not None and not args.graph and os.environ.get('SIDLSG_BENCH_COMM_TIMING', '1') != '0':
        reducer.enable_timing()          # exposed-communication figures of the multi-GPU run (`comm` in the JSON line)
    S = setup_step(args.arch, b, args.resolution, args.kappa, dev, rank=rank, world=world, teacher_weights=args.teacher_weights,
                   reducer=reducer)
    step, phi, cond, sched, gen, batch_size = S.step, S.phi, S.cond, S.sched, S.gen, S.batch_size
    one_iteration = S.one_iteration
    loss_ref = load_loss_reference(args, world)

    S.use_graph = args.graph

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    checks = []
    for it in range(args.warmup):
        lf, lg = one_iteration(it)
        if it == 0 and rank == 0 and loss_ref is not None:          # iteration 0 of rank 0 does not depend on the world size
            checks.append(check_losses(loss_ref, 0, float(lf), float(lg)))
    sync()
    timers = {}
    # the attention entry points the networks use: pre-scaled queries unless switched off (unet._build_prescale_plan)
    ps = any(getattr(m, 'prescaled', False) for m in phi.modules())
    ATTN_FWD, ATTN_BWD = ('sidlsg_attn_fwd_ps', 'sidlsg_attn_bwd_ps') if ps else ('sidlsg_attn_fwd', 'sidlsg_attn_bwd')

    trace = DispatchTrace(lib)
    tracing = not args.no_kernel_timing and rank == 0 and not S.use_graph
    if tracing:      # rooflines of the step's kernel families, sampled live over the timed region
        trace.start()
    t0 = time.time()
    first = None
    for it in range(args.warmup, args.warmup + args.steps):
        lf, lg = one_iteration(it)
        if first is None:
            first = (lf.clone(), lg.clone())      # device scalars of the first timed step, read after the timed region (copies: the graphed
                                                  # iteration returns STATIC scalars that the next replay overwrites)
    sync()
    dt = time.time() - t0
    if rank == 0 and loss_ref is not None and first is not None and (world == 1 or args.warmup == 0):
        checks.append(check_losses(loss_ref, args.warmup, float(first[0]), float(first[1])))
    comm = reducer.timing_report(world) if (reducer is not None and reducer.timing) else None
    if reducer is not None:
        reducer.enable_timing(False)
    if tracing:
        timers = trace.stop()
    if world > 1:      # max over ranks, BEFORE anything rank-dependent: every collective below is executed by every rank
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    t_host = None
    if S.use_graph:                  # per-kernel events cannot see inside a graph launch: sample eager iterations now
        hs = time.time()
        for it in range(args.warmup + args.steps, args.warmup + args.steps + 3):
            one_iteration(it)
        t_host = (time.time() - hs) / 3 * 1e3          # host time to enqueue one graphed iteration (inputs + text encode + replay)
        torch.cuda.synchronize()
        S.use_graph = False
        if not args.no_kernel_timing:                 # (iterations contain the gradient exchange: every rank runs them)
            if rank == 0:
                trace.start()
            for it in range(args.warmup + args.steps + 3, args.warmup + args.steps + 6):
                one_iteration(it)
            torch.cuda.synchronize()
            if rank == 0:
                timers = trace.stop()
    # The timed region runs the teacher (or the early generator forward) on a second HIP stream and the weight gradients on a
    # third (ops._OnWgradStream): kernels of the streams share the chip, so a kernel's duration over the timed region (what
    # rocprofv3 --kernel-trace reports for the same command) is longer than the kernel's own.  A few extra iterations AFTER the
    # timed region, with that overlap switched off, give each family's stand-alone figure (`isolated` in the roofline objects).
    iso = {}
    if not args.no_kernel_timing and step.side is not None:          # rank-independent condition: the iterations exchange gradients
        from sid_lsg_amd import ops as _ops
        side, step.side = step.side, None
        wgrad_side, _ops._WGRAD_SIDE = _ops._WGRAD_SIDE, False      # weight gradients back on the main stream as well
        if rank == 0:
            trace.start()
        for it in range(args.warmup + args.steps, args.warmup + args.steps + 3):
            one_iteration(it)
        torch.cuda.synchronize()
        if rank == 0:
            iso = trace.stop()
        step.side = side
        _ops._WGRAD_SIDE = wgrad_side
    # north_star's second figure: MFMA utilisation of the CFG teacher pass alone (phi forward on the [uncond; cond] batch of
    # 2b samples + guidance + x0), timed with events on a few extra passes after the timed region (rank 0)
    teacher = pair = None
    if timers:
        from sid_lsg_amd.sd_util import hip_denoise, hip_prepare_denoise
        with torch.no_grad():
            prompts = synth_prompts(b, seed=12345)
            img = torch.randn(b, 4, lat, lat, device=dev, generator=gen)
            prep = hip_prepare_denoise(img, torch.randn(b, 4, lat, lat, device=dev, generator=gen),
                                       torch.randint(20, 980, (b,), device=dev, generator=gen), cond.encode(prompts), cond.uncond(b),
                                       sched, args.kappa != 1)
            hip_denoise(phi, prep, args.kappa, predict_x0=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hip_denoise(phi, prep, args.kappa, predict_x0=True)
            e1.record()
            torch.cuda.synchronize()
            teacher = e0.elapsed_time(e1) / 5.0
            # the same for the GROUPED pass of phase B: fake-score network + teacher on the stacked batch of 4b samples, one launch
            # per layer for both (HipUNet2DCondition.forward_pair; sid_step.py)
            pair = None
            if step._can_group():          # (measured whether or not the step itself uses it at this batch size: `grouped_frozen_pass`)
                from sid_lsg_amd import ops as _o

                def pair_pass():
                    ef, er = S.psi.forward_pair(phi, prep.xin, prep.tt, prep.ctx)
                    _o.cfg_x0(ef, prep.xt, prep.s0, prep.s1, args.kappa, True, torch.bfloat16)
                    _o.cfg_x0(er, prep.xt, prep.s0, prep.s1, args.kappa, True, torch.bfloat16)
                pair_pass()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    pair_pass()
                e1.record()
                torch.cuda.synchronize()
                pair = e0.elapsed_time(e1) / 5.0
    if rank != 0:
        _shutdown(world)
        return
    value = args.steps * batch_size / dt
    f_gmac = F_GMAC.get(args.arch, 0.0)
    if args.resolution != 512:          # analytic walk of the oracle for other latent sizes (attention is not linear in pixels)
        from sid_lsg_amd.unet import CONFIGS as _HC, unet_forward_macs
        f_gmac = unet_forward_macs(_HC[args.arch], lat, lat) / 1e9
    f_tflop = 2 * f_gmac / 1000.0
    img_tflop = 18 * f_tflop if args.kappa != 1 else 11 * f_tflop
    out = {
        'metric': 'distillation images/sec (512^2, SD1.5, kappa=1.5)' if (args.arch, args.resolution, args.kappa) == ('sd15', 512, 1.5) else f'distillation images/sec ({args.resolution}^2, {args.arch}, kappa={args.kappa})', 'value': value, 'unit': 'images/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1000.0, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': {'bf16': 'bf16', 'fp8': 'bf16 (teacher forward: e4m3 weights + MX-fp8 MFMA)', 'fp8-frozen': 'bf16 (passes without weight gradients: e4m3 weights + MX-fp8 MFMA)'}[args.teacher_weights], 'data': 'synthetic',
        'config': {'workload': f'{args.arch} SiD-LSG inner step, kappa={args.kappa} on all branches, {args.resolution}x{args.resolution} ({lat}x{lat}x4 latents), '
                               f'batch_gpu={b}, fp32 masters + bf16 MFMA compute, Adam(beta1=0)+EMA, random-init weights',
                   'global_batch': batch_size, 'parallelism': f'dp{world}' + ('+forced-exchange(world-1 RCCL)' if args.force_exchange and world == 1 else ''),
                   'teacher_weights': args.teacher_weights},
        'step_tflops': value * img_tflop, 'step_mfma_frac': value * img_tflop / (PEAK_BF16_TFLOPS * world),
        'graph': bool(args.graph), 'grouped_frozen_pass': bool(step._use_grouped(b)), 'host_enqueue_ms_per_step': t_host, 'loss_fake': float(lf), 'loss_G': float(lg), 'peak_mem_gb': torch.cuda.max_memory_allocated(dev) / 2 ** 30,
    }
    # parity at the BENCH workload: the losses of iteration 0 against the fp32 CPU ORACLE run on this workload's own inputs and weights
    # (tests/golden/bench_loss_reference.json `oracle_fp32`, oracle/make_bench_oracle_reference.py) and of the first timed step against
    # the stored HIP fp32-mode values (tools/make_bench_loss_reference.py; equal to the oracle at iteration 0 to 4e-8); bf16 bounds
    if loss_ref is None:
        out['loss_check'] = 'no reference stored for this configuration'
    elif not checks:
        out['loss_check'] = 'not run'
    else:
        out['loss_check'] = 'ok' if all(c['ok'] for c in checks) else 'FAILED'
        out['loss_check_detail'] = checks
    # multi-GPU: how long the compute stream stood still in FlatGradReducer.wait per step and network (the part of the gradient
    # exchange NOT hidden under compute), and the all-reduce rate the messages saw
    if comm is not None:
        ex = {k: v / (args.warmup + args.steps) for k, v in comm['exposed_ms'].items()}
        out['comm'] = {'exposed_ms_per_step': ex, 'comm_exposed_ms': sum(ex.values()), 'messages_per_step': comm['messages'] / (args.warmup + args.steps),
                       'bytes_per_step': comm['bytes'] / (args.warmup + args.steps), 'allreduce_ms_per_step': comm['comm_ms'] / (args.warmup + args.steps),
                       'allreduce_algbw_GBps': comm['algbw_GBps'], 'allreduce_busbw_GBps': comm['busbw_GBps'],
                       'backend': torch.distributed.get_backend(), 'note': 'events on the waiting stream around FlatGradReducer.wait (exposed) and on the '
                       'communication stream around every message (rate); warmup + timed iterations'}
        # What the first multi-GPU run should see, next to what it did see: ring all-reduce of the two networks' fp32
        # gradients, 2 (n - 1) / n x bytes over the slowest link pair; ASSUMED bus bandwidth 300 GB/s (RCCL rings over 7 x ~153 GB/s xGMI
        # links per GPU; nobody has measured it on this pool).  Six segments per network start under the backward that produces them; the
        # last segment (4.4 % of a network) cannot start before its backward ends: that tail is the exposed part by construction.
        if world > 1:
            bus = 300.0
            tot = out['comm']['bytes_per_step'] * 2.0 * (world - 1) / world / (bus * 1e9) * 1e3
            out['comm']['expected'] = {'assumed_busbw_GBps': bus, 'allreduce_ms_per_step': tot, 'exposed_ms_per_step': 0.044 * tot,
                                       'hidden_if_step_ms_above': tot, 'note': 'ring model, 6 segments per network overlapped with its backward; '
                                       'compare with allreduce_ms_per_step / comm_exposed_ms (measured)'}
    else:
        out['comm'] = None
    if teacher is not None:
        n_fwd = (2 if args.kappa != 1 else 1) * b
        tf = n_fwd * f_tflop / (teacher * 1e-3)
        out['teacher_pass'] = {'what': f'phi forward on the CFG batch of {n_fwd} samples + guidance + x0 (no grad)', 'ms': teacher,
                               'tflops': tf, 'mfma_frac': tf / PEAK_BF16_TFLOPS}
        if pair is not None:
            tf2 = 2 * n_fwd * f_tflop / (pair * 1e-3)
            out['frozen_pair_pass'] = {'what': f'GROUPED pass of phase B: fake-score network + teacher on the stacked batch of {2 * n_fwd} samples '
                                               '(one launch per layer for both networks) + guidance + x0 (no grad)', 'ms': pair, 'tflops': tf2,
                                       'mfma_frac': tf2 / PEAK_BF16_TFLOPS, 'vs_two_separate_passes': 2 * teacher / pair}
    if timers:
        KERNELS = {'conv': 'implicit-GEMM conv3x3 fwd + dgrad (gemm_v3_kernel<1>, gemm_bf16_kernel<*,*,1|2>)',
                   'gemm': 'dense GEMM fwd + dgrad: Linear / 1x1 conv over tokens (gemm_v3_kernel<0>, gemm_bf16_kernel<*,*,0>, gemm_finish_kernel)',
                   'attn': f'flash attention forward, self + cross ({ATTN_FWD}: attn_q_kernel<*,*,0,*,*>)',
                   'attn_bwd': f'flash attention backward ({ATTN_BWD}: attn_q_kernel<*,*,1,*,*> + attn_dkdv_kernel)',
                   'wgrad': 'dense weight gradient dW += dY^T A (wgrad_v2_kernel<0>, wgrad_v2s_kernel + wgrad_reduce_kernel)',
                   'conv_wgrad': 'conv3x3 weight gradient (wgrad_v2f_kernel / wgrad_v2wf_kernel, wgrad_v2_kernel<1>, wgrad_v2w_kernel<1> + wgrad_reduce_kernel)',
                   'gn': 'GroupNorm(32)+SiLU forward (gn_stats_kernel + gn_apply_kernel, one entry point)',
                   'gn_bwd': 'GroupNorm(32)+SiLU backward (gn_bwd_stats_kernel + gn_bwd_apply_kernel [+ colsum_reduce2_kernel])',
                   'ln': 'LayerNorm forward (ln_fwd_kernel)', 'ln_bwd': 'LayerNorm backward (ln_bwd_kernel [+ colsum_reduce2_kernel])'}

        def roof(key):
            r = timers[key]
            hbm = key in ('gn', 'gn_bwd', 'ln', 'ln_bwd')
            div, peak, unit = (1e9, PEAK_HBM_GBS, 'GB/s') if hbm else (1e12, PEAK_BF16_TFLOPS, 'TFLOP/s')
            ach = r['flops'] / (r['ms'] * 1e-3) / div if r['ms'] > 0 else 0.0
            o = {'bound': 'hbm' if hbm else 'mfma', 'kernel': KERNELS[key], 'achieved': ach, 'peak': peak, 'unit': unit, 'frac': ach / peak,
                 'timing': 'per-dispatch start/stop timestamps of every sampled call\'s kernels over the timed region (hipExtLaunchKernelGGL: the '
                           'interval rocprofv3 --kernel-trace reports; all kernels of a call, e.g. split-K GEMM + its finish kernel)',
                 # every call graded against ITS OWN bound: max(flop / MFMA peak, algorithmic bytes / HBM peak) summed over the sampled calls,
                 # over their measured time (a K = 320 GEMM is HBM-bound, a K = 5760 conv MFMA-bound) + the family's two plain fractions
                 'frac_of_per_call_roofline': r['bound_ms'] / r['ms'] if r['ms'] > 0 else 0.0,
                 'algorithmic_GBps': r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else 0.0,
                 'frac_hbm': r['bytes'] / (r['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS if r['ms'] > 0 else 0.0,
                 'traffic': None, 'launches': r['launches'], 'launches_total': r['calls'], 'kernels_timed': r['kernels'], 'avg_launch_ms': r['ms'] / max(r['launches'], 1),
                 # share of the step: sampled average duration x all launches of the timed region (streams overlap: the shares of all
                 # families add up to more than the wall time)
                 'est_ms_per_step': r['ms'] / max(r['launches'], 1) * r['calls'] / args.steps,
                 ('algorithmic_bytes_per_launch' if hbm else 'algorithmic_tflop_per_launch'): r['flops'] / max(r['launches'], 1) / (1 if hbm else 1e12)}
            q = iso.get(key)
            if q is not None and q['ms'] > 0:
                a2 = q['flops'] / (q['ms'] * 1e-3) / div
                o['isolated'] = {'achieved': a2, 'frac': a2 / peak, 'avg_launch_ms': q['ms'] / max(q['launches'], 1), 'launches': q['launches'],
                                 'what': '3 iterations after the timed region with the side-stream and weight-gradient-stream overlap off'}
            return o
        objs = {k: roof(k) for k in timers}
        # `roofline` = the family that takes the largest share of the step (dense GEMM since round 2's conv work; the MFMA
        # families are compared by est_ms_per_step); every family keeps its own object
        mfma = ['conv', 'gemm', 'attn_bwd', 'attn', 'wgrad', 'conv_wgrad']
        dominant = max(mfma, key=lambda k: objs[k]['est_ms_per_step'])
        out['roofline'] = dict(objs[dominant], family=dominant, why='largest est_ms_per_step of the MFMA kernel families in the timed region')
        for k in timers:
            out['roofline_' + k] = objs[k]
    if not args.no_cpu_baseline and world == 1:
        # torch's CPU backend degrades badly when oversubscribed on many-core hosts (256 threads: 160 s per forward);
        # 32 threads is the measured sweet spot class for one fp32 conv-heavy forward -> cores = threads actually used
        out['cpu_baseline'] = cpu_baseline(args.arch, min(os.cpu_count() or 1, 32), kappa=args.kappa)
    detail = os.environ.get('SIDLSG_BENCH_DETAIL', os.path.join(ROOT, 'bench_detail.json'))
    try:          # everything (every family's full roofline object, loss_check_detail, comm, ...) goes to a file ...
        with open(detail, 'w') as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        print(f'bench.py: could not write {detail}: {e}', file=sys.stderr)
    print(json.dumps(compact_line(out)), flush=True)          # ... and ONE line of < 4 KB to stdout (the driver keeps ~8 KB of tail)
    _shutdown(world)


def _shutdown(world):
    """All ranks leave together: the others wait here while rank 0 finishes its rank-local measurements (teacher pass)."""
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
