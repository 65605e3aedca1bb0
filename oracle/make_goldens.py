"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference).

Run in the authoring container only:  python -m oracle.make_goldens
The reference itself never travels; only these small input/output vectors do.
Every file records the weight checksum of the seeded random nets it was made with.

  glue_*.npz   reference sid_sd_sampler / sid_sd_denoise (training/sid_sd_util.py:163-274)
               on duck-typed oracle nets: pins rows A3, A4, A6 of SURVEY.md section 8.
  loop_*.npz   the UNMODIFIED reference training_loop (training/sid_training_loop.py:148-672)
               run for a few iterations on CPU: per-step loss values and final parameter
               checksums: pins A1, A2, A7-A10, A13 (RNG order, accumulation, Adam, EMA).
  networks_blocks.npz  the reference's own in-tree network primitives (training/networks.py: GroupNorm :96, AttentionOp
               :113, UNetBlock :134, Conv2d :47, Linear :30) run forward + backward on seeded inputs: the only
               reference-held arithmetic for GroupNorm+SiLU+conv3x3+time-embedding-add+1x1-shortcut (= the SD
               ResnetBlock2D when adaptive_scale=False) and for softmax(QK^T/sqrt(d))V attention:
               pins those pieces of row A5 for both the oracle UNet layers and the HIP ops.
  loop_k15_a1_n50.npz  the same loop for 50 iterations (loss-curve drift check of the bf16 path).
  metrics_ref.npz  the reference's evaluation path (metrics/sid_metric_utils.py: generator feature statistics, PIL LANCZOS resize,
               InfiniteSampler prompt order, FeatureStats) + the Frechet distance formula on stand-in generator / detector objects.
  bias_act.npz reference torch_utils/ops/bias_act.py::_bias_act_ref (+ autograd grads).
  sampler.npz  reference torch_utils/misc.py::InfiniteSampler order.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import fixtures, ref_harness  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
PROMPTS = [
    'a photo of a cat sitting on a sofa', 'an astronaut riding a horse', 'a bowl of fruit on a wooden table',
    'mountains at sunset, oil painting', 'a red bicycle leaning on a wall', 'portrait of an old fisherman',
    'a city street in the rain at night', 'two dogs playing in the snow', 'a cup of coffee and a book',
    'a lighthouse on a cliff', 'sailing boats in a harbor', 'a field of sunflowers under blue sky',
    'a steam train crossing a bridge', 'a child flying a kite', 'close-up of a dragonfly', 'a castle in the clouds',
    'a bowl of ramen', 'a robot painting a picture', 'a forest path in autumn', 'a vintage car on route 66',
]


def gen_glue():
    ref = ref_harness.import_reference()
    for cfg_name, lat in (('tiny', 8), ('tiny40', 8)):
        unet, vae, sched, te, tok = fixtures.factory(cfg_name)
        unet2 = fixtures.make_unet(cfg_name, seed=99)
        unet.eval().requires_grad_(False)
        out = dict(cfg=cfg_name, weight_checksum=np.array(fixtures.checksum(unet)),
                   weight_checksum2=np.array(fixtures.checksum(unet2)))
        g = torch.Generator().manual_seed(7)
        case = 0
        for b in (1, 2):
            prompts = PROMPTS[case:case + b]
            z = torch.randn(b, 4, lat, lat, generator=g)
            noise = torch.randn(b, 4, lat, lat, generator=g)
            t = torch.randint(20, 980, (b,), generator=g)
            init_t = torch.full((b,), 625, dtype=torch.long)
            xhat = ref.sd_util.sid_sd_sampler(unet=unet, latents=z, contexts=prompts, init_timesteps=init_t,
                                              noise_scheduler=sched, text_encoder=te, tokenizer=tok, resolution=lat * 8,
                                              dtype=torch.float32, return_images=False, vae=None, num_steps=1)
            out[f'b{b}_z'], out[f'b{b}_noise'], out[f'b{b}_t'] = z.numpy(), noise.numpy(), t.numpy()
            out[f'b{b}_prompts'] = np.array(prompts)
            out[f'b{b}_xhat'] = xhat.numpy()
            for kappa in (1.0, 1.5, 4.5):
                for px0 in (True, False):
                    y = ref.sd_util.sid_sd_denoise(unet=unet2, images=xhat, noise=noise, contexts=prompts, timesteps=t,
                                                   noise_scheduler=sched, text_encoder=te, tokenizer=tok,
                                                   resolution=lat * 8, dtype=torch.float32, predict_x0=px0,
                                                   guidance_scale=kappa)
                    out[f'b{b}_k{kappa}_x0{int(px0)}'] = y.detach().numpy()
            case += b
        np.savez_compressed(os.path.join(OUT, f'glue_{cfg_name}.npz'), **out)
        print('glue', cfg_name, 'done')


def _run_loop(name, **kw):
    with tempfile.TemporaryDirectory() as tmp:
        pdir = os.path.join(tmp, 'prompts')
        os.makedirs(pdir)
        with open(os.path.join(pdir, 'aesthetics_6_plus.txt'), 'wt') as f:
            f.write('\n'.join(PROMPTS) + '\n')
        run_dir = os.path.join(tmp, 'run')
        os.makedirs(run_dir)
        cfg_name = kw.pop('cfg_name', 'tiny')
        with ref_harness.cpu_process_group():
            res = ref_harness.run_reference_training_loop(lambda: fixtures.factory(cfg_name), pdir, run_dir, **kw)
    out = dict(cfg=cfg_name, prompts=np.array(PROMPTS),
               loss_names=np.array([n for n, _ in res['losses']]),
               loss_values=np.array([v for _, v in res['losses']], dtype=np.float64),
               weight_checksum=np.array(fixtures.checksum(fixtures.make_unet(cfg_name))),
               fake_score_checksum=np.array(fixtures.checksum(res['fake_score_params'])),
               G_checksum=np.array(fixtures.checksum(res['G_params'])),
               # a few raw parameter values for a sharper pin than the checksums
               G_conv_in_w=res['G_params'][0].numpy(), fake_conv_in_w=res['fake_score_params'][0].numpy(),
               G_last_b=res['G_params'][-1].numpy(), fake_last_b=res['fake_score_params'][-1].numpy())
    for k, v in kw.items():
        out['kw_' + k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, f'loop_{name}.npz'), **out)
    print('loop', name, [f'{v:.6g}' for v in out['loss_values']])


def gen_loops():
    # kappa=1.5 everywhere (BASELINE config #1 setting), 2 accumulation rounds, alpha=1
    _run_loop('k15_a1', iterations=4, batch_size=4, batch_gpu=2, seed=3, alpha=1.0, kappa=(1.5, 1.5, 1.5),
              lr=1e-4, glr=1e-4, resolution=128)
    # no guidance (kappa=1: single-branch path, no prompt dropout), alpha=1.2 general branch, default lrs
    _run_loop('k1_a12', iterations=3, batch_size=2, batch_gpu=2, seed=5, alpha=1.2, kappa=(1.0, 1.0, 1.0),
              lr=1e-5, glr=1e-5, resolution=128)
    # kappa 4.5 (BASELINE config #3 guidance), batch_gpu 1
    _run_loop('k45_a1', iterations=3, batch_size=2, batch_gpu=1, seed=11, alpha=1.0, kappa=(4.5, 4.5, 4.5),
              lr=1e-4, glr=1e-4, resolution=128)


def gen_loop_long():
    """tests/golden/loop_k15_a1_n50.npz: FIFTY iterations of the unmodified reference loop (kappa = 1.5, alpha = 1): the loss
    CURVE over a stretch long enough for rounding drift of a reduced-precision path to show (the networks move ~50 Adam steps
    apart from their common start, so the generator loss leaves its near-zero start)."""
    _run_loop('k15_a1_n50', iterations=50, batch_size=2, batch_gpu=2, seed=21, alpha=1.0, kappa=(1.5, 1.5, 1.5),
              lr=1e-4, glr=1e-4, resolution=128)


LOOP2_KW = dict(iterations=3, batch_size=4, batch_gpu=1, seed=7, alpha=1.0, kappa=(1.5, 1.5, 1.5), lr=1e-4, glr=1e-4, resolution=128)


def _loop2_worker(prefix):
    """One rank of the 2-rank reference run (launched by torch.distributed.run): the UNMODIFIED reference training_loop with
    world_size 2 on gloo -- DistributedDataParallel gradient averaging, `misc.ddp_sync` no_sync rounds (2 accumulation
    rounds per rank), rank-strided prompt stream, per-rank seeds (sid_training_loop.py:238, 274, 316-323, 416, 487)."""
    torch.set_num_threads(4)
    torch.distributed.init_process_group('gloo')
    rank = torch.distributed.get_rank()
    with tempfile.TemporaryDirectory() as tmp:
        pdir = os.path.join(tmp, 'prompts')
        os.makedirs(pdir)
        with open(os.path.join(pdir, 'aesthetics_6_plus.txt'), 'wt') as f:
            f.write('\n'.join(PROMPTS) + '\n')
        run_dir = os.path.join(tmp, 'run')
        os.makedirs(run_dir)
        res = ref_harness.run_reference_training_loop(lambda: fixtures.factory('tiny'), pdir, run_dir, **LOOP2_KW)
    np.savez(f'{prefix}.rank{rank}.npz', loss_names=np.array([n for n, _ in res['losses']]),
             loss_values=np.array([v for _, v in res['losses']], dtype=np.float64),
             G_conv_in_w=res['G_params'][0].numpy(), fake_conv_in_w=res['fake_score_params'][0].numpy(),
             G_last_b=res['G_params'][-1].numpy(), G_checksum=np.array(fixtures.checksum(res['G_params'])),
             fake_score_checksum=np.array(fixtures.checksum(res['fake_score_params'])))
    torch.distributed.destroy_process_group()


def gen_loop_2rank():
    """tests/golden/loop2_k15_a1.npz: per-rank loss curves + final weights of a 2-rank run of the reference loop."""
    import subprocess
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, 'loop2')
        env = dict(os.environ, PYTHONPATH=ROOT)
        subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                               '--master-port', '29547', os.path.abspath(__file__), '_loop2_worker', prefix], env=env, cwd=ROOT)
        r0, r1 = np.load(f'{prefix}.rank0.npz'), np.load(f'{prefix}.rank1.npz')
        assert np.array_equal(r0['G_conv_in_w'], r1['G_conv_in_w']), 'ranks must end with identical weights'
        out = dict(cfg='tiny', prompts=np.array(PROMPTS), loss_names=r0['loss_names'], loss_values_rank0=r0['loss_values'],
                   loss_values_rank1=r1['loss_values'], G_conv_in_w=r0['G_conv_in_w'], fake_conv_in_w=r0['fake_conv_in_w'],
                   G_last_b=r0['G_last_b'], G_checksum=r0['G_checksum'], fake_score_checksum=r0['fake_score_checksum'],
                   weight_checksum=np.array(fixtures.checksum(fixtures.make_unet('tiny'))))
        for k, v in LOOP2_KW.items():
            out['kw_' + k] = np.array(v)
        np.savez_compressed(os.path.join(OUT, 'loop2_k15_a1.npz'), **out)
        print('loop2', out['loss_values_rank0'], out['loss_values_rank1'])


def gen_bias_act():
    ref = ref_harness.import_reference()
    g = torch.Generator().manual_seed(3)
    out = {}
    x = torch.randn(2, 40, 3, 5, generator=g)
    b = torch.randn(40, generator=g)
    dy = torch.randn(2, 40, 3, 5, generator=g)
    seed2 = torch.randn(2, 40, 3, 5, generator=g)
    out['x'], out['b'], out['dy'], out['seed2'] = x.numpy(), b.numpy(), dy.numpy(), seed2.numpy()
    for act in ref.bias_act.activation_funcs.keys():
        for gain, clamp in ((None, None), (1.0, None), (1.5, 0.7)):
            xx = x.clone().requires_grad_(True)
            bb = b.clone().requires_grad_(True)
            y = ref.bias_act._bias_act_ref(xx, bb, dim=1, act=act, gain=gain, clamp=clamp)
            dyy = dy.clone().requires_grad_(True)
            dx, db = torch.autograd.grad(y, [xx, bb], dyy, create_graph=True)
            key = f'{act}_g{gain}_c{clamp}'
            out[key + '_y'], out[key + '_dx'], out[key + '_db'] = y.detach().numpy(), dx.detach().numpy(), db.detach().numpy()
            # second order (reference plugin grad=2, bias_act.py:190-212): seed on dx -> gradients w.r.t. x and dy
            d2x, d2dy = torch.autograd.grad(dx, [xx, dyy], seed2, allow_unused=True)
            out[key + '_d2x'] = (d2x if d2x is not None else torch.zeros_like(x)).numpy()
            out[key + '_d2dy'] = d2dy.numpy()
    np.savez_compressed(os.path.join(OUT, 'bias_act.npz'), **out)
    print('bias_act done')


def gen_blocks():
    """training/networks.py::UNetBlock(adaptive_scale=False) forward/backward with every parameter seeded non-trivially
    (the reference zero-initialises conv1 / proj).  Cases: channel change + 1x1 skip conv; identity skip; + attention with
    head dim 40 (the SD1.5 64x64-stage head dim) and with head dim 64 (SD2.1)."""
    ref_harness.import_reference()
    import training.networks as N
    out = {}
    for tag, cin, cout, emb, attn, heads, hw in (('res_proj', 32, 64, 48, False, None, 12), ('res_id', 64, 64, 48, False, None, 8),
                                                 ('attn40', 32, 80, 48, True, 2, 12), ('attn64', 32, 64, 32, True, 1, 10)):
        g = torch.Generator().manual_seed({'res_proj': 1, 'res_id': 2, 'attn40': 3, 'attn64': 4}[tag])
        blk = N.UNetBlock(in_channels=cin, out_channels=cout, emb_channels=emb, attention=attn, num_heads=heads, adaptive_scale=False)
        with torch.no_grad():
            for n, p in blk.named_parameters():
                if n.startswith('norm'):
                    p.copy_((1.0 if n.endswith('weight') else 0.0) + 0.3 * torch.randn(p.shape, generator=g))
                elif n == 'qkv.bias':
                    p.zero_()                                   # SD's to_q / to_k / to_v have no bias
                elif n.endswith('bias'):
                    p.copy_(0.2 * torch.randn(p.shape, generator=g))
                else:
                    fan_in = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) * fan_in ** -0.5)
        x = torch.randn(2, cin, hw, hw, generator=g).requires_grad_()
        # the SD ResnetBlock2D feeds silu(temb) to its projection; UNetBlock takes the projection input directly
        raw = torch.randn(2, emb, generator=g)
        out[f'{tag}_emb_raw'] = raw.numpy()
        e = torch.nn.functional.silu(raw).requires_grad_()
        y = blk(x, e)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out[f'{tag}_x'], out[f'{tag}_emb'], out[f'{tag}_y'], out[f'{tag}_dy'] = x.detach().numpy(), e.detach().numpy(), y.detach().numpy(), dy.numpy()
        out[f'{tag}_dx'], out[f'{tag}_demb'] = x.grad.numpy(), e.grad.numpy()
        out[f'{tag}_groups'] = np.array([blk.norm0.num_groups, blk.norm1.num_groups, blk.norm2.num_groups if attn else 0, heads or 0])
        for n, p in blk.named_parameters():
            out[f'{tag}_p_{n}'] = p.detach().numpy()
            out[f'{tag}_g_{n}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'networks_blocks.npz'), **out)
    print('blocks done', sorted(k for k in out if k.endswith('_y')))


def gen_metrics():
    """tests/golden/metrics_ref.npz: the reference's OWN evaluation code path (metrics/sid_metric_utils.py, loaded from its file:
    `compute_feature_stats_for_generator` :412-510 with `FeatureStats` :112-188, `resize_images_in_tensor` :353-375 and the
    InfiniteSampler prompt order) run on stand-in generator / detector objects, plus the Frechet distance with the three lines of
    metrics/sid_fid_and_clip.py:65-67 (restated here: they sit inside a function that needs the COCO images).
    Absent packages are stubbed where the module imports them: networks.clip (open_clip / timm), torchvision.transforms[.functional]
    -- `to_pil_image` / `to_tensor` are restated from torchvision's documented semantics for uint8 CHW tensors / RGB images."""
    import importlib.util
    import types

    import scipy.linalg
    from PIL import Image
    ref_harness.import_reference()                          # dnnlib, torch_utils on sys.path; Sampler drift patch
    ref_harness._stub('networks')
    ref_harness._stub('networks.clip', CLIP=ref_harness._Missing)
    tv = ref_harness._stub('torchvision')
    tvt = ref_harness._stub('torchvision.transforms')
    tv.transforms = tvt

    def to_pil_image(t):                                    # uint8 [C, H, W] -> PIL RGB
        return Image.fromarray(t.permute(1, 2, 0).cpu().numpy())

    def to_tensor(img):                                     # PIL RGB -> float32 [C, H, W] in [0, 1]
        return torch.from_numpy(np.array(img)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    tvt.functional = ref_harness._stub('torchvision.transforms.functional', to_pil_image=to_pil_image, to_tensor=to_tensor)
    spec = importlib.util.spec_from_file_location('ref_sid_metric_utils', os.path.join(ref_harness.REFERENCE_ROOT, 'metrics', 'sid_metric_utils.py'))
    mu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mu)
    import dnnlib
    R, N = 512, 12
    det = fixtures.StandInDetector()
    seen = dict(contexts=[], resized=[])

    def G(latents, contexts, init_timesteps):
        assert latents.shape[1:] == (4, R // 8, R // 8) and len(contexts) == latents.shape[0] and int(init_timesteps[0]) == 625
        seen['contexts'].extend(str(c) for c in contexts)
        return fixtures.text_images(contexts, R)

    def detector(images, **kw):
        assert kw == dict(return_features=True) and images.dtype == torch.uint8 and images.shape[1:] == (3, 256, 256)
        seen['resized'].append(images.clone())
        return det(images)
    mu.get_feature_detector = lambda **kw: detector
    opts = types.SimpleNamespace(G=G, init_timestep=625, rank=0, num_gpus=1, device=torch.device('cpu'), progress=mu.ProgressMonitor(),
                                 dataset_kwargs=dnnlib.EasyDict(class_name='oracle.fixtures.CaptionSet', resolution=R))
    torch.manual_seed(0)
    stats = mu.compute_feature_stats_for_generator(opts=opts, detector_url='stand-in', detector_kwargs=dict(return_features=True),
                                                   batch_size=8, batch_gen=4, capture_mean_cov=True, max_items=N)
    mu_gen, sigma_gen = stats.get_mean_cov()
    F_ = mu_gen.shape[0]
    rng = np.random.RandomState(5)
    mu_real = rng.normal(size=F_) * 0.05
    a = rng.normal(size=(F_, F_)) * 0.05
    sigma_real = a @ a.T + 0.01 * np.eye(F_)
    m = np.square(mu_gen - mu_real).sum()                                      # metrics/sid_fid_and_clip.py:65
    sq, _ = scipy.linalg.sqrtm(np.dot(sigma_gen, sigma_real), disp=False)      # :66
    fid = float(np.real(m + np.trace(sigma_gen + sigma_real - sq * 2)))        # :67
    resized = torch.cat(seen['resized'])
    np.savez_compressed(os.path.join(OUT, 'metrics_ref.npz'), captions=np.array(fixtures.METRIC_CAPTIONS), resolution=R, num_items=N,
                        contexts=np.array(seen['contexts'][:N]), resized_first=resized[0].numpy(),
                        resized_sums=resized[:N].to(torch.float64).sum(dim=(2, 3)).numpy(),
                        features_all=torch.cat([det(r[None]) for r in resized[:N]]).numpy(),
                        mu_gen=mu_gen, sigma_gen=sigma_gen, mu_real=mu_real, sigma_real=sigma_real, fid=fid, stats_num_items=stats.num_items)
    print('metrics: fid', fid, 'num_items', stats.num_items, 'contexts', seen['contexts'][:4])


def gen_sampler():
    ref = ref_harness.import_reference()
    out = {}
    for n, rank, world, seed in ((20, 0, 1, 0), (20, 1, 2, 3), (7, 2, 4, 9)):
        it = iter(ref.misc.InfiniteSampler(dataset=list(range(n)), rank=rank, num_replicas=world, seed=seed))
        out[f'n{n}_r{rank}_w{world}_s{seed}'] = np.array([int(next(it)) for _ in range(64)])
    np.savez_compressed(os.path.join(OUT, 'sampler.npz'), **out)
    print('sampler done')


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 2 and sys.argv[1] == '_loop2_worker':
        _loop2_worker(sys.argv[2])
        sys.exit(0)
    which = sys.argv[1:] or ['glue', 'loops', 'bias_act', 'sampler', 'blocks', 'loop2', 'loop_long', 'metrics']
    if 'loop2' in which:
        gen_loop_2rank()
    if 'blocks' in which:
        gen_blocks()
    if 'glue' in which:
        gen_glue()
    if 'bias_act' in which:
        gen_bias_act()
    if 'sampler' in which:
        gen_sampler()
    if 'loops' in which:
        gen_loops()
    if 'loop_long' in which:
        gen_loop_long()
    if 'metrics' in which:
        gen_metrics()
