"""SD glue with the reference's names and signatures (training/sid_sd_util.py):
    load_sd15        :51-118   -> (unet, vae, noise_scheduler, text_encoder, tokenizer)
    sid_sd_sampler   :163-211  one-step (or few-step) generator  z -> x_hat (or decoded images)
    sid_sd_denoise   :214-274  add_noise -> (CFG-batched) UNet -> guided eps or x0 prediction

`unet` must be a HipUNet2DCondition (bare or wrapped in DistributedDataParallel, as the reference's loop passes it):
the whole glue runs as fused HIP kernels (sidlsg_noisy_input / UNet / sidlsg_cfg_x0: no per-sample python loop, no
host syncs, the CFG pair [uncond ; cond] shares one x_t).  There is no generic / CPU branch: any other network raises.
"""
import os
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import ops
from .scheduler import DDPMScheduler
from .text import TEXT_CONFIGS, CLIPBPETokenizer, CLIPTextModel, HashTokenizer
from .unet import CONFIGS, HipUNet2DCondition


def _unwrap(net):
    return net.module if hasattr(net, 'module') and isinstance(net.module, torch.nn.Module) else net


def _ddp_exchange(net, out):
    """INTEGRATION.md mode 2: the reference wraps fake_score / G in DistributedDataParallel (sid_training_loop.py:316-323)
    and hands the WRAPPER to sid_sd_sampler / sid_sd_denoise.  The HIP networks accumulate weight gradients in place in
    one flat buffer (autograd never sees per-parameter gradients), so DDP's bucket hooks cannot fire; the wrapper is
    honoured as a marker instead: when it is in sync mode (outside `no_sync()`, i.e. the last accumulation round of
    misc.ddp_sync, torch_utils/misc.py:168-175) the flat gradient buffer is all-reduced to the mean once this backward
    pass has finished -- the same result DDP's buckets produce, as one exchange."""
    inner = _unwrap(net)
    if inner is net or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return out
    world = torch.distributed.get_world_size()
    if world == 1 or not getattr(net, 'require_backward_grad_sync', True) or not inner._train_params:
        return out

    def exchange():
        # engine callbacks run in FIFO order and this one was queued by the FIRST node of the backward, i.e. before the
        # weight-gradient stream's own join callback (armed at the first wgrad launch): wait for that stream here
        g = inner.flat_grads
        ops.flush_deferred()          # queued dgamma / dbeta reductions first (their own end-of-backward callback may run after this one)
        if g.is_cuda:
            cur = torch.cuda.current_stream(g.device)
            for s in ops.grad_streams(g.device):
                cur.wait_stream(s)
        torch.distributed.all_reduce(g)
        g.div_(world)
    return ops.after_backward(out, exchange)


def _is_hip(net):
    return isinstance(_unwrap(net), HipUNet2DCondition)


def _require_hip(net):
    """The glue below drives the HIP kernels directly (NHWC bf16, fused scheduler arithmetic).  It has no generic /
    CPU branch: a foreign UNet belongs with the reference's own sid_sd_util (mode 2)."""
    if not _is_hip(net):
        raise TypeError(f'sid_lsg_amd.sd_util works on HipUNet2DCondition networks only, got {type(_unwrap(net)).__name__}')


def _arch_from_dir(path):
    """Architecture of a local diffusers-layout directory from its unet/config.json (block_out_channels,
    cross_attention_dim, use_linear_projection -- the fields that distinguish the supported configurations)."""
    import json
    cj = os.path.join(path, 'unet', 'config.json')
    if not os.path.isfile(cj):
        return None
    c = json.load(open(cj))
    for name, cfg in CONFIGS.items():
        if (tuple(c.get('block_out_channels', ())) == tuple(cfg.block_out_channels) and c.get('cross_attention_dim') == cfg.cross_attention_dim
                and bool(c.get('use_linear_projection', False)) == cfg.use_linear_projection):
            return name
    raise ValueError(f'{cj}: not one of the supported UNet configurations {sorted(CONFIGS)}')


def _arch_of(name):
    if os.path.isdir(name):
        arch = _arch_from_dir(name)
        if arch is not None:
            return arch
    n = os.path.basename(os.path.normpath(name)).lower() if os.path.isdir(name) else name.lower()
    if n.startswith('random:'):
        return n.split(':', 1)[1]
    if '2-1-base' in n or 'sd21' in n or 'stable-diffusion-2' in n:
        return 'sd21-base'
    return 'sd15'


def resolve_compute_dtype(compute_dtype=None):
    if compute_dtype is None:
        compute_dtype = os.environ.get('SIDLSG_COMPUTE_DTYPE', 'bf16')
    if isinstance(compute_dtype, str):
        try:
            compute_dtype = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'fp32': torch.float32, 'float32': torch.float32}[compute_dtype.lower()]
        except KeyError:
            raise ValueError(f'compute dtype {compute_dtype!r}: expected bf16 or fp32') from None
    return compute_dtype


def load_sd15(pretrained_model_name_or_path, pretrained_vae_model_name_or_path, device, weight_dtype, revision=None,
              variant=None, lora_config=None, enable_xformers=False, gradient_checkpointing=False, seed=0, compute_dtype=None):
    """Same contract as the reference factory.  Sources, in order:
      * a local diffusers-layout directory (unet/diffusion_pytorch_model.safetensors, text_encoder/model.safetensors,
        tokenizer/{vocab.json,merges.txt}) -> real weights through `load_state_dict` (key names are diffusers');
      * 'random:<arch>' (arch in sd15 | sd21-base | tiny | tiny40), or any hub id when SIDLSG_ALLOW_RANDOM_INIT=1 ->
        seeded random weights of that architecture (no network here: benchmarks and parity tests use this).
    `weight_dtype` is accepted for signature compatibility (the reference passes its fp16/fp32 switch here): masters are
    always fp32.  The COMPUTE dtype is `compute_dtype` (torch.bfloat16 = production MFMA path, torch.float32 = the
    fp32-accurate mode of csrc/fp32.hip), default from $SIDLSG_COMPUTE_DTYPE ('bf16' | 'fp32'), else bf16.
    `enable_xformers` / `gradient_checkpointing` are accepted and ignored (attention is always the fused HIP kernel;
    the reference itself never forwards gradient_checkpointing, sid_training_loop.py:224-228)."""
    name = str(pretrained_model_name_or_path)
    arch = _arch_of(name)
    device = torch.device(device)
    local = os.path.isdir(name)
    if not local and not name.lower().startswith('random:') and os.environ.get('SIDLSG_ALLOW_RANDOM_INIT', '0') != '1':
        raise FileNotFoundError(f'{name}: not a local diffusers directory and there is no network; pass a directory, '
                                f"'random:{arch}', or set SIDLSG_ALLOW_RANDOM_INIT=1")
    cfg = CONFIGS[arch]
    src = None
    if local:
        from safetensors.torch import load_file
        src = load_file(os.path.join(name, 'unet', 'diffusion_pytorch_model.safetensors'))
    unet = HipUNet2DCondition(cfg, compute_dtype=resolve_compute_dtype(compute_dtype))
    unet.materialize(device, seed=seed, source=src)
    tcfg = TEXT_CONFIGS.get(arch, dict(hidden=cfg.cross_attention_dim, layers=2, heads=2, dff=2 * cfg.cross_attention_dim,
                                       act='quick_gelu'))
    g = torch.random.get_rng_state()
    torch.manual_seed(seed + 1)
    text_encoder = CLIPTextModel(max_pos=cfg.text_len, **tcfg)
    torch.random.set_rng_state(g)
    tokenizer = HashTokenizer(model_max_length=cfg.text_len, pad_token_id=0 if arch == 'sd21-base' else 49407)
    if local:
        from safetensors.torch import load_file
        te = os.path.join(name, 'text_encoder', 'model.safetensors')
        if os.path.isfile(te):
            res = text_encoder.load_state_dict(load_file(te), strict=False)
            # transformers checkpoints may carry the (non-parameter) position-id buffer; anything else missing or
            # unexpected would silently leave seeded random weights in the conditioning of all three networks
            bad = [k for k in list(res.missing_keys) + list(res.unexpected_keys) if not k.endswith('position_ids')]
            if bad:
                raise RuntimeError(f'{te}: text-encoder checkpoint does not match the architecture: {bad[:8]}')
        vj, mt = os.path.join(name, 'tokenizer', 'vocab.json'), os.path.join(name, 'tokenizer', 'merges.txt')
        if os.path.isfile(vj) and os.path.isfile(mt):
            tokenizer = CLIPBPETokenizer.from_files(vj, mt, model_max_length=cfg.text_len, pad_token_id=tokenizer.pad_token_id)
    text_encoder.requires_grad_(False).eval().to(device)
    # VAE (decode only; cold path): real weights when the directory has them, seeded random ones otherwise
    vae_dir = pretrained_vae_model_name_or_path if (pretrained_vae_model_name_or_path and os.path.isdir(str(pretrained_vae_model_name_or_path))) else name
    vae_file = os.path.join(str(vae_dir), 'vae', 'diffusion_pytorch_model.safetensors')
    from .vae import HipAutoencoderKLDecoder
    vae = HipAutoencoderKLDecoder('sd' if arch in ('sd15', 'sd21-base') else 'tiny')
    if local and os.path.isfile(vae_file):
        from safetensors.torch import load_file
        vae.load_state_dict(load_file(vae_file))
    else:
        vae.init_parameters(seed + 2)
    vae = vae.to(device)
    return unet, vae, DDPMScheduler().to(device), text_encoder, tokenizer


def encode_contexts(contexts, text_encoder, tokenizer, device):
    """list[str] -> [B, L, D] text states (no grad); a tensor is passed through (pre-computed states)."""
    if torch.is_tensor(contexts):
        return contexts
    ids = tokenizer(list(contexts), padding='max_length', max_length=tokenizer.model_max_length, truncation=True,
                    return_tensors='pt').input_ids
    with torch.no_grad():
        return text_encoder(ids.to(device))[0]


# ------------------------------------------------------------------------------------------------
def hip_generate(unet, z, ctx16, init_t, sched, x0=None):
    """x_t = s0*x0 + s1*z at t_init ; eps = G(x_t) ; x_hat = (x_t - s1*eps)/s0   (sid_sd_util.py:182-185)"""
    s0, s1 = sched.coefficients(init_t)
    net = _unwrap(unet)
    xin, xt = ops.noisy_input(x0, z, s0, s1, 1, net.compute_dtype)
    eps = _ddp_exchange(unet, net.forward_nhwc(xin, init_t, ctx16))
    return ops.cfg_x0(eps, xt, s0, s1, 1.0, True, net.compute_dtype)


def hip_prepare_denoise(images, noise, t, cond16, uncond16, sched, guided, act_dtype=torch.bfloat16):
    """Shared by every network evaluated on the same (images, noise, t): the noisy CFG batch and its conditioning
    (`act_dtype` = the compute dtype of the networks that will consume it)."""
    s0, s1 = sched.coefficients(t)
    dup = 2 if guided else 1
    xin, xt = ops.noisy_input(images, noise, s0, s1, dup, act_dtype)
    ctx = torch.cat([uncond16, cond16]) if guided else cond16          # (sid_sd_util.py:259-261)
    tt = torch.cat([t, t]) if guided else t
    return SimpleNamespace(xin=xin, xt=xt, s0=s0, s1=s1, ctx=ctx, tt=tt)


def hip_denoise(unet, prep, guidance_scale, predict_x0):
    net = _unwrap(unet)
    eps = _ddp_exchange(unet, net.forward_nhwc(prep.xin, prep.tt, prep.ctx))
    return ops.cfg_x0(eps, prep.xt, prep.s0, prep.s1, guidance_scale, predict_x0, net.compute_dtype)  # u + k(c-u), then x0 (:264-272)


# ------------------------------------------------------------------------------------------------
def sid_sd_sampler(unet, latents, contexts, init_timesteps, noise_scheduler, text_encoder, tokenizer, resolution,
                   dtype=torch.float16, return_images=False, vae=None, guidance_scale=1, num_steps=1, train_sampler=True,
                   num_steps_eval=1):
    steps = num_steps if train_sampler else num_steps_eval
    _require_hip(unet)
    emb = encode_contexts(contexts, text_encoder, tokenizer, latents.device).to(_unwrap(unet).compute_dtype).contiguous()
    D_x = None
    ctxmgr = torch.enable_grad() if train_sampler else torch.no_grad()
    with ctxmgr:
        for i in range(steps):
            noise = latents if i == 0 else torch.randn_like(latents)
            t_i = (init_timesteps * (1 - i / steps)).to(torch.long)
            D_x = hip_generate(unet, noise.to(torch.float32).contiguous(), emb, t_i.contiguous(), noise_scheduler, x0=D_x)
    if not return_images:
        return D_x.to(torch.float32)
    upcast = vae.dtype == torch.float16 and getattr(vae.config, 'force_upcast', False)
    if upcast:
        vae.to(dtype=torch.float32)
    images = vae.decode(D_x.to(vae.dtype) / vae.config.scaling_factor, return_dict=False)[0]
    if upcast:
        vae.to(dtype=torch.float16)
    return images.to(torch.float32)


def sid_sd_denoise(unet, images, noise, contexts, timesteps, noise_scheduler, text_encoder, tokenizer, resolution,
                   dtype=torch.float16, predict_x0=True, guidance_scale=1):
    _require_hip(unet)
    b = images.shape[0]
    cond = encode_contexts(contexts, text_encoder, tokenizer, images.device)
    guided = guidance_scale != 1
    uncond = encode_contexts([''] * b, text_encoder, tokenizer, images.device) if guided else None
    bf = _unwrap(unet).compute_dtype
    prep = hip_prepare_denoise(images.to(torch.float32).contiguous(), noise.to(torch.float32).contiguous(),
                               timesteps.contiguous(), cond.to(bf).contiguous(),
                               uncond.to(bf).contiguous() if guided else None, noise_scheduler, guided, act_dtype=bf)
    return hip_denoise(unet, prep, float(guidance_scale), predict_x0)
