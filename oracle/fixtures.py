"""Shared deterministic test objects (seeded random weights).  TEST INFRASTRUCTURE ONLY.

No SD weights, tokenizer vocabulary or diffusers exist offline, so every parity test uses
seeded random-init networks of the reference architecture.  Construction is deterministic
for a given torch build (CPU mt19937 + default nn init), and each golden file stores a
float64 checksum of the weights it was generated with so drift is detected, not ignored.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .scheduler_ref import DDPMSchedulerRef
from .text_ref import CLIPTextRef, HashTokenizerRef
from .unet_ref import CONFIGS, UNet2DConditionRef


class VAEStandIn(torch.nn.Module):
    """Cold-path stand-in: the loop only needs config + decode() for preview PNGs
    (training/sid_training_loop.py:254, 357-363; training/sid_sd_util.py:198-209)."""

    def __init__(self):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=[128, 256, 512, 512], scaling_factor=0.18215, force_upcast=True)
        self.post_quant_conv = torch.nn.Conv2d(4, 4, 1)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def decode(self, z, return_dict=False):
        img = F.interpolate(z[:, :3], scale_factor=8.0, mode='nearest').clamp(-1, 1)
        return (img,)


def text_stack(cfg_name, seed=4321):
    cfg = CONFIGS[cfg_name]
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    if cfg_name in ('sd15',):
        te = CLIPTextRef(768, 12, 12, 3072, max_pos=77)
    elif cfg_name == 'sd21-base':
        te = CLIPTextRef(1024, 23, 16, 4096, max_pos=77, act='gelu')
    else:
        te = CLIPTextRef(cfg.cross_attention_dim, 2, 2, 2 * cfg.cross_attention_dim, max_pos=cfg.text_len)
    torch.random.set_rng_state(g)
    tok = HashTokenizerRef(model_max_length=cfg.text_len, pad_token_id=49407 if cfg_name != 'sd21-base' else 0)
    return te.eval().requires_grad_(False), tok


def make_unet(cfg_name, seed=1234, perturb=0.0):
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    unet = UNet2DConditionRef(CONFIGS[cfg_name])
    if perturb:
        with torch.no_grad():
            for p in unet.parameters():
                p.add_(perturb * torch.randn_like(p))
    torch.random.set_rng_state(g)
    return unet


_UNET_WEIGHTS = {}      # (cfg_name, seed) -> freshly initialised state dict (never handed out: callers get clones)


def make_unet_cached(cfg_name, seed=1234):
    """make_unet(cfg_name, seed), bit for bit, without paying the initialisation twice: seeding + default nn init of the full-size
    networks is ~20 s of serial random-number generation per call, and the full-size parity tests of the GPU suite asked for the
    same four (architecture, seed) pairs 24 times.  The first call builds the network the ordinary way and keeps a private copy
    of its weights; later calls construct the module on the meta device (no init) and adopt clones of that copy."""
    key = (cfg_name, seed)
    if key not in _UNET_WEIGHTS:
        unet = make_unet(cfg_name, seed)
        _UNET_WEIGHTS[key] = {k: v.detach().clone() for k, v in unet.state_dict().items()}
        return unet
    g = torch.random.get_rng_state()
    with torch.device('meta'):
        unet = UNet2DConditionRef(CONFIGS[cfg_name])
    torch.random.set_rng_state(g)
    unet.load_state_dict({k: v.clone() for k, v in _UNET_WEIGHTS[key].items()}, assign=True)
    assert not any(p.is_meta for p in unet.parameters()) and not any(b.is_meta for b in unet.buffers())
    return unet


def factory(cfg_name='tiny', seed=1234):
    """Same 5-tuple as the reference's load_sd15 (training/sid_sd_util.py:118)."""
    te, tok = text_stack(cfg_name)
    return make_unet(cfg_name, seed), VAEStandIn(), DDPMSchedulerRef(), te, tok


def checksum(module_or_tensors):
    ts = list(module_or_tensors.parameters() if isinstance(module_or_tensors, torch.nn.Module) else module_or_tensors)
    return float(sum(p.detach().double().sum() for p in ts)), float(sum(p.detach().double().abs().sum() for p in ts))


# ---- evaluation-metric fixtures (tests/golden/metrics_ref.npz) ---------------------------------------
METRIC_CAPTIONS = [f'evaluation caption number {i}: {w}' for i, w in enumerate(
    'a red barn in the snow|two cats on a sofa|a bowl of oranges|city lights at dusk|a sailing boat|an old bicycle|a mountain lake|'
    'a plate of pasta|a child with a kite|a dog on the beach|a steam engine|tulips in a vase|a desert road|a wooden bridge|'
    'a tea cup and a book|a lighthouse at night|a hot air balloon'.split('|'))]


class CaptionSet(torch.utils.data.Dataset):
    """Item contract of the reference's metric dataset (training/mscoco_dataset.py: `(image, caption)`, attribute `resolution`)."""

    def __init__(self, resolution=512, **_):
        self.resolution, self.name = resolution, 'captions'

    def __len__(self):
        return len(METRIC_CAPTIONS)

    def __getitem__(self, i):
        return torch.zeros(1, 4, 4), METRIC_CAPTIONS[i]


def text_images(contexts, resolution):
    """Stand-in generator output: an image in ~[-1.2, 1.2] that is a pure function of the caption (so a replay does not depend on
    which RNG drew the latents): smooth low-frequency content + noise, [N, 3, R, R] fp32 on the CPU."""
    import zlib
    out = []
    for c in contexts:
        g = torch.Generator().manual_seed(zlib.crc32(str(c).encode()))
        low = F.interpolate(torch.randn(1, 3, 16, 16, generator=g), size=(resolution, resolution), mode='bilinear', align_corners=False)[0]
        out.append((0.8 * low + 0.25 * torch.randn(3, resolution, resolution, generator=g)).clamp(-1.2, 1.2))
    return torch.stack(out)


class StandInDetector(torch.nn.Module):
    """Stand-in for the TorchScript Inception (`detector(uint8 NCHW images, return_features=True) -> [N, F]`): a fixed seeded
    conv + pooling; fp32, so CPU and GPU agree to rounding."""

    def __init__(self, features=24, seed=99):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer('w', torch.randn(features, 3, 8, 8, generator=g) * 0.05)
        self.register_buffer('mix', torch.randn(features, features, generator=g) * 0.3)

    def forward(self, img, return_features=True):
        x = F.conv2d(img.to(torch.float32) / 255.0 - 0.5, self.w, stride=8)
        x = torch.tanh(x).mean(dim=(2, 3))
        return x @ self.mix


# ---- one SiD iteration on seeded inputs: shared by tests/test_gpu_unet.py (live oracle) and oracle/make_fullsize_fixtures.py (stored oracle) ----
def iteration_inputs(cfg_name, lat, b, rounds, gen):
    """The (z, noise, t, cond, uncond) draws of one iteration -- phase A rounds, then phase B rounds -- from `gen` (a CPU generator)."""
    cfg = CONFIGS[cfg_name]
    bf = torch.bfloat16
    inputs = dict(A=[], B=[])
    for ph in ('A', 'B'):
        for _ in range(rounds):
            inputs[ph].append(dict(z=torch.randn(b, 4, lat, lat, generator=gen), noise=torch.randn(b, 4, lat, lat, generator=gen),
                                   t=torch.randint(20, 980, (b,), generator=gen),
                                   cond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=gen).to(bf).float(),
                                   uncond=torch.randn(1, cfg.text_len, cfg.cross_attention_dim, generator=gen).to(bf).float().expand(b, -1, -1).contiguous()))
    return inputs


def iteration_hp(b, rounds, lr, kappa, alpha):
    return dict(alpha=alpha, kappa1=kappa, kappa2=kappa, kappa4=kappa, ls=1.0, lsg=1.0, batch_gpu_total=b * rounds, lr=lr, glr=lr,
                betas=(0.0, 0.999), eps=1e-8, init_t=625, batch_size=b * rounds, ema_halflife_kimg=50, ema_rampup_ratio=0.05)


SAMPLE_STRIDE = 431      # the stored full-size oracle iterations keep every 431st entry of each parameter tensor (all of a small one)


def sample_index(numel):
    return torch.arange(0, numel, SAMPLE_STRIDE) if numel > 10 * SAMPLE_STRIDE else torch.arange(numel)


# (name, architecture, latent size, batch, kappa): the full-size iterations whose oracle result is STORED (tests/golden/fullsize_*.npz)
FULLSIZE_CASES = {
    'sd15_k45_b1': ('sd15', 64, 1, 4.5),
    'sd21_k2_b1': ('sd21-base', 64, 1, 2.0),
    'sd21_k2_768': ('sd21-base', 96, 1, 2.0),
    'sd15_k15_b2': ('sd15', 64, 2, 1.5),
    'sd21_k15_b1': ('sd21-base', 64, 1, 1.5),
}
FULLSIZE_LR, FULLSIZE_SEED, FULLSIZE_EMA_NAMES = 1e-6, 5, ('conv_in.weight', 'conv_out.bias')
