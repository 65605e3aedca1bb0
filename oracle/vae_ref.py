"""CPU restatement of the AutoencoderKL *decoder* of Stable Diffusion (the `vae.decode` the reference calls at
training/sid_sd_util.py:198-209 and generate_onestep.py through sid_sd_sampler(return_images=True)).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity unpinned: the arithmetic lives in diffusers==0.27.2
(`AutoencoderKL`, third party, absent here); this follows its published architecture for SD1.x/2.x [from memory]:
  post_quant_conv 1x1 (4->4) -> conv_in 3x3 (4->512) -> mid: ResBlock, single-head attention (GroupNorm 32 eps 1e-6,
  q/k/v/out Linear 512), ResBlock -> 4 up blocks with (512, 512, 256, 128) channels, 3 ResBlocks each, nearest x2 + conv3x3
  after the first three -> GroupNorm(32, eps 1e-6) -> SiLU -> conv_out 3x3 (128->3).
ResBlock (no time embedding) = GN32(eps 1e-6) -> SiLU -> conv3x3 -> GN32 -> SiLU -> conv3x3 (+ 1x1 shortcut when channels change).
Parameter names equal diffusers' `state_dict` keys so that real weights load into both this and the HIP module.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class VAEConfig(SimpleNamespace):
    pass


VAE_CONFIGS = {
    'sd': VAEConfig(block_out_channels=[128, 256, 512, 512], layers_per_block=2, latent_channels=4, out_channels=3,
                    norm_num_groups=32, scaling_factor=0.18215, force_upcast=True),
    'tiny': VAEConfig(block_out_channels=[32, 64, 64, 64], layers_per_block=1, latent_channels=4, out_channels=3,
                      norm_num_groups=8, scaling_factor=0.18215, force_upcast=True),
}


class ResnetRef(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttnRef(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)          # [B, N, C]
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(B, C, H, W)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetRef(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Up(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([AttnRef(c, groups)])
        self.resnets = nn.ModuleList([ResnetRef(c, c, groups), ResnetRef(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class DecoderRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch = list(reversed(cfg.block_out_channels))
        g = cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[0], 3, padding=1)
        self.mid_block = _Mid(ch[0], g)
        self.up_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.up_blocks.append(_UpBlock(prev, c, cfg.layers_per_block + 1, g, add_up=i < len(ch) - 1))
            prev = c
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        h = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            h = b(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class AutoencoderKLDecoderRef(nn.Module):
    """`.decode(z, return_dict=False)[0]`, `.config`, `.post_quant_conv`, `.dtype` -- the members the reference touches."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = DecoderRef(cfg)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def decode(self, z, return_dict=False):
        return (self.decoder(self.post_quant_conv(z)),)
