"""Oracle: fp32 NCHW restatement of the SD UNet2DConditionModel forward.  TEST INFRASTRUCTURE ONLY.

The reference never contains this arithmetic; it calls
`unet(x, t, encoder_hidden_states=e).sample` on a diffusers==0.27.2
`UNet2DConditionModel` (reference call sites: training/sid_sd_util.py:184,194,245,263;
construction: training/sid_sd_util.py:77-79).  This file restates the published
architecture of that class for the SD1.5 / SD2.1-base `unet/config.json`
 with the same module tree, parameter names and shapes, so
`state_dict()` keys equal the diffusers ones.  parity unpinned vs. diffusers itself
(package absent offline); pinned structurally by the exact parameter totals
asserted in tests/test_oracle_pinned.py and tests/test_host_logic.py::test_unet_structure_matches_diffusers_contract.

Everything here is plain torch.nn.functional on NCHW fp32 tensors; autograd gives the
reference gradients the HIP backward kernels are checked against.
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    # True = CrossAttn{Down,Up}Block2D, False = {Down,Up}Block2D (down order)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    # diffusers calls this "attention_head_dim" but it is the number of HEADS per stage
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    text_len: int = 77

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


SD15 = UNetConfig()
SD21_BASE = UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True)
# reduced-width config with the same topology, for CPU-sized tests
TINY = UNetConfig(block_out_channels=(32, 64, 128, 128), num_heads=(2, 2, 4, 4), cross_attention_dim=64,
                  norm_num_groups=8, text_len=13)
# heads of dim 40/80 at tiny width (exercises the SD1.5 head-dim padding paths)
TINY40 = UNetConfig(block_out_channels=(80, 160, 320, 320), num_heads=(2, 2, 2, 2), cross_attention_dim=96,
                    norm_num_groups=8, text_len=77)

# SD2.1-base topology at tiny width: Linear proj_in/proj_out, head dim 64 everywhere (1/2/4/4 heads), 1024-style wide context
TINY21 = UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=128,
                    norm_num_groups=8, text_len=77, use_linear_projection=True)

CONFIGS = {'sd15': SD15, 'sd21-base': SD21_BASE, 'tiny': TINY, 'tiny40': TINY40, 'tiny21': TINY21}


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    ang = t.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, query_dim, heads, cross_dim=None):
        super().__init__()
        self.heads = heads
        kv = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(kv, query_dim, bias=False)
        self.to_v = nn.Linear(kv, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Identity()])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, N, C = x.shape
        h, d = self.heads, C // self.heads
        q = self.to_q(x).view(B, N, h, d).transpose(1, 2)
        k = self.to_k(ctx).view(B, -1, h, d).transpose(1, 2)
        v = self.to_v(ctx).view(B, -1, h, d).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), v)
        o = o.transpose(1, 2).reshape(B, N, C)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        x = x + self.ff(self.norm3(x))
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, cross_dim, groups, linear_proj):
        super().__init__()
        self.linear_proj = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        if linear_proj:
            self.proj_in = nn.Linear(dim, dim)
            self.proj_out = nn.Linear(dim, dim)
        else:
            self.proj_in = nn.Conv2d(dim, dim, 1)
            self.proj_out = nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, cross_dim)])

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x)
        if self.linear_proj:
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = self.proj_in(h)
        else:
            h = self.proj_in(h)
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if self.linear_proj:
            h = self.proj_out(h)
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
            h = self.proj_out(h)
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class DownBlock(nn.Module):
    def __init__(self, cfg, cin, cout, heads, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList()
        if has_attn:
            self.attentions = nn.ModuleList()
        for j in range(cfg.layers_per_block):
            self.resnets.append(ResnetBlock2D(cin if j == 0 else cout, cout, cfg.time_embed_dim,
                                              cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2DModel(cout, heads, cfg.cross_attention_dim,
                                                          cfg.norm_num_groups, cfg.use_linear_projection))
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for j, res in enumerate(self.resnets):
            x = res(x, temb)
            if hasattr(self, 'attentions'):
                x = self.attentions[j](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg, c, heads):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, cfg.cross_attention_dim,
                                                            cfg.norm_num_groups, cfg.use_linear_projection)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
                                      for _ in range(2)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cfg, cin, cout, cprev, heads, has_attn, add_up):
        super().__init__()
        n = cfg.layers_per_block + 1
        self.resnets = nn.ModuleList()
        if has_attn:
            self.attentions = nn.ModuleList()
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = cprev if j == 0 else cout
            self.resnets.append(ResnetBlock2D(rin + skip, cout, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2DModel(cout, heads, cfg.cross_attention_dim,
                                                          cfg.norm_num_groups, cfg.use_linear_projection))
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips: List[torch.Tensor], temb, ctx):
        for j, res in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = res(x, temb)
            if hasattr(self, 'attentions'):
                x = self.attentions[j](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionRef(nn.Module):
    """`unet(x, t, encoder_hidden_states=e).sample`."""

    def __init__(self, cfg: UNetConfig = SD15):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(DownBlock(cfg, cin, cout, cfg.num_heads[i], cfg.down_has_attn[i],
                                              add_down=(i != len(ch) - 1)))
        self.mid_block = MidBlock(cfg, ch[-1], cfg.num_heads[-1])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        rev_heads = list(reversed(cfg.num_heads))
        rev_attn = list(reversed(cfg.down_has_attn))
        cout = rev[0]
        for i in range(len(ch)):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(UpBlock(cfg, cin, cout, cprev, rev_heads[i], rev_attn[i],
                                          add_up=(i != len(ch) - 1)))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states=None, **_):
        cfg = self.cfg
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        if t.ndim == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        temb = self.time_embedding(timestep_embedding(t, cfg.block_out_channels[0]).to(sample.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips.extend(outs)
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return UNetOutput(x)


def unet_forward_macs(cfg: UNetConfig, h: int, w: int) -> int:
    """Analytic MAC count of one UNet forward on one sample; contractions only."""
    macs = 0
    L = cfg.text_len
    ch = cfg.block_out_channels

    def conv(cin, cout, hh, ww, k=3):
        return hh * ww * cout * cin * k * k

    def res(cin, cout, hh, ww):
        m = conv(cin, cout, hh, ww) + conv(cout, cout, hh, ww) + cfg.time_embed_dim * cout
        if cin != cout:
            m += conv(cin, cout, hh, ww, 1)
        return m

    def tr(c, heads, hh, ww):
        n = hh * ww
        m = 2 * n * c * c                       # proj_in / proj_out
        m += 4 * n * c * c                      # self q,k,v,out
        m += 2 * n * n * c                      # QK^T + PV
        m += 2 * n * c * c + 2 * L * cfg.cross_attention_dim * c   # cross q,out + k,v
        m += 2 * n * L * c
        m += n * c * 8 * c + n * 4 * c * c      # GEGLU in, out
        return m

    macs += ch[0] * cfg.time_embed_dim + cfg.time_embed_dim ** 2
    macs += conv(cfg.in_channels, ch[0], h, w)
    hh, ww = h, w
    cout = ch[0]
    skip_ch = [ch[0]]
    for i, c in enumerate(ch):
        cin, cout = cout, c
        for j in range(cfg.layers_per_block):
            macs += res(cin if j == 0 else cout, cout, hh, ww)
            if cfg.down_has_attn[i]:
                macs += tr(cout, cfg.num_heads[i], hh, ww)
            skip_ch.append(cout)
        if i != len(ch) - 1:
            hh, ww = hh // 2, ww // 2
            macs += conv(cout, cout, hh, ww)
            skip_ch.append(cout)
    macs += 2 * res(ch[-1], ch[-1], hh, ww) + tr(ch[-1], cfg.num_heads[-1], hh, ww)
    rev = list(reversed(ch))
    rev_heads = list(reversed(cfg.num_heads))
    rev_attn = list(reversed(cfg.down_has_attn))
    cout = rev[0]
    for i in range(len(ch)):
        cprev, cout = cout, rev[i]
        for j in range(cfg.layers_per_block + 1):
            rin = cprev if j == 0 else cout
            macs += res(rin + skip_ch.pop(), cout, hh, ww)
            if rev_attn[i]:
                macs += tr(cout, rev_heads[i], hh, ww)
        if i != len(ch) - 1:
            hh, ww = hh * 2, ww * 2
            macs += conv(cout, cout, hh, ww)
    macs += conv(ch[0], cfg.out_channels, hh, ww)
    return macs
