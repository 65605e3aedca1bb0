"""AutoencoderKL decoder on the HIP kernels (inference only): `vae.decode(z / scaling_factor, return_dict=False)[0]`.

Callers in the reference: sid_sd_sampler(return_images=True) (training/sid_sd_util.py:198-209) from the snapshot preview
(sid_training_loop.py:357-363) and generate_onestep.py:218-311 -- SURVEY.md section 8(f), row 1.  Same parameter names as
diffusers' AutoencoderKL, so a real `vae/diffusion_pytorch_model.safetensors` loads with `load_state_dict` (encoder keys
are ignored: nothing on this path encodes images).

Everything runs on the UNet's kernels (NHWC bf16): implicit-GEMM conv3x3 (nearest x2 fused into the loader), GroupNorm+SiLU,
GEMM.  The one exception is the mid-block attention: a single head of width 512 over 4096 tokens, outside the head sizes
(<= 160) the flash kernel is built for -- it uses torch's scaled_dot_product_attention (one call per decode).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

BF16, F32 = torch.bfloat16, torch.float32

VAE_CONFIGS = {
    'sd': dict(block_out_channels=[128, 256, 512, 512], layers_per_block=2, latent_channels=4, out_channels=3,
               norm_num_groups=32, scaling_factor=0.18215, force_upcast=True),
    'tiny': dict(block_out_channels=[32, 64, 64, 64], layers_per_block=1, latent_channels=4, out_channels=3,
                 norm_num_groups=8, scaling_factor=0.18215, force_upcast=True),
}


def _pad8(n):
    return (n + 7) // 8 * 8


class _Conv3(nn.Module):
    def __init__(self, cin, cout, ups=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.empty(cout))
        self.cin, self.cout, self.ups = cin, cout, ups
        self.w16 = self.b32 = None

    def prepare(self):
        cip, cop = _pad8(self.cin), _pad8(self.cout)                    # conv_in (4 -> 8 in) / conv_out (3 -> 8 out)
        w = torch.zeros((cop, 9, cip), device=self.weight.device, dtype=BF16)
        w[:self.cout, :, :self.cin] = self.weight.detach().permute(0, 2, 3, 1).reshape(self.cout, 9, self.cin).to(BF16)
        b = torch.zeros(cop, device=self.weight.device, dtype=F32)
        b[:self.cout] = self.bias.detach().float()
        self.w16, self.b32 = w.view(cop, 9 * cip), b

    def forward(self, x, res=None, out_f32=False):
        return ops.conv3x3(x, self.w16, bias=self.b32, res=res, ups=self.ups, out_f32=out_f32)


class _Lin(nn.Module):
    def __init__(self, cin, cout, conv1x1=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((cout, cin, 1, 1) if conv1x1 else (cout, cin)))
        self.bias = nn.Parameter(torch.empty(cout))
        self.w16 = self.b32 = None

    def prepare(self):
        self.w16 = self.weight.detach().reshape(self.weight.shape[0], -1).to(BF16).contiguous()
        self.b32 = self.bias.detach().float().contiguous()

    def forward(self, x2d, res=None):
        return ops.gemm(x2d, self.w16, bias=self.b32, res=res)


class _GN(nn.Module):
    def __init__(self, groups, c):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.empty(c)), nn.Parameter(torch.empty(c))
        self.groups = groups

    def forward(self, x, silu):
        return ops.group_norm(x, self.weight, self.bias, self.groups, 1e-6, silu)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1, self.conv1 = _GN(groups, cin), _Conv3(cin, cout)
        self.norm2, self.conv2 = _GN(groups, cout), _Conv3(cout, cout)
        self.conv_shortcut = _Lin(cin, cout, conv1x1=True) if cin != cout else None

    def forward(self, x):
        B, H, W, C = x.shape
        sc = x if self.conv_shortcut is None else self.conv_shortcut(x.view(B * H * W, C)).view(B, H, W, -1)
        return self.conv2(self.norm2(self.conv1(self.norm1(x, True)), True), res=sc)


class _Attn(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = _GN(groups, c)
        self.to_q, self.to_k, self.to_v = _Lin(c, c), _Lin(c, c), _Lin(c, c)
        self.to_out = nn.ModuleList([_Lin(c, c), nn.Identity()])

    def forward(self, x):
        B, H, W, C = x.shape
        h = self.group_norm(x, False).view(B * H * W, C)
        q, k, v = (m(h).view(B, 1, H * W, C) for m in (self.to_q, self.to_k, self.to_v))
        o = F.scaled_dot_product_attention(q, k, v)                      # one head, width C (512): see module docstring
        return self.to_out[0](o.reshape(B * H * W, C), res=x.view(B * H * W, C)).view(B, H, W, C)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv3(c, c, ups=1)

    def forward(self, x):
        return self.conv(x)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Up(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn(c, groups)])
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch = list(reversed(cfg.block_out_channels))
        g = cfg.norm_num_groups
        self.conv_in = _Conv3(cfg.latent_channels, ch[0])
        self.mid_block = _Mid(ch[0], g)
        self.up_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.up_blocks.append(_UpBlock(prev, c, cfg.layers_per_block + 1, g, add_up=i < len(ch) - 1))
            prev = c
        self.conv_norm_out = _GN(g, ch[-1])
        self.conv_out = _Conv3(ch[-1], cfg.out_channels)

    def forward(self, x):
        h = self.mid_block(self.conv_in(x))
        for b in self.up_blocks:
            h = b(h)
        return self.conv_out(self.conv_norm_out(h, True), out_f32=True)


class HipAutoencoderKLDecoder(nn.Module):
    """Duck-types the members the reference touches: `.decode`, `.config`, `.post_quant_conv`, `.dtype`."""

    def __init__(self, arch='sd'):
        super().__init__()
        self.config = SimpleNamespace(**VAE_CONFIGS[arch])
        self.post_quant_conv = nn.Conv2d(self.config.latent_channels, self.config.latent_channels, 1)
        self.decoder = _Decoder(self.config)
        self._ready = False
        for p in self.parameters():
            p.requires_grad_(False)

    def init_parameters(self, seed=0):
        """Seeded random weights (no VAE weights ship offline): conv/linear default-style uniform, norms = identity."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (_Conv3, _Lin, nn.Conv2d)):
                    fan_in = m.weight[0].numel()
                    bound = fan_in ** -0.5
                    m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 2 - 1) * bound)
                    m.bias.copy_((torch.rand(m.bias.shape, generator=g) * 2 - 1) * bound)
                elif isinstance(m, _GN):
                    m.weight.fill_(1.0)
                    m.bias.zero_()
        self._ready = False
        return self

    def load_state_dict(self, state_dict, strict=False, **kw):            # encoder.* / quant_conv.* keys are not used here
        sd = {k: v for k, v in state_dict.items() if k.startswith(('decoder.', 'post_quant_conv.'))}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._ready = False
        return out

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def _prepare(self):
        for m in self.modules():
            if isinstance(m, (_Conv3, _Lin)):
                m.prepare()
        self._ready = True

    @torch.no_grad()
    def decode(self, z, return_dict=False):
        """z: [B, 4, h, w] (already divided by scaling_factor by the caller) -> images [B, 3, 8h, 8w] fp32 in ~[-1, 1]."""
        if z.device.type != 'cuda':
            raise RuntimeError('HipAutoencoderKLDecoder runs on the MI355X only (no CPU fallback)')
        if not self._ready:
            self._prepare()
        z = F.conv2d(z.to(self.post_quant_conv.weight.dtype), self.post_quant_conv.weight, self.post_quant_conv.bias)
        B, C, h, w = z.shape
        x = torch.zeros((B, h, w, _pad8(C)), device=z.device, dtype=BF16)
        x[..., :C] = z.permute(0, 2, 3, 1).to(BF16)
        y = self.decoder(x)                                               # [B, 8h, 8w, 8] fp32, channels 3..7 are padding
        img = y[..., :self.config.out_channels].permute(0, 3, 1, 2).contiguous()
        if return_dict:
            return SimpleNamespace(sample=img)
        return (img,)
