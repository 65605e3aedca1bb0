#!/usr/bin/env python
"""Attention kernels with pre-scaled queries (sidlsg_attn_fwd_ps / _bwd_ps) next to the plain entry points: time and
error against an fp32 torch reference.  GPU only.   python tools/ab/attn_ps.py [B]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd._lib import lib  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def call(ps, qkv, o, lse, heads, D, do=None, dqkv=None, delta=None):
    B, N, C3 = qkv.shape
    C = C3 // 3
    s = torch.cuda.current_stream().cuda_stream
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C
    sfx = '_ps' if ps else ''
    if do is None:
        getattr(lib, 'sidlsg_attn_fwd' + sfx)(q, k, v, o.data_ptr(), lse.data_ptr(), B, heads, N, N, D, C3, C3, C3, C, N * C3, N * C3, N * C3, N * C, s)
    else:
        dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 2 * C, dqkv.data_ptr() + 4 * C
        getattr(lib, 'sidlsg_attn_bwd' + sfx)(q, k, v, o.data_ptr(), do.data_ptr(), lse.data_ptr(), dq, dk, dv, delta.data_ptr(), B, heads, N, N, D,
                                              C3, C3, C3, C, N * C3, N * C3, N * C3, N * C, s)


def reference(qkv, heads, D, do):
    """fp32 autograd reference on the bf16 inputs: softmax(q k^T / sqrt(D)) v"""
    B, N, C3 = qkv.shape
    C = C3 // 3
    x = qkv.float().requires_grad_()
    q, k, v = [t.view(B, N, heads, D).transpose(1, 2) for t in x.split(C, dim=2)]
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, C)
    o.backward(do.float())
    return o.detach(), x.grad


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.manual_seed(0)
    for N, heads, D in ((4096, 8, 40), (1024, 8, 80), (256, 8, 160), (4096, 5, 64), (1000, 2, 40), (77, 8, 40)):
        C = heads * D
        c2 = 1.4426950408889634 / math.sqrt(D)
        Bs = B if N >= 1024 else max(B, 4)
        qkv = torch.randn(Bs, N, 3 * C, device=dev).to(BF16)
        qkv_ps = qkv.clone()
        qkv_ps[:, :, :C] = (qkv[:, :, :C].float() * c2).to(BF16)
        do = torch.randn(Bs, N, C, device=dev).to(BF16)
        res = {}
        for ps, x in ((False, qkv), (True, qkv_ps)):
            o = torch.empty(Bs, N, C, device=dev, dtype=BF16)
            lse = torch.empty(Bs, heads, N, device=dev, dtype=F32)
            dqkv = torch.zeros_like(x)
            delta = torch.empty(Bs, heads, N, device=dev, dtype=F32)
            tf = timeit(lambda: call(ps, x, o, lse, heads, D))
            tb = timeit(lambda: call(ps, x, o, lse, heads, D, do, dqkv, delta), iters=5)
            res[ps] = (tf, tb, o.clone(), dqkv.clone())
        fl = 4.0 * Bs * heads * N * N * D
        line = f'N{N} h{heads} d{D} B{Bs}: '
        for ps in (False, True):
            tf, tb = res[ps][:2]
            line += f"{'PS ' if ps else 'old'} fwd {tf * 1e6:7.1f} us {fl / tf / 1e12:6.1f} TF/s bwd {tb * 1e6:7.1f} us {2.5 * fl / tb / 1e12:6.1f} TF/s | "
        print(line, flush=True)
        # accuracy on a slice of the batch (the fp32 reference materialises N x N)
        nb = 1 if N >= 4096 else 2
        for ps, x in ((False, qkv), (True, qkv_ps)):
            xin = x[:nb].clone()
            if ps:   # reference of the PS call: the un-scaled queries are q' / c2 (what the kernel's math sees)
                xr = xin.float()
                xr[:, :, :C] /= c2
            else:
                xr = xin.float()
            o_ref, g_ref = reference(xr, heads, D, do[:nb])      # the PS backward returns the gradient wrt the unscaled q too
            o, dqkv = res[ps][2][:nb].float(), res[ps][3][:nb].float()
            eo = float((o - o_ref).abs().max() / o_ref.abs().max())
            eg = [float((dqkv[:, :, i * C:(i + 1) * C] - g_ref[:, :, i * C:(i + 1) * C]).norm() / g_ref[:, :, i * C:(i + 1) * C].norm()) for i in range(3)]
            print(f"    {'PS ' if ps else 'old'} O max-rel {eo:.2e}; dq/dk/dv rel-l2 {eg[0]:.2e} {eg[1]:.2e} {eg[2]:.2e}", flush=True)


if __name__ == '__main__':
    main()
