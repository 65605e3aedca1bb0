"""Worker of tests/test_gpu_ops.py::test_gemm_short_k_a_stationary (run with SIDLSG_GEMM_AS_MIN_N=160 in a fresh process):
the A-stationary short-K GEMM kernel on every shape class against the fp32 product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16)


def close(got, ref, tol, name):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert torch.isfinite(got).all(), f'{name}: non-finite output'
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item() + 1e-12
    assert err <= tol * scale, f'{name}: max err {err:.4g} vs scale {scale:.4g}'


K = 320
for M, N in ((8192, 320), (65536, 320), (16384, 960), (8200, 2560), (32768, 1280), (9000, 160), (49152, 320), (40000, 5120)):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    res = rnd(M, N, seed=4)
    ad, wd = a.to(dev), w.to(dev)
    ref = a.float() @ w.float().t()
    close(ops.gemm(ad, wd), ref, 1.2e-2, f'{M}x{N} plain')
    close(ops.gemm(ad, wd, bias=bias.to(dev)), ref + bias, 1.2e-2, f'{M}x{N} bias')
    close(ops.gemm(ad, wd, bias=bias.to(dev), res=res.to(dev)), ref + bias + res.float(), 1.2e-2, f'{M}x{N} bias + residual')
    close(ops.gemm(ad, wd, alpha=0.5), 0.5 * ref, 1.2e-2, f'{M}x{N} alpha')
    wide = torch.zeros(M, 3 * K, dtype=BF16)
    wide[:, K:2 * K] = a
    wided = wide.to(dev)
    close(ops.gemm(wided[:, K:2 * K], wd, bias=bias.to(dev), lda=3 * K), ref + bias, 1.2e-2, f'{M}x{N} strided A')
    out = torch.full((M, N + 64), 7.0, device=dev, dtype=BF16)                 # strided C: columns beyond N stay untouched
    lib.sidlsg_gemm_bf16(ad.data_ptr(), K, wd.data_ptr(), out.data_ptr(), N + 64, None, None, 0, None, 0, 1, M, N, K, 1.0, 0, ops._s())
    close(out[:, :N], ref, 1.2e-2, f'{M}x{N} strided C')
    assert float((out[:, N:].float() - 7.0).abs().max()) == 0.0
    print(f'{M}x{N}x{K} ok', flush=True)
print('all shapes ok')
