#!/usr/bin/env python
"""sidlsg_gemm_mx8 (both operands e4m3, MX MFMA) against an fp64 reference on the same quantised operands, and its time next
to the bf16 kernel on the same shapes.   python tools/ab/mx8_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sid_lsg_amd import ops  # noqa: E402
from sid_lsg_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def timeit(fn, iters=20, warm=30):
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


torch.manual_seed(0)
for M, N, K in ((4096, 320, 128), (1232, 640, 768), (65536, 960, 320), (65536, 2560, 320), (16384, 1920, 640), (16384, 5120, 640), (4096, 10240, 1280),
                (4096, 3840, 1280), (65536, 320, 320)):
    a = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    bias = torch.randn(N, device=dev)
    a8 = torch.empty(M, K, device=dev, dtype=torch.uint8)
    lib.sidlsg_cast_fp8(a.data_ptr(), a8.data_ptr(), M * K, ops._s())
    w8 = ops.Fp8Weight(w)
    out = torch.empty(M, N, device=dev, dtype=BF16)

    def mx8():
        lib.sidlsg_gemm_mx8(a8.data_ptr(), K, w8.q.data_ptr(), w8.scale.data_ptr(), out.data_ptr(), N, bias.data_ptr(), None, 0, None, 0, 1, M, N, K, 1.0, 0, ops._s())
    mx8()
    torch.cuda.synchronize()
    rows = slice(0, min(M, 2048))
    ref = a8[rows].view(torch.float8_e4m3fn).double() @ w8.dequantize().double().t() + bias.double()
    got = out[rows].double()
    err = float((got - ref).abs().max() / ref.abs().max())
    # the cast itself: every element within half an e4m3 ulp (2^-4 relative) of the bf16 value
    cast_err = float(((a8.view(torch.float8_e4m3fn).float() - a.float()).abs() / a.float().abs().clamp_min(2.0 ** -6)).max())
    c16 = torch.empty(M, N, device=dev, dtype=BF16)
    t16 = timeit(lambda: ops.gemm(a, w, out=c16, bias=bias))
    t8 = timeit(mx8)
    fl = 2.0 * M * N * K
    print(f'{M}x{N}x{K}: mx8 {t8 * 1e6:7.1f} us {fl / t8 / 1e12:7.0f} TF/s | bf16 {t16 * 1e6:7.1f} us {fl / t16 / 1e12:7.0f} TF/s | x{t16 / t8:4.2f} | '
          f'max err vs fp64 on the quantised operands {err:.2e} (bf16 output rounding 4e-3) | cast err {cast_err:.3f} ulp-rel', flush=True)
