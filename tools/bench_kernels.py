#!/usr/bin/env python
"""Per-shape kernel micro-benchmarks on the SD1.5 work list -- the optimisation harness.
Prints achieved TFLOP/s (contractions) or GB/s (HBM-bound kernels) per shape.  GPU only.

    python tools/bench_kernels.py [conv] [gemm] [wgrad] [attn] [norm] [geglu] [fp8] [--batch 16]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sid_lsg_amd import ops  # noqa: E402
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


def r(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(BF16)


def bench_conv(B):
    print('--- conv3x3 (B, H, Cin, Cout, stride, ups): TFLOP/s')
    cases = [(64, 320, 320, 1, 0), (64, 640, 320, 1, 0), (64, 960, 320, 1, 0), (32, 640, 640, 1, 0), (32, 1280, 640, 1, 0),
             (32, 1920, 640, 1, 0), (16, 1280, 1280, 1, 0), (16, 2560, 1280, 1, 0), (8, 1280, 1280, 1, 0), (8, 2560, 1280, 1, 0),
             (64, 320, 320, 2, 0), (32, 640, 640, 1, 1), (16, 1280, 1280, 1, 1), (64, 320, 8, 1, 0), (64, 8, 320, 1, 0)]
    tot_f = tot_t = 0
    for H, cin, cout, stride, ups in cases:
        Hs = H // 2 if ups else H
        x, w = r(B, Hs, Hs, cin), r(cout, 9 * cin, scale=0.02)
        bias = torch.randn(cout, device=dev)
        t = timeit(lambda: ops.conv3x3(x, w, bias=bias, stride=stride, ups=ups))
        Ho = (H - 1) // stride + 1
        fl = 2.0 * B * Ho * Ho * cout * 9 * cin
        tot_f += fl; tot_t += t
        print(f'  B{B} {H}x{H} {cin}->{cout} s{stride} u{ups}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s')
    print(f'  aggregate {tot_f / tot_t / 1e12:.1f} TF/s')


def bench_gemm(B):
    print('--- dense GEMM (M, N, K): TFLOP/s')
    cases = []
    for hw, c in ((4096, 320), (1024, 640), (256, 1280)):
        M = B * hw
        cases += [(M, 3 * c, c), (M, c, c), (M, 8 * c, c), (M, c, 4 * c)]
    cases += [(B * 77, 640, 768), (B * 4096, 320, 640), (B * 64, 1280, 1280), (B * 128, 1280, 1280), (B * 128, 5120, 1280),
              (B * 256, 1280, 5120), (B * 128, 1280, 5120), (B * 512, 640, 640)]
    tot_f = tot_t = 0
    for M, N, K in cases:
        a, w = r(M, K), r(N, K, scale=0.02)
        bias = torch.randn(N, device=dev)
        t = timeit(lambda: ops.gemm(a, w, bias=bias))
        fl = 2.0 * M * N * K
        tot_f += fl; tot_t += t
        print(f'  {M}x{N}x{K}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s')
    print(f'  aggregate {tot_f / tot_t / 1e12:.1f} TF/s')


def bench_fp8(B):
    """fp8-weight forward contractions next to the bf16 kernels on the same shapes."""
    print('--- fp8-weight vs bf16 forward: TFLOP/s')
    for M, N, K in ((B * 4096, 320, 320), (B * 4096, 2560, 320), (B * 4096, 320, 1280), (B * 1024, 640, 640), (B * 1024, 5120, 640),
                    (B * 256, 1280, 1280), (B * 256, 10240, 1280), (B * 256, 1280, 5120)):
        a, w = r(M, K), r(N, K, scale=0.02)
        q = ops.Fp8Weight(w)
        t16, t8 = timeit(lambda: ops.gemm(a, w)), timeit(lambda: ops.gemm(a, q))
        fl = 2.0 * M * N * K
        print(f'  gemm {M}x{N}x{K}: bf16 {t16 * 1e6:8.1f} us {fl / t16 / 1e12:6.1f} | fp8w {t8 * 1e6:8.1f} us {fl / t8 / 1e12:6.1f} TF/s')
    for H, cin, cout in ((64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (8, 2560, 1280)):
        x, w = r(B, H, H, cin), r(cout, 9 * cin, scale=0.02)
        q = ops.Fp8Weight(w)
        t16, t8 = timeit(lambda: ops.conv3x3(x, w)), timeit(lambda: ops.conv3x3(x, q))
        fl = 2.0 * B * H * H * cout * 9 * cin
        print(f'  conv {H}x{H} {cin}->{cout}: bf16 {t16 * 1e6:8.1f} us {fl / t16 / 1e12:6.1f} | fp8w {t8 * 1e6:8.1f} us {fl / t8 / 1e12:6.1f} TF/s')


def bench_wgrad(B):
    print('--- wgrad: TFLOP/s')
    for M, N, K in ((B * 4096, 320, 320), (B * 4096, 2560, 320), (B * 1024, 640, 2560), (B * 256, 1280, 1280)):
        dy, a = r(M, N), r(M, K)
        dw = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        t = timeit(lambda: lib.sidlsg_wgrad_bf16(dy.data_ptr(), N, a.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K, ops._s()))
        print(f'  dense {M}x{N}x{K}: {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TF/s')
    for H, cin, cout in ((64, 320, 320), (32, 640, 640), (16, 1280, 1280), (32, 1920, 640)):
        x, dy = r(B, H, H, cin), r(B, H, H, cout)
        dw = torch.zeros(cout, 9 * cin, device=dev)
        db = torch.zeros(cout, device=dev)
        t = timeit(lambda: lib.sidlsg_conv3x3_wgrad_bf16(dy.data_ptr(), cout, x.data_ptr(), cin, dw.data_ptr(), db.data_ptr(), B, H, H, cin, cout, 1, 0, ops._s()))
        print(f'  conv {H}x{H} {cin}->{cout}: {t * 1e6:8.1f} us  {2.0 * B * H * H * cout * 9 * cin / t / 1e12:7.1f} TF/s')


def bench_attn(B):
    print('--- attention (N, heads, D): TFLOP/s fwd / bwd')
    for N, heads, D in ((4096, 8, 40), (1024, 8, 80), (256, 8, 160), (4096, 5, 64)):
        C = heads * D
        qkv = r(B, N, 3 * C).requires_grad_()
        do = r(B, N, C)
        tf = timeit(lambda: ops.self_attention(qkv.detach(), heads))
        y = ops.self_attention(qkv, heads)

        def bwd():
            y.backward(do, retain_graph=True)
        tb = timeit(bwd, iters=5)
        fl = 4.0 * B * heads * N * N * D
        print(f'  self N{N} h{heads} d{D}: fwd {tf * 1e6:8.1f} us {fl / tf / 1e12:6.1f} TF/s | bwd {tb * 1e6:8.1f} us {2.5 * fl / tb / 1e12:6.1f} TF/s')
    for N, heads, D in ((4096, 8, 40), (1024, 8, 80)):
        C = heads * D
        q, kv = r(B, N, C), r(B, 77, 2 * C)
        tf = timeit(lambda: ops.cross_attention(q, kv, heads))
        print(f'  cross N{N} L77 h{heads} d{D}: fwd {tf * 1e6:8.1f} us {4.0 * B * heads * N * 77 * D / tf / 1e12:6.1f} TF/s')


def bench_norm(B):
    """Raw C-ABI calls on preallocated buffers (kernel time, not Python/autograd overhead)."""
    print('--- GroupNorm+SiLU / LayerNorm: us and GB/s of algorithmic bytes (fwd: x,y; bwd: x,dy,dx; bf16)')
    F32 = torch.float32
    for HW, C in ((4096, 320), (4096, 640), (1024, 640), (1024, 1280), (1024, 1920), (256, 1280), (256, 2560), (64, 1280), (64, 2560)):
        x, dy = r(B, HW, C), r(B, HW, C)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = torch.empty(lib.sidlsg_groupnorm_ws_floats.raw(B, HW, C, 32), device=dev, dtype=F32)
        st = torch.empty(B, 32, 2, device=dev, dtype=F32)
        tf = timeit(lambda: lib.sidlsg_groupnorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), ws.data_ptr(),
                                                     B, HW, C, 32, 1e-5, 1, ops._s()))
        tb = timeit(lambda: lib.sidlsg_groupnorm_bwd(x.data_ptr(), dy.data_ptr(), st.data_ptr(), g.data_ptr(), b.data_ptr(), None, dx.data_ptr(),
                                                     gg.data_ptr(), gb.data_ptr(), ws.data_ptr(), B, HW, C, 32, 1, ops._s()))
        by = 1.0 * B * HW * C * 2
        print(f'  GN B{B} HW{HW} C{C}: fwd {tf * 1e6:7.1f} us {2 * by / tf / 1e9:6.0f} GB/s | bwd {tb * 1e6:7.1f} us {3 * by / tb / 1e9:6.0f} GB/s')
    for rows, C in ((B * 4096, 320), (B * 1024, 640), (B * 256, 1280)):
        x, dy = r(rows, C), r(rows, C)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        st = torch.empty(rows, 2, device=dev, dtype=F32)
        ws = torch.empty(lib.sidlsg_layernorm_bwd_nblocks.raw(rows) * C * 2, device=dev, dtype=F32)
        tf = timeit(lambda: lib.sidlsg_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), rows, C, 1e-5, ops._s()))
        tb = timeit(lambda: lib.sidlsg_layernorm_bwd(x.data_ptr(), dy.data_ptr(), st.data_ptr(), g.data_ptr(), None, dx.data_ptr(), gg.data_ptr(),
                                                     gb.data_ptr(), ws.data_ptr(), rows, C, ops._s()))
        by = 1.0 * rows * C * 2
        print(f'  LN {rows}x{C}: fwd {tf * 1e6:7.1f} us {2 * by / tf / 1e9:6.0f} GB/s | bwd {tb * 1e6:7.1f} us {3 * by / tb / 1e9:6.0f} GB/s')


def bench_geglu(B):
    print('--- GEGLU fwd / bwd: us and GB/s (fwd: h 2F in, y F out; bwd: h, dy in, dh out)')
    for M, F in ((B * 4096, 1280), (B * 1024, 2560), (B * 256, 5120)):
        h, dy = r(M, 2 * F), r(M, F)
        y, dh = torch.empty(M, F, device=dev, dtype=BF16), torch.empty(M, 2 * F, device=dev, dtype=BF16)
        tf = timeit(lambda: lib.sidlsg_geglu_fwd(h.data_ptr(), y.data_ptr(), M, F, ops._s()))
        tb = timeit(lambda: lib.sidlsg_geglu_bwd(h.data_ptr(), dy.data_ptr(), dh.data_ptr(), M, F, ops._s()))
        print(f'  {M}x{F}: fwd {tf * 1e6:7.1f} us {3.0 * M * F * 2 / tf / 1e9:6.0f} GB/s | bwd {tb * 1e6:7.1f} us {5.0 * M * F * 2 / tb / 1e9:6.0f} GB/s')


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--') and not a.endswith('.json')]
    B = 16
    if '--batch' in sys.argv:
        B = int(sys.argv[sys.argv.index('--batch') + 1])
        args = [a for a in args if a != str(B)]
    which = args or ['conv', 'gemm', 'wgrad', 'attn', 'norm']
    lib.load()
    trace_json = sys.argv[sys.argv.index('--trace-json') + 1] if '--trace-json' in sys.argv else None
    if trace_json:       # the library's own per-dispatch timing + ALGORITHMIC bytes of exactly these launches (include/sidlsg_hip.h)
        lib.sidlsg_trace_enable(1 << 16)
    for w in which:
        if not w.endswith('.json'):
            globals()['bench_' + w](B)
    if trace_json:
        import json
        torch.cuda.synchronize()
        FAM = dict(gemm=0, conv=1, attn=2, attn_bwd=3, wgrad=4, conv_wgrad=5, gn=6, gn_bwd=7, ln=8, ln_bwd=9)
        buf = torch.zeros(9, dtype=torch.float64)
        out = {}
        for k, f in FAM.items():
            lib.sidlsg_trace_read(f, buf.data_ptr())
            ms, work, sampled, calls, kernels, nbytes, bound = buf.tolist()[:7]
            if sampled:
                out[k] = dict(calls=int(sampled), algorithmic_bytes_per_call=nbytes / sampled, work_per_call=work / sampled, ms_per_call=ms / sampled)
        lib.sidlsg_trace_enable(0)
        json.dump(out, open(trace_json, 'w'), indent=1)
        print('trace:', {k: (v['calls'], round(v['algorithmic_bytes_per_call'] / 1e6, 1), round(v['ms_per_call'] * 1e3, 1)) for k, v in out.items()})
