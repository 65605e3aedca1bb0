#!/usr/bin/env python
"""Reference point: what does the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on the dense shapes of the step,
and on dense GEMMs with the conv layers' M/N/K?  (Not used by the product path; for DESIGN.md.)"""
import torch

dev = torch.device('cuda')


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


shapes = [(65536, 320, 320), (65536, 2560, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 5120, 640), (4096, 10240, 1280),
          (65536, 320, 2880), (65536, 320, 5760), (16384, 640, 5760), (16384, 640, 11520), (4096, 1280, 11520), (4096, 1280, 23040)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: torch.matmul(a, w.t()))
    print(f'{M}x{N}x{K}: {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TF/s')
