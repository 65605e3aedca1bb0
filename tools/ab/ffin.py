"""short-K GEMM anatomy: full kernel vs one-K-tile (prologue+epilogue) vs no-store builds, in one GPU session."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    from sid_lsg_amd import ops
    from sid_lsg_amd._lib import lib
    lib.load()
    dev = torch.device('cuda:0')
    def timeit(fn, iters=20, warm=5):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    out = []
    for M, N, K in ((65536, 2560, 320), (65536, 320, 320), (65536, 960, 320), (65536, 1280, 320), (16384, 5120, 640), (16384, 640, 640), (65536, 320, 1280), (65536, 320, 2560)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out.append(f'{timeit(lambda: ops.gemm(a, w, out=c)):8.1f}')
    print(sys.argv[2], ' '.join(out))
else:
    print('shapes: FFin64 NxN320 qkv64 1280x320 FFin32 640sq K1280 K2560')
    for name in (sys.argv[1:] or ['base']):
        env = dict(os.environ)
        if name != 'base':
            env['SIDLSG_LIB'] = os.path.join(ROOT, 'tools', 'ab', f'lib{name}.so')
        subprocess.run([sys.executable, __file__, 'child', name], env=env)
