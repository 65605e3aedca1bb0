#!/usr/bin/env python
"""Where does the dense-GEMM time of one distillation iteration go?  HIP events around every sidlsg_gemm_bf16 call of a few
bench iterations, aggregated by (M, N, K).   python tools/gemm_shape_profile.py [--entry sidlsg_gemm_bf16|sidlsg_conv3x3_bf16]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    entry = sys.argv[sys.argv.index('--entry') + 1] if '--entry' in sys.argv else 'sidlsg_gemm_bf16'
    from sid_lsg_amd._lib import lib
    lib.load()
    orig = getattr(lib, entry)
    events = []

    def timed(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(*a); e1.record()
        if entry == 'sidlsg_gemm_bf16':
            key = (a[11], a[12], a[13])                     # M, N, K
            fl = 2.0 * a[11] * a[12] * a[13]
        else:
            key = tuple(a[10:17])                            # B, H, W, Cin, Cout, stride, ups
            B, H, W, Cin, Cout, stride, ups = key
            Hi, Wi = (2 * H, 2 * W) if ups else (H, W)
            fl = 2.0 * B * ((Hi - 1) // stride + 1) * ((Wi - 1) // stride + 1) * Cout * 9 * Cin
        events.append((key, fl, e0, e1))
    lib.__dict__[entry] = timed
    sys.argv = [sys.argv[0], '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-kernel-timing']
    bench.main()
    torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for key, fl, e0, e1 in events:
        a = agg[key]
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
    tot = sum(v[1] for v in agg.values())
    print(f'{entry}: {len(events)} calls, {tot:.1f} ms over 3 iterations')
    for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'  {str(key):40s} calls {n:5d}  {ms:8.2f} ms  {100 * ms / tot:5.1f} %  avg {1e3 * ms / n:7.1f} us  {fl / ms / 1e9:7.1f} TF/s')


if __name__ == '__main__':
    main()
