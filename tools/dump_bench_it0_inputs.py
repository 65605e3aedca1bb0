#!/usr/bin/env python
"""Iteration-0 inputs of the BENCH workload as a fixture (tests/golden/bench_it0_inputs.npz).

bench.py draws z / noise / t with the device generator and encodes its synthetic prompts with the PyTorch-ROCm CLIP text
encoder in bf16 on the GPU -- neither can be re-derived in the GPU-less build container.  This tool stores exactly what
`bench.setup_step(...).prepare(0)` hands to the step (phases A and B; the text states as their bf16 bit patterns), so that
oracle/make_bench_oracle_reference.py can run the fp32 CPU oracle on the SAME inputs and the SAME (CPU-seeded) weights.

    python tools/dump_bench_it0_inputs.py [--out gpurun_out/bench_it0_inputs.npz]      (needs an MI355X, ~1 min)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='sd15')
    ap.add_argument('--batch-gpu', type=int, default=8)
    ap.add_argument('--resolution', type=int, default=512)
    ap.add_argument('--kappa', type=float, default=1.5)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'bench_it0_inputs.npz'))
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    from sid_lsg_amd._lib import lib
    lib.load()
    S = bench.setup_step(args.arch, args.batch_gpu, args.resolution, args.kappa, dev)
    inputs, _ = S.prepare(0)
    torch.cuda.synchronize()
    out = dict(key=np.array(bench.loss_reference_key(args.arch, args.batch_gpu, args.resolution, args.kappa)))
    for ph in ('A', 'B'):
        (r,) = inputs[ph]
        out[f'{ph}_z'] = r['z'].float().cpu().numpy()
        out[f'{ph}_noise'] = r['noise'].float().cpu().numpy()
        out[f'{ph}_t'] = r['t'].cpu().numpy()
        assert r['cond'].dtype == torch.bfloat16 and r['uncond'].dtype == torch.bfloat16
        out[f'{ph}_cond_bf16'] = r['cond'].contiguous().view(torch.int16).cpu().numpy()
        u = r['uncond'].contiguous()
        assert bool((u == u[:1]).all()), 'the "" state is one row repeated'
        out[f'{ph}_uncond_bf16'] = u[:1].view(torch.int16).cpu().numpy()
    # a checksum of the weights this box built from the CPU generator (the oracle script asserts it reproduces them)
    p = S.phi.flat_params
    out['weights_sum'] = np.array(float(p.double().sum()))
    out['weights_abs_sum'] = np.array(float(p.double().abs().sum()))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez(args.out, **out)
    print('wrote', args.out, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == '__main__':
    main()
