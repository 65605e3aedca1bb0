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
    ap.add_argument('--iterations', type=int, default=1, help='also iterations 1 .. n-1 (bench_it<i>_inputs.npz next to --out): the inputs do not depend on the weights')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    from sid_lsg_amd._lib import lib
    lib.load()
    S = bench.setup_step(args.arch, args.batch_gpu, args.resolution, args.kappa, dev)
    p = S.phi.flat_params
    wsum, wabs = float(p.double().sum()), float(p.double().abs().sum())
    for it in range(args.iterations):
        inputs, _ = S.prepare(it)
        torch.cuda.synchronize()
        out = dict(key=np.array(bench.loss_reference_key(args.arch, args.batch_gpu, args.resolution, args.kappa)), iteration=np.array(it))
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
        out['weights_sum'] = np.array(wsum)
        out['weights_abs_sum'] = np.array(wabs)
        path = args.out if it == 0 else os.path.join(os.path.dirname(args.out), f'bench_it{it}_inputs.npz')
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, **out)
        print('wrote', path, {k: (v.shape, str(v.dtype)) for k, v in out.items()})
        committed = os.path.join(ROOT, 'tests', 'golden', os.path.basename(path))
        if os.path.isfile(committed) and os.path.abspath(committed) != os.path.abspath(path):
            old = np.load(committed)
            same = all(np.array_equal(old[k], out[k]) for k in out if k in old.files and k not in ('iteration',))
            print(f'  committed fixture {os.path.basename(committed)}: {"IDENTICAL" if same else "DIFFERS"}')

if __name__ == '__main__':
    main()
