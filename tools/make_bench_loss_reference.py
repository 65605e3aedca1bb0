#!/usr/bin/env python
"""Stored loss reference of the BENCH workload (tests/golden/bench_loss_reference.json).

Runs bench.py's own networks and synthetic input stream (bench.setup_step: seeded random SD1.5 weights, batch_gpu 8, kappa 1.5,
64x64x4 latents) in the HIP **fp32-accurate mode** (csrc/fp32.hip; itself pinned to the fp32 CPU oracle at the full-size
architecture by tests/test_gpu_unet.py::test_sid_iteration_full_size_*) for the first iterations and stores loss_fake / loss_G of
each.  bench.py compares the losses of its iteration 0 and of its first timed step with these values (`loss_check` in the JSON
line); tests/test_gpu_bench_parity.py replays every stored iteration in bf16 and re-derives iteration 0 in fp32.

    python tools/make_bench_loss_reference.py [--iters 7] [--arch sd15 --batch-gpu 8 --resolution 512 --kappa 1.5]
Needs an MI355X (~1 min).  The values do not depend on the box (fp32 accumulation everywhere; atomics order ~1e-7)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=7)
    ap.add_argument('--arch', default='sd15')
    ap.add_argument('--batch-gpu', type=int, default=8)
    ap.add_argument('--resolution', type=int, default=512)
    ap.add_argument('--kappa', type=float, default=1.5)
    ap.add_argument('--out', default=bench.LOSS_REFERENCE)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    from sid_lsg_amd._lib import lib
    lib.load()
    S = bench.setup_step(args.arch, args.batch_gpu, args.resolution, args.kappa, dev, compute_dtype=torch.float32)
    lf, lg = [], []
    for it in range(args.iters):
        a, b = S.one_iteration(it)
        lf.append(float(a))
        lg.append(float(b))
        print(f'iteration {it}: loss_fake {lf[-1]:.6f} loss_G {lg[-1]:.6f}', flush=True)
    data = {}
    if os.path.isfile(args.out):
        with open(args.out) as f:
            data = json.load(f)
    ent = data.setdefault(bench.loss_reference_key(args.arch, args.batch_gpu, args.resolution, args.kappa), {})
    ent.update(made_by='tools/make_bench_loss_reference.py: bench.setup_step(compute_dtype=float32), HIP fp32-accurate mode on one MI355X',
               loss_fake=lf, loss_G=lg)          # (`oracle_fp32`, written by oracle/make_bench_oracle_reference.py, is kept)
    with open(args.out, 'w') as f:
        json.dump(data, f, indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
