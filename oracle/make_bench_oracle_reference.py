#!/usr/bin/env python
"""ORACLE pin of the bench workload (test infrastructure; runs in the GPU-less build container, ~20-30 min on 8 cores).

BASELINE.json configs[1] -- full-size SD1.5, kappa = 1.5 on every branch, batch_gpu 8, 64x64x4 latents -- iteration 0 on the
fp32 CPU oracle (oracle/sid_ref.py on oracle/unet_ref.py), on the inputs bench.py itself draws (tests/golden/
bench_it0_inputs.npz, written on the GPU by tools/dump_bench_it0_inputs.py) and on bench.py's weights (seeded CPU generator:
sid_lsg_amd.unet.random_state_dict).  The batch of 8 is evaluated as 8 accumulation rounds of one sample with
batch_gpu_total = 8 -- the reference's own gradient accumulation (training/sid_training_loop.py:246-249, 389-450): the losses are
sums over samples x scale / batch_gpu_total and GroupNorm / LayerNorm / attention never mix samples, so the round losses add up
to the one-round loss bench.py reports and the accumulated gradient is the one-round gradient (fp32 summation order aside);
a CFG batch of 16 with autograd state does not fit this container's 62 GB.

Writes `oracle_fp32` (loss_fake / loss_G per iteration) into tests/golden/bench_loss_reference.json next to the HIP fp32-mode values;
tests/test_gpu_bench_parity.py and bench.py's `loss_check` compare against it.

`--iterations n` (round 6): iterations 0 .. n-1 in sequence, each on its own committed inputs (tests/golden/bench_it<i>_inputs.npz,
tools/dump_bench_it0_inputs.py --iterations n), carrying the fake-score / generator weights and their Adam state
(training/sid_training_loop.py:443-445, 522-530) from one iteration to the next, so that the stored references after optimizer
steps are the ORACLE's too, not the HIP fp32 mode's.  The file is rewritten after every finished iteration (~25-30 min each on 8 cores).

    python oracle/make_bench_oracle_reference.py [--threads 8] [--iterations 3]"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--inputs', default=os.path.join(ROOT, 'tests', 'golden', 'bench_it0_inputs.npz'))
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'bench_loss_reference.json'))
    ap.add_argument('--arch', default='sd15')
    ap.add_argument('--kappa', type=float, default=1.5)
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    ap.add_argument('--samples', type=int, default=None, help='debug: only the first n samples (the result is then NOT stored)')
    ap.add_argument('--iterations', type=int, default=1)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import sid_ref
    from oracle.scheduler_ref import DDPMSchedulerRef
    from oracle.unet_ref import CONFIGS, UNet2DConditionRef
    from sid_lsg_amd.unet import CONFIGS as HC, random_state_dict
    d = np.load(args.inputs)
    key = str(d['key'])
    b = d['A_z'].shape[0]
    t0 = time.time()
    phi = UNet2DConditionRef(CONFIGS[args.arch]).eval().requires_grad_(False)
    phi.load_state_dict(random_state_dict(HC[args.arch], seed=0))          # bench.setup_step: load_sd15('random:<arch>', seed=0)
    s = sum(float(p.double().sum()) for p in phi.parameters())
    sa = sum(float(p.double().abs().sum()) for p in phi.parameters())
    print(f'weights: sum {s:.6f} abs {sa:.6f}  (GPU box: {float(d["weights_sum"]):.6f} / {float(d["weights_abs_sum"]):.6f})  [{time.time() - t0:.0f} s]', flush=True)
    assert abs(s - float(d['weights_sum'])) < 1e-6 * sa and abs(sa - float(d['weights_abs_sum'])) < 1e-9 * sa, 'not the weights bench.py built'
    psi, G = copy.deepcopy(phi), copy.deepcopy(phi)                          # bench: psi, G = clones of phi
    sched = DDPMSchedulerRef()
    st_psi, st_G = [{} for _ in psi.parameters()], [{} for _ in G.parameters()]      # Adam state, carried across iterations

    def bf16(a):
        return torch.from_numpy(a.copy()).view(torch.bfloat16).float()
    k = args.kappa
    losses_f, losses_g = [], []
    for it in range(args.iterations):
        if it:
            path = os.path.join(os.path.dirname(args.inputs), f'bench_it{it}_inputs.npz')
            d = np.load(path)
            assert str(d['key']) == key and int(d['iteration']) == it, path

        def rounds(ph):
            cond, unc = bf16(d[f'{ph}_cond_bf16']), bf16(d[f'{ph}_uncond_bf16'])
            n = b if args.samples is None else args.samples
            return [dict(z=torch.from_numpy(d[f'{ph}_z'][i:i + 1]), noise=torch.from_numpy(d[f'{ph}_noise'][i:i + 1]),
                         t=torch.from_numpy(d[f'{ph}_t'][i:i + 1]), cond=cond[i:i + 1], uncond=unc) for i in range(n)]
        # ---- phase A (sid_iteration_ref's phase A, per-round losses summed)
        psi.requires_grad_(True)
        loss_fake = 0.0
        for i, r in enumerate(rounds('A')):
            init_t = torch.full((1,), 625, dtype=torch.long)
            with torch.no_grad():
                images = sid_ref.sampler_ref(G, r['z'], r['cond'], init_t, sched)
            nf = sid_ref.denoise_ref(psi, images, r['noise'], r['cond'], r['uncond'], r['t'], sched, predict_x0=False, guidance_scale=k)
            loss, n = sid_ref.fake_score_loss_ref(nf, r['noise'], 1.0, b)
            assert n == 1
            loss.backward()
            loss_fake += float(loss.detach())
            print(f'iteration {it} phase A sample {i}: {float(loss.detach()):.6f}  [{time.time() - t0:.0f} s]', flush=True)
        psi.requires_grad_(False)
        with torch.no_grad():
            for p, stt in zip(psi.parameters(), st_psi):
                sid_ref.adam_step_ref(p, p.grad, stt, 1e-6, (0.0, 0.999), 1e-8)
                p.grad = None
        # ---- phase B
        G.requires_grad_(True)
        loss_G = 0.0
        for i, r in enumerate(rounds('B')):
            init_t = torch.full((1,), 625, dtype=torch.long)
            images = sid_ref.sampler_ref(G, r['z'], r['cond'], init_t, sched)
            y_fake = sid_ref.denoise_ref(psi, images, r['noise'], r['cond'], r['uncond'], r['t'], sched, guidance_scale=k)
            y_real = sid_ref.denoise_ref(phi, images, r['noise'], r['cond'], r['uncond'], r['t'], sched, guidance_scale=k)
            loss, n = sid_ref.generator_loss_ref(images, y_real, y_fake, 1.0, 1.0, b)
            assert n == 1
            loss.backward()
            loss_G += float(loss.detach())
            print(f'iteration {it} phase B sample {i}: {float(loss.detach()):.6f}  [{time.time() - t0:.0f} s]', flush=True)
        G.requires_grad_(False)
        print(f'oracle fp32, {key}, iteration {it}: loss_fake {loss_fake:.6f} loss_G {loss_G:.6f}', flush=True)
        losses_f.append(loss_fake); losses_g.append(loss_G)
        if args.samples is not None:
            return
        data = {}
        if os.path.isfile(args.out):
            with open(args.out) as f:
                data = json.load(f)
        ent = data.setdefault(key, {})
        ent['oracle_fp32'] = dict(made_by='oracle/make_bench_oracle_reference.py: fp32 CPU oracle (oracle/sid_ref.py) on tests/golden/bench_it<i>_inputs.npz, '
                                          f'{b} accumulation rounds of one sample per iteration, weights + Adam state carried across iterations',
                                  loss_fake=list(losses_f), loss_G=list(losses_g))
        with open(args.out, 'w') as f:
            json.dump(data, f, indent=1)
        print('wrote', args.out, flush=True)
        if it + 1 < args.iterations:           # the generator's step (the last iteration's is not needed for any stored loss)
            with torch.no_grad():
                for p, stt in zip(G.parameters(), st_G):
                    sid_ref.adam_step_ref(p, p.grad, stt, 1e-6, (0.0, 0.999), 1e-8)
        for p in G.parameters():
            p.grad = None

if __name__ == '__main__':
    main()
