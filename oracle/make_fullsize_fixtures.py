"""Stored results of the FULL-SIZE oracle iterations of the GPU suite.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_fullsize_fixtures [case ...]      -> tests/golden/fullsize_<case>.npz

Each case is one complete SiD-LSG iteration of oracle/sid_ref.py (the fp32 CPU restatement of training/sid_training_loop.py:383-571,
pinned by tests/test_oracle_pinned.py) on the full-size network of oracle/unet_ref.py with the seeded weights and inputs of
tests/test_gpu_unet.py::_iteration_parity (oracle/fixtures.py: make_unet_cached, iteration_inputs, iteration_hp).  Running them live cost
the GPU suite ~500 s of host time per run (1150 s of the driver's 1200 s limit in round 5); one of them (SD1.5, kappa 1.5, batch 1)
still runs live in every suite, and SIDLSG_LIVE_ORACLE=1 makes the others run live too.  Stored per case:
  loss_fake, loss_G                       the two losses
  ema/<name>                              the EMA generator's weights after the iteration, for the names the test compares
  <net>/sign, <net>/big  (bit-packed)     per sampled weight (every 431st entry of every parameter, fixtures.sample_index; named_parameters
                                          order): sign of the oracle's update p_after - p_before and whether |update| > lr / 2 (Adam with beta1 = 0
                                          moves a weight by ~lr sign(g); entries with a ~0 gradient are excluded from the sign statistics)
  weight_checksum                         float64 checksums of the initial weights (drift of the seeded construction is detected, not ignored)
"""
import copy
import os
import sys
import time

import numpy as np
import torch

from . import fixtures, sid_ref
from .scheduler_ref import DDPMSchedulerRef

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def make(case):
    cfg_name, lat, b, kappa = fixtures.FULLSIZE_CASES[case]
    lr = fixtures.FULLSIZE_LR
    t0 = time.time()
    phi_r = fixtures.make_unet_cached(cfg_name).eval().requires_grad_(False)
    psi_r = fixtures.make_unet_cached(cfg_name, seed=77).requires_grad_(False)
    G_r = copy.deepcopy(phi_r)
    Gema_r = copy.deepcopy(G_r)
    init = {'fake_score': [p.detach().clone() for p in psi_r.parameters()], 'G': [p.detach().clone() for p in G_r.parameters()]}
    cks = np.array(fixtures.checksum(phi_r) + fixtures.checksum(psi_r))
    nets_r = dict(true_score=phi_r, fake_score=psi_r, G=G_r, G_ema=Gema_r)
    st = dict(fake_score=[{} for _ in psi_r.parameters()], G=[{} for _ in G_r.parameters()])
    hp = fixtures.iteration_hp(b, 1, lr, kappa, 1.0)
    hp['cur_nimg'] = 0
    inputs = fixtures.iteration_inputs(cfg_name, lat, b, 1, torch.Generator().manual_seed(fixtures.FULLSIZE_SEED))
    if b > 1:
        # A CFG batch of 2b samples with autograd state does not fit this container's 62 GB: the batch is evaluated as b accumulation rounds of
        # one sample with batch_gpu_total = b -- the reference's own gradient accumulation (training/sid_training_loop.py:246-249, 389-450).  The
        # losses are sums over samples x scale / batch_gpu_total and GroupNorm / LayerNorm / attention never mix samples, so the round losses add up
        # to the one-round loss and the accumulated gradient is the one-round gradient (fp32 summation order aside).
        inputs = {ph: [{k: v[i:i + 1].contiguous() for k, v in inputs[ph][0].items()} for i in range(b)] for ph in inputs}
        hp['sum_round_losses'] = True
    out_r = sid_ref.sid_iteration_ref(nets_r, st, DDPMSchedulerRef(), inputs, hp)
    rec = dict(loss_fake=np.float64(out_r['loss_fake']), loss_G=np.float64(out_r['loss_G']), weight_checksum=cks,
               case=np.array([cfg_name, str(lat), str(b), str(kappa), str(lr)]))
    ema = dict(Gema_r.named_parameters())
    for n in fixtures.FULLSIZE_EMA_NAMES:
        rec['ema/' + n] = ema[n].detach().numpy().copy()
    for name, net in (('fake_score', psi_r), ('G', G_r)):
        sign, big = [], []
        for p, p0 in zip(net.parameters(), init[name]):
            idx = fixtures.sample_index(p.numel())
            d = (p.detach().flatten()[idx] - p0.flatten()[idx])
            sign.append((d > 0).numpy())
            big.append((d.abs() > 0.5 * lr).numpy())
        sign, big = np.concatenate(sign), np.concatenate(big)
        rec[name + '/n'] = np.int64(sign.size)
        rec[name + '/sign'] = np.packbits(sign)
        rec[name + '/big'] = np.packbits(big)
        print(f'  {name}: {sign.size} sampled weights, {int(big.sum())} with an update above lr / 2')
    path = os.path.join(OUT, f'fullsize_{case}.npz')
    np.savez_compressed(path, **rec)
    print(f'{case}: loss_fake {out_r["loss_fake"]:.6f} loss_G {out_r["loss_G"]:.6f} -> {path} ({os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s)', flush=True)


if __name__ == '__main__':
    torch.set_num_threads(min(48, os.cpu_count() or 8))      # (one OpenMP thread per core of a 256-core host crawls: 20 min without finishing one case)
    for case in (sys.argv[1:] or list(fixtures.FULLSIZE_CASES)):
        make(case)
