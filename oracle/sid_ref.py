"""Oracle: the SiD-LSG inner step restated in plain torch fp32 (CPU).  TEST INFRASTRUCTURE ONLY.

Each function cites the reference lines it follows (paths relative to /root/reference).
Pinned: tests/test_oracle_pinned.py replays the golden vectors that
`oracle/make_goldens.py` captured by running the reference's own `sid_sd_sampler`,
`sid_sd_denoise` and the unmodified `training_loop` (via oracle/ref_harness.py).
"""
import copy
import math

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# A3: one-step generator   training/sid_sd_util.py:176-185 (train path, num_steps=1)
def sampler_ref(unet, z, ctx, init_t, sched):
    """x_t = add_noise(0, z, t_init) = s1*z ; eps = unet(x_t) ; x_hat = (x_t - s1*eps)/s0."""
    x_t = sched.add_noise(torch.zeros_like(z), z, init_t).to(torch.float32)          # :182
    eps = unet(sched.scale_model_input(x_t, init_t), init_t, encoder_hidden_states=ctx).sample.to(torch.float32)  # :183-184
    return sched.step(eps, init_t[0], x_t).pred_original_sample.to(torch.float32)   # :185 (note: uses init_t[0])


# A4: CFG denoise   training/sid_sd_util.py:242-274
def denoise_ref(unet, images, noise, cond, uncond, t, sched, predict_x0=True, guidance_scale=1.0):
    x_t = sched.add_noise(images, noise, t)                                           # :242
    if guidance_scale == 1:
        eps = unet(x_t, t, encoder_hidden_states=cond).sample.to(torch.float32)       # :244-245
    else:
        e = torch.cat([uncond, cond])                                                 # :259
        tt = torch.cat([t, t])                                                        # :260
        xx = torch.cat([x_t] * 2)                                                     # :261
        out = unet(xx, tt, encoder_hidden_states=e).sample.to(torch.float32)          # :263
        u, c = out.chunk(2)                                                           # :264
        eps = u + guidance_scale * (c - u)                                            # :265
    if predict_x0:                                                                    # :268-272 per-sample step
        return torch.stack([sched.step(n, tt_, z).pred_original_sample for n, tt_, z in zip(eps, t, x_t.to(torch.float32))])
    return eps                                                                        # :274


# A8: fake-score loss   training/sid_training_loop.py:423-445
def fake_score_loss_ref(noise_fake, noise, loss_scaling, batch_gpu_total):
    nan_mask = torch.isnan(noise_fake).flatten(1).any(1)                              # :423
    if nan_mask.any():                                                                # :429-434
        noise_fake, noise = noise_fake[~nan_mask], noise[~nan_mask]
    loss = ((noise_fake - noise) ** 2).sum() * (loss_scaling / batch_gpu_total)      # :443-445
    return loss, len(noise)


# A7: SiD-LSG generator loss   training/sid_training_loop.py:508-530
def generator_loss_ref(images, y_real, y_fake, alpha, loss_scaling_G, batch_gpu_total):
    nan_mask = (torch.isnan(images).flatten(1).any(1) | torch.isnan(y_real).flatten(1).any(1)
                | torch.isnan(y_fake).flatten(1).any(1))                              # :508-511
    if nan_mask.any():                                                                # :514-520
        keep = ~nan_mask
        images, y_real, y_fake = images[keep], y_real[keep], y_fake[keep]
    with torch.no_grad():                                                             # :522-523
        w = (images - y_real).abs().mean(dim=[1, 2, 3], keepdim=True).clip(min=0.00001)
    if alpha == 1:                                                                    # :525-528
        loss = (y_real - y_fake) * (y_fake - images) / w
    else:
        loss = (y_real - y_fake) * ((y_real - images) - alpha * (y_real - y_fake)) / w
    return loss.sum() * (loss_scaling_G / batch_gpu_total), len(y_real)               # :530


# A9: gradient sanitising + Adam   sid_training_loop.py:458-462, 541-549 ; sid_train.py:220-221
def adam_step_ref(p, g, state, lr, betas=(0.0, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False, clip_value=None):
    """torch.optim.Adam / AdamW single-tensor semantics (no amsgrad, no maximize).  In place."""
    g = torch.nan_to_num(g, nan=0.0, posinf=1e5, neginf=-1e5)                         # :458-460
    if clip_value is not None:                                                        # :546-547 (fp16 only)
        g = g.clamp(-clip_value, clip_value)
    state['step'] = state.get('step', 0) + 1
    b1, b2 = betas
    if weight_decay != 0:
        if decoupled:
            p.mul_(1 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m = state.setdefault('exp_avg', torch.zeros_like(p))
    v = state.setdefault('exp_avg_sq', torch.zeros_like(p))
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** state['step']
    bc2 = 1 - b2 ** state['step']
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# A10: EMA   sid_training_loop.py:553-565
def ema_beta_ref(batch_size, cur_nimg, ema_halflife_kimg, ema_rampup_ratio=0.05):
    half = ema_halflife_kimg * 1000
    if ema_rampup_ratio is not None:
        half = min(half, cur_nimg * ema_rampup_ratio)
    return 0.5 ** (batch_size / max(half, 1e-8))


def ema_update_ref(p_ema, p, beta):
    p_ema.copy_(p.detach().lerp(p_ema, beta))


# ---------------------------------------------------------------------------------------------
# A13 / sampler: prompt order   torch_utils/misc.py:110-141
def infinite_sampler_ref(n, rank=0, num_replicas=1, seed=0, window_size=0.5):
    order = np.arange(n)
    rnd = np.random.RandomState(seed)
    rnd.shuffle(order)
    window = int(np.rint(order.size * window_size))
    idx = 0
    while True:
        i = idx % order.size
        if idx % num_replicas == rank:
            yield int(order[i])
        if window >= 2:
            j = (i - rnd.randint(window)) % order.size
            order[i], order[j] = order[j], order[i]
        idx += 1


# ---------------------------------------------------------------------------------------------
# A1 + A2: one whole iteration with EXPLICIT inputs (what the HIP step is compared against)
def sid_iteration_ref(nets, opt_states, sched, inputs, hp):
    """One fake-score step + one generator step on explicit inputs.

    nets: dict(true_score, fake_score, G, G_ema|None) of torch modules (fp32, CPU)
    opt_states: dict(fake_score=[state per param], G=[...])  (mutated)
    inputs: dict with, per phase 'A'/'B', lists over accumulation rounds of
            dict(z, noise, t, cond, uncond)   (cond already has prompt-dropout applied)
    hp: dict(alpha, kappa1, kappa2, kappa4, ls, lsg, batch_gpu_total, lr, glr, betas, eps, init_t,
             batch_size, cur_nimg, ema_halflife_kimg, ema_rampup_ratio, world_size=1)
    Returns dict(loss_fake, loss_G) (values of the LAST accumulation round, as the reference
    reports them: sid_training_loop.py:452,535).
    """
    G, psi, phi = nets['G'], nets['fake_score'], nets['true_score']
    out = {}
    tot_f = tot_g = 0.0
    # ---- phase A  (sid_training_loop.py:389-462)
    psi.requires_grad_(True)
    for p in psi.parameters():
        p.grad = None
    for r in inputs['A']:
        init_t = torch.full((len(r['z']),), hp['init_t'], dtype=torch.long)
        with torch.no_grad():
            images = sampler_ref(G, r['z'], r['cond'], init_t, sched)                 # :406-411
        nf = denoise_ref(psi, images, r['noise'], r['cond'], r['uncond'], r['t'], sched,
                         predict_x0=False, guidance_scale=hp['kappa1'])               # :418-421
        loss, n = fake_score_loss_ref(nf, r['noise'], hp['ls'], hp['batch_gpu_total'])
        if n > 0:
            loss.backward()                                                           # :449-450
        tot_f = tot_f + float(loss.detach())
    # hp['sum_round_losses'] (oracle/make_fullsize_fixtures.py, make_bench_oracle_reference.py): the SUM over the accumulation rounds = the loss of the
    # same samples as one batch (the losses are sums over samples x scale / batch_gpu_total); the reference itself reports the last round's value
    out['loss_fake'] = tot_f if hp.get('sum_round_losses') else float(loss.detach())
    psi.requires_grad_(False)
    for p, st in zip(psi.parameters(), opt_states['fake_score']):
        if p.grad is not None:
            with torch.no_grad():
                adam_step_ref(p, p.grad, st, hp['lr'], hp['betas'], hp['eps'])
    # ---- phase B  (sid_training_loop.py:468-549)
    G.requires_grad_(True)
    for p in G.parameters():
        p.grad = None
    for r in inputs['B']:
        init_t = torch.full((len(r['z']),), hp['init_t'], dtype=torch.long)
        images = sampler_ref(G, r['z'], r['cond'], init_t, sched)                     # :488-491
        y_fake = denoise_ref(psi, images, r['noise'], r['cond'], r['uncond'], r['t'], sched,
                             guidance_scale=hp['kappa2'])                             # :496-499
        y_real = denoise_ref(phi, images, r['noise'], r['cond'], r['uncond'], r['t'], sched,
                             guidance_scale=hp['kappa4'])                             # :503-506
        loss, n = generator_loss_ref(images, y_real, y_fake, hp['alpha'], hp['lsg'], hp['batch_gpu_total'])
        if n > 0:
            loss.backward()                                                           # :532-533
        tot_g = tot_g + float(loss.detach())
    out['loss_G'] = tot_g if hp.get('sum_round_losses') else float(loss.detach())
    G.requires_grad_(False)
    for p, st in zip(G.parameters(), opt_states['G']):
        if p.grad is not None:
            with torch.no_grad():
                adam_step_ref(p, p.grad, st, hp['glr'], hp['betas'], hp['eps'])
    # ---- EMA  (:553-565)
    if nets.get('G_ema') is not None and hp['ema_halflife_kimg'] > 0:
        beta = ema_beta_ref(hp['batch_size'], hp['cur_nimg'], hp['ema_halflife_kimg'], hp.get('ema_rampup_ratio', 0.05))
        with torch.no_grad():
            for pe, p in zip(nets['G_ema'].parameters(), G.parameters()):
                ema_update_ref(pe, p, beta)
    return out


# ---------------------------------------------------------------------------------------------
# Whole-loop restatement with the reference's RNG consumption order (single process, CPU).
def training_loop_ref(factory, prompts, *, iterations, batch_size, batch_gpu, seed=0, alpha=1.0,
                      kappa=(1.5, 1.5, 1.5), lr=1e-6, glr=1e-6, eps=1e-8, resolution=64, init_timestep=625,
                      tmin=20, tmax=980, ema_halflife_kimg=50, ema_rampup_ratio=0.05, ls=1.0, lsg=1.0,
                      rank=0, world_size=1, grid_n=None, on_iteration=None):
    """Restates training/sid_training_loop.py:238-567 for world_size ranks simulated one at a
    time is NOT attempted: this is the single-rank restatement (world_size=1) used to pin the
    RNG order, accumulation, loss scaling, Adam and EMA against the golden loss curves.

    `prompts`: list[str] (the prompt file lines).  `grid_n`: number of preview prompts the
    reference draws `grid_z` for under seed 2024 on rank 0 (sid_training_loop.py:259-271) --
    irrelevant to the stream because the seed is restored by `torch.manual_seed(original_seed)`,
    which RESETS the generator (Appendix C of SURVEY.md); kept for documentation.
    """
    unet, vae, sched, text_encoder, tokenizer = factory()
    np.random.seed((seed * world_size + rank) % (1 << 31))                            # :238
    torch.manual_seed(np.random.randint(1 << 31))                                     # :239
    batch_gpu_total = batch_size // world_size                                        # :246
    if batch_gpu is None or batch_gpu > batch_gpu_total:
        batch_gpu = batch_gpu_total
    rounds = batch_gpu_total // batch_gpu                                             # :249
    lat = resolution // 8                                                             # :254-255
    # :259-271 -> generator reset to the same seed: no net effect on the stream
    torch.manual_seed(torch.initial_seed())
    sampler = infinite_sampler_ref(len(prompts), rank, world_size, seed)              # :274
    # :275 iter(DataLoader(...)) draws the iterator's base seed from the default generator
    # (torch.utils.data.dataloader._BaseDataLoaderIter.__init__); part of the RNG contract.
    torch.empty((), dtype=torch.int64).random_()

    def next_prompts():
        return [prompts[next(sampler)] for _ in range(batch_gpu)]                    # :275 (DataLoader batch)
    for _ in range(16):                                                               # :277-281 consumes 16 batches
        next_prompts()
    phi = unet.eval().requires_grad_(False)                                           # :284-287
    psi = copy.deepcopy(phi)
    G = copy.deepcopy(phi)
    G_ema = copy.deepcopy(G) if ema_halflife_kimg > 0 else None                       # :324-327
    nets = dict(true_score=phi, fake_score=psi, G=G, G_ema=G_ema)
    st = dict(fake_score=[{} for _ in psi.parameters()], G=[{} for _ in G.parameters()])
    use_dropout = (kappa[0] != 1 or kappa[1] != 1)                                    # :208-211

    def embed(ps):
        with torch.no_grad():
            ids = tokenizer(ps, padding='max_length', max_length=tokenizer.model_max_length, truncation=True,
                            return_tensors='pt').input_ids
            return text_encoder(ids)[0]

    losses = []
    cur_nimg = 0
    hp = dict(alpha=alpha, kappa1=kappa[0], kappa2=kappa[1], kappa4=kappa[2], ls=ls, lsg=lsg,
              batch_gpu_total=batch_gpu_total, lr=lr, glr=glr, betas=(0.0, 0.999), eps=eps, init_t=init_timestep,
              batch_size=batch_size, ema_halflife_kimg=ema_halflife_kimg, ema_rampup_ratio=ema_rampup_ratio)
    for it in range(iterations):
        inputs = dict(A=[], B=[])
        # NOTE: the reference interleaves RNG draws with compute; compute consumes no RNG, so
        # drawing phase-A inputs first, then phase-B inputs, reproduces the same stream.
        for _ in range(rounds):                                                       # :391-413
            ps = next_prompts()
            if use_dropout:
                flags = (torch.rand(len(ps)) < 0.1).tolist()                          # :394
                ps = ['' if f else p for f, p in zip(flags, ps)]
            z = torch.randn([len(ps), 4, lat, lat])                                   # :398
            noise = torch.randn_like(z)                                               # :399
            t = torch.randint(tmin, tmax, (len(ps),), dtype=torch.long)               # :413
            inputs['A'].append(dict(z=z, noise=noise, t=t, cond=embed(ps), uncond=embed([''] * len(ps))))
        for _ in range(rounds):                                                       # :472-484
            ps = next_prompts()
            z = torch.randn([len(ps), 4, lat, lat])                                   # :479
            noise = torch.randn_like(z)                                               # :480
            t = torch.randint(tmin, tmax, (len(ps),), dtype=torch.long)               # :484
            inputs['B'].append(dict(z=z, noise=noise, t=t, cond=embed(ps), uncond=embed([''] * len(ps))))
        hp['cur_nimg'] = cur_nimg
        out = sid_iteration_ref(nets, st, sched, inputs, hp)
        losses.append(('fake_score_Loss/loss', out['loss_fake']))
        losses.append(('G_Loss/loss', out['loss_G']))
        cur_nimg += batch_size                                                        # :567
        if on_iteration is not None:
            on_iteration(it, nets, inputs, out)
    return dict(losses=losses, nets=nets, opt_states=st)
