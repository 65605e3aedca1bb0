"""Grouped launches (include/sidlsg_hip.h "grouped launches"): two networks of one architecture on one stacked batch.

The reference evaluates the frozen fake-score network and the frozen teacher on identical inputs with two `sid_sd_denoise` calls
(sid_training_loop.py:494-506).  Here they are ONE pass: every weight-bearing kernel gets both parameter sets and picks one per
block.  What is pinned:
  * kernels: a grouped launch produces, for each half of the stacked rows, the bits of the ordinary launch on that half with
    that set (same tiles, same K order) -- wherever the dispatcher picks the same kernel / split for both (asserted bit-equal),
    and within one bf16 rounding where the larger grid changes the split-K decision;
  * network: HipUNet2DCondition.forward_pair == two forward_nhwc calls, forward and input gradients;
  * step: SiDStep with the grouped pass == SiDStep with the two-stream path (losses and updated weights)."""
import contextlib
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from sid_lsg_amd._lib import lib
    lib.load()
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0, dev=None):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(*shape, generator=g) * scale).to(BF16)
    return t.to(dev) if dev is not None else t


def same_or_close(got, ref, what):
    if torch.equal(got, ref):
        return 'bit-equal'
    err = float((got.float() - ref.float()).abs().max() / (ref.float().abs().max() + 1e-12))
    assert err < 1e-2, f'{what}: grouped launch differs from the two ordinary launches by {err:.3g} of max'
    return f'within {err:.1e}'


# (M_half, N, K): tile-aligned halves on the direct-to-LDS kernel, the A-stationary kernel, unaligned halves (time-embedding MLP rows,
# 77-token text rows), split-K shapes (few tiles, long K), ragged N
@pytest.mark.parametrize('Mh,N,K', [(4096, 320, 320), (32768, 2560, 320), (16 * 77, 640, 768), (16, 1280, 320), (2, 1280, 1280), (1024, 1280, 5120),
                                    (300, 136, 72), (1000, 1280, 2560), (64, 640, 5760)])
@pytest.mark.parametrize('epi', ['bias', 'bias+res', 'bias+rowvec', 'none'])
def test_grouped_gemm_equals_two_launches(dev, Mh, N, K, epi):
    from sid_lsg_amd import ops
    if epi != 'bias' and Mh > 8192:
        pytest.skip('large shape: one epilogue variant is enough')
    a = rnd(2 * Mh, K, seed=1, dev=dev)
    w0, w1 = rnd(N, K, seed=2, scale=K ** -0.5, dev=dev), rnd(N, K, seed=3, scale=K ** -0.5, dev=dev)
    b0 = b1 = res = rv = None
    rpb = 1
    if epi != 'none':
        b0, b1 = rnd(N, seed=4, dev=dev).float(), rnd(N, seed=5, dev=dev).float()
    if epi == 'bias+res':
        res = rnd(2 * Mh, N, seed=6, dev=dev)
    if epi == 'bias+rowvec':
        rpb = 64 if Mh % 64 == 0 else 1
        rv = rnd(2 * Mh // rpb, N, seed=7, dev=dev).float()
    got = ops.gemm(a, ops.Pair(w0, w1), bias=ops.Pair(b0, b1) if b0 is not None else None, res=res, rowvec=rv, rows_per_batch=rpb)
    refs = []
    for h, (w, b) in enumerate(((w0, b0), (w1, b1))):
        sl = slice(h * Mh, (h + 1) * Mh)
        refs.append(ops.gemm(a[sl], w, bias=b, res=res[sl] if res is not None else None,
                             rowvec=rv[h * Mh // rpb:(h + 1) * Mh // rpb] if rv is not None else None, rows_per_batch=rpb))
    how = same_or_close(got, torch.cat(refs), f'gemm {Mh}x{N}x{K} {epi}')
    # and it really used the second set for the second half
    wrong = ops.gemm(a[Mh:], w0, bias=b0, res=res[Mh:] if res is not None else None, rowvec=rv[Mh // rpb:] if rv is not None else None, rows_per_batch=rpb)
    assert not torch.equal(got[Mh:], wrong)
    print(f'gemm {2 * Mh}x{N}x{K} [{epi}]: {how}')


@pytest.mark.parametrize('Bh,H,W,Cin,Cout,stride,ups', [(2, 16, 16, 64, 320, 1, 0), (8, 8, 8, 1280, 1280, 1, 0), (1, 32, 32, 320, 640, 1, 0), (2, 16, 16, 128, 160, 2, 0),
                                                       (2, 16, 16, 64, 160, 1, 1), (2, 8, 8, 8, 320, 1, 0), (2, 8, 8, 320, 8, 1, 0), (1, 5, 7, 72, 136, 1, 0)])
def test_grouped_conv_equals_two_launches(dev, Bh, H, W, Cin, Cout, stride, ups):
    from sid_lsg_amd import ops
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = rnd(2 * Bh, Hs, Ws, Cin, seed=1, dev=dev)
    w0, w1 = rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5, dev=dev), rnd(Cout, 9 * Cin, seed=3, scale=(9 * Cin) ** -0.5, dev=dev)
    b0, b1 = rnd(Cout, seed=4, dev=dev).float(), rnd(Cout, seed=5, dev=dev).float()
    rv = rnd(2 * Bh, Cout, seed=6, dev=dev).float()
    got = ops.conv3x3(x, ops.Pair(w0, w1), bias=ops.Pair(b0, b1), rowvec=rv, stride=stride, ups=ups)
    ref = torch.cat([ops.conv3x3(x[:Bh], w0, bias=b0, rowvec=rv[:Bh], stride=stride, ups=ups),
                     ops.conv3x3(x[Bh:], w1, bias=b1, rowvec=rv[Bh:], stride=stride, ups=ups)])
    print(f'conv B={2 * Bh} {H}x{W} {Cin}->{Cout} s{stride} u{ups}: {same_or_close(got, ref, "conv")}')
    assert not torch.equal(got[Bh:], ops.conv3x3(x[Bh:], w0, bias=b0, rowvec=rv[Bh:], stride=stride, ups=ups))


@pytest.mark.parametrize('Bh,HW,C,G,silu', [(2, 4096, 320, 32, True), (8, 64, 1280, 32, True), (1, 1024, 640, 32, False), (1, 16, 64, 8, True)])
def test_grouped_groupnorm_equals_two_launches(dev, Bh, HW, C, G, silu):
    from sid_lsg_amd import ops
    x = rnd(2 * Bh, HW, C, seed=1, dev=dev)
    par = [torch.nn.Parameter(rnd(C, seed=s, dev=dev).float() + (1.0 if s < 4 else 0.0), requires_grad=False) for s in (2, 3, 4, 5)]
    g0, g1, be0, be1 = par
    dy, dk = rnd(2 * Bh, HW, C, seed=6, dev=dev), rnd(2 * Bh, HW, C, seed=7, dev=dev)
    xg = x.clone().requires_grad_()
    with ops.dual_networks({id(g0): g1, id(be0): be1}):
        y, xk = ops.group_norm(xg, g0, be0, G, 1e-5, silu, fork=True)
    (y.float() * dy.float() + xk.float() * dk.float()).sum().backward()
    ys, dxs = [], []
    for h, (g, be) in enumerate(((g0, be0), (g1, be1))):
        xh = x[h * Bh:(h + 1) * Bh].clone().requires_grad_()
        yh, xkh = ops.group_norm(xh, g, be, G, 1e-5, silu, fork=True)
        (yh.float() * dy[h * Bh:(h + 1) * Bh].float() + xkh.float() * dk[h * Bh:(h + 1) * Bh].float()).sum().backward()
        ys.append(yh.detach())
        dxs.append(xh.grad)
    # the statistics' chunking depends on the batch size of the launch: same arithmetic, possibly another partial-sum order
    print(f'groupnorm B={2 * Bh} HW={HW} C={C}: fwd {same_or_close(y.detach(), torch.cat(ys), "gn fwd")}, bwd {same_or_close(xg.grad, torch.cat(dxs), "gn bwd")}')


@pytest.mark.parametrize('rows_h,C', [(4096, 320), (1024, 1280), (16 * 77, 640), (6, 64), (2, 320)])
def test_grouped_layernorm_equals_two_launches(dev, rows_h, C):
    from sid_lsg_amd import ops
    x = rnd(2 * rows_h, C, seed=1, dev=dev)
    g0, g1, be0, be1 = [torch.nn.Parameter(rnd(C, seed=s, dev=dev).float() + (1.0 if s < 4 else 0.0), requires_grad=False) for s in (2, 3, 4, 5)]
    dy, dk = rnd(2 * rows_h, C, seed=6, dev=dev), rnd(2 * rows_h, C, seed=7, dev=dev)
    xg = x.clone().requires_grad_()
    with ops.dual_networks({id(g0): g1, id(be0): be1}):
        y, xk = ops.layer_norm(xg, g0, be0, 1e-5, fork=True)
    (y.float() * dy.float() + xk.float() * dk.float()).sum().backward()
    ys, dxs = [], []
    for h, (g, be) in enumerate(((g0, be0), (g1, be1))):
        sl = slice(h * rows_h, (h + 1) * rows_h)
        xh = x[sl].clone().requires_grad_()
        yh, xkh = ops.layer_norm(xh, g, be, 1e-5, fork=True)
        (yh.float() * dy[sl].float() + xkh.float() * dk[sl].float()).sum().backward()
        ys.append(yh.detach())
        dxs.append(xh.grad)
    assert torch.equal(y.detach(), torch.cat(ys)) and torch.equal(xg.grad, torch.cat(dxs)), 'row-wise op: must be bit-equal'


# (rows per half, model width C): FeedForward = GEGLU(x W1^T + b1) W2^T + b2 + res with W1 [8 C, C], W2 [C, 4 C].  16384 x 640 / 4096 x 1280: both fused
# kernels admitted (32x32 / 16x16 stages at batch 16); 32768 x 320: forward fusion only without h, fused backward; 300 x 320: neither (few tiles)
@pytest.mark.parametrize('Mh,C', [(16384, 640), (4096, 1280), (32768, 320), (300, 320)])
@pytest.mark.parametrize('grad', [True, False])
def test_grouped_feed_forward_equals_two_single_passes(dev, Mh, C, grad):
    """ops.feed_forward inside dual_networks (one node over sidlsg_gemm_geglu_bf16_g2 / sidlsg_gemm_geglu_bwd_bf16_g2, ops._FeedForwardG2) against
    the two frozen single-set passes on the halves of the stacked batch: output and input gradient; also against the unfused grouped chain
    (SIDLSG_FF_G2=0 path: grouped GEMMs + stand-alone GEGLU kernels), which it replaced."""
    from sid_lsg_amd import ops
    P = lambda t: torch.nn.Parameter(t, requires_grad=False)      # noqa: E731
    sets = []
    for s0 in (10, 20):
        w1, w2 = rnd(8 * C, C, seed=s0, scale=C ** -0.5, dev=dev), rnd(C, 4 * C, seed=s0 + 1, scale=(4 * C) ** -0.5, dev=dev)
        b1, b2 = rnd(8 * C, seed=s0 + 2, dev=dev).float(), rnd(C, seed=s0 + 3, dev=dev).float()
        sets.append(dict(w1=P(w1.float()), b1=P(b1), w1_16=w1, w1_16t=w1.t().contiguous(), w2=P(w2.float()), b2=P(b2), w2_16=w2, w2_16t=w2.t().contiguous()))
    a, b = sets
    pmap = {id(a[k]): b[k] for k in a}
    x, res, dout = rnd(2 * Mh, C, seed=1, dev=dev), rnd(2 * Mh, C, seed=2, dev=dev), rnd(2 * Mh, C, seed=3, dev=dev)

    def run(xin, rin, st, dual, dslice):
        xg, rg = xin.clone().requires_grad_(grad), rin.clone().requires_grad_(grad)
        ctxm = ops.dual_networks(pmap) if dual else contextlib.nullcontext()
        with torch.set_grad_enabled(grad), ctxm:
            out = ops.feed_forward(xg, st['w1'], st['b1'], st['w1_16'], st['w1_16t'], st['w2'], st['b2'], st['w2_16'], st['w2_16t'], rg)
        if grad:
            (out.float() * dout[dslice].float()).sum().backward()
        return out.detach(), (xg.grad, rg.grad) if grad else (None, None)
    full = slice(0, 2 * Mh)
    got, (gx, gr) = run(x, res, a, True, full)
    parts = [run(x[sl], res[sl], st, False, sl) for sl, st in ((slice(0, Mh), a), (slice(Mh, 2 * Mh), b))]
    ref = torch.cat([p[0] for p in parts])
    msg = [f'out {same_or_close(got, ref, "feed-forward output")}']
    if grad:
        msg.append(f'dx {same_or_close(gx, torch.cat([p[1][0] for p in parts]), "feed-forward dx")}')
        assert torch.equal(gr, dout), 'the residual gradient is the output gradient'
    saved, ops._FF_G2 = ops._FF_G2, False
    try:
        old, (ox, _) = run(x, res, a, True, full)
    finally:
        ops._FF_G2 = saved
    msg.append(f'vs unfused grouped chain: out {same_or_close(got, old, "fused vs unfused")}' + (f', dx {same_or_close(gx, ox, "fused vs unfused dx")}' if grad else ''))
    print(f'feed-forward 2 x {Mh} x {C} grad={grad}: ' + '; '.join(msg))


@pytest.mark.parametrize('cfg_name,lat,B', [('tiny', 8, 2), ('tiny40', 16, 4), ('tiny21', 16, 2)])
def test_forward_pair_equals_two_forwards(dev, cfg_name, lat, B):
    """The whole network: one grouped pass over [x ; x] with (psi, phi) == psi(x), phi(x); outputs and input gradients."""
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    cfg = CONFIGS[cfg_name]
    psi = HipUNet2DCondition(cfg).materialize(dev, seed=11).requires_grad_(False)
    phi = HipUNet2DCondition(cfg).materialize(dev, seed=12).requires_grad_(False)
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(B, lat, lat, 8)
    x[..., :4] = torch.randn(B, lat, lat, 4, generator=g)
    x = x.to(dev).to(BF16)
    t = torch.randint(20, 980, (B,), generator=g).to(dev)
    ctx = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g).to(dev).to(BF16)
    dy = torch.randn(2, B, lat * lat, 8, generator=g).to(dev)
    xa = x.clone().requires_grad_()
    ea, eb = psi.forward_pair(phi, xa, t, ctx)
    (ea * dy[0] + eb * dy[1]).sum().backward()
    xb = x.clone().requires_grad_()
    ra, rb = psi.forward_nhwc(xb, t, ctx), phi.forward_nhwc(xb, t, ctx)
    (ra * dy[0] + rb * dy[1]).sum().backward()

    def rel(a, b):
        return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    e = (rel(ea, ra), rel(eb, rb), rel(xa.grad, xb.grad))
    print(f'{cfg_name}: eps psi {e[0]:.2e}  eps phi {e[1]:.2e}  input gradient {e[2]:.2e}')
    # same kernels on the same tiles; only split-K decisions (grid-size dependent) may differ -> one bf16 rounding at a few places
    assert max(e[0], e[1]) < 2e-3 and e[2] < 5e-3
    assert rel(eb, ra) > 0.1, 'the second half must have been evaluated with the second network'


def test_grouped_frozen_pass_changes_nothing_in_the_step(dev, monkeypatch):
    """SiDStep with the grouped fake-score + teacher pass against the two-stream path it replaces: same losses (to the bf16 rounding of
    split-K order) and the same updated weights, over two iterations with two accumulation rounds."""
    from sid_lsg_amd.optim import FusedAdamEMA
    from sid_lsg_amd.scheduler import DDPMScheduler
    from sid_lsg_amd.sid_step import SiDStep
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    cfg_name, lat, b, lr = 'tiny40', 16, 2, 2e-5
    cfg = CONFIGS[cfg_name]
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('SIDLSG_GROUPED_FROZEN', mode)
        phi = HipUNet2DCondition(cfg).materialize(dev, seed=1).requires_grad_(False)
        psi = HipUNet2DCondition(cfg).materialize(dev, seed=2)
        G = phi.clone_network()
        G_ema = phi.clone_network(with_grad_buffers=False)
        step = SiDStep(G, psi, phi, G_ema, DDPMScheduler().to(dev), FusedAdamEMA(psi.parameters(), lr=lr), FusedAdamEMA(G.parameters(), lr=lr),
                       alpha=1.0, cfg_train_fake=1.5, cfg_eval_fake=1.5, cfg_eval_real=4.5, batch_gpu_total=2 * b, init_timestep=625)
        assert step._use_grouped(b) == (mode == '1')
        gen = torch.Generator().manual_seed(3)
        losses = []
        for it in range(2):
            inputs = {ph: [dict(z=torch.randn(b, 4, lat, lat, generator=gen).to(dev), noise=torch.randn(b, 4, lat, lat, generator=gen).to(dev),
                                t=torch.randint(20, 980, (b,), generator=gen).to(dev),
                                cond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=gen).to(dev).to(BF16),
                                uncond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=gen).to(dev).to(BF16)) for _ in range(2)]
                      for ph in ('A', 'B')}
            lf, lg = step.iteration(inputs, ema_beta=0.5)
            losses += [float(lf), float(lg)]
        torch.cuda.synchronize()
        out[mode] = dict(losses=np.array(losses), G=G.flat_params.clone(), psi=psi.flat_params.clone(), ema=G_ema.flat_params.clone())
    a, g = out['0'], out['1']
    rel = np.abs(a['losses'] - g['losses']) / np.abs(a['losses'])
    print(f'two-stream {a["losses"]}  grouped {g["losses"]}  rel {rel}')
    assert rel.max() < 2e-3
    for k in ('G', 'psi', 'ema'):
        d = float((a[k] - g[k]).abs().max())
        frac = float(((a[k] - g[k]).abs() > 0.5 * lr).float().mean())
        print(f'{k}: max weight difference {d / lr:.2f} lr, {frac:.4%} of the weights differ by more than lr / 2')
        assert d <= 2.01 * lr * 2 and frac < 0.02     # an Adam(beta1 = 0) step is +-lr: only near-zero gradients flip
