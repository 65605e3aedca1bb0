"""Pin the oracle restatement (oracle/sid_ref.py) against golden vectors captured from the
reference itself (oracle/make_goldens.py).  CPU only; does not need /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, sid_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _embed(te, tok, prompts):
    ids = tok(list(prompts), padding='max_length', max_length=tok.model_max_length, truncation=True,
              return_tensors='pt').input_ids
    return te(ids)[0]


@pytest.mark.parametrize('cfg', ['tiny', 'tiny40'])
def test_glue_matches_reference(golden_dir, cfg):
    g = _load(golden_dir, f'glue_{cfg}.npz')
    unet, _, sched, te, tok = fixtures.factory(cfg)
    unet2 = fixtures.make_unet(cfg, seed=99)
    # weights regenerate bit-identically from the seed (else the fixture cannot be replayed)
    np.testing.assert_allclose(np.array(fixtures.checksum(unet)), g['weight_checksum'], rtol=1e-12)
    np.testing.assert_allclose(np.array(fixtures.checksum(unet2)), g['weight_checksum2'], rtol=1e-12)
    unet.eval().requires_grad_(False)
    unet2.eval().requires_grad_(False)
    for b in (1, 2):
        z, noise, t = (torch.from_numpy(g[f'b{b}_{k}']) for k in ('z', 'noise', 't'))
        prompts = [str(p) for p in g[f'b{b}_prompts']]
        cond, uncond = _embed(te, tok, prompts), _embed(te, tok, [''] * b)
        init_t = torch.full((b,), 625, dtype=torch.long)
        xhat = sid_ref.sampler_ref(unet, z, cond, init_t, sched)
        np.testing.assert_allclose(xhat.numpy(), g[f'b{b}_xhat'], rtol=1e-5, atol=1e-5)
        xhat = torch.from_numpy(g[f'b{b}_xhat'])
        for kappa in (1.0, 1.5, 4.5):
            for px0 in (True, False):
                y = sid_ref.denoise_ref(unet2, xhat, noise, cond, uncond, t, sched, predict_x0=px0, guidance_scale=kappa)
                np.testing.assert_allclose(y.numpy(), g[f'b{b}_k{kappa}_x0{int(px0)}'], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', ['k15_a1', 'k1_a12', 'k45_a1'])
def test_loop_matches_reference(golden_dir, name):
    g = _load(golden_dir, f'loop_{name}.npz')
    cfg = str(g['cfg'])
    np.testing.assert_allclose(np.array(fixtures.checksum(fixtures.make_unet(cfg))), g['weight_checksum'], rtol=1e-12)
    kw = {k[3:]: g[k].tolist() for k in g.files if k.startswith('kw_')}
    kw['kappa'] = tuple(kw['kappa'])
    res = sid_ref.training_loop_ref(lambda: fixtures.factory(cfg), [str(p) for p in g['prompts']], **kw)
    names = [n for n, _ in res['losses']]
    vals = np.array([v for _, v in res['losses']])
    assert names == [str(n) for n in g['loss_names']]
    # loss curve within 1e-3 rel of the reference (north_star); observed agreement is ~1e-6
    np.testing.assert_allclose(vals, g['loss_values'], rtol=1e-4, atol=1e-6)
    G, psi = res['nets']['G'], res['nets']['fake_score']
    np.testing.assert_allclose(list(G.parameters())[0].detach().numpy(), g['G_conv_in_w'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(list(psi.parameters())[0].detach().numpy(), g['fake_conv_in_w'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(list(G.parameters())[-1].detach().numpy(), g['G_last_b'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(np.array(fixtures.checksum(G))[1], g['G_checksum'][1], rtol=1e-6)
    np.testing.assert_allclose(np.array(fixtures.checksum(psi))[1], g['fake_score_checksum'][1], rtol=1e-6)


def test_loop_50_iterations_matches_reference(golden_dir):
    """The 50-iteration curve of the unmodified reference loop (tests/golden/loop_k15_a1_n50.npz): the restatement stays within
    north_star's 1e-3 of every recorded loss over the whole stretch (two fp32 implementations of 50 Adam(beta1 = 0) steps)."""
    g = _load(golden_dir, 'loop_k15_a1_n50.npz')
    cfg = str(g['cfg'])
    kw = {k[3:]: g[k].tolist() for k in g.files if k.startswith('kw_')}
    kw['kappa'] = tuple(kw['kappa'])
    res = sid_ref.training_loop_ref(lambda: fixtures.factory(cfg), [str(p) for p in g['prompts']], **kw)
    vals = np.array([v for _, v in res['losses']])
    ref = g['loss_values']
    assert vals.shape == ref.shape == (100,)
    rel = np.abs(vals - ref) / np.abs(ref)
    print('max rel error over 50 iterations: fake', rel[0::2].max(), 'G', rel[1::2].max())
    assert rel.max() < 1e-3


def test_infinite_sampler_matches_reference(golden_dir):
    g = _load(golden_dir, 'sampler.npz')
    for key in g.files:
        n, r, w, s = (int(x[1:]) for x in key.split('_'))
        it = sid_ref.infinite_sampler_ref(n, r, w, s)
        assert [next(it) for _ in range(64)] == g[key].tolist()


def test_scheduler_constants():
    # alpha_bar(625)=0.13776892, s0=0.37117236, s1=0.92856399; t=20, t=979
    from oracle.scheduler_ref import DDPMSchedulerRef
    ac = DDPMSchedulerRef().alphas_cumprod
    assert abs(float(ac[625]) - 0.13776892) < 2e-7
    assert abs(float(ac[20]) - 0.98131430) < 2e-7
    assert abs(float(ac[979]) - 0.00591277) < 2e-7


def _load_ref_block(g, tag):
    """Parameters / tensors of one reference UNetBlock case of networks_blocks.npz as torch tensors."""
    t = {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + '_')}
    return t, [int(v) for v in g[f'{tag}_groups']]


def qkv_rows_to_sd(w, heads):
    """reference UNetBlock qkv rows are (head, channel, {q,k,v}) interleaved (networks.py:177); SD's to_q|to_k|to_v are
    ({q,k,v}, head, channel)."""
    c3 = w.shape[0]
    d = c3 // (3 * heads)
    return w.reshape(heads, d, 3, -1).permute(2, 0, 1, 3).reshape(c3, -1)


@pytest.mark.parametrize('tag', ['res_proj', 'res_id', 'attn40', 'attn64'])
def test_unet_layers_match_reference_networks_py(golden_dir, tag):
    """Row A5 pieces that the reference DOES hold in-tree: training/networks.py UNetBlock (GroupNorm :96 + SiLU + Conv2d :47
    + Linear :30 + 1x1 skip + AttentionOp :113), run by oracle/make_goldens.py::gen_blocks.  The oracle's ResnetBlock2D /
    Attention layers (oracle/unet_ref.py, restating diffusers) must reproduce it with the same weights: forward,
    input gradients and every parameter gradient."""
    from oracle.unet_ref import Attention, ResnetBlock2D
    import torch.nn as nn
    import torch.nn.functional as F
    g = _load(golden_dir, 'networks_blocks.npz')
    t, (g0, g1, g2, heads) = _load_ref_block(g, tag)
    cin, cout, emb = t['x'].shape[1], t['y'].shape[1], t['emb'].shape[1]
    blk = ResnetBlock2D(cin, cout, emb, g0, 1e-5)
    blk.norm2 = nn.GroupNorm(g1, cout, eps=1e-5)
    names = {'norm1': 'norm0', 'conv1': 'conv0', 'time_emb_proj': 'affine', 'norm2': 'norm1', 'conv2': 'conv1', 'conv_shortcut': 'skip'}
    with torch.no_grad():
        for n, p in blk.named_parameters():
            mod, leaf = n.rsplit('.', 1)
            p.copy_(t[f'p_{names[mod]}.{leaf}'])
    x, raw = t['x'].clone().requires_grad_(), t['emb_raw'].clone().requires_grad_()
    y = blk(x, raw)
    attn = None
    if heads:
        norm = nn.GroupNorm(g2, cout, eps=1e-5)
        attn = Attention(cout, heads)
        with torch.no_grad():
            norm.weight.copy_(t['p_norm2.weight']); norm.bias.copy_(t['p_norm2.bias'])
            w = qkv_rows_to_sd(t['p_qkv.weight'].flatten(1), heads)
            attn.to_q.weight.copy_(w[:cout]); attn.to_k.weight.copy_(w[cout:2 * cout]); attn.to_v.weight.copy_(w[2 * cout:])
            attn.to_out[0].weight.copy_(t['p_proj.weight'].flatten(1)); attn.to_out[0].bias.copy_(t['p_proj.bias'])
        B, C, H, W = y.shape
        tok = norm(y).flatten(2).transpose(1, 2)
        y = y + attn(tok).transpose(1, 2).reshape(B, C, H, W)
    np.testing.assert_allclose(y.detach().numpy(), g[f'{tag}_y'], rtol=2e-4, atol=2e-5)
    y.backward(t['dy'])
    np.testing.assert_allclose(x.grad.numpy(), g[f'{tag}_dx'], rtol=2e-4, atol=2e-5)
    # d/d raw = d/d emb * silu'(raw)
    s = torch.sigmoid(t['emb_raw'])
    np.testing.assert_allclose(raw.grad.numpy(), (t['demb'] * s * (1 + t['emb_raw'] * (1 - s))).numpy(), rtol=2e-4, atol=2e-5)
    for n, p in blk.named_parameters():
        mod, leaf = n.rsplit('.', 1)
        np.testing.assert_allclose(p.grad.numpy(), g[f'{tag}_g_{names[mod]}.{leaf}'].reshape(p.shape), rtol=5e-4, atol=5e-5, err_msg=n)
    if heads:
        gw = torch.cat([attn.to_q.weight.grad, attn.to_k.weight.grad, attn.to_v.weight.grad])
        np.testing.assert_allclose(gw.numpy(), qkv_rows_to_sd(t['g_qkv.weight'].flatten(1), heads).numpy(), rtol=5e-4, atol=5e-5)
        np.testing.assert_allclose(attn.to_out[0].weight.grad.numpy(), g[f'{tag}_g_proj.weight'].reshape(cout, cout), rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize('case', sorted(fixtures.FULLSIZE_CASES))
def test_stored_fullsize_oracle_iterations_are_well_formed(golden_dir, case):
    """tests/golden/fullsize_<case>.npz (oracle/make_fullsize_fixtures.py; consumed by tests/test_gpu_unet.py::_iteration_parity): the
    record names the case it was made for, the bit-packed update signs cover exactly the sampled weights of the architecture
    (fixtures.sample_index over named_parameters order, network built on the meta device: no weights needed), the EMA tensors
    have the parameters' shapes, and the tiny-network counterpart of the stored quantities -- computed here from a live oracle
    iteration -- obeys what the GPU test assumes about them (almost every sampled weight moves by more than lr / 2 under
    Adam(beta1 = 0))."""
    from oracle.unet_ref import CONFIGS, UNet2DConditionRef
    fx = _load(golden_dir, f'fullsize_{case}.npz')
    cfg_name, lat, b, kappa = fixtures.FULLSIZE_CASES[case]
    assert [str(v) for v in fx['case']] == [cfg_name, str(lat), str(b), str(kappa), str(fixtures.FULLSIZE_LR)]
    assert np.isfinite(float(fx['loss_fake'])) and np.isfinite(float(fx['loss_G'])) and float(fx['loss_fake']) > 0
    with torch.device('meta'):
        net = UNet2DConditionRef(CONFIGS[cfg_name])
    n = sum(int(fixtures.sample_index(p.numel()).numel()) for p in net.parameters())
    shapes = {k: tuple(p.shape) for k, p in net.named_parameters()}
    for name in ('fake_score', 'G'):
        assert int(fx[name + '/n']) == n
        assert fx[name + '/sign'].shape == fx[name + '/big'].shape == ((n + 7) // 8,)
        big = np.unpackbits(fx[name + '/big'])[:n]
        assert big.mean() > 0.999          # beta1 = 0: |update| ~ lr for every weight with a non-zero gradient
    for k in fixtures.FULLSIZE_EMA_NAMES:
        assert tuple(fx['ema/' + k].shape) == shapes[k]
    assert fx['weight_checksum'].shape == (4,)


def test_sum_of_round_losses_equals_the_one_batch_loss():
    """oracle/make_fullsize_fixtures.py evaluates the batch-2 case as two accumulation rounds of one sample (`sum_round_losses`): on the
    tiny network the summed round losses and the updated weights equal the one-batch iteration's (fp32 summation order aside)."""
    import copy
    from oracle.scheduler_ref import DDPMSchedulerRef
    outs = []
    for split in (False, True):
        phi = fixtures.make_unet('tiny').eval().requires_grad_(False)
        psi = fixtures.make_unet('tiny', seed=77).requires_grad_(False)
        G = copy.deepcopy(phi)
        nets = dict(true_score=phi, fake_score=psi, G=G, G_ema=copy.deepcopy(G))
        st = dict(fake_score=[{} for _ in psi.parameters()], G=[{} for _ in G.parameters()])
        hp = fixtures.iteration_hp(2, 1, 2e-5, 1.5, 1.0)
        hp['cur_nimg'] = 0
        inputs = fixtures.iteration_inputs('tiny', 8, 2, 1, torch.Generator().manual_seed(5))
        if split:
            inputs = {ph: [{k: v[i:i + 1].contiguous() for k, v in inputs[ph][0].items()} for i in range(2)] for ph in inputs}
            hp['sum_round_losses'] = True
        out = sid_ref.sid_iteration_ref(nets, st, DDPMSchedulerRef(), inputs, hp)
        outs.append((out, [p.detach().clone() for p in G.parameters()]))
    (a, pa), (b2, pb) = outs
    assert abs(a['loss_fake'] - b2['loss_fake']) <= 1e-5 * abs(a['loss_fake'])
    assert abs(a['loss_G'] - b2['loss_G']) <= 1e-4 * abs(a['loss_fake'])
    same = sum(int((torch.sign(x - y) == 0).sum()) for x, y in zip(pa, pb))
    total = sum(x.numel() for x in pa)
    assert same > 0.99 * total          # Adam(beta1 = 0) steps are +-lr: identical wherever the gradient sign is
