"""HIP ops against the reference's OWN in-tree network primitives (training/networks.py: GroupNorm :96, AttentionOp :113,
UNetBlock :134), through tests/golden/networks_blocks.npz (made by oracle/make_goldens.py::gen_blocks from the imported
reference).  These are the only pieces of SURVEY section 8 row A5 the reference holds itself; the chain below is the
product's ResnetBlock2D.forward + Transformer2DModel attention path (sid_lsg_amd/unet.py) written out on the ops:
    GroupNorm+SiLU -> conv3x3 (+bias +time-embedding projection broadcast) -> GroupNorm+SiLU -> conv3x3 (+bias +shortcut)
    [-> GroupNorm -> fused q|k|v projection -> softmax(QK^T d^-1/2)V -> output projection (+bias +residual)]
forward, input gradients and every parameter gradient; bf16 (production) and fp32 mode."""
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


def qkv_rows_to_sd(w, heads):
    c3 = w.shape[0]
    d = c3 // (3 * heads)
    return w.reshape(heads, d, 3, -1).permute(2, 0, 1, 3).reshape(c3, -1)


class P:
    """fp32 master + pre-bound gradient + compute copies, the way HipUNet2DCondition.materialize binds a layer."""

    def __init__(self, w, dev, cd, conv=False, bias=False):
        from sid_lsg_amd import ops
        w = w.to(dev).float()
        if conv:                                    # logical [Cout,Cin,3,3], physical [Cout,3,3,Cin]
            phys = w.permute(0, 2, 3, 1).contiguous()
            self.p = torch.nn.Parameter(phys.permute(0, 3, 1, 2))
            self.p.grad = torch.zeros_like(phys).permute(0, 3, 1, 2)
            self.w = phys.reshape(w.shape[0], -1).to(cd)
            self.wt = ops.transpose_w(phys, w.shape[0], w.shape[1], 9, dtype=cd)
        else:
            self.p = torch.nn.Parameter(w.contiguous())
            self.p.grad = torch.zeros_like(self.p)
            if not bias and w.ndim == 2:
                self.w = w.to(cd).contiguous()
                self.wt = ops.transpose_w(self.p, w.shape[0], w.shape[1], 1, dtype=cd)

    def grad(self):
        g = self.p.grad
        return g.detach().float().cpu()


@pytest.mark.parametrize('cd', [BF16, F32], ids=['bf16', 'fp32'])
@pytest.mark.parametrize('tag', ['res_proj', 'res_id', 'attn40', 'attn64'])
def test_hip_block_matches_reference_networks_py(dev, golden_dir, tag, cd):
    from sid_lsg_amd import ops
    g = np.load(os.path.join(golden_dir, 'networks_blocks.npz'))
    t = {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + '_')}
    g0, g1, g2, heads = (int(v) for v in g[f'{tag}_groups'])
    B, cin, H, W = t['x'].shape
    cout = t['y'].shape[1]
    M = B * H * W
    if cd == BF16:          # operands rounded to bf16 on BOTH sides would hide nothing here: the golden is fp32, so state bf16 bounds
        tol_y, tol_g, tol_p = 2.5e-2, 4e-2, 3e-2
    else:
        tol_y, tol_g, tol_p = 1e-4, 2e-4, 5e-4
    n0w, n0b = P(t['p_norm0.weight'], dev, cd, bias=True), P(t['p_norm0.bias'], dev, cd, bias=True)
    n1w, n1b = P(t['p_norm1.weight'], dev, cd, bias=True), P(t['p_norm1.bias'], dev, cd, bias=True)
    c0, c0b = P(t['p_conv0.weight'], dev, cd, conv=True), P(t['p_conv0.bias'], dev, cd, bias=True)
    c1, c1b = P(t['p_conv1.weight'], dev, cd, conv=True), P(t['p_conv1.bias'], dev, cd, bias=True)
    aff, affb = P(t['p_affine.weight'], dev, cd), P(t['p_affine.bias'], dev, cd, bias=True)
    x = t['x'].permute(0, 2, 3, 1).contiguous().to(dev).to(cd).requires_grad_()
    e = t['emb'].to(dev).to(cd).requires_grad_()
    hn, xk = ops.group_norm(x, n0w.p, n0b.p, g0, 1e-5, True, fork=True)
    tproj = ops.linear(e, aff.p, affb.p, aff.w, aff.wt, out_f32=True)
    h = ops.conv3x3_op(hn, c0.p, c0b.p, c0.w, c0.wt, None, tproj, 1, 0, False, None)
    sc = xk
    if 'p_skip.weight' in t:
        sk, skb = P(t['p_skip.weight'].flatten(1), dev, cd), P(t['p_skip.bias'], dev, cd, bias=True)
        sc = ops.linear(xk.view(M, cin), sk.p, skb.p, sk.w, sk.wt).view(B, H, W, cout)
    y = ops.conv3x3_op(ops.group_norm(h, n1w.p, n1b.p, g1, 1e-5, True), c1.p, c1b.p, c1.w, c1.wt, sc, None, 1, 0, False, None)
    if heads:
        n2w, n2b = P(t['p_norm2.weight'], dev, cd, bias=True), P(t['p_norm2.bias'], dev, cd, bias=True)
        qkv = P(qkv_rows_to_sd(t['p_qkv.weight'].flatten(1), heads), dev, cd)
        pr, prb = P(t['p_proj.weight'].flatten(1), dev, cd), P(t['p_proj.bias'], dev, cd, bias=True)
        hn2, yk = ops.group_norm(y, n2w.p, n2b.p, g2, 1e-5, False, fork=True)
        q3 = ops.linear(hn2.view(M, cout), qkv.p, None, qkv.w, qkv.wt)
        o = ops.self_attention(q3.view(B, H * W, 3 * cout), heads)
        y = ops.linear(o.view(M, cout), pr.p, prb.p, pr.w, pr.wt, res=yk.view(M, cout)).view(B, H, W, cout)

    def close(got, ref, tol, name):
        got, ref = got.detach().float().cpu(), ref.float()
        assert torch.isfinite(got).all(), name
        err = float((got - ref).abs().max() / (ref.abs().max() + 1e-12))
        assert err <= tol, f'{tag} {name}: {err:.3g} > {tol}'
        return err
    ey = close(y.permute(0, 3, 1, 2), t['y'], tol_y, 'forward')
    y.backward(t['dy'].permute(0, 2, 3, 1).contiguous().to(dev).to(cd))
    ex = close(x.grad.permute(0, 3, 1, 2), t['dx'], tol_g, 'dx')
    close(e.grad, t['demb'], tol_g, 'd emb')
    worst = 0.0
    for name, obj, ref in (('norm0.weight', n0w, None), ('norm0.bias', n0b, None), ('norm1.weight', n1w, None), ('norm1.bias', n1b, None),
                           ('conv0.weight', c0, None), ('conv0.bias', c0b, None), ('conv1.weight', c1, None), ('conv1.bias', c1b, None),
                           ('affine.weight', aff, None), ('affine.bias', affb, None)):
        worst = max(worst, close(obj.grad(), t[f'g_{name}'], tol_p, f'grad {name}'))
    if 'p_skip.weight' in t:
        worst = max(worst, close(sk.grad(), t['g_skip.weight'].flatten(1), tol_p, 'grad skip.weight'), close(skb.grad(), t['g_skip.bias'], tol_p, 'grad skip.bias'))
    if heads:
        worst = max(worst, close(qkv.grad(), qkv_rows_to_sd(t['g_qkv.weight'].flatten(1), heads), tol_p, 'grad qkv.weight'),
                    close(pr.grad(), t['g_proj.weight'].flatten(1), tol_p, 'grad proj.weight'), close(prb.grad(), t['g_proj.bias'], tol_p, 'grad proj.bias'),
                    close(n2w.grad(), t['g_norm2.weight'], tol_p, 'grad norm2.weight'), close(n2b.grad(), t['g_norm2.bias'], tol_p, 'grad norm2.bias'))
    print(f'{tag} {cd}: fwd {ey:.2e} dx {ex:.2e} worst param grad {worst:.2e}')
