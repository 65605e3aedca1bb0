"""fp8-weight forward path (BASELINE.json configs[4]; include/sidlsg_hip.h "fp8-weight contractions").

The reference has no fp8 mode (sid_training_loop.py:205 knows fp32 / fp16), so there is no reference golden to pin: the
kernels are checked EXACTLY against what they are specified to compute -- e4m3 per-row weights, activations converted
to e4m3 with unit scale, fp32 accumulation -- restated with torch's float8_e4m3fn casts in fp32, and the end-to-end
deviation of a frozen network from its own bf16 path is bounded and printed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
F8 = torch.float8_e4m3fn


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from sid_lsg_amd._lib import lib
    lib.load()
    return torch.device('cuda:0')


def q_act(a):
    """activation side of the contract: bf16 -> e4m3 (unit scale, round-to-nearest-even, saturating) -> fp32"""
    return a.float().clamp(-448, 448).to(F8).float()


def test_quantize_rows_matches_torch_e4m3(dev):
    from sid_lsg_amd import ops
    torch.manual_seed(0)
    w = (torch.randn(200, 320, device=dev) * torch.rand(200, 1, device=dev) * 0.1).to(BF16)
    w[7] = 0
    q = ops.Fp8Weight(w)
    amax = w.float().abs().amax(1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert float(((q.scale - scale).abs() / scale).max()) <= 2.4e-7        # one ulp: a / 448 vs a * (1 / 448)
    ref = (w.float() * (1.0 / q.scale)[:, None]).to(F8)
    got = q.q.view(F8)
    same = (got.view(torch.uint8) == ref.view(torch.uint8)).float().mean().item()
    assert same == 1.0, same
    # and the representation error is the format's: <= 2^-4 relative to the row maximum
    err = (q.dequantize() - w.float()).abs().amax(1) / amax.clamp_min(1e-30)
    assert float(err.max()) <= 2 ** -4 + 1e-6


@pytest.mark.parametrize('M,N,K', [(256, 320, 320), (1000, 200, 64), (77, 640, 768), (4096, 1280, 1280), (130, 24, 16), (513, 136, 2560)])
def test_gemm_fp8w_exact_against_its_contract(dev, M, N, K):
    from sid_lsg_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(BF16)
    rv = torch.randn(4, N, device=dev)
    rpb = (M + 3) // 4
    q = ops.Fp8Weight(w)
    ref = (q_act(a).double() @ q.dequantize().double().t()).float()
    got = ops.gemm(a, q, out_f32=True)
    scale = float(ref.abs().max())
    # the fp8 MFMA aligns its 32 products to the largest one before adding (observed ~2^-14.5 relative), it is not an fp32 FMA chain
    assert float((got - ref).abs().max()) <= 1e-4 * scale * max(1.0, (K / 256) ** 0.5), 'plain'
    got = ops.gemm(a, q, bias=bias, res=res, rowvec=rv, rows_per_batch=rpb)
    full = ref + bias + res.float() + rv.repeat_interleave(rpb, 0)[:M]
    assert got.dtype == BF16
    assert float((got.float() - full).abs().max()) <= 2 ** -8 * float(full.abs().max()) * 1.01, 'epilogue'


def test_gemm_fp8w_saturates_large_activations(dev):
    """Activations past the e4m3 range (the un-normalised residual stream of real weights can exceed 448) are clamped to
    +-448 before the conversion -- the contract's q_act -- instead of turning into NaN."""
    from sid_lsg_amd import ops
    torch.manual_seed(3)
    M, N, K = 256, 64, 128
    a = torch.randn(M, K, device=dev) * 300.0
    a[0, :8] = torch.tensor([449., -449., 1e4, -1e4, 3e38, -3e38, 448., -448.], device=dev)
    a = a.to(BF16)
    assert float(a.float().abs().max()) > 448
    q = ops.Fp8Weight((torch.randn(N, K, device=dev) * 0.05).to(BF16))
    got = ops.gemm(a, q, out_f32=True)
    assert torch.isfinite(got).all(), 'overflowing activations must saturate, not become NaN'
    ref = (q_act(a).double() @ q.dequantize().double().t()).float()
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize('B,H,cin,cout,stride,ups', [(2, 16, 64, 64, 1, 0), (1, 32, 320, 320, 1, 0), (2, 16, 32, 48, 2, 0), (2, 16, 128, 64, 1, 1),
                                                     (1, 8, 2560, 1280, 1, 0), (3, 9, 16, 8, 1, 0)])
def test_conv3x3_fp8w_exact_against_its_contract(dev, B, H, cin, cout, stride, ups):
    from sid_lsg_amd import ops
    torch.manual_seed(cin + cout)
    Hs = H // 2 if ups else H
    x = torch.randn(B, Hs, Hs, cin, device=dev).to(BF16)
    w = (torch.randn(cout, 3, 3, cin, device=dev) * 0.05).to(BF16)
    bias = torch.randn(cout, device=dev)
    q = ops.Fp8Weight(w.view(cout, 9 * cin))
    xin = q_act(x).permute(0, 3, 1, 2)
    if ups:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode='nearest')
    wd = q.dequantize().view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin.double(), wd.double(), bias.double(), stride=stride, padding=1).float().permute(0, 2, 3, 1)
    got = ops.conv3x3(x, q, bias=bias, stride=stride, ups=ups, out_f32=True)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) * max(1.0, (9 * cin / 256) ** 0.5)


def test_fp8_is_refused_where_it_does_not_apply(dev):
    from sid_lsg_amd import ops, unet as U
    with pytest.raises(RuntimeError):
        ops.Fp8Weight(torch.zeros(8, 24, device=dev, dtype=BF16))           # K % 16
    net = U.HipUNet2DCondition(U.CONFIGS['tiny40']).materialize(dev, seed=1).requires_grad_(True)
    with pytest.raises(RuntimeError):
        net.enable_fp8_weights()                                             # trains -> keeps bf16
    q = ops.Fp8Weight(torch.zeros(8, 32, device=dev, dtype=BF16))
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 32, device=dev), q)                          # fp32 activations


@pytest.mark.parametrize('cfg_name', ['tiny', 'tiny40', 'tiny21'])
def test_frozen_unet_fp8_weights_stay_close_to_bf16(dev, cfg_name):
    from sid_lsg_amd import unet as U
    cfg = U.CONFIGS[cfg_name]
    net = U.HipUNet2DCondition(cfg).materialize(dev, seed=3, with_grad_buffers=False).requires_grad_(False)
    torch.manual_seed(0)
    B, lat = 4, 16
    x = torch.zeros(B, lat, lat, net.CIN_PAD, device=dev)
    x[..., :cfg.in_channels] = torch.randn(B, lat, lat, cfg.in_channels, device=dev)
    x = x.to(BF16)
    t = torch.tensor([999, 500, 250, 20], device=dev)
    ctx = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, device=dev).to(BF16)
    with torch.no_grad():
        y16 = net.forward_nhwc(x, t, ctx).float()
        n = net.enable_fp8_weights()
        assert n > 10 and net.enable_fp8_weights() == n
        y8 = net.forward_nhwc(x, t, ctx).float()
        # a reload of the same weights must re-quantise, not keep stale bytes
        net.refresh_compute_weights()
        y8b = net.forward_nhwc(x, t, ctx).float()
    assert torch.equal(y8, y8b)
    rel = float((y8 - y16).norm() / y16.norm())
    print(f'{cfg_name}: {n} layers in fp8, output deviation from bf16 (relative L2) {rel:.3e}')
    # W8A8 with 3 mantissa bits on both operands: ~3 % per contraction; through the stacked layers of a random-init
    # network: 0.076-0.078 with the default 'normalized' policy (0.096-0.10 with 'all'); 0.15 is the bound
    assert torch.isfinite(y8).all() and 0 < rel < 0.15


def test_fp8_policy_keeps_unnormalised_inputs_in_bf16(dev):
    """Default policy 'normalized': only contractions fed by a GroupNorm(+SiLU) / LayerNorm output (or the text states) run
    in e4m3; the ones fed by the residual stream or an activation product (to_out, the GEGLU output projection, proj_out,
    shortcut / resampling convs, the time embedding) keep bf16 weights AND bf16 activations -- an activation beyond +-448
    there would otherwise saturate (the cast clamps instead of producing NaN, but the value is lost).  'all' converts every
    contraction the format allows and deviates more."""
    from sid_lsg_amd import ops, unet as U
    cfg = U.CONFIGS['tiny40']
    torch.manual_seed(0)
    B, lat = 4, 16
    x = torch.zeros(B, lat, lat, 8, device=dev)
    x[..., :cfg.in_channels] = torch.randn(B, lat, lat, cfg.in_channels, device=dev)
    x = x.to(BF16)
    t = torch.tensor([999, 500, 250, 20], device=dev)
    ctx = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, device=dev).to(BF16)
    out = {}
    for policy in ('normalized', 'all'):
        net = U.HipUNet2DCondition(cfg).materialize(dev, seed=3, with_grad_buffers=False).requires_grad_(False)
        with torch.no_grad():
            y16 = net.forward_nhwc(x, t, ctx).float()
            n = net.enable_fp8_weights(policy)
            y8 = net.forward_nhwc(x, t, ctx).float()
        out[policy] = (n, float((y8 - y16).norm() / y16.norm()))
        is8 = lambda w: isinstance(w, ops.Fp8Weight)      # noqa: E731
        for name, mod in net.named_modules():
            if isinstance(mod, U.Attention):
                assert is8(mod.fused['w16'])
                assert policy == 'all' or not is8(mod.to_out[0].w16), name
            elif isinstance(mod, U.FeedForward):
                assert is8(mod.net[0].proj.w16)
                assert policy == 'all' or not is8(mod.net[2].w16), name
            elif isinstance(mod, U.ResnetBlock2D):
                assert is8(mod.conv1.w16) and is8(mod.conv2.w16)
                if mod.conv_shortcut is not None:
                    assert policy == 'all' or not is8(mod.conv_shortcut.w16), name
            elif isinstance(mod, U.Transformer2DModel):
                assert is8(mod.proj_in.w16)
                assert policy == 'all' or not is8(mod.proj_out.w16), name
        assert is8(net.fused['w16']) == (policy == 'all')                  # the time-embedding projections
    print({k: (n, f'{r:.3e}') for k, (n, r) in out.items()})
    assert 10 < out['normalized'][0] < out['all'][0]
    assert 0 < out['normalized'][1] < out['all'][1]
    with pytest.raises(ValueError):
        U.HipUNet2DCondition(cfg).materialize(dev, seed=3, with_grad_buffers=False).requires_grad_(False).enable_fp8_weights('some')


@pytest.mark.parametrize('M,N,K', [(300, 160, 128), (1232, 640, 768), (4096, 960, 320), (2048, 320, 1280), (130, 480, 48)])
def test_gemm_mx8_contract(dev, M, N, K):
    """sidlsg_gemm_mx8: both operands e4m3 on the MX MFMA (unit block scales), per-output-channel weight scale in the epilogue.
    Exact to its contract: fp64 on the SAME quantised operands, up to the fp32 accumulation order and the bf16 output
    rounding (2^-9 relative).  Ragged M and K tails (K = 320, 48: the last 128-byte K-tile is partly past the row)."""
    from sid_lsg_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev).to(BF16)
    a8 = ops.cast_fp8(a)
    # the cast: round to nearest within half an e4m3 ulp (2^-4 relative for normals), saturating at +-448
    av, a8v = a.float(), a8.view(torch.float8_e4m3fn).float()
    assert float(((a8v - av).abs() / av.abs().clamp_min(2.0 ** -6)).max()) <= 2.0 ** -4 + 1e-6
    big = ops.cast_fp8(torch.tensor([[1e4, -1e4, 448.0, -500.0, 0.0, 1.0, -1.0, 3e38]], device=dev, dtype=BF16)).view(torch.float8_e4m3fn).float()
    assert big.tolist() == [[448.0, -448.0, 448.0, -448.0, 0.0, 1.0, -1.0, 448.0]]
    w8 = ops.Fp8Weight(w)
    ref = a8v.double() @ w8.dequantize().double().t() + bias.double()
    got = ops.gemm_mx8(a8, w8, bias=bias, out_f32=True)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * max(1.0, (K / 256) ** 0.5)
    got16 = ops.gemm_mx8(a8, w8, bias=bias, res=res)
    ref16 = ref + res.double()
    assert float((got16.double() - ref16).abs().max()) <= 2.0 ** -8 * float(ref16.abs().max())
    with pytest.raises(RuntimeError):
        ops.gemm_mx8(a8, ops.Fp8Weight(torch.zeros(128, K, device=dev, dtype=BF16)))        # N % 160


def test_norm_linear_mx8_matches_the_unfused_ops(dev):
    """ops.norm_linear_mx8 (norm writes e4m3, MX GEMM, one autograd node) against the same computation spelled out: forward
    = gemm_mx8(cast_fp8(norm(x))) up to the rounding of the bf16 intermediate the fused op does not have; backward = the
    plain norm + linear backward (bf16 backward-data operand), fork gradient included."""
    from sid_lsg_amd import ops
    g = torch.Generator().manual_seed(5)
    for groups, shape, N in ((0, (512, 320), 960), (32, (2, 16, 16, 320), 320)):
        C = shape[-1]
        x = torch.randn(*shape, generator=g).to(dev).to(BF16)
        gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        weight = (torch.randn(N, C, generator=g) * 0.05).to(dev)
        w16 = weight.to(BF16)
        w16t = w16.t().contiguous()
        bias = torch.randn(N, generator=g).to(dev)
        w8 = ops.Fp8Weight(w16)
        dy = torch.randn(x.numel() // C, N, generator=g).to(dev).to(BF16)
        dk = torch.randn(*shape, generator=g).to(dev).to(BF16)
        xa = x.clone().requires_grad_()
        y, keep = ops.norm_linear_mx8(xa, gamma, beta, 1e-5, w8, bias, w16t, weight, groups=groups, silu=False, fork=True)
        torch.autograd.backward([y, keep], [dy, dk])
        xb = x.clone().requires_grad_()
        if groups:
            hn, keep_b = ops.group_norm(xb, gamma, beta, groups, 1e-5, False, fork=True)
        else:
            hn, keep_b = ops.layer_norm(xb, gamma, beta, 1e-5, fork=True)
        yb8 = ops.gemm_mx8(ops.cast_fp8(hn.detach().view(-1, C)), w8, bias=bias)
        rel = float((y.detach().float() - yb8.float()).norm() / yb8.float().norm())
        print(f'groups {groups}: forward deviation fused vs (bf16 norm -> cast -> mx8) {rel:.2e}')
        assert rel < 2e-2          # double rounding (fp32 -> bf16 -> e4m3) flips a few e4m3 roundings
        yb = ops.linear(hn.view(-1, C), weight, bias, w16, w16t)
        torch.autograd.backward([yb, keep_b], [dy, dk])
        assert torch.equal(xa.grad, xb.grad)                     # the backward IS the bf16 one
        dev8 = float((y.detach().float() - yb.detach().float()).norm() / yb.detach().float().norm())
        print(f'groups {groups}: e4m3 x e4m3 projection vs bf16 projection {dev8:.2e}')
        assert 0 < dev8 < 6e-2


def test_fp8_teacher_data_gradient_with_mx8_activations(dev, monkeypatch):
    """Phase B differentiates THROUGH the frozen teacher (data gradient only).  With e4m3 weights the normalised activations
    travel as e4m3 too where the shapes allow (tiny40: the 160- and 320-channel stages): output and input gradient must stay
    close to the bf16 network's, and the MX path must actually be in use (and removable with SIDLSG_MX8=0)."""
    from sid_lsg_amd import unet as U
    cfg = U.CONFIGS['tiny40']
    torch.manual_seed(0)
    B, lat = 4, 16
    x = torch.zeros(B, lat, lat, 8, device=dev)
    x[..., :cfg.in_channels] = torch.randn(B, lat, lat, cfg.in_channels, device=dev)
    x = x.to(BF16)
    t = torch.tensor([999, 500, 250, 20], device=dev)
    ctx = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, device=dev).to(BF16)
    dy = torch.randn(B, lat * lat, 8, device=dev)
    res = {}
    for mode in ('bf16', 'mx8', 'fp8w'):
        monkeypatch.setenv('SIDLSG_MX8', '0' if mode == 'fp8w' else '1')
        net = U.HipUNet2DCondition(cfg).materialize(dev, seed=3, with_grad_buffers=False).requires_grad_(False)
        if mode != 'bf16':
            net.enable_fp8_weights()
        n_mx = sum(int(getattr(m, 'mx8', False)) for m in net.modules())
        xa = x.clone().requires_grad_()
        y = net.forward_nhwc(xa, t, ctx)
        y.backward(dy)
        res[mode] = (y.detach().float(), xa.grad.float(), n_mx)
    assert res['bf16'][2] == 0 and res['fp8w'][2] == 0 and res['mx8'][2] > 4
    for mode in ('mx8', 'fp8w'):
        ry = float((res[mode][0] - res['bf16'][0]).norm() / res['bf16'][0].norm())
        rg = float((res[mode][1] - res['bf16'][1]).norm() / res['bf16'][1].norm())
        print(f'{mode}: {res[mode][2]} MX ops; output deviation from bf16 {ry:.3e}, input-gradient deviation {rg:.3e}')
        assert 0 < ry < 0.15 and 0 < rg < 0.15


@pytest.mark.parametrize('B,H,W,cin,cout', [(2, 16, 16, 320, 320), (1, 8, 12, 128, 160), (2, 9, 7, 48, 160), (1, 32, 32, 640, 320)])
def test_conv3x3_mx8_contract(dev, B, H, W, cin, cout):
    """sidlsg_conv3x3_mx8: implicit GEMM on an e4m3 NHWC image, K-tiles = 128 channels of one tap (Cin = 320 / 48: the last
    chunk of every tap is partly past Cin), halo, time-embedding row vector and residual in the epilogue.  fp64 on the same
    quantised operands."""
    from sid_lsg_amd import ops
    g = torch.Generator().manual_seed(B * H + cin + cout)
    x = torch.randn(B, H, W, cin, generator=g).to(dev).to(BF16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    w16 = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(BF16).contiguous()
    bias = torch.randn(cout, generator=g).to(dev)
    rowvec = torch.randn(B, cout, generator=g).to(dev)
    res = torch.randn(B, H, W, cout, generator=g).to(dev).to(BF16)
    x8 = ops.cast_fp8(x)
    w8 = ops.Fp8Weight(w16)
    wd = w8.dequantize().view(cout, 3, 3, cin).permute(0, 3, 1, 2).double()
    xin = x8.view(torch.float8_e4m3fn).double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin, wd, bias.double(), padding=1).permute(0, 2, 3, 1) + rowvec.double()[:, None, None, :]
    got = ops.conv3x3_mx8(x8, w8, bias=bias, rowvec=rowvec, out_f32=True)
    assert got.shape == ref.shape
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * max(1.0, (9 * cin / 256) ** 0.5)
    got16 = ops.conv3x3_mx8(x8, w8, bias=bias, rowvec=rowvec, res=res)
    ref16 = ref + res.double()
    assert float((got16.double() - ref16).abs().max()) <= 2.0 ** -8 * float(ref16.abs().max())


def test_step_with_fp8_teacher(dev):
    """BASELINE.json configs[4] end to end at test size: the whole SiD-LSG iteration with an e4m3 teacher (MX-fp8 activations
    where the shapes allow), including the generator update's backward THROUGH the teacher.  The fake-score phase never sees
    the teacher (identical losses); the generator loss moves by the teacher's quantisation noise -- phi = psi = G at the start,
    so (y_real - y_fake) IS that noise in the first iteration: the test checks that the step runs, stays finite and that the
    generator's update direction largely agrees with the bf16-teacher run."""
    from sid_lsg_amd.optim import FusedAdamEMA
    from sid_lsg_amd.scheduler import DDPMScheduler
    from sid_lsg_amd.sid_step import SiDStep
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    cfg, lat, b, lr = CONFIGS['tiny40'], 16, 2, 2e-5
    out = {}
    for mode in ('bf16', 'fp8'):
        phi = HipUNet2DCondition(cfg).materialize(dev, seed=1)
        psi = HipUNet2DCondition(cfg).materialize(dev, seed=2)
        G, G_ema = HipUNet2DCondition(cfg).materialize(dev, seed=4), None
        G_ema = G.clone_network(with_grad_buffers=False)
        g0 = G.flat_params.clone()
        if mode == 'fp8':
            phi.requires_grad_(False)
            assert phi.enable_fp8_weights() > 10 and sum(int(getattr(m, 'mx8', False)) for m in phi.modules()) > 4
        opt_f = FusedAdamEMA(psi.parameters(), lr=lr, betas=(0.0, 0.999), eps=1e-8)
        opt_g = FusedAdamEMA(G.parameters(), lr=lr, betas=(0.0, 0.999), eps=1e-8)
        step = SiDStep(G, psi, phi, G_ema, DDPMScheduler().to(dev), opt_f, opt_g, alpha=1.0, cfg_train_fake=1.5, cfg_eval_fake=1.5,
                       cfg_eval_real=1.5, batch_gpu_total=b, init_timestep=625)
        gen = torch.Generator().manual_seed(3)
        losses = []
        for it in range(2):
            inputs = {ph: [dict(z=torch.randn(b, 4, lat, lat, generator=gen).to(dev), noise=torch.randn(b, 4, lat, lat, generator=gen).to(dev),
                                t=torch.randint(20, 980, (b,), generator=gen).to(dev),
                                cond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=gen).to(dev).to(BF16),
                                uncond=torch.randn(b, cfg.text_len, cfg.cross_attention_dim, generator=gen).to(dev).to(BF16))] for ph in ('A', 'B')}
            lf, lg = step.iteration(inputs, ema_beta=0.5)
            losses += [float(lf), float(lg)]
        torch.cuda.synchronize()
        out[mode] = dict(losses=np.array(losses), dG=(G.flat_params - g0).clone(), psi=psi.flat_params.clone())
    a, f = out['bf16'], out['fp8']
    print(f'losses bf16 teacher {a["losses"]}  fp8 teacher {f["losses"]}')
    assert np.isfinite(f['losses']).all() and torch.isfinite(f['dG']).all()
    assert f['losses'][0] == a['losses'][0]                                  # phase A of iteration 0 does not involve the teacher
    rel_g = abs(f['losses'][1] - a['losses'][1]) / abs(a['losses'][1])
    agree = float((torch.sign(a['dG']) == torch.sign(f['dG'])).float().mean())
    print(f'generator loss deviation {rel_g:.3e}; update-sign agreement of G {agree:.4f}')
    assert rel_g < 0.2 and agree > 0.85


def test_fp8_copies_of_a_trainable_network_serve_its_frozen_passes(dev):
    """enable_fp8_weights(frozen_passes_only=True) on a network that trains: (a) outside `fp8_forward()` nothing changes -- forward and
    all gradients are bit-equal to a network without e4m3 copies; (b) inside, the forward is bit-equal to a frozen clone whose e4m3
    weights are always active (the teacher's mode) and the input gradient still flows through the bf16 backward-data operands;
    (c) after an optimizer step the e4m3 copies follow the new weights (re-quantised by refresh_compute_weights)."""
    from sid_lsg_amd.optim import FusedAdamEMA
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    cfg = CONFIGS['tiny40']
    net = HipUNet2DCondition(cfg).materialize(dev, seed=3).requires_grad_(True)
    plain = net.clone_network().requires_grad_(True)
    assert net.enable_fp8_weights(frozen_passes_only=True) > 20
    g = torch.Generator().manual_seed(0)
    B, lat = 2, 16
    x = torch.zeros(B, lat, lat, 8)
    x[..., :4] = torch.randn(B, lat, lat, 4, generator=g)
    x = x.to(dev).to(BF16)
    t = torch.tensor([625, 37], device=dev)
    ctx = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g).to(dev).to(BF16)
    dy = torch.randn(B, lat * lat, 8, generator=g).to(dev)
    # (a) training pass: untouched
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = net.forward_nhwc(xa, t, ctx), plain.forward_nhwc(xb, t, ctx)
    ya.backward(dy)
    yb.backward(dy)
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)
    assert float((net.flat_grads - plain.flat_grads).abs().max()) <= 1e-5 * float(plain.flat_grads.abs().max())     # fp32 atomics order
    # (b) frozen pass through the e4m3 copies == an always-active frozen clone
    frozen = net.clone_network(with_grad_buffers=False).requires_grad_(False)
    frozen.enable_fp8_weights()
    net.requires_grad_(False)
    xc, xd = x.clone().requires_grad_(), x.clone().requires_grad_()
    with net.fp8_forward():
        yc = net.forward_nhwc(xc, t, ctx)
    yd = frozen.forward_nhwc(xd, t, ctx)
    yc.backward(dy)
    yd.backward(dy)
    assert torch.equal(yc, yd) and torch.equal(xc.grad, xd.grad)
    assert not torch.equal(yc, ya), 'the e4m3 forward must differ from the bf16 one'
    assert float((yc - ya).norm() / ya.norm()) < 0.2
    # a forward that wants weight gradients through the e4m3 copies is refused
    net.requires_grad_(True)
    with pytest.raises(RuntimeError, match='without weight gradients'):
        with net.fp8_forward():
            net.forward_nhwc(x, t, ctx)
    with torch.no_grad(), net.fp8_forward():          # ... but fine under no_grad (the generator's pass of phase A)
        assert torch.equal(net.forward_nhwc(x, t, ctx), yc)
    # (c) an optimizer step moves the weights; the e4m3 copies follow
    opt = FusedAdamEMA(net.parameters(), lr=1e-3)
    opt.attach(w16=net.flat_w16, owner=net)
    net.flat_grads.copy_(torch.randn(net.flat_grads.shape, generator=torch.Generator().manual_seed(1)).to(dev))
    opt.step()
    net.refresh_compute_weights(cast=False)
    frozen2 = net.clone_network(with_grad_buffers=False).requires_grad_(False)
    frozen2.enable_fp8_weights()
    with torch.no_grad(), net.fp8_forward():
        ye = net.forward_nhwc(x, t, ctx)
    with torch.no_grad():
        yf = frozen2.forward_nhwc(x, t, ctx)
    assert torch.equal(ye, yf) and not torch.equal(ye, yc)
