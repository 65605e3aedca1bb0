"""fp8-weight forward path (BASELINE.json configs[4]; include/sidlsg_hip.h "fp8-weight contractions").

The reference has no fp8 mode (sid_training_loop.py:205 knows fp32 / fp16), so there is no reference golden to pin: the
kernels are checked EXACTLY against what they are specified to compute -- e4m3 per-row weights, activations converted
to e4m3 with unit scale, fp32 accumulation -- restated with torch's float8_e4m3fn casts in fp32, and the end-to-end
deviation of a frozen network from its own bf16 path is bounded and printed."""
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
