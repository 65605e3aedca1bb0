"""AutoencoderKL decoder on the HIP kernels vs the CPU restatement (oracle/vae_ref.py), seeded random weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    return torch.device('cuda')


@pytest.mark.parametrize('lat', [8, 12])
def test_vae_decode_matches_oracle(dev, lat):
    from oracle.vae_ref import VAE_CONFIGS, AutoencoderKLDecoderRef
    from sid_lsg_amd.vae import HipAutoencoderKLDecoder
    hip = HipAutoencoderKLDecoder('tiny').init_parameters(seed=3)
    ref = AutoencoderKLDecoderRef(VAE_CONFIGS['tiny']).requires_grad_(False)
    missing = ref.load_state_dict(hip.state_dict(), strict=True)       # same key names as diffusers' AutoencoderKL decoder
    assert not missing.missing_keys and not missing.unexpected_keys
    with torch.no_grad():
        for n, p in ref.named_parameters():                            # bf16-representable matmul weights on both sides
            if p.ndim >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    z = torch.randn(2, 4, lat, lat, generator=torch.Generator().manual_seed(0)) * 3.0
    want = ref.decode(z)[0]
    got = hip.decode(z.to(dev))[0]
    assert got.shape == want.shape == (2, 3, 8 * lat, 8 * lat)
    err = (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
    print(f'vae decode lat{lat}: rel max err {err:.4f}')
    assert err < 4e-2          # bf16 activations through 14 conv layers vs fp32


def test_vae_refuses_cpu():
    from sid_lsg_amd.vae import HipAutoencoderKLDecoder
    vae = HipAutoencoderKLDecoder('tiny').init_parameters(0)
    with pytest.raises(RuntimeError):
        vae.decode(torch.zeros(1, 4, 8, 8))
