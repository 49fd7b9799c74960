"""GPU parity of the on-device SD image autoencoder (diffusion_pullback_amd/vae.py) against the CPU oracle
(oracle/vae.py; PARITY UNPINNED: diffusers' AutoencoderKL is not available offline), rows f4 of SURVEY.md section 8."""
import pytest
import torch

from _util import rel

pytestmark = pytest.mark.gpu


def _mk(cfg, dtype, seed=3):
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd.vae import AutoencoderKL
    p = cf.vae_init_params(cfg, seed=seed)
    return p, AutoencoderKL(cfg, p, dtype=dtype, device="cuda:0", max_batch=2)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
def test_small_vae_encode_decode(dtype, tol):
    from diffusion_pullback_amd import configs as cf
    from oracle import vae as ov
    cfg = cf.VAEConfig(block_out_channels=(32, 64, 64), layers_per_block=1, groups=8, sample_size=32)
    p, net = _mk(cfg, dtype)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 3, 32, 32, generator=g)                 # batch 3 with max_batch 2: exercises the chunking
    mom = net.encode_moments(x).cpu()
    assert mom.shape == (3, 8, 8, 8)
    assert rel(mom, ov.encode_moments(p, cfg, x)) < tol
    z = torch.randn(3, 4, 8, 8, generator=g)
    img = net.decode(z).cpu()
    assert img.shape == (3, 3, 32, 32)
    assert rel(img, ov.decode(p, cfg, z)) < tol
    # posterior sampling: mean + std * noise with a seeded generator, and the mean-only path
    s0 = net.encode(x, sample_posterior=False).cpu()
    assert rel(s0, mom[:, :4]) < tol                         # a second pass: GroupNorm sums use atomics, not bitwise
    s1 = net.encode(x, generator=torch.Generator(device="cuda").manual_seed(5)).cpu()
    noise = torch.randn(mom[:, :4].shape, generator=torch.Generator(device="cuda").manual_seed(5), device="cuda").cpu()
    assert rel(s1, mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * noise) < tol


def test_sd15_vae_widths_bf16():
    """The real channel widths (128, 256, 512, 512; 3 ResBlocks per up block, 512-channel single-head attention) at a
    256x256 image so the CPU oracle finishes in seconds; bf16 engine vs fp32 oracle."""
    from diffusion_pullback_amd import configs as cf
    from oracle import vae as ov
    cfg = cf.VAEConfig(sample_size=256)
    p, net = _mk(cfg, torch.bfloat16, seed=1)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 4, 32, 32, generator=g)
    img = net.decode(z).cpu()
    ref = ov.decode(p, cfg, z)
    assert img.shape == (1, 3, 256, 256) and torch.isfinite(img).all()
    assert rel(img, ref) < 5e-2, rel(img, ref)
    x = torch.randn(1, 3, 256, 256, generator=g)
    mom = net.encode_moments(x).cpu()
    assert rel(mom, ov.encode_moments(p, cfg, x)) < 5e-2


def test_sd15_vae_full_size_roundtrip_shapes():
    """BASELINE size: 4x64x64 latent -> 512x512 image -> moments [8,64,64]; finite, deterministic across two runs."""
    from diffusion_pullback_amd import configs as cf
    p, net = _mk(cf.SD15_VAE, torch.bfloat16, seed=2)
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2))
    a = net.decode(z)                                             # fixed-order GroupNorm statistics (default): bitwise repeatable runs,
    b = net.decode(z)                                             # here through the large-map path (gn_reduce_kernel: 4096 statistics blocks)
    assert a.shape == (1, 3, 512, 512) and torch.isfinite(a).all()
    assert torch.equal(a, b)
    m = net.encode_moments(a.clamp(-1, 1))
    assert m.shape == (1, 8, 64, 64) and torch.isfinite(m).all()
