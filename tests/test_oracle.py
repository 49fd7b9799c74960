"""Pin the CPU oracle against the golden vectors generated from the imported
reference (tests/golden/make_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import pullback as opb
from oracle import scheduler as osch
from oracle import unet_ddpm, unet_sd

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return torch.load(os.path.join(G, name), weights_only=False)


def test_scheduler_matches_reference():
    f = _load("scheduler.pt")
    ac, betas = osch.linear_alphas_cumprod(torch.float32)
    assert torch.equal(ac, f["alphas_cumprod"]) and torch.equal(betas, f["betas"])
    for n in (100, 50, 10):
        t, tn = osch.timesteps(n)
        assert torch.equal(t, f[f"fwd{n}_t"]) and torch.equal(tn, f[f"fwd{n}_tn"])
        t, tn = osch.timesteps(n, is_inversion=True)
        assert torch.equal(t, f[f"inv{n}_t"]) and torch.equal(tn, f[f"inv{n}_tn"])
    ts, tn = osch.timesteps(100)
    assert len(ts) == 99
    for i in (0, 30, 98):
        xn, x0 = osch.step(ac, ts, tn, f["et"], ts[i], f["xt"])
        assert torch.equal(xn, f[f"step_fwd_{i}"]) and torch.equal(x0, f[f"x0_fwd_{i}"])
    for e, idx in f["edit_idx"].items():
        assert int((ts - e * 1000).abs().argmin()) == idx
    assert f["edit_idx"][0.7] == 30 and f["edit_idx"][1.0] == 0 and f["edit_idx"][0.6] == 40
    ts, tn = osch.timesteps(100, is_inversion=True)
    for i in (0, 50, 97):
        xn, _ = osch.step(ac, ts, tn, f["et"], ts[i], f["xt"])
        assert torch.equal(xn, f[f"step_inv_{i}"])


def test_ddpm_unet_matches_vendored_reference():
    f = _load("ddpm_small.pt")
    cfg = unet_ddpm.DDPMConfig(**f["cfg"])
    p = unet_ddpm.init_params(cfg, seed=f["seed"])
    with torch.no_grad():
        for op, idx in [("down", 0), ("down", 1), ("down", 2), ("mid", 0), ("up", 2), ("up", 1), ("up", 0)]:
            h = unet_ddpm.forward(p, cfg, f["x"], f["t"], stop=(op, idx))
            torch.testing.assert_close(h, f[f"h_{op}_{idx}"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(unet_ddpm.forward(p, cfg, f["x"], f["t"]), f["eps"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(unet_ddpm.forward(p, cfg, f["xb"], f["t"]), f["eps_b"], rtol=1e-5, atol=1e-5)


def test_ddpm_param_count_full_size():
    n = sum(torch.Size(s).numel() for s in unet_ddpm.param_shapes(unet_ddpm.CELEBA_HQ_256).values())
    assert n == 113_673_219          # SURVEY: "113.7 M params"


def test_sd15_param_count():
    shapes = unet_sd.param_shapes(unet_sd.SD15)
    n = sum(torch.Size(s).numel() for s in shapes.values())
    assert n == 859_520_964          # published SD-v1.5 U-Net size (SURVEY §8c "859.5 M")
    enc = sum(torch.Size(s).numel() for k, s in shapes.items()
              if k.startswith(("time_embedding", "conv_in", "down_blocks", "mid_block")))
    assert abs(enc - 348.7e6) < 0.1e6


def test_pullback_xt_matches_vendored_reference():
    f = _load("pullback_xt_ddpm.pt")
    cfg = unet_ddpm.DDPMConfig(**f["cfg"])
    p = unet_ddpm.init_params(cfg, seed=f["seed"])
    get_h = lambda xb: unet_ddpm.forward(p, cfg, xb, f["t"], stop=("mid", 0))
    u, s, vT = opb.pullback(get_h, f["x"], pca_rank=f["k"], chunk_size=f["chunk_size"], min_iter=f["min_iter"],
                            max_iter=f["max_iter"], convergence_threshold=f["thr"], variant="xt", V0=f["V0"])
    torch.testing.assert_close(s, f["s"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(vT, f["vT"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(u, f["u"], rtol=1e-3, atol=1e-4)
    # V0=None must reproduce the reference's own RNG draw
    torch.manual_seed(f["rng_seed"])
    _, s2, _ = opb.pullback(get_h, f["x"], pca_rank=f["k"], chunk_size=f["chunk_size"], min_iter=f["min_iter"],
                            max_iter=f["max_iter"], convergence_threshold=f["thr"], variant="xt")
    torch.testing.assert_close(s2, f["s"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_pullback_zt_xt_match_reference_utils(case):
    f = _load("pullback_zt_tiny.pt")
    c = f["cases"][case]
    cfg = unet_sd.SDConfig(**f["cfg"])
    p = unet_sd.init_params(cfg, seed=f["seed"], gain=f["gain"])
    get_h = lambda zb: unet_sd.forward(p, cfg, zb, f["t"], f["ctx"].expand(zb.shape[0], -1, -1), stop=(c["op"], c["idx"]))
    for variant in ("zt", "xt"):
        u, s, vT = opb.pullback(get_h, f["z"], pca_rank=c["k"], chunk_size=c["chunk"], min_iter=c["min_iter"],
                                max_iter=c["max_iter"], convergence_threshold=c["thr"], variant=variant, V0=c["V0"])
        ur, sr, vr = c[variant]
        assert u.shape == ur.shape and vT.shape == vr.shape
        torch.testing.assert_close(s, sr, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(vT, vr, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(u, ur, rtol=1e-3, atol=1e-4)


def test_vae_oracle_shapes_and_param_count():
    """AutoencoderKL restatement (PARITY UNPINNED, diffusers absent): published parameter count of the SD-1.5 VAE and
    the encode / decode shape contract of reference src/modules/edit.py:144-146, :476-480."""
    import math
    from oracle import vae as ov
    assert sum(math.prod(v) for v in ov.param_shapes(ov.SD15_VAE).values()) == 83_653_863
    cfg = ov.VAEConfig(block_out_channels=(16, 32, 32), layers_per_block=1, groups=4, sample_size=16)
    p = ov.init_params(cfg, seed=0)
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(0))
    z, mean, logvar = ov.encode(p, cfg, x, noise=torch.ones(2, 4, 4, 4))
    assert z.shape == (2, 4, 4, 4) and torch.allclose(z, mean + torch.exp(0.5 * logvar))
    assert torch.equal(ov.encode(p, cfg, x)[0], mean)                      # no noise -> posterior mean
    img = ov.decode(p, cfg, z)
    assert img.shape == (2, 3, 16, 16) and torch.isfinite(img).all()
    # batch independence (GroupNorm is per sample): decoding sample 0 alone gives the same image
    torch.testing.assert_close(ov.decode(p, cfg, z[:1]), img[:1], rtol=1e-3, atol=1e-4)


def test_clip_text_oracle_param_count_and_causality():
    """CLIP ViT-L/14 text model restatement (PARITY UNPINNED, transformers weights absent): published parameter count and the
    causal-mask property of the encoder behind pipe._encode_prompt (reference src/modules/edit.py:505-522)."""
    import math
    from oracle import clip_text as oc
    assert sum(math.prod(v) for v in oc.param_shapes(oc.SD15_CLIP).values()) == 123_060_480
    cfg = oc.CLIPTextConfig(vocab_size=50, hidden=16, layers=2, heads=2, intermediate=32, max_position=8)
    p = oc.init_params(cfg, seed=0)
    ids = torch.randint(0, 50, (2, 8), generator=torch.Generator().manual_seed(0))
    y = oc.encode(p, cfg, ids)
    ids2 = ids.clone(); ids2[:, 5:] = (ids2[:, 5:] + 3) % 50
    y2 = oc.encode(p, cfg, ids2)
    assert y.shape == (2, 8, 16)
    torch.testing.assert_close(y2[:, :5], y[:, :5], rtol=1e-5, atol=1e-6)
    assert not torch.allclose(y2[:, 5:], y[:, 5:], atol=1e-3)
