"""GPU edge cases of the drop-in surface: rank 1 and the maximum rank, error behaviour, timestep forms, chunking."""
import pytest
import torch

from _util import abs_cos, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def toy():
    from diffusion_pullback_amd import PullbackUNet
    from oracle import unet_sd
    f = load_golden("pullback_zt_tiny.pt")
    cfg = unet_sd.SDConfig(**f["cfg"])
    p = unet_sd.init_params(cfg, seed=f["seed"], gain=f["gain"])
    return f, cfg, p, PullbackUNet("sd", cfg, p, dtype=torch.float32, device=DEV, max_batch=2, max_rank=50, verbose=False)


def test_rank_one_and_max_rank_match_full_jacobian_svd(toy):
    f, cfg, p, net = toy
    from oracle import unet_sd
    J = torch.autograd.functional.jacobian(
        lambda a: unet_sd.forward(p, cfg, a, f["t"], f["ctx"], stop=("mid", 0)).reshape(-1), f["z"]).reshape(-1, 256)
    sv = torch.linalg.svdvals(J)
    _, _, Vh = torch.linalg.svd(J, full_matrices=False)
    for k in (1, 16, 50):                            # 50 = the reference signature's default pca_rank
        V0 = torch.linalg.qr(torch.randn(256, k, generator=torch.Generator().manual_seed(k)))[0].T.contiguous()
        u, s, vT = net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=k, chunk_size=5,
                                                 min_iter=10, max_iter=60, convergence_threshold=1e-5, V0=V0)
        assert u.shape == (J.shape[0], k) and s.shape == (k,) and vT.shape == (k, 256)
        n = 1 if k == 1 else 8                      # leading part of the spectrum has converged
        if k == 50:
            assert torch.allclose(s.cpu()[:20], sv[:20], rtol=5e-3)
        assert torch.allclose(s.cpu()[:n], sv[:n], rtol=2e-3), (k, s.cpu()[:n], sv[:n])
        assert (abs_cos(vT[:n], Vh[:n]) > 0.99).all()
        assert torch.allclose((vT @ vT.T).cpu(), torch.eye(k), atol=1e-3)


def test_error_behaviour(toy):
    f, cfg, p, net = toy
    from diffusion_pullback_amd import DpbError
    with pytest.raises(ValueError):                 # reference message path, utils.py:527
        net.get_h(f["z"], f["t"], f["ctx"], op="down", block_idx=9)
    with pytest.raises(ValueError):
        net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=57)
    with pytest.raises(ValueError):
        net.local_encoder_pullback_zt(f["z"].repeat(2, 1, 1, 1), f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=2)
    with pytest.raises(DpbError):                   # wrong latent size
        net.get_h(torch.zeros(1, 4, 9, 9), f["t"], f["ctx"], op="mid", block_idx=0)
    with pytest.raises(DpbError):                   # conditional net without encoder_hidden_states
        net.get_h(f["z"], f["t"], None, op="mid", block_idx=0)
    with pytest.raises(DpbError):                   # batch above max_batch
        net(f["z"].repeat(3, 1, 1, 1), f["t"], f["ctx"].repeat(3, 1, 1))


def test_timestep_forms_and_chunking_agree(toy):
    f, cfg, p, net = toy
    h0 = net.get_h(f["z"], f["t"], f["ctx"], op="mid", block_idx=0)
    h1 = net.get_h(f["z"], float(f["t"]), f["ctx"], op="mid", block_idx=0)
    h2 = net.get_h(f["z"], f["t"].reshape(1), f["ctx"], op="mid", block_idx=0)
    # (GroupNorm statistics use atomics: repeated runs agree to fp32 round-off, not bit for bit)
    assert torch.allclose(h0, h1, rtol=1e-4, atol=1e-4) and torch.allclose(h0, h2, rtol=1e-4, atol=1e-4)
    V0 = torch.linalg.qr(torch.randn(256, 6, generator=torch.Generator().manual_seed(1)))[0].T.contiguous()
    outs = [net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=6, chunk_size=c, min_iter=1, max_iter=3,
                                          convergence_threshold=1e-9, V0=V0) for c in (25, 2, 3)]     # 1, 3 and 2 chunks
    for u, s, vT in outs[1:]:
        assert torch.allclose(s, outs[0][1], rtol=1e-4) and torch.allclose(vT, outs[0][2], atol=1e-4)
