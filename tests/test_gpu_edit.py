"""GPU tests of the DDIM / edit loop on the HIP engine (rows a9-a13 of the scope table): scheduler step vs the
reference golden vectors, DDIM trajectories vs the CPU oracle, and the two drivers end to end on reduced nets."""
import os

import pytest
import torch

from _util import load_golden, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_scheduler_step_matches_reference_golden():
    from diffusion_pullback_amd import scheduler as sch
    f = load_golden("scheduler.pt")
    s = sch.YHCustomScheduler()
    xt, et = f["xt"].to(DEV), f["et"].to(DEV)
    s.set_timesteps(100)
    for i in (0, 30, 98):
        r = s.step(et, s.timesteps[i], xt, eta=0.0)
        torch.testing.assert_close(r.prev_sample.cpu(), f[f"step_fwd_{i}"], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(r.x0.cpu(), f[f"x0_fwd_{i}"], rtol=2e-6, atol=2e-6)
    s.set_timesteps(100, is_inversion=True)
    for i in (0, 50, 97):
        r = s.step(et, s.timesteps[i], xt, eta=0.0)
        torch.testing.assert_close(r.prev_sample.cpu(), f[f"step_inv_{i}"], rtol=2e-6, atol=2e-6)


def test_scheduler_step_eta_matches_oracle():
    from diffusion_pullback_amd import scheduler as sch
    from oracle import scheduler as osch
    f = load_golden("scheduler.pt")
    s = sch.YHCustomScheduler()
    s.set_timesteps(100)
    noise = torch.randn(f["xt"].shape, generator=torch.Generator().manual_seed(3))
    ac, _ = osch.linear_alphas_cumprod()
    ts, tn = osch.timesteps(100)
    for i in (85, 97):
        r = s.step(f["et"].to(DEV), s.timesteps[i], f["xt"].to(DEV), eta=1.0, noise=noise.to(DEV))
        ref, x0 = osch.step(ac, ts, tn, f["et"], ts[i], f["xt"], eta=1.0, noise=noise)
        torch.testing.assert_close(r.prev_sample.cpu(), ref, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(r.x0.cpu(), x0, rtol=1e-5, atol=1e-5)


def _uncond_args(tmp, **kw):
    from diffusion_pullback_amd import main as m
    argv = ["--note", "t", "--model_name", "CelebA_HQ_HF", "--dataset_name", "CelebA_HQ", "--result_folder", str(tmp), "--device", DEV,
            "--performance_boosting_t", "0.2", "--x_space_guidance_edit_step", "1", "--x_space_guidance_scale", "0.1",
            "--x_space_guidance_num_step", "16", "--edit_t", "0.6", "--net_scale", "small", "--pca_rank", "3"]
    a = m.preset(m.parse_args(argv))
    a.input_root = os.path.join(str(tmp), "inputs")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_uncond_ddim_trajectory_matches_oracle(tmp_path):
    """inversion steps + forward steps on the engine vs the same loop with the CPU oracle U-Net / scheduler."""
    from diffusion_pullback_amd import main as m
    from diffusion_pullback_amd.edit import EditUncondDiffusion
    from oracle import scheduler as osch
    from oracle import unet_ddpm
    a = _uncond_args(tmp_path)
    unet = m.build_unet(a)
    ed = EditUncondDiffusion(a, unet=unet)
    cfg, p = unet.config, __import__("diffusion_pullback_amd.configs", fromlist=["x"]).ddpm_init_params(unet.config, seed=a.seed)
    x = torch.randn(2, 3, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(5))
    got, t, idx = ed.DDIMforwardsteps(x.to(DEV), t_start_idx=0, t_end_idx=6)
    assert idx == 6 and float(t) == float(ed.scheduler.timesteps[6])
    ac, _ = osch.linear_alphas_cumprod()
    ts, tn = osch.timesteps(100)
    ref = x
    with torch.no_grad():
        for i in range(6):
            ref, _ = osch.step(ac, ts, tn, unet_ddpm.forward(p, cfg, ref, ts[i]), ts[i], ref)
    assert rel(got, ref) < 1e-4, rel(got, ref)


def test_uncond_driver_end_to_end(tmp_path):
    from diffusion_pullback_amd import main as m
    from diffusion_pullback_amd.edit import EditUncondDiffusion
    a = _uncond_args(tmp_path)
    ed = EditUncondDiffusion(a, unet=m.build_unet(a))
    assert int(ed.edit_t_idx) == 40 and int(ed.performance_boosting_t_idx) == 79   # |t - 200| minimal at 201.8
    xt = ed.run_edit_local_encoder_pullback_zt(idx=0, vis_num=4, vis_num_pc=1, pca_rank=3, op="mid", block_idx=0)
    assert xt.shape[0] == 5                                                   # 17 latents subsampled [::4] (edit.py:301-302)
    u, vT = ed.last_basis
    assert u.shape[1] == 3 and vT.shape == (3, 3 * 32 * 32)
    g = (vT @ vT.T).cpu()
    assert torch.allclose(g, torch.eye(3), atol=1e-3)
    d = os.path.join(a.input_root, "local_encoder_pullback_uncond-model_CelebA_HQ_HF-dataset_CelebA_HQ-num_steps_100-pca_rank_3")
    name = "local_basis-CelebA_HQ_0-0.6T-mid-block_0-seed_0.pt"
    assert os.path.exists(os.path.join(d, "u-" + name)) and os.path.exists(os.path.join(d, "vT-" + name))
    outs = os.listdir(a.result_folder)
    assert any(o.startswith("x0_gen-Edit_xt-CelebA_HQ_0-edit_0.6T-mid-block_0-pc_000_pos") for o in outs)
    assert any(o.startswith("x0_gen-Edit_xt-CelebA_HQ_0-edit_0.6T-mid-block_0-pc_000_neg") for o in outs)
    # second call hits the .pt cache and the "already done" skip
    ed.run_edit_local_encoder_pullback_zt(idx=0, vis_num=4, vis_num_pc=1, pca_rank=3, op="mid", block_idx=0)


def test_sd_driver_end_to_end(tmp_path):
    from diffusion_pullback_amd import main as m
    argv = ["--note", "t", "--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Examples", "--result_folder", str(tmp_path),
            "--device", DEV, "--edit_prompt", "sitting dog", "--x_space_guidance_scale", "1", "--x_space_guidance_num_step", "8",
            "--edit_t", "0.7", "--for_steps", "20", "--inv_steps", "20", "--net_scale", "small", "--pca_rank", "2", "--dtype", "bf16",
            "--run_edit_local_encoder_pullback_zt", "False"]
    a = m.preset(m.parse_args(argv))
    a.input_root = os.path.join(str(tmp_path), "inputs")
    from diffusion_pullback_amd.edit import EditStableDiffusion
    ed = EditStableDiffusion(a, unet=m.build_unet(a))
    res = ed.run_edit_local_encoder_pullback_zt(idx=5, op="mid", block_idx=0, vis_num=4, vis_num_pc=1, pca_rank=2, edit_prompt="tiger")
    assert len(res) == 2 and res[0].shape == (5, 4, 16, 16)                  # 9 latents subsampled [::9 // 4] -> rows 0,2,4,6,8
    d = os.path.join(a.input_root, "local_encoder_pullback_stable_diffusion-dataset_Examples-num_steps_20-pca_rank_2")
    assert os.path.exists(os.path.join(d, 'vT-local_basis-Examples_5-0.7T-"tiger"-mid-block_0-seed_0.pt'))
    assert torch.isfinite(res[0]).all()


def test_sd_driver_with_on_device_vae_and_prompt_encoder(tmp_path):
    """Row f4: the edit driver with the HIP AutoencoderKL (decoded PNGs instead of latents) and an injected on-device prompt
    encoder (reference src/modules/edit.py:476-480, :505-522) at reduced network sizes."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd import main as m
    from diffusion_pullback_amd.edit import EditStableDiffusion
    from diffusion_pullback_amd.text_encoder import ClipTextEncoder
    argv = ["--note", "t", "--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Examples", "--result_folder", str(tmp_path),
            "--device", DEV, "--edit_prompt", "sitting dog", "--x_space_guidance_scale", "1", "--x_space_guidance_num_step", "4",
            "--edit_t", "0.7", "--for_steps", "10", "--inv_steps", "10", "--net_scale", "small", "--pca_rank", "2", "--dtype", "bf16",
            "--run_edit_local_encoder_pullback_zt", "False", "--vae", "synthetic"]
    a = m.preset(m.parse_args(argv))
    a.input_root = os.path.join(str(tmp_path), "inputs")
    unet = m.build_unet(a)
    vae = m.build_vae(a)
    assert vae is not None and vae.cfg.latent_size == a.image_size
    ccfg = cf.CLIPTextConfig(vocab_size=300, hidden=64, layers=2, heads=4, intermediate=128, max_position=77)       # cross_dim of the small U-Net
    tok = lambda s: ([298] + [b % 256 for b in s.encode()][:75] + [299] * 77)[:77]
    enc = ClipTextEncoder(ccfg, cf.clip_init_params(ccfg, seed=1), dtype=torch.bfloat16, device=DEV, max_batch=1, tokenizer=tok)
    ed = EditStableDiffusion(a, unet=unet, vae=vae, prompt_encoder=enc.encode_prompt)
    assert ed.edit_prompt_emb.shape == (1, 77, 64) and torch.isfinite(ed.edit_prompt_emb).all()
    ed.run_edit_local_encoder_pullback_zt(idx=5, op="mid", block_idx=0, vis_num=4, vis_num_pc=1, pca_rank=2, edit_prompt="tiger")
    pngs = [f for f in os.listdir(ed.result_folder) if f.endswith(".png")]
    assert len(pngs) >= 2, pngs                                         # x0_gen-*_pos / _neg decoded by the HIP autoencoder
    from PIL import Image
    im = Image.open(os.path.join(ed.result_folder, sorted(pngs)[0]))
    assert im.size[1] == 2 * a.image_size                               # small VAE: 2x upsampling of the 16x16 latents


# ---------------------------------------------------------------- the drivers vs fixtures recorded from the reference's own methods
class _Rec:
    """records every U-Net call (t, input) of a driver run; everything else is delegated to the wrapped PullbackUNet"""

    def __init__(self, net):
        self._net, self.calls = net, []

    def __call__(self, x, t, *a, **k):
        self.calls.append((float(t), x.detach().float().cpu().clone()))
        return self._net(x, t, *a, **k)

    def __getattr__(self, n):
        return getattr(self._net, n)


def _compare_trace(calls, f, flipped, n_pre, blk, tol):
    """the product's U-Net inputs against the reference's; ``flipped``: the product's basis vector has the opposite sign (singular
    vectors are defined up to sign), so its '+v' edit is the reference's '-v' edit and the two edit blocks swap"""
    order = list(range(n_pre)) + (list(range(n_pre + blk, n_pre + 2 * blk)) + list(range(n_pre, n_pre + blk)) if flipped
                                   else list(range(n_pre, n_pre + 2 * blk)))
    assert len(calls) == len(f["trace_t"]) == n_pre + 2 * blk, (len(calls), len(f["trace_t"]))
    worst = 0.0
    for i, j in enumerate(order):
        t, x = calls[i]
        assert t == f["trace_t"][j], (i, j, t, f["trace_t"][j])
        assert x.shape == f["trace_x"][j].shape, (i, x.shape, f["trace_x"][j].shape)
        worst = max(worst, rel(x, f["trace_x"][j]))
    assert worst < tol, worst
    return worst


def test_uncond_driver_matches_reference_driver_fixture(tmp_path, monkeypatch):
    """Rows a10-a13: EditUncondDiffusion.run_edit_local_encoder_pullback_zt on the HIP engine against the run recorded from the
    reference's own class on the vendored PullBackDDPM (tests/golden/make_golden_edit.py): all 64 U-Net inputs (18 inversion steps,
    8 forward steps to edit_t, 2 x 8 x-space-guidance steps, 2 x 11 decode steps), the basis (V0 drawn under the recorded seed), the
    .pt cache names and the images handed to save_image."""
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    from diffusion_pullback_amd import edit as E
    from oracle import unet_ddpm
    f = load_golden("edit_uncond_small.pt")
    a = f["args"]
    cfg = unet_ddpm.DDPMConfig(**f["cfg"])
    params = cf.ddpm_init_params(cfg, seed=f["seed"], spectrum=cf.Spectrum(**f["spectrum"]))
    net = _Rec(PullbackUNet("ddpm", cfg, params, dtype=torch.float32, device=DEV, max_batch=5, max_rank=2, verbose=False))
    args = _uncond_args(tmp_path, for_steps=a["for_steps"], inv_steps=a["inv_steps"], edit_t=a["edit_t"], x_space_guidance_edit_step=a["x_space_guidance_edit_step"],
                        x_space_guidance_scale=a["x_space_guidance_scale"], x_space_guidance_num_step=a["x_space_guidance_num_step"], image_size=32, seed=a["seed"])
    saved = []
    monkeypatch.setattr(E, "save_image", lambda x, path, nrow=None: saved.append((os.path.basename(path), x.detach().float().cpu().clone())))
    ed = E.EditUncondDiffusion(args, unet=net, dataset={a["idx"]: f["x0"]})
    ed.performance_boosting_t_idx = 1000                   # as in the fixture: the eta = 1 tail draws device noise
    assert int(ed.edit_t_idx) == f["edit_t_idx"]
    torch.manual_seed(a["rng_seed"])
    ed.run_edit_local_encoder_pullback_zt(idx=a["idx"], vis_num=a["vis_num"], vis_num_pc=a["vis_num_pc"], pca_rank=a["pca_rank"], op="mid", block_idx=0)
    u, vT = ed.last_basis
    cos = abs_cos_(vT, f["vT"])
    assert (cos > 0.999).all() and (abs_cos_(u.T, f["u"].T) > 0.999).all(), cos
    flipped = bool((vT[0].cpu() * f["vT"][0]).sum() < 0)
    n_pre = (a["inv_steps"] - 2) + f["edit_t_idx"]
    blk = a["x_space_guidance_num_step"] + (a["for_steps"] - 1 - f["edit_t_idx"])
    worst = _compare_trace(net.calls, f, flipped, n_pre, blk, 2e-3)
    ref_saved = dict(f["saved"])
    names = [n for n, _ in saved]
    assert names[:2] == [n for n, _ in f["saved"]][:2]                                # original_x0-*, xT-* of the inversion
    got = dict(saved)
    for tag, rtag in (("pos", "neg" if flipped else "pos"), ("neg", "pos" if flipped else "neg")):
        n = f"x0_gen-Edit_xt-CelebA_HQ_0-edit_0.6T-mid-block_0-pc_000_{tag}.png"
        assert rel(got[n], ref_saved[n.replace(tag, rtag)]) < 2e-3
    d = os.path.join(args.input_root, os.path.basename(f["basis_dir"]))
    assert sorted(x for x in os.listdir(d) if x.endswith(".pt") and not x.startswith("s-")) == [x for x in f["basis_files"] if x.endswith(".pt")]
    print("uncond driver: worst relative U-Net input error", worst, "flipped", flipped)


@pytest.mark.parametrize("trajectory_batch", [0, 10], ids=["one_experiment_at_a_time", "trajectories_together"])
def test_sd_driver_matches_reference_driver_fixture(tmp_path, monkeypatch, trajectory_batch):
    """Rows a10-a13, SD variant: EditStableDiffusion on the HIP engine against the reference's class driving a toy SD-style net with
    fixed prompt embeddings / VAE stand-ins: inversion under the inversion prompt, forward under the forward prompt, guidance under
    the edit prompt (edit.py:112-183, :385-502, :185-307).  trajectory_batch = 0: the reference's order, every U-Net input compared call by call;
    = 10 (round 4, the CLI default is 20): the (pc, +-) experiments advance together -- guidance as one batch-4 call per step, the 2 x 5 decode
    trajectories as one batch-10 call per step -- and must hand the same tensors to save_image under the same names."""
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    from diffusion_pullback_amd import edit as E
    from diffusion_pullback_amd import main as m
    from oracle import unet_sd
    f = load_golden("edit_sd_toy.pt")
    a = f["args"]
    cfg = unet_sd.SDConfig(**f["cfg"])
    params = cf.sd_init_params(cfg, seed=f["seed"], gain=f["gain"], spectrum=cf.Spectrum(**f["spectrum"]))
    net = _Rec(PullbackUNet("sd", cfg, params, dtype=torch.float32, device=DEV, max_batch=max(5, trajectory_batch), max_rank=2, verbose=False))
    argv = ["--note", "t", "--trajectory_batch", str(trajectory_batch), "--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Examples", "--result_folder", str(tmp_path), "--device", DEV,
            "--edit_prompt", "sitting dog", "--x_space_guidance_scale", str(a["x_space_guidance_scale"]), "--x_space_guidance_num_step",
            str(a["x_space_guidance_num_step"]), "--x_space_guidance_edit_step", str(a["x_space_guidance_edit_step"]), "--edit_t", str(a["edit_t"]),
            "--for_steps", str(a["for_steps"]), "--inv_steps", str(a["inv_steps"]), "--net_scale", "small", "--pca_rank", "2"]
    args = m.preset(m.parse_args(argv))
    args.input_root = os.path.join(str(tmp_path), "inputs")
    args.image_size = 8
    emb = f["emb"]

    class VAE:
        def encode(self, x): return (f["z0"] / 0.18215).to(DEV)
        def decode(self, l): return l[:, :3]
    saved = []
    monkeypatch.setattr(E, "save_image", lambda x, path, nrow=None: saved.append((os.path.basename(path), x.detach().float().cpu().clone())))
    ed = E.EditStableDiffusion(args, unet=net, vae=VAE(), prompt_encoder=lambda p: emb["edit"] if p == "tiger" else emb["for"], dataset={a["idx"]: torch.zeros(1, 3, 16, 16)})
    ed.for_prompt_emb, ed.neg_prompt_emb, ed.null_prompt_emb, ed.inv_prompt_emb = (emb[k].to(DEV) for k in ("for", "neg", "null", "inv"))
    assert torch.equal(ed.scheduler.alphas_cumprod, f["alphas_cumprod"]) and int(ed.edit_t_idx) == f["edit_t_idx"]
    torch.manual_seed(a["rng_seed"])
    res = ed.run_edit_local_encoder_pullback_zt(idx=a["idx"], op="mid", block_idx=0, vis_num=a["vis_num"], vis_num_pc=a["vis_num_pc"], pca_rank=a["pca_rank"], edit_prompt="tiger")
    u, vT = ed.last_basis
    assert (abs_cos_(vT, f["vT"]) > 0.999).all() and (abs_cos_(u.T, f["u"].T) > 0.999).all()
    flipped = bool((vT[0].cpu() * f["vT"][0]).sum() < 0)
    n_pre = (a["inv_steps"] - 2) + f["edit_t_idx"]
    blk = a["x_space_guidance_num_step"] + (a["for_steps"] - 1 - f["edit_t_idx"])
    if trajectory_batch > 1:
        # the same experiments in fewer, fatter U-Net calls: the shared prefix call by call, then 8 guidance calls of batch 4 and 11 decode calls of batch 10
        nG, nD = a["x_space_guidance_num_step"], a["for_steps"] - 1 - f["edit_t_idx"]
        assert len(net.calls) == n_pre + nG + nD and [c[1].shape[0] for c in net.calls[n_pre:]] == [4] * nG + [10] * nD
        worst = max(rel(x, f["trace_x"][i]) for i, (t, x) in enumerate(net.calls[:n_pre]))
        first = 1 if flipped else 0          # the product's first chain ('+v') is the reference's second block when the vector's sign is flipped
        for s_ in range(nG):                 # rows [z_+, z_-, z_+ + s v, z_- - s v] of guidance call s_ against the two reference batch-2 calls
            x = net.calls[n_pre + s_][1]
            for c in range(2):
                ref = f["trace_x"][n_pre + ((c + first) % 2) * blk + s_]
                worst = max(worst, rel(torch.stack([x[c], x[2 + c]]), ref))
        assert worst < 2e-3, worst
    else:
        worst = _compare_trace(net.calls, f, flipped, n_pre, blk, 2e-3)
    ref_saved, got = dict(f["saved"]), dict(saved)
    for tag, rtag in (("pos", "neg" if flipped else "pos"), ("neg", "pos" if flipped else "neg")):
        n = f"x0_gen-Edit_zt-Examples_5-edit_0.7T-mid-block_0-pc_000_{tag}-edit_prompt_tiger.png"
        assert rel(got[n], ref_saved[n.replace("_" + tag + "-", "_" + rtag + "-")]) < 2e-3
    d = os.path.join(args.input_root, os.path.basename(f["basis_dir"]))
    assert sorted(x for x in os.listdir(d) if x.endswith(".pt")) == [x for x in f["basis_files"] if x.endswith(".pt")]
    assert len(res) == 2 and res[0].shape == (5, 4, 8, 8)
    print("sd driver: worst relative U-Net input error", worst, "flipped", flipped)


def abs_cos_(a, b):
    from _util import abs_cos
    return abs_cos(a, b)
