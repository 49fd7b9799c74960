"""CPU tests of the host-side mirror: scheduler tables vs the reference golden vectors, CLI preset rules,
driver bookkeeping that does not need the GPU."""
import pytest
import torch

from _util import load_golden
from diffusion_pullback_amd import scheduler as sch


def test_scheduler_tables_match_reference():
    f = load_golden("scheduler.pt")
    s = sch.YHCustomScheduler()
    assert torch.equal(s.alphas_cumprod, f["alphas_cumprod"]) and torch.equal(s.betas, f["betas"])
    for n in (100, 50, 10):
        s.set_timesteps(n)
        assert torch.equal(s.timesteps, f[f"fwd{n}_t"]) and torch.equal(s.timesteps_next, f[f"fwd{n}_tn"])
        s.set_timesteps(n, is_inversion=True)
        assert torch.equal(s.timesteps, f[f"inv{n}_t"]) and torch.equal(s.timesteps_next, f[f"inv{n}_tn"])
    s.set_timesteps(100)
    assert len(s.timesteps) == 99
    for e, idx in f["edit_idx"].items():
        assert int((s.timesteps - e * 1000).abs().argmin()) == idx
    a = sch.extract(s.alphas_cumprod, s.timesteps[30], (2, 3, 8, 8))
    assert a.shape == (2, 1, 1, 1) and a[0, 0, 0, 0] == s.alphas_cumprod[696]        # 696.27 truncates to 696


def test_scheduler_step_rejects_cpu_tensors():
    from diffusion_pullback_amd import DpbError
    s = sch.YHCustomScheduler()
    s.set_timesteps(100)
    with pytest.raises(DpbError):
        s.step(torch.zeros(1, 3, 4, 4), s.timesteps[0], torch.zeros(1, 3, 4, 4))


def test_cli_preset_rules(tmp_path):
    from diffusion_pullback_amd import main as m
    a = m.preset(m.parse_args(["--note", "t", "--model_name", "stabilityai/stable-diffusion-2-1-base", "--dataset_name", "Examples",
                               "--result_folder", str(tmp_path), "--edit_t", "0.7", "--some_dead_flag", "1"]))
    assert a.is_stable_diffusion and (a.c_in, a.image_size, a.memory_bound) == (4, 64, 5)
    b = m.preset(m.parse_args(["--note", "u", "--model_name", "CelebA_HQ_HF", "--dataset_name", "CelebA_HQ", "--result_folder", str(tmp_path),
                               "--performance_boosting_t", "0.2", "--use_x_space_guidance", "True", "--h_t", "0.6"]))
    assert not b.is_stable_diffusion and b.memory_bound == 50 and b.x_space_guidance_scale == 4
    with pytest.raises(ValueError):
        m.preset(m.parse_args(["--note", "x", "--model_name", "CelebA_HQ", "--result_folder", str(tmp_path)]))
    with pytest.raises(AssertionError):       # uncond requires performance_boosting_t == 0.2 (define_argparser.py:229-231)
        m.preset(m.parse_args(["--note", "x", "--model_name", "CelebA_HQ_HF", "--result_folder", str(tmp_path)]))


def test_tape_structure_sd15_and_ddpm():
    """The op tape of the full-size nets can be built without a GPU (weights on CPU): shapes and taps."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd.tape import build_ddpm, build_sd
    cfg = cf.DDPMConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=32)
    t = build_ddpm(cfg, cf.ddpm_init_params(cfg), torch.float32, "cpu")
    assert t.tap_shape[t.taps[("mid", 0)]] == (64, 8, 8) and t.tap_shape[t.taps["eps"]] == (3, 32, 32)
    assert t.buffers[t.taps["eps"]][1] == 8 and t.valid[t.taps["eps"]] == 3          # 3 channels padded to 8
    scfg = cf.SDConfig(block_out_channels=(32, 64), layers_per_block=1, down_attn=(True, False), up_attn=(False, True), heads=(2, 2),
                       cross_dim=16, groups=8, sample_size=8, ctx_len=5)
    ts = build_sd(scfg, cf.sd_init_params(scfg), torch.bfloat16, "cpu", upto=("mid", 0))
    assert ("up", 0) not in ts.taps and ts.tap_shape[ts.taps[("mid", 0)]] == (64, 4, 4)
    assert all(c % 8 == 0 for _, c, _ in ts.buffers)


def test_tape_structure_vae_and_cli_default():
    """Image autoencoder tapes build without a GPU; the CLI keeps latents unless --vae is given."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd import main as M
    from diffusion_pullback_amd.tape import build_vae_decoder, build_vae_encoder
    cfg = cf.VAEConfig(block_out_channels=(32, 64, 64), layers_per_block=1, groups=8, sample_size=32)
    p = cf.vae_init_params(cfg)
    d = build_vae_decoder(cfg, p, torch.bfloat16, "cpu")
    assert d.tap_shape[d.taps["image"]] == (3, 32, 32) and d.buffers[d.x][:2] == (64, 8)          # 8x8 latent, 4 channels padded to 8
    assert d.buffers[d.taps["image"]][1] == 8 and d.valid[d.taps["image"]] == 3
    e = build_vae_encoder(cfg, p, torch.bfloat16, "cpu")
    assert e.tap_shape[e.taps["moments"]] == (8, 8, 8) and e.buffers[e.x][:2] == (32 * 32, 8)
    assert sum(1 for o in d.ops if o["kind"] == 4) == 1 and sum(1 for o in e.ops if o["kind"] == 4) == 1   # one attention op each
    assert all(o["w"][1] == 0 for o in d.ops + e.ops if o["kind"] == 1)                              # primal only: no adjoint weights
    full = cf.vae_param_shapes(cf.SD15_VAE)
    assert cf.SD15_VAE.latent_size == 64 and full["decoder.conv_in.weight"] == (512, 4, 3, 3) and full["encoder.conv_out.weight"] == (8, 512, 3, 3)
    args = M.preset(M.parse_args(["--note", "t", "--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Examples", "--edit_prompt", "x",
                                  "--x_space_guidance_scale", "1", "--x_space_guidance_num_step", "4", "--edit_t", "0.7",
                                  "--run_edit_local_encoder_pullback_zt", "True"]))
    assert args.vae == "none" and M.build_vae(args) is None
    assert args.text_encoder == "none" and M.build_prompt_encoder(args) is None


def test_tape_structure_clip_text():
    """Prompt-encoder tape: causal attention ops, quick-GELU ops, primal-only weights."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd.tape import build_clip_text
    cfg = cf.CLIPTextConfig(vocab_size=50, hidden=16, layers=2, heads=2, intermediate=32, max_position=8)
    t = build_clip_text(cfg, cf.clip_init_params(cfg), torch.bfloat16, "cpu")
    att = [o for o in t.ops if o["kind"] == 4]
    assert len(att) == 2 and all(o["ip"][0] == 2 and o["ip"][4] == 1 for o in att)                 # heads, causal flag
    assert sum(1 for o in t.ops if o["kind"] == 6 and o["ip"][0] == 1) == 2                           # quick-GELU
    assert t.tap_shape[t.taps["last_hidden_state"]] == (16, 8, 1) and t.buffers[t.x][:2] == (8, 16)
    assert all(o["w"][1] == 0 for o in t.ops if o["kind"] == 1)
