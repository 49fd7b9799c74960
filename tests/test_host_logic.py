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
    assert a.trajectory_batch == 20                                   # default: the (pc, +-) trajectories advance together
    # an EXPLICIT --memory_bound (not a flag of the reference, which only knows the per-model constant) is the bound of every U-Net call: it replaces the
    # constant, caps the trajectory batching, the guidance chains per call and the engine's max_batch (floor: the pair of x-space guidance)
    c = m.preset(m.parse_args(["--note", "t", "--model_name", "stabilityai/stable-diffusion-2-1-base", "--dataset_name", "Examples",
                               "--result_folder", str(tmp_path), "--memory_bound", "2", "--dead", "2"]))
    assert c.memory_bound == 2 and c.trajectory_batch == 2 and c.memory_bound_given == 2
    d = m.preset(m.parse_args(["--note", "t", "--model_name", "stabilityai/stable-diffusion-2-1-base", "--dataset_name", "Examples",
                               "--result_folder", str(tmp_path), "--memory_bound=3", "--trajectory_batch", "8"]))
    assert d.trajectory_batch == 3 and d.memory_bound == 3
    for bad in (["--memory_bound", "x"], ["--memory_bound", "0"], ["--memory_bound", "--other"]):       # argparse errors, not a bare ValueError / a silent 1
        with pytest.raises(SystemExit):
            m.parse_args(["--note", "t", "--model_name", "stabilityai/stable-diffusion-2-1-base"] + bad)
    # the chunking and the guidance chains per call follow the explicit bound
    from diffusion_pullback_amd import edit as E

    class _Eng:
        max_batch = 40

    class _U:
        engine = _Eng()
    eb = E._EditBase()
    eb.unet, eb.memory_bound, eb.trajectory_batch = _U(), c.memory_bound, c.trajectory_batch
    assert [x.shape[0] for x in eb._chunks(torch.zeros(5, 1))] == [2, 2, 1]
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


def test_hf_unet2dmodel_key_map_roundtrip():
    """google/ddpm-ema-celebahq-256 is loaded through diffusers by the reference's live path (utils.py:101-104): its UNet2DModel
    state-dict keys must map one-to-one onto the names tape.build_ddpm reads (pure rename, weights untouched)."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd import weights as W
    from diffusion_pullback_amd.tape import build_ddpm
    cfg = cf.DDPMConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=32)
    p = cf.ddpm_init_params(cfg, seed=3)
    hf = W.ddpm_vendored_to_hf_names(p, cfg)
    assert "time_embedding.linear_1.weight" in hf and "down_blocks.0.resnets.0.time_emb_proj.bias" in hf
    assert "mid_block.attentions.0.query.weight" in hf and hf["mid_block.attentions.0.query.weight"].dim() == 2   # Linear in diffusers
    assert "up_blocks.0.resnets.0.conv_shortcut.weight" in hf            # up_blocks.0 = bottleneck level = vendored up.2
    assert "up_blocks.0.upsamplers.0.conv.weight" in hf and "up_blocks.2.upsamplers.0.conv.weight" not in hf
    assert "conv_norm_out.weight" in hf and not any(k.startswith(("down.", "up.", "mid.", "temb.")) for k in hf)
    back = W.ddpm_hf_to_vendored_names(hf, cfg)
    assert set(back) == set(p)
    for k in p:
        assert torch.equal(back[k].reshape(p[k].shape), p[k]), k
    W.check_shapes(back, cf.ddpm_param_shapes(cfg), "small DDPM")
    t = build_ddpm(cfg, back, torch.float32, "cpu")                       # 2-D attention projections are accepted by the tape
    assert t.tap_shape[t.taps[("mid", 0)]] == (64, 8, 8)
    # later diffusers releases renamed the attention projections
    hf2 = {k.replace(".query.", ".to_q.").replace(".key.", ".to_k.").replace(".value.", ".to_v.").replace(".proj_attn.", ".to_out.0."): v
           for k, v in hf.items()}
    assert set(W.ddpm_hf_to_vendored_names(hf2, cfg)) == set(p)
    assert W.ddpm_hf_to_vendored_names(p, cfg).keys() == p.keys()         # vendored names pass through
    with pytest.raises(KeyError):
        W.ddpm_hf_to_vendored_names({"down_blocks.0.resnets.0.bogus.weight": torch.zeros(1)}, cfg)
    # full-size key set: every parameter of the 256x256 architecture is reachable from an HF-named dict
    full = cf.ddpm_param_shapes(cf.CELEBA_HQ_256)
    fake = {k: torch.empty(0) for k in full}
    assert set(W.ddpm_hf_to_vendored_names(W.ddpm_vendored_to_hf_names(fake, cf.CELEBA_HQ_256), cf.CELEBA_HQ_256)) == set(full)


def test_sd_model_id_selects_architecture(tmp_path):
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd import weights as W
    from diffusion_pullback_amd.tape import build_sd
    assert cf.sd_config_for("runwayml/stable-diffusion-v1-5") is cf.SD15
    assert cf.sd_config_for("CompVis/stable-diffusion-v1-4") is cf.SD15
    c21 = cf.sd_config_for("stabilityai/stable-diffusion-2-1-base")       # the id the reference's SD scripts pass
    assert c21 is cf.SD21_BASE and c21.cross_dim == 1024 and c21.heads == (5, 10, 20, 20) and c21.use_linear_projection
    assert cf.sd_config_for("stabilityai/stable-diffusion-2-base") is cf.SD21_BASE
    for bad in ("stabilityai/stable-diffusion-2-1", "stabilityai/stable-diffusion-xl-base-1.0", "foo/stable-diffusion-3"):
        with pytest.raises(ValueError):
            cf.sd_config_for(bad)
    s21 = cf.sd_param_shapes(c21)
    assert s21["mid_block.attentions.0.proj_in.weight"] == (1280, 1280)                       # Linear, not 1x1 conv
    assert s21["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 1024)
    assert sum(int(torch.tensor(v).prod()) for v in s21.values()) == 865_910_724              # SD-2.1-base U-Net parameter count
    # SD-2.1-shaped weights under an SD-1.5 config (and vice versa) are rejected with a readable message
    tiny15 = cf.SDConfig(block_out_channels=(32, 64), layers_per_block=1, down_attn=(True, False), up_attn=(False, True), heads=(2, 2),
                         cross_dim=16, groups=8, sample_size=8, ctx_len=5)
    tiny21 = cf.SDConfig(block_out_channels=(32, 64), layers_per_block=1, down_attn=(True, False), up_attn=(False, True), heads=(1, 2),
                         cross_dim=24, groups=8, sample_size=8, ctx_len=5, use_linear_projection=True)
    p21 = cf.sd_init_params(tiny21)
    with pytest.raises(ValueError, match="does not match"):
        W.check_shapes(p21, cf.sd_param_shapes(tiny15), "tiny SD-1.5")
    W.check_shapes(p21, cf.sd_param_shapes(tiny21), "tiny SD-2.1")
    t = build_sd(tiny21, p21, torch.bfloat16, "cpu", upto=("mid", 0))      # per-block head counts + Linear projections build
    heads = [o["ip"][0] for o in t.ops if o["kind"] == 4]
    assert heads == [1, 1, 2, 2]                                           # down0 self/cross (1 head), mid self/cross (heads[-1] = 2)


def test_gemm_dispatch_plans_respect_the_slab_scratch_and_the_tile_contracts():
    """dpb_debug_gemm_plan (host-only) over the products of the SD-1.5 / SD-2.1 pullback path at k = 1..10 and 1..8 samples advanced together:
    every plan's split count fits the fp32 slab scratch (a rule once returned before the capacity clamp: 10240 x 1280 x 5120 split two-fold
    wrote 105 MB of partials into 64 MB), the 256x256 tiles are never split and only take products they tile with < 7 % (ring) / <= 25 % (8-phase) padding, fused GEGLU
    epilogues only go to kernels that implement them, fp32 never leaves the register-staged kernel."""
    import ctypes as C
    from diffusion_pullback_amd import lib as L
    lib = L.load()
    slab = 64 << 20

    def plan(dtype, M, N, K, hw=0, cin=0, epi=0):
        k, t, s = C.c_int(), C.c_int(), C.c_int()
        L.check(lib.dpb_debug_gemm_plan(dtype, M, N, K, hw, cin, epi, slab, C.byref(k), C.byref(t), C.byref(s)))
        return k.value, t.value, s.value

    chans = (320, 640, 960, 1280, 1920, 2560, 3840, 5120, 10240)
    n_plans = n_p8 = n_wres = 0
    for nt in (1, 3, 5, 10, 20, 40, 80):
        for hw in (8, 16, 32, 64):
            M = nt * hw * hw
            for N in chans:
                for K in (320, 640, 1280, 2560, 5120, 10240):
                    for dt in (L.DPB_BF16, L.DPB_F16, L.DPB_F32):
                        kind, tile, s = plan(dt, M, N, K)
                        n_plans += 1
                        assert s >= 1 and s * M * N * 4 <= slab or s == 1, (dt, M, N, K, kind, tile, s)
                        if dt == L.DPB_F32:
                            assert kind in (0, 1), (M, N, K, kind, tile)
                        if tile == 518:
                            t256 = -(-M // 256) * -(-N // 256)
                            assert s == 1 and K >= 640 and t256 >= 160 and t256 * 65536 <= 1.07 * M * N, (M, N, K, s)
                        if tile == 540:                                   # weights-resident streaming kernel: K = 320 linear layers from the row count at which streaming the activations past resident weights beats the tile kernels
                            assert dt != L.DPB_F32 and s == 1 and K == 320 and N % 320 == 0 and M >= (49152 if N == 320 else 147456), (M, N, K, s)
                            n_wres += 1
                        if tile == 530:                                   # 8-phase tile: fills the chip, <= 25 % padding (40 % from 2048 tiles on), last round >= 58 % occupied, unsplit
                            t256 = -(-M // 256) * -(-N // 256)
                            assert dt != L.DPB_F32 and s == 1 and t256 >= 160 and (K >= 640 or (K >= 320 and t256 >= 4096)), (M, N, K, s)
                            assert (0.6 if t256 >= 2048 else 0.75) * t256 * 65536 <= M * N, (M, N, K, t256)
                            assert t256 >= 1024 or t256 >= 0.58 * 256 * -(-t256 // 256), (M, N, K, t256)
                            n_p8 += 1
            for cin in (320, 640, 1280, 1920, 2560):          # 3x3 convolutions of the ResBlocks
                for cout in (320, 640, 1280):
                    kind, tile, s = plan(L.DPB_BF16, M, cout, 9 * cin, hw, cin)
                    n_plans += 1
                    assert s >= 1 and (s == 1 or s * M * cout * 4 <= slab), (M, cout, cin, kind, tile, s)
                    assert kind in (2, 3), (M, cout, cin, kind)           # bf16 convolutions of these sizes: ring or halo-tile kernel
                    if kind == 3:
                        assert hw >= 16 and s <= cin // 64
                    if tile == 530:                                       # the 8-phase tile as an implicit-GEMM convolution: whole 64-channel K tiles
                        assert cin % 64 == 0 and s == 1 and M * cout >= 0.75 * (-(-M // 256) * -(-cout // 256)) * 65536
                        n_p8 += 1
    assert n_plans > 4000 and n_p8 > 100 and n_wres >= 4
    # the 320-channel linear layers of the 64x64 level (tangent / adjoint passes, plain epilogue) at many tangents: the weights-resident kernel, never split;
    # not at 5 tangents (20480 rows: the rings are ahead), not with a fused epilogue, not for the 640-channel level
    assert plan(L.DPB_BF16, 81920, 320, 320) == (2, 540, 1) and plan(L.DPB_F16, 327680, 960, 320) == (2, 540, 1)
    assert plan(L.DPB_BF16, 20480, 320, 320)[1] != 540 and plan(L.DPB_BF16, 81920, 960, 320)[1] != 540
    assert plan(L.DPB_BF16, 81920, 640, 640)[1] != 540 and plan(L.DPB_BF16, 327680, 2560, 320, epi=1)[1] != 540
    try:                                                                   # a forced split count never reaches it (it has no slab path), a forced tile only where it applies
        L.check(lib.dpb_debug_set(b"gemm_splitk", 4)); L.check(lib.dpb_debug_set(b"gemm_tile", 540))
        assert plan(L.DPB_BF16, 20480, 320, 320) == (2, 540, 1)
        assert plan(L.DPB_BF16, 20480, 640, 640)[1] == 515
    finally:
        L.check(lib.dpb_debug_set(b"gemm_splitk", 0)); L.check(lib.dpb_debug_set(b"gemm_tile", 0))
    # the launch that overflowed: now the 256x256 tile, unsplit; and the same product forced onto the 128x128 ring keeps within the scratch
    assert plan(L.DPB_BF16, 10240, 1280, 5120) == (2, 530, 1)
    assert plan(L.DPB_BF16, 20480, 320, 2880, 64, 320)[0] == 3            # N = 320 (37.5 % padding on 256-column tiles): the halo-tile kernel keeps the 64x64-level convolutions
    assert plan(L.DPB_BF16, 20480, 640, 5760, 32, 640)[:2] == (2, 530)     # 20 tangents: the 8-phase tile takes the 32x32-level 3x3 convolutions
    assert plan(L.DPB_BF16, 81920, 640, 5760, 32, 640)[0] == 3             # 80 tangents, N = 640 (17 % padding): the halo-tile kernel is ahead again
    assert plan(L.DPB_BF16, 20480, 1280, 11520, 16, 1280)[:2] == (2, 530)  # N = 1280: no padding, the 8-phase tile
    # fused GEGLU epilogues: FF-in tangent (N = 2F interleaved) and FF-out adjoint (N = F) of every level go to ring kernels, unsplit
    for M, F, Cc in ((20480, 1280, 320), (5120, 2560, 640), (1280, 5120, 1280), (40960, 2560, 640), (2560, 5120, 1280)):
        for epi, N, K in ((1, 2 * F, Cc), (2, F, Cc)):
            kind, tile, s = plan(L.DPB_BF16, M, N, K, epi=epi)
            assert kind == 2 and s == 1 and (tile in (128, 130, 132, 256, 518, 530) or 512 <= tile <= 517), (M, N, K, epi, kind, tile, s)
            if tile in (518, 530):
                assert N % 256 == 0


def test_gemm_override_environment_changes_the_plan_of_the_named_shape_only():
    """DPB_GEMM_OVERRIDE="MxNxK:gather=code/split,..." (the in-pipeline tuning hook of tools/gpu_gemm_override.py) is read once per process:
    checked in a child process through dpb_debug_gemm_plan -- the named shape gets the forced kernel and split (still clamped to the slab
    scratch), every other shape keeps the heuristic's plan."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import ctypes as C, json\n"
        "from diffusion_pullback_amd import lib as L\n"
        "lib = L.load()\n"
        "def plan(M, N, K, slab=64 << 20):\n"
        "    k, t, s = C.c_int(), C.c_int(), C.c_int()\n"
        "    L.check(lib.dpb_debug_gemm_plan(L.DPB_BF16, M, N, K, 0, 0, 0, slab, C.byref(k), C.byref(t), C.byref(s)))\n"
        "    return [k.value, t.value, s.value]\n"
        "print(json.dumps([plan(320, 1280, 1280), plan(1280, 1280, 1280), plan(10240, 1280, 5120), plan(10240, 1280, 5120, 1 << 20)]))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(override):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("DPB_GEMM_OVERRIDE", None)
        if override:
            env["DPB_GEMM_OVERRIDE"] = override
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    base = run("")
    forced = run("320x1280x1280:0=515/4,10240x1280x5120:0=515/2")
    assert base[0] == [2, 64, 1] and forced[0] == [2, 515, 4]            # the named shape: 64x64 ring unsplit -> BK=64 ring, four-fold
    assert forced[1] == base[1]                                            # an unnamed shape keeps its plan
    assert base[2] == [2, 530, 1] and forced[2] == [2, 515, 1]             # 2 x 52 MB of slabs do not fit 64 MB: the forced split is clamped to 1
    assert forced[3][2] == 1                                               # and with 1 MB of scratch nothing splits


def test_bench_names_the_baseline_config_it_measures():
    """bench.py's config.workload: every --op down/up run is BASELINE configs[4] with its tap, k = 10 / edit ctx / strong mode is configs[3],
    DDPM is configs[1], the default is configs[2] (r02 review: 16 sweep lines were labelled configs[2])."""
    import argparse
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    A = lambda **kw: argparse.Namespace(**{**dict(workload="sd15", k=5, ctx="null"), **kw})
    assert bench.workload_name(A(), False, ("mid", 0)).startswith("BASELINE configs[2]")
    assert "configs[4]" in bench.workload_name(A(), False, ("up", 3)) and "up_block_3" in bench.workload_name(A(), False, ("up", 3))
    assert "configs[4]" in bench.workload_name(A(), False, ("down", 0)) and "down_block_0" in bench.workload_name(A(), False, ("down", 0))
    assert bench.workload_name(A(k=10, ctx="edit"), False, ("mid", 0)).startswith("BASELINE configs[3]")
    assert bench.workload_name(A(), True, ("mid", 0)).startswith("BASELINE configs[3]")
    assert bench.workload_name(A(workload="ddpm256"), False, ("mid", 0)).startswith("BASELINE configs[1]")


def test_bench_quotes_pmc_traffic_only_for_the_running_build(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC file ONLY if that file records the source hash of the library that is running; otherwise null."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from diffusion_pullback_amd import lib as L
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kern = {"kernels": {"k<1>": {"launches": 3, "fetch_kb_per_launch": 10.0, "write_kb_per_launch": 4.0}}}
    (prof / "r99_pmc_traffic_sd15_mid_k5_bf16.json").write_text(json.dumps({**kern, "_src_hash": "not-this-build"}))
    monkeypatch.setattr(L, "_built_hash", lambda: "abc123")
    t, note = bench.pmc_traffic("k<1>", True)
    assert t is None and "null" in note
    (prof / "r98_pmc_traffic_sd15_mid_k5_bf16.json").write_text(json.dumps({**kern, "_src_hash": "abc123"}))
    t, note = bench.pmc_traffic("k<1>", True)
    assert t == (2 * 10.0 + 4.0) * 1024 and "r98" in note
    assert bench.pmc_traffic("k<1>", False)[0] is None


def test_spectrum_for_tap_shapes_the_last_self_attention_of_the_prefix():
    from diffusion_pullback_amd import configs as cf
    assert cf.Spectrum.for_tap("mid", 0).also == ()
    assert cf.Spectrum.for_tap("down", 1).also == ("down_blocks.1.attentions.1",)
    assert cf.Spectrum.for_tap("down", 3).also == ("down_blocks.2.attentions.1",)          # down3 has no attention of its own
    up = cf.Spectrum.for_tap("up", 3)
    assert up.also == ("up_blocks.3.attentions.2",) and up.amp == 100.0                   # two shaped layers in series: fp16 range
    cfg = cf.SDConfig(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), up_attn=(False, True), heads=(2, 2),
                      cross_dim=64, sample_size=16, ctx_len=77)
    a = cf.sd_init_params(cfg, seed=0, spectrum=cf.Spectrum())
    b = cf.sd_init_params(cfg, seed=0, spectrum=cf.Spectrum(also=("down_blocks.0.attentions.0",)))
    changed = sorted(k for k in a if not torch.equal(a[k], b[k]))
    assert changed == ["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"]


def test_orth_scratch_contract_follows_the_block_layout():
    """dpb_orth_scratch_bytes (include/dpb.h) is the caller's side of the re-orthonormalisation's fixed-order reductions: Cm [k][k], Gram / overlap
    partials of up to 64 column slices, (distance, violation) partials of up to 256 slices of 256 columns -- a pure host function (no GPU needed)."""
    from diffusion_pullback_amd import lib as L
    lib = L.load()
    f = lambda k, n: int(lib.dpb_orth_scratch_bytes(k, n))
    assert f(5, 4 * 64 * 64) == 8 * (25 * (1 + 2 * 64) + 2 * 64)            # 64 slices of 256 columns: the SD latent
    assert f(5, 3 * 256 * 256) == 8 * (25 * (1 + 2 * 64) + 2 * 256)         # both block counts capped: the DDPM-256 image
    assert f(3, 100) == 8 * (9 * (1 + 2 * 1) + 2 * 1)                       # one ragged slice
    assert f(56, 4 * 64 * 64) == 8 * (56 * 56 * (1 + 2 * 64) + 2 * 64)      # the largest rank of rounds 1-5
    assert f(128, 4 * 64 * 64) == 8 * (128 * 128 * (1 + 2 * 64) + 2 * 64)   # the library's limit since round 6 (same layout: the eigen-solve reuses partial 0's slots)
    assert f(129, 4 * 64 * 64) == 0 and f(0, 4 * 64 * 64) == 0 and f(5, 0) == 0
    assert all(f(k, n) <= f(k, 2 * n) for k in (1, 5, 50) for n in (64, 1000, 16384, 100000))


def test_flash_forward_isa_keeps_the_rescale_and_the_row_sum_off_the_hot_path(tmp_path):
    """attn_fwd_kernel (csrc/attn_fused.hip, round 6) was VALU-bound 2-3x over its MFMAs; what took it to 1.67x is structural and a compiler or an edit can
    silently undo it: (i) the accumulator rescale (32 multiplies per 32 keys) must sit behind a scalar branch, not be if-converted back into the straight-line
    code of the MFMAs; (ii) at d = 40 the row sum comes out of the P V product (ones column), so the MFMA segments carry no chain of adds; (iii) the shuffle
    that hands l to the other half-wave is executed by ALL lanes -- under an exec mask it reads its inactive source lanes as 0 (l = 0 -> NaN: the first
    build of the round, caught only on the GPU).  Read off the gfx950 ISA hipcc emits here."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    out = tmp_path / "attn_fused.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                        os.path.join(root, "diffusion_pullback_amd", "csrc", "attn_fused.hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = re.split(r"^(_ZN3dpb15attn_fwd_kernelILi40ELi[01]ELi8EEEvNS_9FusedArgsEPNS_4bf16EPf):", text, flags=re.M)[1:]
    assert len(kernels) == 2 * 2, "expected attn_fwd_kernel<40, 0 | 1, 8>"
    for name, body in zip(kernels[0::2], kernels[1::2]):
        body = body.split(".Lfunc_end")[0]
        assert "scratch_" not in body, name
        ins = [ln.split(";")[0].strip() for ln in body.splitlines()]
        ins = [i for i in ins if i and not (i.startswith(".") and not i.endswith(":"))]
        segs, cur, lead = [], [], ""                        # straight-line segments; `lead` = what ended the previous one (a label or a branch)
        for i in ins:
            if i.endswith(":") or i.startswith("s_cbranch") or i.startswith("s_branch"):
                segs.append((lead, cur)); cur, lead = [], i
            else:
                cur.append(i)
        segs.append((lead, cur))
        n = lambda seg, *pre: sum(1 for i in seg if i.startswith(pre))
        hot = [seg for _, seg in segs if n(seg, "v_mfma_")]
        assert len(hot) >= 4 and sum(n(seg, "v_mfma_") for seg in hot) == 4 * 7, name          # 3 score + 4 output MFMAs per 32 keys, four blocks per stage
        for seg in hot:
            assert n(seg, "v_pk_mul_f32", "v_mul_f32") <= 2, (name, "accumulator rescale inside an MFMA segment", n(seg, "v_pk_mul_f32", "v_mul_f32"))
            assert n(seg, "v_add_f32", "v_pk_add_f32") <= 2, (name, "row-sum adds inside an MFMA segment")
        rescale = [(lead, seg) for lead, seg in segs if n(seg, "v_pk_mul_f32") + n(seg, "v_mul_f32") // 2 >= 12 and not n(seg, "v_mfma_")]
        assert len(rescale) >= 4, (name, len(rescale))
        assert all(lead.startswith("s_cbranch_scc") or lead.startswith("s_cbranch_vcc") for lead, _ in rescale), (name, [lead for lead, _ in rescale])
        assert sum(n(seg, "v_exp_f32") for seg in hot) == 4 * 16, name                             # one exp per score; the four alpha exps live in the rescale segments
        last_mfma = max(k for k, (_, seg) in enumerate(segs) if n(seg, "v_mfma_"))
        tail = [i for _, seg in segs[last_mfma:] for i in seg]
        shf = [k for k, i in enumerate(tail) if i.startswith("ds_bpermute_b32") or "permlane32_swap" in i or "row_" in i and "dpp" in i]
        assert shf, (name, "the half-wave exchange of l was not found behind the loop")
        masked = 0
        for i in tail[:shf[-1]]:
            masked += i.startswith("s_and_saveexec_b64") - (i.startswith("s_or_b64 exec"))
        assert masked <= 0, (name, "the exchange of l runs under an exec mask")


def test_measurement_scripts_compile():
    """tools/*.py and bench.py are not imported by any test (they need a GPU box): at least their syntax is checked here."""
    import glob, os, py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, doraise=True, cfile=os.path.join("/tmp", "dpb_pyc_" + os.path.basename(f) + "c"))


def test_p8_isa_keeps_fragment_registers_untouched_until_their_wait(tmp_path):
    """gemm_p8.hip reads its MFMA fragments with `asm volatile ds_read_b128` and orders them only by threading the registers through a later
    `s_waitcnt lgkmcnt(0)` asm (csrc/gemm_p8.hip:44-58).  Nothing in the language stops a future compiler from placing a copy, a spill or a
    consumer of a not-yet-landed fragment between the two; the bitwise / race-screen GPU tests cannot see that class of regression.  This check
    reads the gfx950 ISA hipcc emits here (cross-compiles without a GPU): between every ds_read_b128 and the next lgkmcnt(0) wait no
    instruction may name one of its destination registers, the kernels must not use scratch, and the counted DMA waits must be the ones the
    schedule in the file header derives (two LDS-DMA pieces per stage: vmcnt(10) / vmcnt(6))."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    out = tmp_path / "gemm_p8.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                        os.path.join(root, "diffusion_pullback_amd", "csrc", "gemm_p8.hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = re.split(r"^(_ZN3dpb14gemm_p8_kernel\w+):", text, flags=re.M)[1:]
    assert len(kernels) >= 2 * 4, "expected the gemm_p8_kernel instantiations"
    reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")

    def regs(s):
        o = set()
        for m in reg.finditer(s):
            if m.group(1) is not None:
                o.add(int(m.group(1)))
            else:
                o.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return o
    n_reads = 0
    for name, body in zip(kernels[0::2], kernels[1::2]):
        body = body.split(".Lfunc_end")[0]
        assert "scratch_" not in body, name
        lines = body.splitlines()
        last_mfma = max(i for i, ln in enumerate(lines) if "v_mfma_" in ln.split(";")[0])
        pending = set()
        waits = []
        for ln in lines[:last_mfma + 1]:          # the K loop: the epilogue behind it stages through LDS with the compiler's own (counted) waits
            ins = ln.split(";")[0].strip()
            if not ins or ins.startswith(".") or ins.endswith(":"):
                continue
            if ins.startswith("s_waitcnt"):
                waits.append(ins)
                if "lgkmcnt(0)" in ins:
                    pending.clear()
                continue
            touched = regs(ins)
            if ins.startswith("ds_read_b128"):
                dst = regs(ins.split(",")[0])
                assert not (touched - dst) & pending, (name, ins)          # its address register is not a pending fragment either
                pending |= dst
                n_reads += 1
                continue
            assert not touched & pending, f"{name}: `{ins}` names a fragment register before its s_waitcnt lgkmcnt(0)"
        vm = [int(m) for w in waits for m in re.findall(r"vmcnt\((\d+)\)", w)]
        assert vm.count(10) >= 2 and vm.count(6) >= 3, (name, sorted(set(vm)))   # two unrolled K tiles + the prologue (drains: prologue / epilogue only)
    assert n_reads >= 8 * 2 * 24
    # output stores (round 6, common.h): write-through.  The epilogue's 16-byte 16-bit stores are hipcc-scheduled buffer stores with the sc1 bit (through the
    # output tensor's descriptor); the asm fallback (tensors of 4 GiB or more) must carry its pad INSIDE the asm block -- hipcc pads nothing behind an asm
    # statement, and a VALU write of the data registers right behind a > 64-bit store corrupts it
    assert len(re.findall(r"buffer_store_dwordx4 .* sc1", text)) >= len(kernels) // 2
    asm_blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, flags=re.S)
    stores = [b for b in asm_blocks if "global_store_dwordx4" in b]
    assert stores and all(re.search(r"global_store_dwordx4 [^\n]* sc1\s*\n\s*s_nop 1", b) for b in stores), "an asm write-through store without its s_nop 1"
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", text):
        assert m.group(1) == "0"
