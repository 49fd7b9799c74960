"""Command line of the pullback editing path -- ``python -m diffusion_pullback_amd.main --flags``.

Keeps the reference's live-path flags and preset rules so its scripts run unchanged
(reference src/main.py:13-34, src/utils/define_argparser.py:15-242, src/scripts/*.sh):
``--run_edit_local_encoder_pullback_zt True`` dispatches to ``EditStableDiffusion`` when
``'stable-diffusion' in model_name`` and to ``EditUncondDiffusion`` otherwise, with the call-site
constants of main.py:30-34 (op='mid', block_idx=0, vis_num=4, vis_num_pc=2, pca_rank=2) as defaults.
Flags of the reference's dead experiments are accepted and ignored with a note.
New flags (not in the reference): --pca_rank, --op, --block_idx, --dtype bf16, --weights (state-dict
file; default: seeded synthetic weights, there are no checkpoints offline), --net_scale (reduced nets),
--vae (none | synthetic | <state-dict file>: decode the edited latents to PNGs with the on-device AutoencoderKL),
--text_encoder (none | synthetic | <state-dict file>) + --tokenizer_dir (CLIP vocab.json / merges.txt): on-device prompt encoder.
"""
from __future__ import annotations

import argparse
import os
import random
import sys

import numpy as np
import torch

from . import configs as cf
from .edit import EditStableDiffusion, EditUncondDiffusion
from .pullback import PullbackUNet

# reference src/configs/params.py:1-27
X_SPACE_GUIDANCE_SCALE_DICT = {
    "stable-diffusion": {1.0: 0.5, 0.9: 0.5, 0.8: 1, 0.7: 1, 0.6: 2, 0.5: 2, 0.4: 2, 0.3: 2, 0.2: 2, 0.1: 2, 0.0: 0},
    "uncond": {1.0: 0.5, 0.8: 1, 0.6: 4, 0.4: 16, 0.2: 16},
}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("true",):
        return True
    if v.lower() in ("false",):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


_FLAGS = [  # (name, type, default) -- define_argparser.py:20-110, live path only
    ("sh_file_name", str, ""), ("device", str, "cuda:0"), ("dtype", str, "fp32"), ("seed", int, 0), ("result_folder", str, "./runs/"),
    ("model_name", str, ""), ("dataset_name", str, ""), ("image_size", int, 256), ("c_in", int, 3), ("sample_idx", int, 0),
    ("for_prompt", str, ""), ("inv_prompt", str, ""), ("neg_prompt", str, ""), ("for_steps", int, 100), ("inv_steps", int, 100),
    ("performance_boosting_t", float, 0.0), ("use_yh_custom_scheduler", str2bool, True), ("guidance_scale", float, 0),
    ("edit_prompt", str, ""), ("use_x_space_guidance", str2bool, False), ("x_space_guidance_edit_step", float, 1),
    ("x_space_guidance_scale", float, 0), ("x_space_guidance_num_step", int, 0), ("x_space_guidance_use_edit_prompt", str2bool, True),
    ("h_t", float, 0.8), ("edit_t", float, 1.0), ("x_edit_step_size", float, 0), ("pca_device", str, "cpu"), ("buffer_device", str, "cpu"),
    ("save_result_as", str, "image"), ("run_ddim_forward", str2bool, False), ("run_ddim_inversion", str2bool, False),
    ("run_edit_local_encoder_pullback_zt", str2bool, False),
    # new
    ("pca_rank", int, 2), ("op", str, "mid"), ("block_idx", int, 0), ("vis_num", int, 4), ("vis_num_pc", int, 2), ("weights", str, ""),
    ("net_scale", str, "full"), ("vae", str, "none"), ("text_encoder", str, "none"), ("tokenizer_dir", str, ""),
    ("timing", str2bool, False),      # wall-clock breakdown of the run by phase (timing.py; synchronises at phase boundaries)
    # U-Net batch of the edit trajectories: the 2 * vis_num_pc independent (pc, +-) edits of edit.py:276-307 run together (x-space guidance as one batch-2n call per
    # step, the n * (vis_num + 1) decode trajectories as one batch).  Same files, tensors equal up to 16-bit rounding (tile / split-K choices depend on the batch);
    # 0 / 1 = one experiment after another, memory_bound latents per call (the setting for bitwise-reproducible runs); an explicit --memory_bound caps it
    # (and replaces the per-model memory_bound constant: it is then the bound of every U-Net call, the 2 of an x-space-guidance pair being the floor)
    ("trajectory_batch", int, 20),
]


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    for name, typ, default in _FLAGS:
        p.add_argument("--" + name, type=typ, default=default, required=False)
    p.add_argument("--note", type=str, required=True)
    # --memory_bound is NOT a flag of the reference (define_argparser.py:211-221 sets memory_bound to a per-model constant, kept in preset()); here an explicit
    # value is the user's bound on the U-Net batch -- decode chunks, trajectory batching, the guidance chains per call and the engine's max_batch all respect it
    p.add_argument("--memory_bound", type=int, default=None)
    args, extra = p.parse_known_args(argv)
    if args.memory_bound is not None and args.memory_bound < 1:
        p.error("--memory_bound must be a positive integer")
    args.memory_bound_given = args.memory_bound
    if extra:
        print(f"note: ignoring flags of experiments outside the pullback path: {extra}")
    return args


def seed_everything(seed: int):
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)


def preset(args):
    """define_argparser.py:145-233 (folders, derived args, asserts)."""
    seed_everything(args.seed)
    args.is_stable_diffusion = "stable-diffusion" in args.model_name
    if args.is_stable_diffusion:
        args.exp = f"Stable_Diffusion-{args.dataset_name}-{args.note}"
    else:
        if args.model_name not in ("CelebA_HQ_HF", "LSUN_church_HF", "FFHQ_HF"):
            raise ValueError("model_name choice: [CelebA_HQ_HF, LSUN_church_HF, FFHQ_HF]")
        args.exp = f"{args.model_name}-{args.dataset_name}-{args.note}"
    args.exp_folder = os.path.join(args.result_folder, args.exp)
    args.obs_folder = os.path.join(args.exp_folder, "obs")
    args.result_folder = os.path.join(args.exp_folder, "results")
    os.makedirs(args.obs_folder, exist_ok=True)
    os.makedirs(args.result_folder, exist_ok=True)
    args.device = torch.device(args.device)
    args.compute_dtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}.get(args.dtype)   # define_argparser.py:196 (+ bf16)
    if args.compute_dtype is None:
        raise ValueError("dtype choice: [fp32, fp16, bf16]")
    args.dtype = torch.float32                      # boundary dtype of latents
    if args.use_x_space_guidance:
        args.x_space_guidance_scale = X_SPACE_GUIDANCE_SCALE_DICT["stable-diffusion" if args.is_stable_diffusion else "uncond"][args.h_t]
    given = getattr(args, "memory_bound_given", None)
    if given:
        args.trajectory_batch = min(args.trajectory_batch, given)
    if args.is_stable_diffusion:
        args.c_in, args.image_size, args.memory_bound = 4, 64, given or 5
        assert args.use_yh_custom_scheduler
        assert args.performance_boosting_t <= 0
    else:
        args.c_in, args.image_size, args.memory_bound, args.noise_schedule = 3, 256, given or 50, "linear"
        assert args.use_yh_custom_scheduler
        assert args.for_steps == 100
        assert args.performance_boosting_t == 0.2
    return args


def build_unet(args) -> PullbackUNet:
    from . import weights as W
    small = args.net_scale != "full"
    # U-Net batch: memory_bound latents of the decode loop (x2 under classifier-free guidance), never below the 2 of x-space guidance
    max_batch = max(2, min(args.memory_bound, args.vis_num + 1) * (2 if args.guidance_scale > 1.0 else 1))
    if getattr(args, "trajectory_batch", 0) > 1 and args.is_stable_diffusion and args.run_edit_local_encoder_pullback_zt:
        # 2 * vis_num_pc chains together: batch 4 * vis_num_pc for the guidance step, 2 * vis_num_pc * (vis_num + 1) decode trajectories
        max_batch = max(max_batch, min(args.trajectory_batch, 2 * args.vis_num_pc * (args.vis_num + 1)) * (2 if args.guidance_scale > 1.0 else 1),
                        min(args.trajectory_batch, 4 * args.vis_num_pc))
    if args.is_stable_diffusion:
        cfg = cf.sd_config_for(args.model_name)           # SD-v1.x or SD-2(.1)-base; anything else raises
        if small:
            cfg = cf.SDConfig(block_out_channels=(32, 64), layers_per_block=1, down_attn=(True, False), up_attn=(False, True),
                              heads=(2, 2), cross_dim=64, groups=8, sample_size=16, use_linear_projection=cfg.use_linear_projection)
        params = torch.load(args.weights, map_location="cpu") if args.weights else cf.sd_init_params(cfg, seed=args.seed)
        W.check_shapes(params, cf.sd_param_shapes(cfg), f"{args.model_name} U-Net")
        if small:
            args.image_size = cfg.sample_size
        return PullbackUNet("sd", cfg, params, dtype=args.compute_dtype, device=args.device, max_batch=max_batch, max_rank=max(args.pca_rank, 2))
    cfg = cf.CELEBA_HQ_256 if not small else cf.DDPMConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=32)
    if args.weights:                                      # diffusers UNet2DModel keys (google/ddpm-ema-celebahq-256) or vendored names
        params = W.ddpm_hf_to_vendored_names(torch.load(args.weights, map_location="cpu"), cfg)
    else:
        params = cf.ddpm_init_params(cfg, seed=args.seed)
    W.check_shapes(params, cf.ddpm_param_shapes(cfg), f"{args.model_name} U-Net")
    if small:
        args.image_size = cfg.resolution
    return PullbackUNet("ddpm", cfg, params, dtype=args.compute_dtype, device=args.device, max_batch=max_batch, max_rank=max(args.pca_rank, 2))


def build_vae(args):
    """``pipe.vae`` of the reference (edit.py:144-146, :476-480) on the HIP engine; None keeps latents as the output."""
    if args.vae == "none" or not args.is_stable_diffusion:
        return None
    from .vae import AutoencoderKL
    small = args.net_scale != "full"
    cfg = cf.SD15_VAE if not small else cf.VAEConfig(block_out_channels=(32, 64), layers_per_block=1, groups=8, sample_size=2 * args.image_size)
    params = cf.vae_init_params(cfg, seed=args.seed) if args.vae == "synthetic" else torch.load(args.vae, map_location="cpu")
    return AutoencoderKL(cfg, params, dtype=args.compute_dtype, device=args.device, max_batch=1)


def build_prompt_encoder(args):
    """``pipe._encode_prompt`` of the reference (edit.py:505-522) on the HIP engine; None keeps the seeded stand-in
    embeddings.  The BPE tokenizer needs CLIP's vocabulary files (``--tokenizer_dir``, loaded with transformers'
    CLIPTokenizer); without them a byte-level stand-in tokenizer is used, which is only meaningful with synthetic weights."""
    if args.text_encoder == "none" or not args.is_stable_diffusion or args.net_scale != "full":
        return None
    from .text_encoder import ClipTextEncoder
    cfg = cf.clip_config_for(args.model_name)
    if args.tokenizer_dir:
        from transformers import CLIPTokenizer
        tk = CLIPTokenizer.from_pretrained(args.tokenizer_dir)
        tokenizer = lambda s: tk(s, padding="max_length", max_length=cfg.max_position, truncation=True).input_ids
    else:
        tokenizer = lambda s: ([cfg.vocab_size - 2] + [256 + b for b in s.encode()][: cfg.max_position - 2] + [cfg.vocab_size - 1] * cfg.max_position)[: cfg.max_position]
    params = cf.clip_init_params(cfg, seed=args.seed) if args.text_encoder == "synthetic" else torch.load(args.text_encoder, map_location="cpu")
    return ClipTextEncoder(cfg, params, dtype=args.compute_dtype, device=args.device, max_batch=1, tokenizer=tokenizer).encode_prompt


def main(argv=None):
    import time
    from . import timing as T
    t_start = time.perf_counter()
    args = preset(parse_args(argv))
    if getattr(args, "timing", False):
        T.enable(True)
    with T.phase("U-Net weights (synthetic draw / load) + engine build + upload"):
        unet = build_unet(args)
    if args.is_stable_diffusion:
        print("is stable-diffusion")
        with T.phase("VAE weights + engine build"):
            vae = build_vae(args)
        with T.phase("text-encoder weights + engine build"):
            penc = build_prompt_encoder(args)
        edit = EditStableDiffusion(args, unet=unet, vae=vae, prompt_encoder=penc)
    else:
        print("is NOT stable-diffusion")
        edit = EditUncondDiffusion(args, unet=unet)
    if args.run_edit_local_encoder_pullback_zt:                                  # main.py:30-34
        if args.is_stable_diffusion:
            edit.run_edit_local_encoder_pullback_zt(idx=args.sample_idx, op=args.op, block_idx=args.block_idx, vis_num=args.vis_num,
                                                    vis_num_pc=args.vis_num_pc, pca_rank=args.pca_rank, edit_prompt=args.edit_prompt)
        else:
            edit.run_edit_local_encoder_pullback_zt(idx=args.sample_idx, op=args.op, block_idx=args.block_idx, vis_num=args.vis_num,
                                                    vis_num_pc=args.vis_num_pc, pca_rank=args.pca_rank)
    if args.run_ddim_forward:
        edit.run_DDIMforward(num_samples=5)
    if args.run_ddim_inversion:
        edit.run_DDIMinversion(idx=args.sample_idx)
    if T.ENABLED:
        print(T.report(time.perf_counter() - t_start))
    return edit


if __name__ == "__main__":
    main(sys.argv[1:])
