"""ORACLE (test infrastructure, never imported by the product package).

PARITY UNPINNED.  Functional plain-PyTorch restatement of the image autoencoder
``AutoencoderKL`` of Stable Diffusion v1.5 as published in ``diffusers==0.11.0``
(third-party; pinned in /root/reference/requirements.txt:4; NOT vendored under
/root/reference and not installed here).  The reference touches it in two places:

  * src/modules/edit.py:144-146  ``z0 = vae.encode(x0).latent_dist.sample() * 0.18215``
  * src/modules/edit.py:476-480  ``x0 = vae.decode(1/0.18215 * latents).sample``,
    then ``(x0 / 2 + 0.5).clamp(0, 1)`` and ``save_image``

Architecture facts restated here (public vae/config.json of runwayml/stable-diffusion-v1-5 [ext]):
encoder conv_in 3->128, four DownEncoderBlock2D (128, 256, 512, 512; 2 ResBlocks each, no time
embedding; stride-2 3x3 downsampler with F.pad (0,1,0,1) after the first three), mid
(Res, single-head AttentionBlock over the 64x64 positions, Res), GroupNorm(32, eps 1e-6) + SiLU +
conv_out -> 8 moments channels, quant_conv 1x1; decoder post_quant_conv 1x1, conv_in 4->512, mid,
four UpDecoderBlock2D (512, 512, 256, 128; 3 ResBlocks each, nearest x2 + conv3x3 after the
first three), GroupNorm + SiLU + conv_out -> 3.  Posterior = diagonal Gaussian with logvar clamped to
[-30, 20].  Parameter names equal diffusers' ``state_dict`` keys so real weights can be fed when
present.  Anchor: parameter count 83,653,863 (tests/test_oracle.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from diffusion_pullback_amd.configs import SD15_VAE, Params, VAEConfig  # noqa: F401
from diffusion_pullback_amd.configs import vae_init_params as init_params  # noqa: F401
from diffusion_pullback_amd.configs import vae_param_shapes as param_shapes  # noqa: F401


def _gn(p, n, x, cfg):
    return F.group_norm(x, cfg.groups, p[n + ".weight"], p[n + ".bias"], cfg.gn_eps)


def _conv(p, n, x, stride=1, padding=1):
    return F.conv2d(x, p[n + ".weight"], p[n + ".bias"], stride=stride, padding=padding)


def _resnet(p, pre, x, cfg):
    h = _conv(p, pre + ".conv1", F.silu(_gn(p, pre + ".norm1", x, cfg)))
    h = _conv(p, pre + ".conv2", F.silu(_gn(p, pre + ".norm2", h, cfg)))
    if pre + ".conv_shortcut.weight" in p:
        x = _conv(p, pre + ".conv_shortcut", x, padding=0)
    return x + h


def _attn(p, pre, x, cfg):
    """diffusers 0.11 AttentionBlock, num_head_channels=None -> one head over all channels."""
    b, c, hh, ww = x.shape
    h = _gn(p, pre + ".group_norm", x, cfg).reshape(b, c, hh * ww).transpose(1, 2)
    q = F.linear(h, p[pre + ".query.weight"], p[pre + ".query.bias"])
    k = F.linear(h, p[pre + ".key.weight"], p[pre + ".key.bias"])
    v = F.linear(h, p[pre + ".value.weight"], p[pre + ".value.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / (c ** 0.5), dim=-1) @ v
    a = F.linear(a, p[pre + ".proj_attn.weight"], p[pre + ".proj_attn.bias"])
    return x + a.transpose(1, 2).reshape(b, c, hh, ww)


def _mid(p, pre, x, cfg):
    x = _resnet(p, pre + ".resnets.0", x, cfg)
    x = _attn(p, pre + ".attentions.0", x, cfg)
    return _resnet(p, pre + ".resnets.1", x, cfg)


def encode_moments(p: Params, cfg: VAEConfig, x: torch.Tensor) -> torch.Tensor:
    """image [B,3,S,S] in [-1,1] -> moments [B, 2*latent, S/8, S/8] (mean | logvar), after quant_conv."""
    h = _conv(p, "encoder.conv_in", x)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = _resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, cfg)
        if i != n - 1:
            h = _conv(p, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _mid(p, "encoder.mid_block", h, cfg)
    h = _conv(p, "encoder.conv_out", F.silu(_gn(p, "encoder.conv_norm_out", h, cfg)))
    return _conv(p, "quant_conv", h, padding=0)


def encode(p: Params, cfg: VAEConfig, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (sample, mean, logvar); sample = mean + exp(0.5 logvar) * noise (edit.py:145 ``latent_dist.sample()``)."""
    mean, logvar = encode_moments(p, cfg, x).chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    if noise is None:
        noise = torch.zeros_like(mean)
    return mean + torch.exp(0.5 * logvar) * noise, mean, logvar


def decode(p: Params, cfg: VAEConfig, z: torch.Tensor) -> torch.Tensor:
    """latent [B,4,S/8,S/8] (already divided by the scaling factor, edit.py:477) -> image [B,3,S,S]."""
    h = _conv(p, "post_quant_conv", z, padding=0)
    h = _conv(p, "decoder.conv_in", h)
    h = _mid(p, "decoder.mid_block", h, cfg)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", h, cfg)
        if i != n - 1:
            h = _conv(p, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(p, "decoder.conv_out", F.silu(_gn(p, "decoder.conv_norm_out", h, cfg)))
