"""ORACLE (test infrastructure, never imported by the product package).

PARITY UNPINNED.  Functional plain-PyTorch restatement of the latent-space
``UNet2DConditionModel`` of Stable Diffusion v1.5 as published in
``diffusers==0.11.0`` (third-party; pinned in /root/reference/requirements.txt:4;
NOT vendored under /root/reference and not installed here).  The reference only
drives that class through its sub-modules:

  * src/utils/utils.py:438-527  get_h: time_proj -> time_embedding -> conv_in ->
    down_blocks -> mid_block -> up_blocks, returning the activations at (op, idx)
  * src/modules/edit.py:454-458 full ``unet(x, t, encoder_hidden_states=).sample``

Architecture facts restated here: block_out_channels
(320,640,1280,1280), 2 layers/block, CrossAttnDown x3 + Down, mid
(Res, Transformer, Res), Up + CrossAttnUp x3 with skip concat and nearest x2 +
conv3x3; 8 heads; GroupNorm(32) eps 1e-5 in ResBlocks / 1e-6 before the
transformer; time embedding [cos | sin] of dim 320, freq exponent -ln(1e4)*i/160.
``('down', i)`` returns the block output *after its downsampler* -- the intent of
the reference's unreachable utils.py:489-490 (its live branch :476-480 cannot run).

Parameter names equal diffusers' ``state_dict`` keys so real weights can be fed
when present.  Anchors: parameter count 859,520,964 (tests/test_oracle.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from diffusion_pullback_amd.configs import SD15, Params, SDConfig  # noqa: F401  (shared config + synthetic weights)
from diffusion_pullback_amd.configs import sd_init_params as init_params  # noqa: F401
from diffusion_pullback_amd.configs import sd_param_shapes as param_shapes  # noqa: F401


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _gn(p, n, x, groups, eps):
    return F.group_norm(x, groups, p[n + ".weight"], p[n + ".bias"], eps)


def _conv(p, n, x, stride=1, padding=1):
    return F.conv2d(x, p[n + ".weight"], p[n + ".bias"], stride=stride, padding=padding)


def _lin(p, n, x):
    return F.linear(x, p[n + ".weight"], p.get(n + ".bias"))


def _resnet(p, pre, x, temb, cfg):
    h = _conv(p, pre + ".conv1", F.silu(_gn(p, pre + ".norm1", x, cfg.groups, 1e-5)))
    h = h + _lin(p, pre + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = _conv(p, pre + ".conv2", F.silu(_gn(p, pre + ".norm2", h, cfg.groups, 1e-5)))
    if (pre + ".conv_shortcut.weight") in p:
        x = _conv(p, pre + ".conv_shortcut", x, padding=0)
    return x + h


def _mha(p, pre, x, ctx, heads):
    """CrossAttention: to_q/k/v without bias, softmax(q k^T d^-1/2) v, to_out.0 with bias."""
    b, n, c = x.shape
    d = c // heads
    q = _lin(p, pre + ".to_q", x).reshape(b, n, heads, d).transpose(1, 2)
    k = _lin(p, pre + ".to_k", ctx).reshape(b, ctx.shape[1], heads, d).transpose(1, 2)
    v = _lin(p, pre + ".to_v", ctx).reshape(b, ctx.shape[1], heads, d).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(b, n, c)
    return _lin(p, pre + ".to_out.0", o)


def _transformer(p, pre, x, ctx, heads, cfg):
    b, c, hh, ww = x.shape
    h = _gn(p, pre + ".norm", x, cfg.groups, 1e-6)
    if cfg.use_linear_projection:
        h = _lin(p, pre + ".proj_in", h.permute(0, 2, 3, 1).reshape(b, hh * ww, c))
    else:
        h = _conv(p, pre + ".proj_in", h, padding=0).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    tb = pre + ".transformer_blocks.0"
    ln = lambda n, z: F.layer_norm(z, (c,), p[n + ".weight"], p[n + ".bias"], 1e-5)
    z = ln(tb + ".norm1", h)
    h = h + _mha(p, tb + ".attn1", z, z, heads)
    h = h + _mha(p, tb + ".attn2", ln(tb + ".norm2", h), ctx, heads)
    f = _lin(p, tb + ".ff.net.0.proj", ln(tb + ".norm3", h))
    a, g = f.chunk(2, dim=-1)
    h = h + _lin(p, tb + ".ff.net.2", a * F.gelu(g))
    if cfg.use_linear_projection:
        h = _lin(p, pre + ".proj_out", h).reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
    else:
        h = _conv(p, pre + ".proj_out", h.reshape(b, hh, ww, c).permute(0, 3, 1, 2), padding=0)
    return h + x


def forward(p: Params, cfg: SDConfig, x, t, ctx, stop: Optional[Tuple[str, int]] = None):
    """``stop=(op, idx)`` -> feature map of get_h (utils.py:438-527); None -> eps."""
    if not torch.is_tensor(t):
        t = torch.tensor([float(t)])
    t = t.reshape(-1) if t.dim() else t[None]
    t = t.expand(x.shape[0]) if t.shape[0] != x.shape[0] else t
    emb = timestep_embedding(t, cfg.block_out_channels[0]).to(x.dtype)
    emb = _lin(p, "time_embedding.linear_2", F.silu(_lin(p, "time_embedding.linear_1", emb)))

    h = _conv(p, "conv_in", x)
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = _resnet(p, f"down_blocks.{i}.resnets.{j}", h, emb, cfg)
            if cfg.down_attn[i]:
                h = _transformer(p, f"down_blocks.{i}.attentions.{j}", h, ctx, cfg.heads[i], cfg)
            skips.append(h)
        if i != nb - 1:
            h = _conv(p, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
        if stop == ("down", i):
            return h

    h = _resnet(p, "mid_block.resnets.0", h, emb, cfg)
    h = _transformer(p, "mid_block.attentions.0", h, ctx, cfg.heads[-1], cfg)
    h = _resnet(p, "mid_block.resnets.1", h, emb, cfg)
    if stop == ("mid", 0):
        return h

    rheads = tuple(reversed(cfg.heads))
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(p, f"up_blocks.{i}.resnets.{j}", torch.cat([h, skips.pop()], dim=1), emb, cfg)
            if cfg.up_attn[i]:
                h = _transformer(p, f"up_blocks.{i}.attentions.{j}", h, ctx, rheads[i], cfg)
        if i != nb - 1:
            h = _conv(p, f"up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
        if stop == ("up", i):
            return h
    if stop is not None:
        raise ValueError(f"(op, block_idx) = {stop} is not valid")
    return _conv(p, "conv_out", F.silu(_gn(p, "conv_norm_out", h, cfg.groups, 1e-5)))
