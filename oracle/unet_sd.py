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
when present.  Anchors: parameter counts 859,520,964 (SD-v1.5) and 865,910,724
(SD-2.1-base) of this module's OWN shape table (tests/test_oracle.py).  The config
dataclass and the shape table are the oracle's own (round 4): nothing but the seeded
weight VALUES is shared with diffusion_pullback_amd/configs.py, and ``forward`` checks
the shapes of the parameters it is handed against its own table.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class SDConfig:
    """The oracle's OWN reading of diffusers' ``unet/config.json`` -- deliberately not imported from the product
    package: a wrong head count, width or projection kind in ``diffusion_pullback_amd/configs.py`` must not be shared by the checker.
    Field names follow the product's dataclass so that fixtures (``SDConfig(**f["cfg"])``) and duck-typed callers work with either."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    down_attn: Tuple[bool, ...] = (True, True, True, False)      # CrossAttnDownBlock2D x3, DownBlock2D
    up_attn: Tuple[bool, ...] = (False, True, True, True)        # UpBlock2D, CrossAttnUpBlock2D x3
    heads: Tuple[int, ...] = (8, 8, 8, 8)                        # ``attention_head_dim`` of diffusers 0.11 = head COUNT per down block
    cross_dim: int = 768
    groups: int = 32
    sample_size: int = 64
    use_linear_projection: bool = False
    ctx_len: int = 77

    @property
    def temb_ch(self) -> int:
        return 4 * self.block_out_channels[0]


# runwayml/stable-diffusion-v1-5 unet/config.json (BASELINE.json configs[2..4])
SD15 = SDConfig()
# stabilityai/stable-diffusion-2-1-base unet/config.json -- the model id of the reference's SD scripts
# (src/scripts/main_various_local_encoder_pullback_with_edit_prompt.sh:11): attention_head_dim [5, 10, 20, 20] (64-wide heads),
# cross_attention_dim 1024 (OpenCLIP-H), use_linear_projection true
SD21_BASE = SDConfig(heads=(5, 10, 20, 20), cross_dim=1024, use_linear_projection=True)


def param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """``state_dict`` keys and shapes of diffusers' UNet2DConditionModel for ``cfg`` -- the oracle's own table, written from the module
    tree of the published class (anchors: 859,520,964 parameters for SD-v1.5, 865,910,724 for SD-2.1-base; tests/test_oracle.py)."""
    out: Dict[str, Tuple[int, ...]] = {}
    C = tuple(cfg.block_out_channels)
    T = 4 * C[0]

    def affine(name, n):                               # GroupNorm / LayerNorm
        out[f"{name}.weight"] = (n,)
        out[f"{name}.bias"] = (n,)

    def conv2d(name, cin, cout, ks):
        out[f"{name}.weight"] = (cout, cin, ks, ks)
        out[f"{name}.bias"] = (cout,)

    def linear(name, cin, cout, bias=True):
        out[f"{name}.weight"] = (cout, cin)
        if bias:
            out[f"{name}.bias"] = (cout,)

    def ResnetBlock2D(name, cin, cout):
        affine(f"{name}.norm1", cin)
        conv2d(f"{name}.conv1", cin, cout, 3)
        linear(f"{name}.time_emb_proj", T, cout)
        affine(f"{name}.norm2", cout)
        conv2d(f"{name}.conv2", cout, cout, 3)
        if cin != cout:
            conv2d(f"{name}.conv_shortcut", cin, cout, 1)

    def CrossAttention(name, width, kv_width):
        linear(f"{name}.to_q", width, width, bias=False)
        linear(f"{name}.to_k", kv_width, width, bias=False)
        linear(f"{name}.to_v", kv_width, width, bias=False)
        linear(f"{name}.to_out.0", width, width)

    def Transformer2DModel(name, width):
        affine(f"{name}.norm", width)
        for proj in ("proj_in", "proj_out"):
            if cfg.use_linear_projection:
                linear(f"{name}.{proj}", width, width)
            else:
                conv2d(f"{name}.{proj}", width, width, 1)
        blk = f"{name}.transformer_blocks.0"           # BasicTransformerBlock, depth 1
        affine(f"{blk}.norm1", width)
        CrossAttention(f"{blk}.attn1", width, width)
        affine(f"{blk}.norm2", width)
        CrossAttention(f"{blk}.attn2", width, cfg.cross_dim)
        affine(f"{blk}.norm3", width)
        linear(f"{blk}.ff.net.0.proj", width, 2 * 4 * width)      # GEGLU: value | gate
        linear(f"{blk}.ff.net.2", 4 * width, width)

    linear("time_embedding.linear_1", C[0], T)
    linear("time_embedding.linear_2", T, T)
    conv2d("conv_in", cfg.in_channels, C[0], 3)
    skip_widths = [C[0]]                               # what the up path will pop, in push order
    width = C[0]
    for i, cout in enumerate(C):
        for j in range(cfg.layers_per_block):
            ResnetBlock2D(f"down_blocks.{i}.resnets.{j}", width, cout)
            width = cout
            if cfg.down_attn[i]:
                Transformer2DModel(f"down_blocks.{i}.attentions.{j}", width)
            skip_widths.append(width)
        if i + 1 < len(C):
            conv2d(f"down_blocks.{i}.downsamplers.0.conv", width, width, 3)
            skip_widths.append(width)
    ResnetBlock2D("mid_block.resnets.0", width, width)
    Transformer2DModel("mid_block.attentions.0", width)
    ResnetBlock2D("mid_block.resnets.1", width, width)
    for i, cout in enumerate(reversed(C)):
        for j in range(cfg.layers_per_block + 1):
            ResnetBlock2D(f"up_blocks.{i}.resnets.{j}", width + skip_widths.pop(), cout)
            width = cout
            if cfg.up_attn[i]:
                Transformer2DModel(f"up_blocks.{i}.attentions.{j}", width)
        if i + 1 < len(C):
            conv2d(f"up_blocks.{i}.upsamplers.0.conv", width, width, 3)
    assert not skip_widths
    affine("conv_norm_out", C[0])
    conv2d("conv_out", C[0], cfg.out_channels, 3)
    return out


def check_params(p: Params, cfg) -> None:
    """Every tensor of ``p`` has the key and shape the oracle's own table gives for ``cfg`` (``p`` may be a prefix-restricted subset)."""
    want = param_shapes(cfg)
    for k, v in p.items():
        if k not in want:
            raise KeyError(f"parameter {k!r} is not a key of UNet2DConditionModel for {cfg}")
        if tuple(v.shape) != want[k]:
            raise ValueError(f"parameter {k!r} has shape {tuple(v.shape)}, the oracle's table for {cfg} says {want[k]}")


def init_params(cfg, seed: int = 0, gain: float = 1.0, dtype=torch.float32, only_prefix=None, spectrum=None) -> Params:
    """Seeded synthetic weights.  Only the VALUES are shared with the product (one generator stream, so the HIP engine and this oracle see the
    same bits: diffusion_pullback_amd.configs.sd_init_params); the names and shapes they arrive under are checked against this module's own
    table, and a full draw must cover it exactly."""
    from diffusion_pullback_amd.configs import sd_init_params
    p = sd_init_params(cfg, seed=seed, gain=gain, dtype=dtype, only_prefix=only_prefix, spectrum=spectrum)
    check_params(p, cfg)
    if only_prefix is None and set(p) != set(param_shapes(cfg)):
        raise KeyError(f"synthetic weights miss {sorted(set(param_shapes(cfg)) - set(p))[:5]} ...")
    return p


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


def forward(p: Params, cfg, x, t, ctx, stop: Optional[Tuple[str, int]] = None):
    """``stop=(op, idx)`` -> feature map of get_h (utils.py:438-527); None -> eps."""
    check_params(p, cfg)
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
