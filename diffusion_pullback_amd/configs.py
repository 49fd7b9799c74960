"""Network configurations and seeded synthetic weights for the two U-Net families of the pullback path.

No checkpoints or datasets are reachable offline, so throughput runs and tests use random-init weights
at the exact architecture shapes, generated on the CPU generator (identical bits for the HIP engine and
for the CPU oracle).  Parameter names equal the upstream ``state_dict`` keys (vendored DDPM:
reference src/models/ddpm/diffusion.py:22-129; SD: diffusers UNet2DConditionModel), so real weights
drop in unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

Params = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class DDPMConfig:
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4, 4)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = (16,)
    in_channels: int = 3
    out_ch: int = 3
    resolution: int = 256
    groups: int = 32
    gn_eps: float = 1e-6

    @property
    def temb_ch(self) -> int:
        return self.ch * 4


# configs/custom_celeba_ddpm.yml:21-30
CELEBA_HQ_256 = DDPMConfig()



@dataclass(frozen=True)
class SDConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    down_attn: Tuple[bool, ...] = (True, True, True, False)
    up_attn: Tuple[bool, ...] = (False, True, True, True)
    heads: Tuple[int, ...] = (8, 8, 8, 8)          # per down block; mid uses heads[-1]
    cross_dim: int = 768
    groups: int = 32
    sample_size: int = 64
    use_linear_projection: bool = False            # SD-2.x stores proj_in/out as Linear
    ctx_len: int = 77

    @property
    def temb_ch(self) -> int:
        return self.block_out_channels[0] * 4


SD15 = SDConfig()
# stabilityai/stable-diffusion-2-1-base (and 2-base), the id the reference's SD scripts pass
# (src/scripts/main_various_local_encoder_pullback_with_edit_prompt.sh:11): OpenCLIP-H context (1024 wide), 64-wide heads
# (diffusers' ``attention_head_dim`` [5,10,20,20] is the per-block head COUNT in 0.11), Linear proj_in / proj_out.
SD21_BASE = SDConfig(heads=(5, 10, 20, 20), cross_dim=1024, use_linear_projection=True)


def sd_config_for(model_name: str) -> SDConfig:
    """U-Net architecture for a Hugging Face model id; unknown ``stable-diffusion`` ids raise instead of silently
    building the wrong network (define_argparser.py:153 accepts any id containing 'stable-diffusion')."""
    n = model_name.lower()
    if "stable-diffusion-v1" in n or "stable-diffusion-1" in n:
        return SD15
    if "stable-diffusion-2" in n and n.rstrip("/").endswith("base"):
        return SD21_BASE
    raise ValueError(f"unknown Stable Diffusion model id {model_name!r}: supported are the SD-v1.x family (runwayml/stable-diffusion-v1-5, "
                     "CompVis/stable-diffusion-v1-4, ...) and stabilityai/stable-diffusion-2-base / -2-1-base (epsilon-prediction, "
                     "64x64 latents); the 768-v models use v-prediction, which the reference's scheduler step does not implement either")



def ddpm_param_shapes(cfg: DDPMConfig) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of every parameter (equals the vendored module's state_dict layout)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(n, i, o):
        s[n + ".weight"] = (o, i); s[n + ".bias"] = (o,)

    def conv(n, i, o, k):
        s[n + ".weight"] = (o, i, k, k); s[n + ".bias"] = (o,)

    def gn(n, c):
        s[n + ".weight"] = (c,); s[n + ".bias"] = (c,)

    def resblock(n, i, o):
        gn(n + ".norm1", i); conv(n + ".conv1", i, o, 3); lin(n + ".temb_proj", cfg.temb_ch, o)
        gn(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".nin_shortcut", i, o, 1)

    def attn(n, c):
        gn(n + ".norm", c)
        for w in ("q", "k", "v", "proj_out"):
            conv(n + "." + w, c, c, 1)

    lin("temb.dense.0", cfg.ch, cfg.temb_ch); lin("temb.dense.1", cfg.temb_ch, cfg.temb_ch)
    conv("conv_in", cfg.in_channels, cfg.ch, 3)
    nres = len(cfg.ch_mult)
    in_mult = (1,) + tuple(cfg.ch_mult)
    res = cfg.resolution
    bi = cfg.ch
    for lvl in range(nres):
        bi, bo = cfg.ch * in_mult[lvl], cfg.ch * cfg.ch_mult[lvl]
        for blk in range(cfg.num_res_blocks):
            resblock(f"down.{lvl}.block.{blk}", bi, bo)
            bi = bo
            if res in cfg.attn_resolutions:
                attn(f"down.{lvl}.attn.{blk}", bi)
        if lvl != nres - 1:
            conv(f"down.{lvl}.downsample.conv", bi, bi, 3)
            res //= 2
    resblock("mid.block_1", bi, bi); attn("mid.attn_1", bi); resblock("mid.block_2", bi, bi)
    for lvl in reversed(range(nres)):
        bo = cfg.ch * cfg.ch_mult[lvl]
        skip = bo
        for blk in range(cfg.num_res_blocks + 1):
            if blk == cfg.num_res_blocks:
                skip = cfg.ch * in_mult[lvl]
            resblock(f"up.{lvl}.block.{blk}", bi + skip, bo)
            bi = bo
            if res in cfg.attn_resolutions:
                attn(f"up.{lvl}.attn.{blk}", bi)
        if lvl != 0:
            conv(f"up.{lvl}.upsample.conv", bi, bi, 3)
            res *= 2
    gn("norm_out", bi); conv("conv_out", bi, cfg.out_ch, 3)
    return s


@dataclass(frozen=True)
class Spectrum:
    """Spectrum shaping of the synthetic weights (SURVEY.md section 7 "parity is subtle (ii)").

    Random-init U-Nets have a nearly flat top of the Jacobian spectrum (sigma_1..5 within 1 % of each other on SD-1.5), so
    individual singular vectors are ill-conditioned and "top-5 |cos| >= 0.99" cannot be tested; trained networks have the
    fast-decaying spectra the paper reports.  No weight of these architectures is indexed by position, so the only place a
    synthetic network can hold a few dominant GLOBAL directions is a token-pooling op: the mid-block self-attention.  Shaping
    adds ``rank`` seeded channel directions to its output projection,  W_o += amp * sum_i decay^i a_i b_i^T  (a_i, b_i
    orthonormal), and scales its query projection by ``q_scale`` (near-uniform attention = mean pooling over the 8x8 tokens,
    as in an untrained attention layer).  The Jacobian then has ``rank`` leading singular values ~ amp * decay^i * c whose
    right vectors are (upstream Jacobian)^T applied to the pooled b_i directions -- they run through every down-block kernel --
    above the flat bulk.  Architecture, shapes and FLOPs are untouched."""
    rank: int = 12
    decay: float = 0.8
    amp: float = 400.0
    q_scale: float = 0.1
    # SD only: further transformer blocks (parameter prefixes such as "down_blocks.1.attentions.1") whose self-attention is shaped the same way.
    # The mid-block shaping is not in the prefix of a ('down', i) tap, so those taps keep the flat spectrum unless the LAST self-attention inside
    # their own prefix is shaped too: Spectrum.for_tap(op, idx).  Default () = the headline weights of rounds 2+ (bench.py, the goldens).
    also: Tuple[str, ...] = ()

    @staticmethod
    def for_tap(op: str, idx: int, **kw) -> "Spectrum":
        """Shaping that separates the top singular values AT tap (op, idx) of SD-v1.x/2.x: mid block + the last self-attention of the tap's prefix."""
        last = {("down", 0): "down_blocks.0.attentions.1", ("down", 1): "down_blocks.1.attentions.1", ("down", 2): "down_blocks.2.attentions.1",
                ("down", 3): "down_blocks.2.attentions.1", ("up", 1): "up_blocks.1.attentions.2", ("up", 2): "up_blocks.2.attentions.2",
                ("up", 3): "up_blocks.3.attentions.2"}.get((op, idx))
        if op == "up":
            kw.setdefault("amp", 100.0)      # two shaped layers in series (mid and the tap's own): amp 400 twice puts sigma_1 at 1.8e3 and
                                             # overflows the fp16 engine's cotangents; 100 gives 380 / 278 / 236 / 197 / 145 at up_3 (measured)
        return Spectrum(also=(last,) if last else (), **kw)


def _shape_spectrum(p: Params, w_out: str, w_q: str, sp: Spectrum, seed: int, salt: int = 0) -> None:
    if w_out not in p:                       # prefix-restricted parameter sets that stop before the mid block
        return
    g = torch.Generator().manual_seed(seed + 7919 + 104729 * salt)   # own stream: every other parameter keeps its unshaped bits
    w = p[w_out]
    c = w.shape[0]
    r = min(sp.rank, c)
    a = torch.linalg.qr(torch.randn(c, r, generator=g))[0]
    b = torch.linalg.qr(torch.randn(c, r, generator=g))[0]
    spike = (a * (sp.amp * sp.decay ** torch.arange(r, dtype=torch.float32))[None, :]) @ b.T
    p[w_out] = (w.float() + spike.reshape(w.shape)).to(w.dtype)
    p[w_q] = (p[w_q].float() * sp.q_scale).to(w.dtype)


def ddpm_init_params(cfg: DDPMConfig, seed: int = 0, gain: float = 1.0, dtype=torch.float32, spectrum: "Spectrum | None" = None) -> Params:
    """Seeded synthetic weights (no checkpoints are reachable offline).

    Fan-in scaled normal for matrices, GroupNorm affine near identity.  Generated on
    CPU so the oracle and the HIP engine see identical bits."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shp in ddpm_param_shapes(cfg).items():
        if name.endswith(".weight") and len(shp) == 1:          # norm gamma
            p[name] = (1.0 + 0.1 * torch.randn(shp, generator=g)).to(dtype)
        elif name.endswith(".bias"):
            p[name] = (0.05 * torch.randn(shp, generator=g)).to(dtype)
        else:
            fan_in = math.prod(shp[1:])
            p[name] = (gain * torch.randn(shp, generator=g) / math.sqrt(fan_in)).to(dtype)
    if spectrum is not None:
        _shape_spectrum(p, "mid.attn_1.proj_out.weight", "mid.attn_1.q.weight", spectrum, seed)
    return p


def sd_param_shapes(cfg: SDConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(n, i, o, bias=True):
        s[n + ".weight"] = (o, i)
        if bias:
            s[n + ".bias"] = (o,)

    def conv(n, i, o, k):
        s[n + ".weight"] = (o, i, k, k); s[n + ".bias"] = (o,)

    def norm(n, c):
        s[n + ".weight"] = (c,); s[n + ".bias"] = (c,)

    def resnet(n, i, o):
        norm(n + ".norm1", i); conv(n + ".conv1", i, o, 3); lin(n + ".time_emb_proj", cfg.temb_ch, o)
        norm(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)

    def attn(n, c, kv):
        lin(n + ".to_q", c, c, False); lin(n + ".to_k", kv, c, False); lin(n + ".to_v", kv, c, False)
        lin(n + ".to_out.0", c, c)

    def transformer(n, c):
        norm(n + ".norm", c)
        if cfg.use_linear_projection:
            lin(n + ".proj_in", c, c); lin(n + ".proj_out", c, c)
        else:
            conv(n + ".proj_in", c, c, 1); conv(n + ".proj_out", c, c, 1)
        tb = n + ".transformer_blocks.0"
        norm(tb + ".norm1", c); attn(tb + ".attn1", c, c)
        norm(tb + ".norm2", c); attn(tb + ".attn2", c, cfg.cross_dim)
        norm(tb + ".norm3", c); lin(tb + ".ff.net.0.proj", c, 8 * c); lin(tb + ".ff.net.2", 4 * c, c)

    boc = cfg.block_out_channels
    nb = len(boc)
    lin("time_embedding.linear_1", boc[0], cfg.temb_ch); lin("time_embedding.linear_2", cfg.temb_ch, cfg.temb_ch)
    conv("conv_in", cfg.in_channels, boc[0], 3)
    ch = boc[0]
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch, boc[i]); ch = boc[i]
            if cfg.down_attn[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", ch)
        if i != nb - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    resnet("mid_block.resnets.0", ch, ch); transformer("mid_block.attentions.0", ch); resnet("mid_block.resnets.1", ch, ch)
    rev = tuple(reversed(boc))
    prev = rev[0]
    for i in range(nb):
        out = rev[i]
        inp = rev[min(i + 1, nb - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = inp if j == cfg.layers_per_block else out
            rin = prev if j == 0 else out
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, out)
            if cfg.up_attn[i]:
                transformer(f"up_blocks.{i}.attentions.{j}", out)
        if i != nb - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
        prev = out
    norm("conv_norm_out", boc[0]); conv("conv_out", boc[0], cfg.out_channels, 3)
    return s


def sd_init_params(cfg: SDConfig, seed: int = 0, gain: float = 1.0, dtype=torch.float32, only_prefix=None,
                   spectrum: "Spectrum | None" = None) -> Params:
    """Seeded synthetic weights at the exact architecture shapes (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shp in sd_param_shapes(cfg).items():
        if name.endswith(".weight") and len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        else:
            t = gain * torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        if only_prefix is None or name.startswith(only_prefix):
            p[name] = t.to(dtype)
    if spectrum is not None:
        tb = "mid_block.attentions.0.transformer_blocks.0.attn1."
        _shape_spectrum(p, tb + "to_out.0.weight", tb + "to_q.weight", spectrum, seed)
        for n, pre in enumerate(spectrum.also):
            tb = pre + ".transformer_blocks.0.attn1."
            _shape_spectrum(p, tb + "to_out.0.weight", tb + "to_q.weight", spectrum, seed, salt=n + 1)
    return p


# =================================================================== SD image autoencoder (AutoencoderKL)
@dataclass(frozen=True)
class VAEConfig:
    """diffusers==0.11 AutoencoderKL as shipped with runwayml/stable-diffusion-v1-5 (vae/config.json) -- the object
    behind ``pipe.vae`` at reference src/modules/edit.py:144-146 (encode) and :476-480 (decode)."""
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    groups: int = 32
    gn_eps: float = 1e-6
    sample_size: int = 512                         # image side; latents are sample_size / 2^(len(boc)-1)
    scaling_factor: float = 0.18215                # edit.py:146 / :477

    @property
    def latent_size(self) -> int:
        return self.sample_size >> (len(self.block_out_channels) - 1)


SD15_VAE = VAEConfig()


def vae_param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    boc = cfg.block_out_channels
    L2 = 2 * cfg.latent_channels

    def conv(n, ci, co, k=3):
        s[n + ".weight"] = (co, ci, k, k); s[n + ".bias"] = (co,)

    def lin(n, ci, co):
        s[n + ".weight"] = (co, ci); s[n + ".bias"] = (co,)

    def norm(n, c):
        s[n + ".weight"] = (c,); s[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", ci, co); norm(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    def mid(n, c):
        resnet(n + ".resnets.0", c, c)
        a = n + ".attentions.0"
        norm(a + ".group_norm", c)
        for q in ("query", "key", "value", "proj_attn"):
            lin(a + "." + q, c, c)
        resnet(n + ".resnets.1", c, c)

    conv("encoder.conv_in", cfg.in_channels, boc[0])
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch, co)
            ch = co
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch, ch)
    mid("encoder.mid_block", ch)
    norm("encoder.conv_norm_out", ch)
    conv("encoder.conv_out", ch, L2)
    conv("quant_conv", L2, L2, 1)
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = tuple(reversed(boc))
    conv("decoder.conv_in", cfg.latent_channels, rev[0])
    mid("decoder.mid_block", rev[0])
    ch = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch, co)
            ch = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch)
    norm("decoder.conv_norm_out", ch)
    conv("decoder.conv_out", ch, cfg.out_channels)
    return s


def vae_init_params(cfg: VAEConfig, seed: int = 0, gain: float = 1.0, dtype=torch.float32) -> Params:
    """Seeded synthetic weights at the exact architecture shapes (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shp in vae_param_shapes(cfg).items():
        if name.endswith(".weight") and len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        else:
            t = gain * torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        p[name] = t.to(dtype)
    return p


# =================================================================== SD prompt encoder (CLIP ViT-L/14 text model)
@dataclass(frozen=True)
class CLIPTextConfig:
    """transformers CLIPTextModel as shipped with runwayml/stable-diffusion-v1-5 (text_encoder/config.json) -- the object
    behind ``pipe._encode_prompt`` at reference src/modules/edit.py:505-522 (last_hidden_state of the 77 padded tokens)."""
    vocab_size: int = 49408
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_position: int = 77
    eps: float = 1e-5
    act: str = "quick_gelu"                        # MLP activation: "quick_gelu" (CLIP ViT-L/14) | "gelu" (OpenCLIP ViT-H/14)


SD15_CLIP = CLIPTextConfig()
# stabilityai/stable-diffusion-2(-1)-base text_encoder/config.json: OpenCLIP ViT-H/14 text tower without its last layer (diffusers ships 23 of the
# 24 layers: SD-2.x conditions on the penultimate hidden state), 1024 wide, 16 heads, erf GELU; its output is the 1024-wide context of SD21_BASE
SD21_CLIP = CLIPTextConfig(hidden=1024, layers=23, heads=16, intermediate=4096, act="gelu")


def clip_config_for(model_name: str) -> CLIPTextConfig:
    return SD21_CLIP if sd_config_for(model_name) is SD21_BASE else SD15_CLIP


def clip_param_shapes(cfg: CLIPTextConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    h = cfg.hidden

    def lin(n, ci, co):
        s[n + ".weight"] = (co, ci); s[n + ".bias"] = (co,)

    def norm(n):
        s[n + ".weight"] = (h,); s[n + ".bias"] = (h,)

    s["text_model.embeddings.token_embedding.weight"] = (cfg.vocab_size, h)
    s["text_model.embeddings.position_embedding.weight"] = (cfg.max_position, h)
    for i in range(cfg.layers):
        pre = f"text_model.encoder.layers.{i}"
        norm(pre + ".layer_norm1")
        for q in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(f"{pre}.self_attn.{q}", h, h)
        norm(pre + ".layer_norm2")
        lin(pre + ".mlp.fc1", h, cfg.intermediate)
        lin(pre + ".mlp.fc2", cfg.intermediate, h)
    norm("text_model.final_layer_norm")
    return s


def clip_init_params(cfg: CLIPTextConfig, seed: int = 0, gain: float = 1.0, dtype=torch.float32) -> Params:
    """Seeded synthetic weights at the exact architecture shapes (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shp in clip_param_shapes(cfg).items():
        if "embedding" in name:
            t = 0.02 * torch.randn(shp, generator=g)
        elif name.endswith(".weight") and len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        else:
            t = gain * torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        p[name] = t.to(dtype)
    return p
