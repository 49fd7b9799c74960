"""Build the op tape (include/dpb.h: dpb_net_desc) of a diffusion U-Net from its config and state dict.

Two architecture families, the ones the reference's live path drives:
  * ``build_ddpm``  -- pixel-space DDPM U-Net (HF google/ddpm-ema-celebahq-256 family; parameter
    naming of reference src/models/ddpm/diffusion.py:22-129), used by get_h_uncond /
    local_encoder_pullback_xt (src/utils/utils.py:114-249).
  * ``build_sd``    -- Stable-Diffusion UNet2DConditionModel (diffusers naming), used by get_h /
    local_encoder_pullback_zt (src/utils/utils.py:438-527, :722-816).

Weights are repacked once at load time into the layouts the HIP GEMM consumes:
  conv   W  [Cout][ky][kx][Cin_pad]     (forward operand, K contiguous)
         Wt [Cin_pad][ky][kx][Cout_pad] (adjoint operand: dX = conv^T(dY) is the same kernel)
  linear W  [out][in] as stored, Wt = W^T.
Activations are NHWC, channel counts padded to multiples of 8 only at the 3/4-channel boundary.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L


def _r8(c: int) -> int:
    return (c + 7) // 8 * 8


class Tape:
    def __init__(self, params: Dict[str, torch.Tensor], dtype: torch.dtype, device):
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        self.p = params
        self.dtype = dtype
        self.device = torch.device(device)
        self.buffers: List[Tuple[int, int, int]] = []
        self.valid: List[int] = []
        self.ops: List[dict] = []
        self.keep: List[torch.Tensor] = []          # device weights (kept alive for the engine)
        self.taps: Dict[object, int] = {}
        self.tap_shape: Dict[int, Tuple[int, int, int]] = {}   # buf -> (C, H, W)
        self.flops_to: Dict[int, float] = {}
        self._wcache: Dict[str, Tuple[int, int, int]] = {}

    # ------------------------------------------------------------- buffers / weights
    def buf(self, rows: int, ch: int, kind: int = L.BUF_ACT, valid: int = 0) -> int:
        self.buffers.append((rows, ch, kind))
        self.valid.append(valid)
        return len(self.buffers) - 1

    def _dev(self, t: torch.Tensor, dtype) -> int:
        t = t.detach().to(dtype).contiguous().to(self.device)
        self.keep.append(t)
        return t.data_ptr()

    def _conv_w(self, name, cin_p: int, cout_p: int, need_adj: bool, interleave: int = 0):
        """``name`` may be a tuple of parameter names: their weights (and biases) are concatenated along Cout,
        which fuses projections that read the same input (q/k/v) into one GEMM.  ``interleave`` = 64 permutes the output rows of a
        GEGLU projection from [a | g] halves to alternating 64-row blocks a0 g0 a1 g1 ..., so one 128-column GEMM tile holds a hidden
        unit's value AND gate (fused GEGLU epilogues, csrc/epilogue.h)."""
        key = f"{name}|{cin_p}|{cout_p}|{need_adj}|{interleave}"
        if key in self._wcache:
            return self._wcache[key]
        names = name if isinstance(name, tuple) else (name,)
        dev = self.device                 # pad / permute / convert on the engine's device (the fp32 tensors are uploaded once; on the CPU this was 5 s of an SD build)
        w = torch.cat([self.p[n + ".weight"].detach().to(device=dev, dtype=torch.float32) for n in names], dim=0)
        perm = None
        if interleave:
            f = w.shape[0] // 2
            blk = torch.arange(f, device=dev).reshape(f // interleave, interleave)
            perm = torch.stack([blk, blk + f], dim=1).reshape(-1)              # a-block b, g-block b, a-block b+1, ...
            w = w[perm]
        if w.dim() == 2:
            w = w[:, :, None, None]
        cout, cin, kh, kw = w.shape
        wf = torch.zeros(cout, kh, kw, cin_p, device=dev)
        wf[..., :cin] = w.permute(0, 2, 3, 1)
        pf = self._dev(wf.reshape(cout, kh * kw * cin_p), self.dtype)
        pa = 0
        if need_adj:
            wa = torch.zeros(cin_p, kh, kw, cout_p, device=dev)
            wa[:cin, :, :, :cout] = w.permute(1, 2, 3, 0)
            pa = self._dev(wa.reshape(cin_p, kh * kw * cout_p), self.dtype)
        bs = [self.p.get(n + ".bias") for n in names]
        b = None
        if any(x is not None for x in bs):
            b = torch.cat([x.detach().to(device=dev, dtype=torch.float32) if x is not None else torch.zeros(self.p[n + ".weight"].shape[0], device=dev)
                           for x, n in zip(bs, names)])
            if perm is not None:
                b = b[perm]
        pb = self._dev(b.float(), torch.float32) if b is not None else 0
        self._wcache[key] = (pf, pa, pb)
        return pf, pa, pb

    # ------------------------------------------------------------- ops
    def _op(self, **kw) -> None:
        d = dict(kind=0, in0=-1, in1=-1, in2=-1, out=-1, res=-1, rowbias=-1, ip=[0] * 12, fp=[0.0] * 4, w=[0, 0, 0, 0])
        d.update(kw)
        self.ops.append(d)

    def conv(self, name: str, x: int, hw: Tuple[int, int], cout: int, ks: int = 3, stride: int = 1, pad: int = 1,
             upsample: bool = False, res: int = -1, rowbias: int = -1, need_adj: bool = True, kind: int = L.BUF_ACT,
             interleave: int = 0, rowbias_off: int = 0) -> int:
        """3x3 / strided / upsampling convolution, 1x1 convolution or Linear (ks=1)."""
        rows, cin_p, _ = self.buffers[x]
        h, w = hw
        if ks == 1:
            ho, wo, gather = h, w, L.GATHER_NONE
        elif upsample:
            ho, wo, gather = 2 * h, 2 * w, L.GATHER_UPCONV
        else:
            ho = (h + 2 * pad - ks) // stride + 1 if pad else (h + 1 - ks) // stride + 1   # pad=0: reference pads (0,1,0,1)
            wo = (w + 2 * pad - ks) // stride + 1 if pad else (w + 1 - ks) // stride + 1
            gather = L.GATHER_CONV
        cout_p = _r8(cout)
        out = self.buf(ho * wo if ks != 1 else rows, cout_p, kind, cout if cout != cout_p else 0)
        pf, pa, pb = self._conv_w(name, cin_p, cout_p, need_adj, interleave)
        self._op(kind=L.OP_CONV, in0=x, out=out, res=res, rowbias=rowbias,
                 ip=[h, w, cin_p, ho, wo, cout, ks, stride, pad, gather, rowbias_off, 0], w=[pf, pa, pb, 0])
        return out

    # ------------------------------------------------------------- projections of one x-independent input, fused across the whole net
    def shared_begin(self, src: int, kind: int) -> dict:
        """Every ResBlock projects the SAME SiLU(time embedding) and every cross-attention layer projects the SAME prompt context: instead of one
        small GEMM per layer (22 + 16 launches per SD-1.5 forward) their weights are concatenated along Cout into ONE product, issued where the
        input becomes available; the consumers read column windows of its output (rowbias column offset / attention k, v offsets).  The width is
        known only when the build stops (``upto``), so the op is created by shared_end and inserted at this position of the tape."""
        rows = self.buffers[src][0]
        return dict(src=src, kind=kind, pos=len(self.ops), names=[], width=0, out=self.buf(rows, 8, kind))

    def shared_add(self, h: dict, names, cout: int) -> int:
        """-> column offset of this layer's ``cout`` outputs inside the fused product"""
        assert cout % 8 == 0, cout
        off = h["width"]
        h["names"] += list(names) if isinstance(names, tuple) else [names]
        h["width"] += cout
        return off

    def shared_end(self, h: dict) -> None:
        if not h["names"]:                        # nothing uses it (prefix builds): leave a tiny unused buffer, no op
            return
        rows, cin_p, _ = self.buffers[h["src"]]
        self.buffers[h["out"]] = (rows, h["width"], h["kind"])
        pf, pa, pb = self._conv_w(tuple(h["names"]), cin_p, h["width"], False)
        d = dict(kind=L.OP_CONV, in0=h["src"], in1=-1, in2=-1, out=h["out"], res=-1, rowbias=-1,
                 ip=[1, 1, cin_p, 1, 1, h["width"], 1, 1, 1, L.GATHER_NONE, 0, 0], fp=[0.0] * 4, w=[pf, pa, pb, 0])
        self.ops.insert(h["pos"], d)

    def groupnorm(self, name: str, x: int, groups: int, eps: float, silu: bool) -> int:
        rows, c, _ = self.buffers[x]
        out = self.buf(rows, c)
        self._op(kind=L.OP_GROUPNORM, in0=x, out=out, ip=[groups, int(silu)] + [0] * 10, fp=[eps, 0, 0, 0],
                 w=[self._dev(self.p[name + ".weight"], torch.float32), self._dev(self.p[name + ".bias"], torch.float32), 0, 0])
        return out

    def layernorm(self, name: str, x: int, eps: float = 1e-5) -> int:
        rows, c, _ = self.buffers[x]
        out = self.buf(rows, c)
        self._op(kind=L.OP_LAYERNORM, in0=x, out=out, fp=[eps, 0, 0, 0],
                 w=[self._dev(self.p[name + ".weight"], torch.float32), self._dev(self.p[name + ".bias"], torch.float32), 0, 0])
        return out

    def attention(self, q: int, k: int, v: int, heads: int, c: int = 0, offsets=(0, 0, 0), causal: bool = False) -> int:
        """q / k / v may be column windows (``offsets``) of wider buffers, e.g. one fused [rows][3C] projection."""
        rows, cq, _ = self.buffers[q]
        c = c or cq
        out = self.buf(rows, c)
        self._op(kind=L.OP_ATTENTION, in0=q, in1=k, in2=v, out=out, ip=[heads, offsets[0], offsets[1], offsets[2], int(causal)] + [0] * 7)
        return out

    def geglu(self, x: int, interleave: int = 0) -> int:
        rows, c, _ = self.buffers[x]
        out = self.buf(rows, c // 2)
        self._op(kind=L.OP_GEGLU, in0=x, out=out, ip=[c // 2, interleave] + [0] * 10)
        return out

    def silu(self, x: int) -> int:
        rows, c, kind = self.buffers[x]
        out = self.buf(rows, c, kind)
        self._op(kind=L.OP_SILU, in0=x, out=out)
        return out

    def quick_gelu(self, x: int) -> int:
        """x * sigmoid(1.702 x) (CLIP MLP activation); primal only."""
        rows, c, kind = self.buffers[x]
        out = self.buf(rows, c, kind)
        self._op(kind=L.OP_SILU, in0=x, out=out, ip=[1] + [0] * 11)
        return out

    def gelu(self, x: int) -> int:
        """exact (erf) GELU (OpenCLIP-H MLP activation); primal only."""
        rows, c, kind = self.buffers[x]
        out = self.buf(rows, c, kind)
        self._op(kind=L.OP_SILU, in0=x, out=out, ip=[2] + [0] * 11)
        return out

    def concat(self, a: int, b: int) -> int:
        rows, ca, _ = self.buffers[a]
        _, cb, _ = self.buffers[b]
        out = self.buf(rows, ca + cb)
        self._op(kind=L.OP_CONCAT, in0=a, in1=b, out=out)
        return out

    def tap(self, key, buf: int, c: int, h: int, w: int) -> None:
        self.taps[key] = buf
        self.tap_shape[buf] = (c, h, w)


# =================================================================== DDPM (pixel space, unconditional)
def build_ddpm(cfg, params, dtype, device, upto: Optional[Tuple[str, int]] = None) -> Tape:
    t = _build_ddpm(cfg, params, dtype, device, upto)
    for h in t._deferred:
        t.shared_end(h)
    return t


def _build_ddpm(cfg, params, dtype, device, upto: Optional[Tuple[str, int]] = None) -> Tape:
    """cfg: any object with the fields of oracle.unet_ddpm.DDPMConfig (ch, ch_mult, num_res_blocks,
    attn_resolutions, in_channels, out_ch, resolution, groups, gn_eps).  ``upto=(op, idx)`` stops
    building after that tap (weights beyond it are not uploaded)."""
    t = Tape(params, dtype, device)
    ch, res = cfg.ch, cfg.resolution
    G, eps = cfg.groups, cfg.gn_eps
    t.temb_in = t.buf(1, _r8(ch), L.BUF_SHARED)
    e0 = t.conv("temb.dense.0", t.temb_in, (1, 1), cfg.temb_ch, ks=1, need_adj=False, kind=L.BUF_SHARED)
    e1 = t.conv("temb.dense.1", t.silu(e0), (1, 1), cfg.temb_ch, ks=1, need_adj=False, kind=L.BUF_SHARED)
    st = t.silu(e1)
    tproj = t.shared_begin(st, L.BUF_SHARED)       # every ResBlock's temb_proj in one product
    t._deferred = [tproj]
    t.x = t.buf(res * res, _r8(cfg.in_channels))

    def resblock(pre, x, cin, cout, r):
        n1 = t.groupnorm(pre + ".norm1", x, G, eps, True)
        c1 = t.conv(pre + ".conv1", n1, (r, r), cout, rowbias=tproj["out"], rowbias_off=t.shared_add(tproj, pre + ".temb_proj", cout))
        n2 = t.groupnorm(pre + ".norm2", c1, G, eps, True)
        sc = t.conv(pre + ".nin_shortcut", x, (r, r), cout, ks=1) if cin != cout else x
        return t.conv(pre + ".conv2", n2, (r, r), cout, res=sc)

    def attn(pre, x, c, r):
        n = t.groupnorm(pre + ".norm", x, G, eps, False)
        qkv = t.conv((pre + ".q", pre + ".k", pre + ".v"), n, (r, r), 3 * c, ks=1)      # fused projection: one GEMM, N = 3C
        a = t.attention(qkv, qkv, qkv, 1, c, (0, c, 2 * c))
        return t.conv(pre + ".proj_out", a, (r, r), c, ks=1, res=x)

    nres = len(cfg.ch_mult)
    in_mult = (1,) + tuple(cfg.ch_mult)
    h = t.conv("conv_in", t.x, (res, res), ch)
    hs = [(h, ch)]
    r = res
    bi = ch
    for lvl in range(nres):
        bo = ch * cfg.ch_mult[lvl]
        bi = ch * in_mult[lvl]
        for blk in range(cfg.num_res_blocks):
            h = resblock(f"down.{lvl}.block.{blk}", hs[-1][0], bi, bo, r)
            bi = bo
            if r in cfg.attn_resolutions:
                h = attn(f"down.{lvl}.attn.{blk}", h, bi, r)
            hs.append((h, bi))
        if lvl != nres - 1:
            h = t.conv(f"down.{lvl}.downsample.conv", hs[-1][0], (r, r), bi, stride=2, pad=0)
            r //= 2
            hs.append((h, bi))
        t.tap(("down", lvl), hs[-1][0], bi, r, r)
        if upto == ("down", lvl):
            return t
    h = resblock("mid.block_1", hs[-1][0], bi, bi, r)
    h = attn("mid.attn_1", h, bi, r)
    h = resblock("mid.block_2", h, bi, bi, r)
    t.tap(("mid", 0), h, bi, r, r)
    if upto == ("mid", 0):
        return t
    for lvl in reversed(range(nres)):
        bo = ch * cfg.ch_mult[lvl]
        for blk in range(cfg.num_res_blocks + 1):
            sk, skc = hs.pop()
            h = resblock(f"up.{lvl}.block.{blk}", t.concat(h, sk), bi + skc, bo, r)
            bi = bo
            if r in cfg.attn_resolutions:
                h = attn(f"up.{lvl}.attn.{blk}", h, bi, r)
        if lvl != 0:
            h = t.conv(f"up.{lvl}.upsample.conv", h, (r, r), bi, upsample=True)
            r *= 2
        t.tap(("up", lvl), h, bi, r, r)
        if upto == ("up", lvl):
            return t
    n = t.groupnorm("norm_out", h, G, eps, True)
    o = t.conv("conv_out", n, (r, r), cfg.out_ch)
    t.tap("eps", o, cfg.out_ch, r, r)
    return t


# =================================================================== Stable Diffusion (latent space, text conditioned)
def build_sd(cfg, params, dtype, device, upto: Optional[Tuple[str, int]] = None) -> Tape:
    t = _build_sd(cfg, params, dtype, device, upto)
    for h in t._deferred:
        t.shared_end(h)
    return t


def _build_sd(cfg, params, dtype, device, upto: Optional[Tuple[str, int]] = None) -> Tape:
    """cfg: fields of oracle.unet_sd.SDConfig.  ('down', i) taps follow the intended semantics of the
    reference (output of the block after its downsampler; utils.py:489-490)."""
    t = Tape(params, dtype, device)
    boc = cfg.block_out_channels
    G = cfg.groups
    s = cfg.sample_size
    t.temb_in = t.buf(1, boc[0], L.BUF_SHARED)
    e0 = t.conv("time_embedding.linear_1", t.temb_in, (1, 1), cfg.temb_ch, ks=1, need_adj=False, kind=L.BUF_SHARED)
    e1 = t.conv("time_embedding.linear_2", t.silu(e0), (1, 1), cfg.temb_ch, ks=1, need_adj=False, kind=L.BUF_SHARED)
    st = t.silu(e1)
    tproj = t.shared_begin(st, L.BUF_SHARED)       # every ResBlock's time_emb_proj in one product
    t.ctx = t.buf(cfg.ctx_len, _r8(cfg.cross_dim), valid=cfg.cross_dim if cfg.cross_dim % 8 else 0)
    kvall = t.shared_begin(t.ctx, L.BUF_ACT)       # every cross-attention layer's to_k / to_v of the prompt context in one product
    t._deferred = [kvall, tproj]                   # (later tape position first: inserting does not shift the earlier one)
    t.x = t.buf(s * s, _r8(cfg.in_channels))

    def resnet(pre, x, cin, cout, r):
        n1 = t.groupnorm(pre + ".norm1", x, G, 1e-5, True)
        c1 = t.conv(pre + ".conv1", n1, (r, r), cout, rowbias=tproj["out"], rowbias_off=t.shared_add(tproj, pre + ".time_emb_proj", cout))
        n2 = t.groupnorm(pre + ".norm2", c1, G, 1e-5, True)
        sc = t.conv(pre + ".conv_shortcut", x, (r, r), cout, ks=1) if cin != cout else x
        return t.conv(pre + ".conv2", n2, (r, r), cout, res=sc)

    def transformer(pre, x, c, r, heads):
        n = t.groupnorm(pre + ".norm", x, G, 1e-6, False)
        h = t.conv(pre + ".proj_in", n, (r, r), c, ks=1)
        tb = pre + ".transformer_blocks.0"
        z = t.layernorm(tb + ".norm1", h)
        qkv = t.conv((tb + ".attn1.to_q", tb + ".attn1.to_k", tb + ".attn1.to_v"), z, (r, r), 3 * c, ks=1)   # fused q/k/v
        h = t.conv(tb + ".attn1.to_out.0", t.attention(qkv, qkv, qkv, heads, c, (0, c, 2 * c)), (r, r), c, ks=1, res=h)
        z = t.layernorm(tb + ".norm2", h)
        q = t.conv(tb + ".attn2.to_q", z, (r, r), c, ks=1)
        ko = t.shared_add(kvall, (tb + ".attn2.to_k", tb + ".attn2.to_v"), 2 * c)                             # k | v of the context: a window of the net-wide product
        h = t.conv(tb + ".attn2.to_out.0", t.attention(q, kvall["out"], kvall["out"], heads, c, (0, ko, ko + c)), (r, r), c, ks=1, res=h)
        z = t.layernorm(tb + ".norm3", h)
        il = 64 if (4 * c) % 64 == 0 else 0                                     # a / g interleaved in 64-column blocks: GEGLU fuses into GEMM epilogues
        f = t.geglu(t.conv(tb + ".ff.net.0.proj", z, (r, r), 8 * c, ks=1, interleave=il), il)
        h = t.conv(tb + ".ff.net.2", f, (r, r), c, ks=1, res=h)
        return t.conv(pre + ".proj_out", h, (r, r), c, ks=1, res=x)

    nb = len(boc)
    h = t.conv("conv_in", t.x, (s, s), boc[0])
    skips = [(h, boc[0])]
    r = s
    ch = boc[0]
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet(f"down_blocks.{i}.resnets.{j}", h, ch, boc[i], r)
            ch = boc[i]
            if cfg.down_attn[i]:
                h = transformer(f"down_blocks.{i}.attentions.{j}", h, ch, r, cfg.heads[i])
            skips.append((h, ch))
        if i != nb - 1:
            h = t.conv(f"down_blocks.{i}.downsamplers.0.conv", h, (r, r), ch, stride=2, pad=1)
            r //= 2
            skips.append((h, ch))
        t.tap(("down", i), h, ch, r, r)
        if upto == ("down", i):
            return t
    h = resnet("mid_block.resnets.0", h, ch, ch, r)
    h = transformer("mid_block.attentions.0", h, ch, r, cfg.heads[-1])
    h = resnet("mid_block.resnets.1", h, ch, ch, r)
    t.tap(("mid", 0), h, ch, r, r)
    if upto == ("mid", 0):
        return t
    rheads = tuple(reversed(cfg.heads))
    rev = tuple(reversed(boc))
    for i in range(nb):
        out = rev[i]
        for j in range(cfg.layers_per_block + 1):
            sk, skc = skips.pop()
            h = resnet(f"up_blocks.{i}.resnets.{j}", t.concat(h, sk), ch + skc, out, r)
            ch = out
            if cfg.up_attn[i]:
                h = transformer(f"up_blocks.{i}.attentions.{j}", h, ch, r, rheads[i])
        if i != nb - 1:
            h = t.conv(f"up_blocks.{i}.upsamplers.0.conv", h, (r, r), ch, upsample=True)
            r *= 2
        t.tap(("up", i), h, ch, r, r)
        if upto == ("up", i):
            return t
    n = t.groupnorm("conv_norm_out", h, G, 1e-5, True)
    o = t.conv("conv_out", n, (r, r), cfg.out_channels)
    t.tap("eps", o, cfg.out_channels, r, r)
    return t


# =================================================================== SD image autoencoder (primal only)
def _vae_blocks(t: Tape, cfg):
    G, eps = cfg.groups, cfg.gn_eps

    def resnet(pre, x, cin, cout, r):
        n1 = t.groupnorm(pre + ".norm1", x, G, eps, True)
        c1 = t.conv(pre + ".conv1", n1, (r, r), cout, need_adj=False)
        n2 = t.groupnorm(pre + ".norm2", c1, G, eps, True)
        sc = t.conv(pre + ".conv_shortcut", x, (r, r), cout, ks=1, need_adj=False) if cin != cout else x
        return t.conv(pre + ".conv2", n2, (r, r), cout, res=sc, need_adj=False)

    def mid(pre, x, c, r):
        x = resnet(pre + ".resnets.0", x, c, c, r)
        a = pre + ".attentions.0"
        n = t.groupnorm(a + ".group_norm", x, G, eps, False)
        qkv = t.conv((a + ".query", a + ".key", a + ".value"), n, (r, r), 3 * c, ks=1, need_adj=False)
        x = t.conv(a + ".proj_attn", t.attention(qkv, qkv, qkv, 1, c, (0, c, 2 * c)), (r, r), c, ks=1, res=x, need_adj=False)
        return resnet(pre + ".resnets.1", x, c, c, r)

    return resnet, mid


def build_vae_encoder(cfg, params, dtype, device) -> Tape:
    """image [S,S,3] -> moments [S/8,S/8,2*latent] (mean | logvar); diffusers AutoencoderKL.encode + quant_conv
    (the object behind reference src/modules/edit.py:144-146).  cfg: configs.VAEConfig."""
    t = Tape(params, dtype, device)
    resnet, mid = _vae_blocks(t, cfg)
    boc = cfg.block_out_channels
    r = cfg.sample_size
    t.temb_in = t.buf(1, 8, L.BUF_SHARED)          # the autoencoder has no time embedding; the engine wants the slot
    t.x = t.buf(r * r, _r8(cfg.in_channels))
    h = t.conv("encoder.conv_in", t.x, (r, r), boc[0], need_adj=False)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            h = resnet(f"encoder.down_blocks.{i}.resnets.{j}", h, ch, co, r)
            ch = co
        if i != len(boc) - 1:
            h = t.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", h, (r, r), ch, stride=2, pad=0, need_adj=False)   # F.pad (0,1,0,1)
            r //= 2
    h = mid("encoder.mid_block", h, ch, r)
    n = t.groupnorm("encoder.conv_norm_out", h, cfg.groups, cfg.gn_eps, True)
    o = t.conv("encoder.conv_out", n, (r, r), 2 * cfg.latent_channels, need_adj=False)
    m = t.conv("quant_conv", o, (r, r), 2 * cfg.latent_channels, ks=1, need_adj=False)
    t.tap("moments", m, 2 * cfg.latent_channels, r, r)
    return t


def build_vae_decoder(cfg, params, dtype, device) -> Tape:
    """latent [S/8,S/8,4] -> image [S,S,3]; post_quant_conv + diffusers AutoencoderKL.decode (edit.py:476-478)."""
    t = Tape(params, dtype, device)
    resnet, mid = _vae_blocks(t, cfg)
    rev = tuple(reversed(cfg.block_out_channels))
    r = cfg.latent_size
    t.temb_in = t.buf(1, 8, L.BUF_SHARED)
    t.x = t.buf(r * r, _r8(cfg.latent_channels))
    h = t.conv("post_quant_conv", t.x, (r, r), cfg.latent_channels, ks=1, need_adj=False)
    h = t.conv("decoder.conv_in", h, (r, r), rev[0], need_adj=False)
    h = mid("decoder.mid_block", h, rev[0], r)
    ch = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(f"decoder.up_blocks.{i}.resnets.{j}", h, ch, co, r)
            ch = co
        if i != len(rev) - 1:
            h = t.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", h, (r, r), ch, upsample=True, need_adj=False)
            r *= 2
    n = t.groupnorm("decoder.conv_norm_out", h, cfg.groups, cfg.gn_eps, True)
    o = t.conv("decoder.conv_out", n, (r, r), cfg.out_channels, need_adj=False)
    t.tap("image", o, cfg.out_channels, r, r)
    return t


# =================================================================== SD prompt encoder (primal only)
def build_clip_text(cfg, params, dtype, device) -> Tape:
    """[tokens][hidden] token+position embeddings (dpb_embed_tokens) -> last_hidden_state [tokens][hidden]:
    transformers CLIPTextModel (pre-LN transformer, causal self-attention, quick-GELU MLP, final LayerNorm), the
    text_encoder behind ``pipe._encode_prompt`` (reference src/modules/edit.py:505-522).  cfg: configs.CLIPTextConfig."""
    t = Tape(params, dtype, device)
    n, h = cfg.max_position, cfg.hidden
    t.temb_in = t.buf(1, 8, L.BUF_SHARED)          # no time embedding; the engine wants the slot
    t.x = t.buf(n, h)
    x = t.x
    for i in range(cfg.layers):
        pre = f"text_model.encoder.layers.{i}"
        z = t.layernorm(pre + ".layer_norm1", x, cfg.eps)
        qkv = t.conv((pre + ".self_attn.q_proj", pre + ".self_attn.k_proj", pre + ".self_attn.v_proj"), z, (n, 1), 3 * h, ks=1, need_adj=False)
        a = t.attention(qkv, qkv, qkv, cfg.heads, h, (0, h, 2 * h), causal=True)
        x = t.conv(pre + ".self_attn.out_proj", a, (n, 1), h, ks=1, res=x, need_adj=False)
        z = t.layernorm(pre + ".layer_norm2", x, cfg.eps)
        act = t.quick_gelu if cfg.act == "quick_gelu" else t.gelu
        f = act(t.conv(pre + ".mlp.fc1", z, (n, 1), cfg.intermediate, ks=1, need_adj=False))
        x = t.conv(pre + ".mlp.fc2", f, (n, 1), h, ks=1, res=x, need_adj=False)
    o = t.layernorm("text_model.final_layer_norm", x, cfg.eps)
    t.tap("last_hidden_state", o, h, n, 1)
    return t
