"""State-dict plumbing at the drop-in boundary: key renaming and shape validation.  Pure host-side renames -- no arithmetic.

The reference's live unconditional path loads ``google/ddpm-ema-celebahq-256`` through diffusers
(``DDIMPipeline.from_pretrained``, reference src/utils/utils.py:101-104), whose ``UNet2DModel.state_dict()`` uses diffusers'
block naming; the tape builder (tape.build_ddpm) uses the naming of the vendored definition of the same network
(reference src/models/ddpm/diffusion.py:22-129: ``down.{i}.block.{j}``, ``mid.block_1``, ``up.{i}.attn.{j}``, ...).
``ddpm_hf_to_vendored_names`` is the rename between the two -- the inverse of diffusers'
``convert_ddpm_original_checkpoint_to_diffusers.py`` -- so ``bind(pipe.unet, "ddpm", ...)`` works on the real checkpoint.
Stable-Diffusion weights need no rename: tape.build_sd consumes diffusers' ``UNet2DConditionModel`` keys directly.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping, Tuple

import torch

_RES = {"norm1": "norm1", "conv1": "conv1", "time_emb_proj": "temb_proj", "norm2": "norm2", "conv2": "conv2", "conv_shortcut": "nin_shortcut"}
# diffusers 0.11 AttentionBlock names, and the names later releases renamed them to
_ATT = {"group_norm": "norm", "query": "q", "key": "k", "value": "v", "proj_attn": "proj_out",
        "to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out"}


def _ddpm_hf_key(key: str, n_levels: int) -> str:
    stem, leaf = key.rsplit(".", 1)                     # leaf: weight | bias
    top = {"time_embedding.linear_1": "temb.dense.0", "time_embedding.linear_2": "temb.dense.1", "conv_in": "conv_in",
           "conv_norm_out": "norm_out", "conv_out": "conv_out"}
    if stem in top:
        return f"{top[stem]}.{leaf}"
    m = re.fullmatch(r"(down|up)_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.+)", stem)
    if m:
        side, i, kind, j, sub = m.group(1), int(m.group(2)), m.group(3), int(m.group(4)), m.group(5)
        lvl = i if side == "down" else n_levels - 1 - i  # diffusers counts up blocks from the bottleneck, the vendored net by level
        if kind == "resnets" and sub in _RES:
            return f"{side}.{lvl}.block.{j}.{_RES[sub]}.{leaf}"
        if kind == "attentions" and sub in _ATT:
            return f"{side}.{lvl}.attn.{j}.{_ATT[sub]}.{leaf}"
    m = re.fullmatch(r"down_blocks\.(\d+)\.downsamplers\.0\.conv", stem)
    if m:
        return f"down.{int(m.group(1))}.downsample.conv.{leaf}"
    m = re.fullmatch(r"up_blocks\.(\d+)\.upsamplers\.0\.conv", stem)
    if m:
        return f"up.{n_levels - 1 - int(m.group(1))}.upsample.conv.{leaf}"
    m = re.fullmatch(r"mid_block\.resnets\.(\d+)\.(.+)", stem)
    if m and m.group(2) in _RES:
        return f"mid.block_{int(m.group(1)) + 1}.{_RES[m.group(2)]}.{leaf}"
    m = re.fullmatch(r"mid_block\.attentions\.0\.(.+)", stem)
    if m and m.group(1) in _ATT:
        return f"mid.attn_1.{_ATT[m.group(1)]}.{leaf}"
    raise KeyError(f"unrecognised UNet2DModel state-dict key {key!r}")


def ddpm_hf_to_vendored_names(state_dict: Mapping[str, torch.Tensor], cfg) -> Dict[str, torch.Tensor]:
    """diffusers ``UNet2DModel.state_dict()`` (google/ddpm-ema-celebahq-256 family) -> the parameter names tape.build_ddpm reads.
    ``cfg``: configs.DDPMConfig (only ``len(cfg.ch_mult)`` is used).  Attention projections stay 2-D ``[C, C]`` Linear
    weights (the vendored net stores them as 1x1 convolutions ``[C, C, 1, 1]``; the tape accepts both).  A state dict that
    already uses the vendored names is returned unchanged."""
    if "temb.dense.0.weight" in state_dict:
        return dict(state_dict)
    n_levels = len(cfg.ch_mult)
    out: Dict[str, torch.Tensor] = {}
    for k, v in state_dict.items():
        nk = _ddpm_hf_key(k, n_levels)
        if nk in out:
            raise KeyError(f"two source keys map to {nk!r}")
        out[nk] = v
    return out


def ddpm_vendored_to_hf_names(params: Mapping[str, torch.Tensor], cfg) -> Dict[str, torch.Tensor]:
    """Inverse rename (diffusers 0.11 attention names, 2-D attention projections): used to build HF-named synthetic state
    dicts for the tests, and to export weights for a diffusers pipeline."""
    n_levels = len(cfg.ch_mult)
    inv_res = {v: k for k, v in _RES.items()}
    inv_att = {"norm": "group_norm", "q": "query", "k": "key", "v": "value", "proj_out": "proj_attn"}
    top = {"temb.dense.0": "time_embedding.linear_1", "temb.dense.1": "time_embedding.linear_2", "conv_in": "conv_in",
           "norm_out": "conv_norm_out", "conv_out": "conv_out"}
    out: Dict[str, torch.Tensor] = {}
    for key, v in params.items():
        stem, leaf = key.rsplit(".", 1)
        if stem in top:
            nk = top[stem]
        elif (m := re.fullmatch(r"(down|up)\.(\d+)\.block\.(\d+)\.(\w+)", stem)):
            i = int(m.group(2)) if m.group(1) == "down" else n_levels - 1 - int(m.group(2))
            nk = f"{m.group(1)}_blocks.{i}.resnets.{m.group(3)}.{inv_res[m.group(4)]}"
        elif (m := re.fullmatch(r"(down|up)\.(\d+)\.attn\.(\d+)\.(\w+)", stem)):
            i = int(m.group(2)) if m.group(1) == "down" else n_levels - 1 - int(m.group(2))
            nk = f"{m.group(1)}_blocks.{i}.attentions.{m.group(3)}.{inv_att[m.group(4)]}"
            if v.dim() == 4:
                v = v[:, :, 0, 0]
        elif (m := re.fullmatch(r"down\.(\d+)\.downsample\.conv", stem)):
            nk = f"down_blocks.{m.group(1)}.downsamplers.0.conv"
        elif (m := re.fullmatch(r"up\.(\d+)\.upsample\.conv", stem)):
            nk = f"up_blocks.{n_levels - 1 - int(m.group(1))}.upsamplers.0.conv"
        elif (m := re.fullmatch(r"mid\.block_(\d)\.(\w+)", stem)):
            nk = f"mid_block.resnets.{int(m.group(1)) - 1}.{inv_res[m.group(2)]}"
        elif (m := re.fullmatch(r"mid\.attn_1\.(\w+)", stem)):
            nk = f"mid_block.attentions.0.{inv_att[m.group(1)]}"
            if v.dim() == 4:
                v = v[:, :, 0, 0]
        else:
            raise KeyError(f"unrecognised vendored DDPM parameter {key!r}")
        out[f"{nk}.{leaf}"] = v
    return out


def check_shapes(params: Mapping[str, torch.Tensor], shapes: Mapping[str, Tuple[int, ...]], what: str, prefixes=None) -> None:
    """Raise a readable error when a state dict does not fit the architecture the engine is about to be built for
    (e.g. SD-2.1 weights under an SD-1.5 config) instead of failing inside the weight repack."""
    bad = []
    for name, shp in shapes.items():
        if prefixes is not None and not name.startswith(tuple(prefixes)):
            continue
        if name not in params:
            bad.append(f"missing {name} {tuple(shp)}")
            continue
        got = tuple(params[name].shape)
        if got != tuple(shp) and not (len(got) == 2 and len(shp) == 4 and got + (1, 1) == tuple(shp)) \
                and not (len(got) == 4 and len(shp) == 2 and got == tuple(shp) + (1, 1)):
            bad.append(f"{name}: checkpoint {got} != architecture {tuple(shp)}")
        if len(bad) >= 8:
            break
    if bad:
        raise ValueError(f"state dict does not match the {what} architecture:\n  " + "\n  ".join(bad))
