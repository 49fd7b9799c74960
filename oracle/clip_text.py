"""ORACLE (test infrastructure, never imported by the product package).

PARITY UNPINNED.  Functional plain-PyTorch restatement of the CLIP ViT-L/14 text model
(``transformers`` ``CLIPTextModel``; third-party, listed unpinned in /root/reference/requirements.txt, NOT vendored
under /root/reference; no weights or tokenizer vocabulary are reachable offline) that
``pipe._encode_prompt`` runs at reference src/modules/edit.py:505-522: token + position embeddings, 12 pre-LN transformer
layers (causal 12-head self-attention with q scaled by d^-1/2, quick-GELU MLP), final LayerNorm; the reference consumes
``last_hidden_state`` of the 77 padded tokens as ``encoder_hidden_states``.  Parameter names equal transformers'
``state_dict`` keys.  Anchor: parameter count 123,060,480 (tests/test_oracle.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from diffusion_pullback_amd.configs import SD15_CLIP, SD21_CLIP, CLIPTextConfig, Params  # noqa: F401
from diffusion_pullback_amd.configs import clip_init_params as init_params  # noqa: F401
from diffusion_pullback_amd.configs import clip_param_shapes as param_shapes  # noqa: F401


def embed(p: Params, cfg: CLIPTextConfig, ids: torch.Tensor) -> torch.Tensor:
    """ids [B, L] -> [B, L, hidden]"""
    pos = p["text_model.embeddings.position_embedding.weight"][: ids.shape[1]]
    return p["text_model.embeddings.token_embedding.weight"][ids] + pos[None]


def encode(p: Params, cfg: CLIPTextConfig, ids: torch.Tensor) -> torch.Tensor:
    """ids [B, L] int64 -> last_hidden_state [B, L, hidden]"""
    x = embed(p, cfg, ids).float()
    b, n, h = x.shape
    d = h // cfg.heads
    mask = torch.full((n, n), float("-inf")).triu(1)
    for i in range(cfg.layers):
        pre = f"text_model.encoder.layers.{i}"
        z = F.layer_norm(x, (h,), p[pre + ".layer_norm1.weight"], p[pre + ".layer_norm1.bias"], cfg.eps)
        q, k, v = (F.linear(z, p[f"{pre}.self_attn.{w}.weight"], p[f"{pre}.self_attn.{w}.bias"]).reshape(b, n, cfg.heads, d).transpose(1, 2)
                   for w in ("q_proj", "k_proj", "v_proj"))
        a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(b, n, h)
        x = x + F.linear(a, p[pre + ".self_attn.out_proj.weight"], p[pre + ".self_attn.out_proj.bias"])
        z = F.layer_norm(x, (h,), p[pre + ".layer_norm2.weight"], p[pre + ".layer_norm2.bias"], cfg.eps)
        f = F.linear(z, p[pre + ".mlp.fc1.weight"], p[pre + ".mlp.fc1.bias"])
        f = f * torch.sigmoid(1.702 * f) if cfg.act == "quick_gelu" else F.gelu(f)
        x = x + F.linear(f, p[pre + ".mlp.fc2.weight"], p[pre + ".mlp.fc2.bias"])
    return F.layer_norm(x, (h,), p["text_model.final_layer_norm.weight"], p["text_model.final_layer_norm.bias"], cfg.eps)
