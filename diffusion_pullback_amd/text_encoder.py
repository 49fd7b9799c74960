"""On-device prompt encoder for the SD edit loop: the CLIP text model behind ``pipe._encode_prompt``
(reference src/modules/edit.py:505-522), i.e. ``text_encoder(input_ids)[0]`` of the 77 padded tokens.

Token + position embedding lookup is ``dpb_embed_tokens``; the 12 transformer layers run on the tape engine (LayerNorm,
fused q/k/v projection, causal softmax attention, quick-GELU MLP); primal passes only, no CPU fallback.
The BPE tokenizer is host-side text processing and needs CLIP's vocabulary files, which are not reachable offline:
``encode_prompt`` takes it by injection (``tokenizer(str) -> list[int]`` of max_position ids, e.g. transformers'
``CLIPTokenizer(...)(prompt, padding="max_length", max_length=77, truncation=True).input_ids``).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence

import torch

from . import configs as cf
from . import lib as L
from .engine import Engine
from .tape import build_clip_text


class ClipTextEncoder:
    def __init__(self, cfg: cf.CLIPTextConfig = cf.SD15_CLIP, params: Optional[cf.Params] = None, dtype=torch.bfloat16,
                 device="cuda:0", max_batch: int = 2, tokenizer: Optional[Callable[[str], Sequence[int]]] = None):
        self.cfg, self.dtype, self.device, self.max_batch = cfg, dtype, torch.device(device), max_batch
        self.tokenizer = tokenizer
        if params is None:
            params = cf.clip_init_params(cfg, seed=0)
        self.lib = L.load()
        tape = build_clip_text(cfg, params, dtype, device)
        self.eng = Engine(tape, 8, False, True, cfg.hidden, max_batch=max_batch, max_tangents=1)
        self._tok = params["text_model.embeddings.token_embedding.weight"].to(device=self.device, dtype=dtype).contiguous()
        self._pos = params["text_model.embeddings.position_embedding.weight"].to(device=self.device, dtype=dtype).contiguous()

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor) -> torch.Tensor:
        """input_ids [B, max_position] integer -> last_hidden_state [B, max_position, hidden] (fp32, on the device)"""
        n, h = self.cfg.max_position, self.cfg.hidden
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        if ids.dim() != 2 or ids.shape[1] != n:
            raise ValueError(f"input_ids must be [B, {n}] (pad / truncate to max_position as the reference's tokenizer call does)")
        outs = []
        for i in range(0, ids.shape[0], self.max_batch):
            chunk = ids[i:i + self.max_batch]
            b = chunk.shape[0]
            with torch.cuda.device(self.device):
                x = torch.empty(b, h, n, 1, dtype=torch.float32, device=self.device)      # [b][c][t]: the engine's NCHW boundary
                st = torch.cuda.current_stream(self.device).cuda_stream
                L.check(self.lib.dpb_embed_tokens(chunk.data_ptr(), self._tok.data_ptr(), self._pos.data_ptr(),
                                                  L.dtype_code(self.dtype), x.data_ptr(), b, n, h,
                                                  self.cfg.vocab_size, C.c_void_p(st)))
            y = self.eng.forward(x, 0.0, None, "last_hidden_state")                       # [b, h, n, 1]
            outs.append(y[:, :, :, 0].transpose(1, 2).contiguous())
        return torch.cat(outs, dim=0)

    def encode_prompt(self, prompt: str) -> torch.Tensor:
        """``pipe._encode_prompt(prompt, ...)`` without classifier-free guidance: [1, 77, 768]."""
        if self.tokenizer is None:
            raise RuntimeError("ClipTextEncoder.encode_prompt needs a tokenizer (CLIP's BPE vocabulary is not available offline): "
                               "pass tokenizer=callable(str) -> token ids")
        ids = torch.as_tensor(list(self.tokenizer(prompt)), dtype=torch.int32)[None]
        return self(ids)
