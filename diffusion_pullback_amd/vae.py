"""On-device image autoencoder for the SD edit loop: the ``pipe.vae`` the reference calls at
src/modules/edit.py:144-146 (``vae.encode(x0).latent_dist.sample() * 0.18215``) and :476-480
(``vae.decode(1 / 0.18215 * latents).sample`` -> ``(x/2+0.5).clamp(0,1)`` -> save_image).

Same tape executor and HIP kernels as the U-Net (implicit-GEMM convolutions, GroupNorm+SiLU, single-head attention
over the 64x64 mid-block positions, nearest-x2 upsampling fused into the conv gather); primal passes only.
``EditStableDiffusion(args, unet=..., vae=AutoencoderKL(...))`` plugs it in; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import configs as cf
from .engine import Engine
from .tape import build_vae_decoder, build_vae_encoder


class AutoencoderKL:
    def __init__(self, cfg: cf.VAEConfig = cf.SD15_VAE, params: Optional[cf.Params] = None, dtype=torch.bfloat16,
                 device="cuda:0", max_batch: int = 1, encoder: bool = True, decoder: bool = True):
        self.cfg, self.dtype, self.device, self.max_batch = cfg, dtype, torch.device(device), max_batch
        if params is None:
            params = cf.vae_init_params(cfg, seed=0)
        self.enc = self.dec = None
        if encoder:
            t = build_vae_encoder(cfg, {k: v for k, v in params.items() if k.startswith(("encoder.", "quant_conv"))}, dtype, device)
            self.enc = Engine(t, 8, False, True, cfg.in_channels, max_batch=max_batch, max_tangents=1)
        if decoder:
            t = build_vae_decoder(cfg, {k: v for k, v in params.items() if k.startswith(("decoder.", "post_quant_conv"))}, dtype, device)
            self.dec = Engine(t, 8, False, True, cfg.latent_channels, max_batch=max_batch, max_tangents=1)

    def _chunks(self, eng: Engine, x: torch.Tensor, tap: str) -> torch.Tensor:
        outs = [eng.forward(x[i:i + self.max_batch], 0.0, None, tap) for i in range(0, x.shape[0], self.max_batch)]
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """image [B,3,S,S] in [-1,1] -> [B, 2*latent, S/8, S/8] = (mean | logvar)"""
        if self.enc is None:
            raise RuntimeError("this AutoencoderKL was built without its encoder")
        return self._chunks(self.enc, x, "moments")

    @torch.no_grad()
    def encode(self, x: torch.Tensor, generator: Optional[torch.Generator] = None, sample_posterior: bool = True) -> torch.Tensor:
        """``latent_dist.sample()`` of the reference call: mean + exp(0.5 logvar) * N(0, I) (logvar clamped to [-30, 20])."""
        mean, logvar = self.encode_moments(x).chunk(2, dim=1)
        if not sample_posterior:
            return mean
        std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
        noise = torch.randn(mean.shape, generator=generator, dtype=mean.dtype, device=mean.device if generator is None else generator.device)
        return mean + std * noise.to(mean.device)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """latent [B,4,S/8,S/8] (already multiplied by 1/0.18215) -> image [B,3,S,S]"""
        if self.dec is None:
            raise RuntimeError("this AutoencoderKL was built without its decoder")
        return self._chunks(self.dec, z, "image")
