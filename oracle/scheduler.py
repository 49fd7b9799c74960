"""ORACLE (test infrastructure, never imported by the product package).

CPU restatement of the reference's DDIM scheduler.

Follows /root/reference/src/utils/utils.py
  * YHCustomScheduler.set_timesteps :1182-1195  (= patched set_timesteps :273-286)
  * YHCustomScheduler.step          :1197-1241  (= patched step :288-315, eta==0 and eta>0/no-logvar)
  * linear beta schedule, fp64 -> dtype, alphas_cumprod :1243-1264
  * extract (timestep truncated with .long()) :1302-1317
"""
from __future__ import annotations

import torch


def linear_alphas_cumprod(dtype=torch.float32, beta_start=1e-4, beta_end=0.02, n=1000):
    betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float64)
    return torch.cumprod(1.0 - betas, dim=0).to(dtype), betas.to(dtype)


def scaled_linear_alphas_cumprod(dtype=torch.float32, beta_start=0.00085, beta_end=0.012, n=1000):
    """Stable Diffusion's scheduler config (diffusers DDIMScheduler 'scaled_linear',
    third-party; the reference reads pipe.scheduler.alphas_cumprod, utils.py:265)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).to(dtype), betas.to(dtype)


def timesteps(num_inferences: int, t_max: float = 999, is_inversion: bool = False):
    """-> (timesteps, timesteps_next)"""
    seq = torch.linspace(0, 1, num_inferences) * t_max
    if is_inversion:
        seq = seq + 1e-6
        seq_prev = torch.cat([torch.tensor([-1]), seq[:-1]], dim=0)
        return seq_prev[1:], seq[1:]
    seq_prev = torch.cat([torch.tensor([-1]), seq[:-1]], dim=0)
    return reversed(seq[1:]), reversed(seq_prev[1:])


def extract(a: torch.Tensor, t: torch.Tensor, x_shape):
    t = t.repeat(x_shape[0])
    out = torch.gather(a, 0, t.long())
    return out.reshape((x_shape[0],) + (1,) * (len(x_shape) - 1))


def step(alphas_cumprod, ts, ts_next, et, t, xt, eta: float = 0.0, noise=None):
    """-> (xt_next, P_xt).  ``noise`` replaces randn_like for eta>0 (testability)."""
    t_idx = ts.tolist().index(t)
    t_next = ts_next[t_idx]
    at = extract(alphas_cumprod, t, xt.shape)
    at_next = extract(alphas_cumprod, t_next, xt.shape)
    p_xt = (xt - et * (1 - at).sqrt()) / at.sqrt()
    if eta == 0:
        return at_next.sqrt() * p_xt + (1 - at_next).sqrt() * et, p_xt
    sigma_t = ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
    d_xt = (1 - at_next - eta * sigma_t ** 2).sqrt() * et
    noise = torch.randn_like(xt) if noise is None else noise
    return at_next.sqrt() * p_xt + d_xt + eta * sigma_t * noise, p_xt
