"""ORACLE (test infrastructure, never imported by the product package).

CPU restatement of the DDIM inversion / forward / x-space-guidance / edit orchestration loops of the reference's two
experiment drivers, as pure functions over an ``eps(x, t[, emb])`` callable and the oracle scheduler.  Pinned:
tests/test_oracle.py replays the fixtures tests/golden/edit_{uncond_small,sd_toy}.pt, which were recorded from the
reference's own methods (tests/golden/make_golden_edit.py), U-Net call by U-Net call.

Follows /root/reference/src/modules/edit.py
  * EditStableDiffusion.run_DDIMinversion :112-183, DDIMforwardsteps :385-482, x_space_guidance :484-502,
    run_edit_local_encoder_pullback_zt :185-307
  * EditUncondDiffusion.run_DDIMinversion :613-678, DDIMforwardsteps :1601-1714, x_space_guidance :1716-1734,
    run_edit_local_encoder_pullback_zt :680-779
Left out (file output only): PNG / .pt writing, the spectrum plot, the vT visualisation, CPU<->device buffer bouncing.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch

from . import scheduler as osch


class Sched:
    """Timestep tables + step of the patched scheduler (utils.py:273-315) on a given alphas_cumprod table."""

    def __init__(self, alphas_cumprod: torch.Tensor):
        self.ac = alphas_cumprod
        self.ts = self.tn = None

    def set_timesteps(self, n: int, is_inversion: bool = False):
        self.ts, self.tn = osch.timesteps(n, is_inversion=is_inversion)

    def step(self, et, t, xt, eta: float = 0.0):
        return osch.step(self.ac, self.ts, self.tn, et, t, xt, eta=eta)[0]


def ddim_inversion(eps: Callable, sch: Sched, z0: torch.Tensor, inv_steps: int) -> torch.Tensor:
    """edit.py:149-181 / :644-664: inv_steps-2 U-Net calls (the loop breaks at the last index before calling the U-Net)."""
    sch.set_timesteps(inv_steps, is_inversion=True)
    x = z0
    for i, t in enumerate(sch.ts):
        if i == len(sch.ts) - 1:
            break
        x = sch.step(eps(x, t), t, x)
    return x


def ddim_forward(eps: Callable, sch: Sched, x: torch.Tensor, for_steps: int, t_start_idx: int, t_end_idx: int, memory_bound: int,
                 uncond_order: bool):
    """edit.py:420-473 (SD: the end test comes after the start test) / :1637-1690 (uncond: the end test comes first).
    Returns ``(x, t, idx)`` when ``t_end_idx`` is reached, else the fully denoised batch."""
    sch.set_timesteps(for_steps)
    for i, t in enumerate(sch.ts):
        if uncond_order:
            if t_end_idx == i:
                return x, t, i
            if i < t_start_idx:
                continue
        else:
            if i < t_start_idx:
                continue
            elif t_start_idx == i:
                pass
            elif i == t_end_idx:
                return x, t, i
        if uncond_order:
            chunks = [x] if x.size(0) // memory_bound == 0 else list(x.chunk(x.size(0) // memory_bound))
        else:
            chunks = [x] if x.size(0) == 1 else list(x.chunk(x.size(0) // memory_bound))
        x = torch.cat([sch.step(eps(c, t), t, c) for c in chunks], dim=0)
    return x


def x_space_guidance(eps: Callable, sch: Sched, x: torch.Tensor, t_idx: int, vk: torch.Tensor, step: float, scale: float) -> torch.Tensor:
    """edit.py:484-502 / :1716-1734: one batch-2 U-Net call, x + scale * (eps(x + step*vk) - eps(x))."""
    t = sch.ts[t_idx]
    e_null, e_edit = eps(torch.cat([x, x + step * vk], dim=0), t).chunk(2)
    return x + scale * (e_edit - e_null)


def run_edit(eps_inv: Callable, eps_for: Callable, eps_edit: Callable, pullback: Optional[Callable], sch: Sched, z0: torch.Tensor, *,
             for_steps: int, inv_steps: int, edit_t: float, num_step: int, edit_step: float, scale: float, vis_num: int, vis_num_pc: int,
             memory_bound: int, uncond_order: bool, latent_scale: float = 1.0, basis: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """edit.py:185-307 / :680-779.  ``pullback(zt, t) -> (u, s, vT)``; ``basis=(u, vT)`` plays the .pt cache branch.
    Returns dict(zT, zt, t_idx, u, vT, results=[decoded batch per (pc, direction)])."""
    sch.set_timesteps(for_steps)
    edit_t_idx = int((sch.ts - edit_t * 1000).abs().argmin())
    zT = ddim_inversion(eps_inv, sch, z0, inv_steps)
    zt, t, t_idx = ddim_forward(eps_for, sch, zT, for_steps, 0, edit_t_idx, memory_bound, uncond_order)
    assert t_idx == edit_t_idx
    if basis is None:
        u, _, vT = pullback(zt, t)
    else:
        u, vT = basis
    un = u / u.norm(dim=0, keepdim=True)
    vn = vT / vT.norm(dim=1, keepdim=True)
    original = zt.clone()
    results: List[torch.Tensor] = []
    for pc in range(vis_num_pc):
        for direction in (1, -1):
            vk = direction * vn[pc, :].view(-1, *zT.shape[1:])
            lst = [original.clone()]
            for _ in range(num_step):
                lst.append(x_space_guidance(eps_edit, sch, lst[-1], edit_t_idx, vk, edit_step, scale))
            zs = torch.cat(lst, dim=0)
            zs = zs[::(zs.size(0) // vis_num)]
            results.append(latent_scale * ddim_forward(eps_for, sch, zs, for_steps, edit_t_idx, -1, memory_bound, uncond_order))
    return dict(zT=zT, zt=original, t=t, t_idx=edit_t_idx, u=u, un=un, vT=vT, results=results)
