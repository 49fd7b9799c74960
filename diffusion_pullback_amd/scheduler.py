"""DDIM scheduler of the reference, with the elementwise update on the HIP engine.

Mirrors (same names / arguments / attributes):
  * ``YHCustomScheduler``  reference src/utils/utils.py:1171-1281
  * patched ``set_timesteps`` / ``step`` for the Stable-Diffusion scheduler, utils.py:273-315
  * ``SchedulerOutput`` utils.py:1166-1169, ``extract`` utils.py:1302-1317

The timestep tables are tiny host-side tensors; ``step`` runs the update on the device through
``dpb_ddim_step`` (eta == 0, same operation order as utils.py:301-306) or ``dpb_lincomb`` (eta > 0).
CPU tensors are rejected: there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


class SchedulerOutput(object):
    def __init__(self, xt_next, P_xt):
        self.prev_sample = xt_next
        self.x0 = P_xt


def extract(a: torch.Tensor, t, x_shape):
    """utils.py:1302-1317 -- coefficient lookup; the float timestep is truncated with .long()."""
    if isinstance(t, int):
        t = torch.tensor([t]).repeat(x_shape[0])
    elif isinstance(t, torch.Tensor):
        t = t.repeat(x_shape[0])
    else:
        raise ValueError(f"t must be int or torch.Tensor, got {type(t)}")
    bs, = t.shape
    assert x_shape[0] == bs, f"{x_shape[0]}, {t.shape}"
    out = torch.gather(a, 0, t.long().to(a.device))
    assert out.shape == (bs,)
    return out.reshape((bs,) + (1,) * (len(x_shape) - 1))


class YHCustomScheduler(object):
    def __init__(self, args=None, alphas_cumprod: torch.Tensor = None, noise_schedule: str = None):
        self.t_max = 999
        ns = noise_schedule if noise_schedule is not None else getattr(args, "noise_schedule", None)
        self.noise_schedule = "linear" if ns is None else ns
        self.timesteps = None
        self.timesteps_next = None
        self.learn_sigma = False
        if alphas_cumprod is not None:                      # SD: the pipeline's table (utils.py:264-266)
            self.alphas_cumprod = alphas_cumprod.detach().float().cpu()
            self.betas = None
        else:
            self.get_alphas_cumprod()

    # utils.py:1243-1264
    def get_alphas_cumprod(self):
        if self.noise_schedule == "linear":
            betas = torch.linspace(0.0001, 0.02, 1000, dtype=torch.float64)
        elif self.noise_schedule == "scaled_linear":        # Stable Diffusion's table (diffusers DDIMScheduler, third party)
            betas = (torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2)
        else:
            raise ValueError(f"noise_schedule {self.noise_schedule} not supported")
        self.betas = betas.to(torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).to(torch.float32)

    # utils.py:1182-1195 (= :273-286)
    def set_timesteps(self, num_inferences, device=None, is_inversion=False):
        seq = torch.linspace(0, 1, num_inferences) * self.t_max
        if is_inversion:
            seq = seq + 1e-6
            seq_prev = torch.cat([torch.tensor([-1]), seq[:-1]], dim=0)
            self.timesteps = seq_prev[1:]
            self.timesteps_next = seq[1:]
        else:
            seq_prev = torch.cat([torch.tensor([-1]), seq[:-1]], dim=0)
            self.timesteps = reversed(seq[1:])
            self.timesteps_next = reversed(seq_prev[1:])

    def scale_model_input(self, sample, t=None):            # diffusers API used at edit.py:158 (identity for DDIM)
        return sample

    # utils.py:1197-1241 (= :288-315)
    def step(self, et, t, xt, eta=0.0, noise=None, **kwargs):
        assert et.shape == xt.shape, "et, xt shape should be same"
        if not xt.is_cuda:
            raise L.DpbError("scheduler.step needs device tensors (no CPU fallback in the product path)")
        lib = L.load()
        t = t if torch.is_tensor(t) else torch.tensor(float(t))
        t_idx = self.timesteps.tolist().index(float(t))
        t_next = self.timesteps_next[t_idx]
        at = float(extract(self.alphas_cumprod, t.reshape(()).cpu(), (1,)).item())
        at_next = float(extract(self.alphas_cumprod, t_next.reshape(()).cpu(), (1,)).item())
        x = xt.detach().to(torch.float32).contiguous()
        e = et.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x)
        x0 = torch.empty_like(x)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            if eta == 0:
                L.check(lib.dpb_ddim_step(x.data_ptr(), e.data_ptr(), out.data_ptr(), x0.data_ptr(), x.numel(), at, at_next, st))
            else:
                a_t = torch.tensor(at, dtype=torch.float32); a_n = torch.tensor(at_next, dtype=torch.float32)
                sigma = ((1 - a_t / a_n) * (1 - a_n) / (1 - a_t)).sqrt()
                d = (1 - a_n - eta * sigma ** 2).sqrt()
                c_x = (a_n.sqrt() / a_t.sqrt()).item()
                c_e = (d - a_n.sqrt() * (1 - a_t).sqrt() / a_t.sqrt()).item()
                noise = torch.randn_like(x) if noise is None else noise.to(x)
                L.check(lib.dpb_lincomb(x.data_ptr(), e.data_ptr(), noise.data_ptr(), out.data_ptr(), x.numel(), c_x, c_e,
                                        float(eta * sigma), st))
                L.check(lib.dpb_lincomb(x.data_ptr(), e.data_ptr(), None, x0.data_ptr(), x.numel(), (1 / a_t.sqrt()).item(),
                                        (-(1 - a_t).sqrt() / a_t.sqrt()).item(), 0.0, st))
        return SchedulerOutput(out.to(xt.dtype), x0.to(xt.dtype))


def get_custom_diffusion_scheduler(args):
    """utils.py:31-54 (only the YH custom scheduler is reachable on the live path)."""
    if getattr(args, "use_yh_custom_scheduler", True):
        return YHCustomScheduler(args)
    raise ValueError("recommend to use yh custom scheduler")


def get_stable_diffusion_scheduler(args, scheduler=None):
    """utils.py:261-271: SD keeps the pipeline's alphas_cumprod and gets the custom set_timesteps/step."""
    ac = getattr(scheduler, "alphas_cumprod", None)
    if ac is not None:
        return YHCustomScheduler(args, alphas_cumprod=ac)
    return YHCustomScheduler(args, noise_schedule="scaled_linear")
