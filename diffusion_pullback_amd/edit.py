"""Experiment drivers of the pullback editing path, latents resident on the device.

Mirrors the public surface of the reference's ``modules/edit.py``:
  * ``EditStableDiffusion``  (reference src/modules/edit.py:31-535)
      run_DDIMinversion :112-183, run_edit_local_encoder_pullback_zt :185-307, DDIMforwardsteps :385-482,
      x_space_guidance :484-502, run_DDIMforward :101-110
  * ``EditUncondDiffusion``  (reference src/modules/edit.py:540-779, :1601-1734)
      run_DDIMinversion :613-678, run_edit_local_encoder_pullback_zt :680-779, DDIMforwardsteps :1601-1714,
      x_space_guidance :1716-1734

Kept from the reference: the step counts (inv_steps-2 inversion steps, ``edit_t_idx`` forward steps,
``x_space_guidance_num_step`` guidance steps, the rest decode), the ``[::len // vis_num]`` subsample,
the u / vT normalisation, the ``.pt`` basis cache naming, EXP_NAME / result file naming, the
"already done" skips and the eta=1 "performance boosting" tail of the unconditional decode.
Changed on purpose: latents never bounce through a CPU buffer (the reference moves them every
step, edit.py:434, :469-472); chunks of ``memory_bound`` samples still bound the U-Net batch.

Third-party pieces the reference takes from diffusers (VAE, CLIP prompt encoder, image datasets) are
injected: ``vae`` (``encode(x)->latent``, ``decode(z)->image``), ``prompt_encoder(str)->[1,L,D]``,
``dataset[idx]->[1,3,H,W]``.  Without them the drivers run on synthetic latents / seeded prompt
embeddings and save latents instead of decoded images (there are no weights offline).
"""
from __future__ import annotations

import os
import zlib
from typing import Optional

import torch

from . import timing as T
from .scheduler import get_custom_diffusion_scheduler, get_stable_diffusion_scheduler


def save_image(x: torch.Tensor, path: str, nrow: Optional[int] = None) -> None:
    """Minimal stand-in for torchvision.utils.save_image (edit.py:480): images in [0,1], one row."""
    x = x.detach().float().clamp(0, 1).cpu()
    if x.dim() == 2:                                   # a matrix (the unconditional driver saves the raw vT): one grey image
        x = x[None, None]
    if x.dim() == 3:
        x = x[None]
    if x.shape[1] not in (1, 3):                       # latents: keep as tensor next to the requested name
        torch.save(x, os.path.splitext(path)[0] + ".pt")
        return
    try:
        from PIL import Image
        row = torch.cat(list(x), dim=2)                # [C, H, B*W]
        arr = (row.permute(1, 2, 0) * 255 + 0.5).to(torch.uint8).numpy()
        Image.fromarray(arr.squeeze(-1) if arr.shape[-1] == 1 else arr).save(path, compress_level=1)   # same pixels, a third of the default level's encode time
    except Exception:
        torch.save(x, os.path.splitext(path)[0] + ".pt")


def save_spectrum_plot(s: torch.Tensor, path: str, dpi=None) -> None:
    """Scatter plot of the singular values next to the cached basis (edit.py:249-251 / :733-735); skipped without matplotlib."""
    try:
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
    except Exception:
        return
    plt.scatter(range(s.size(0)), s.detach().float().cpu().tolist(), s=1)
    plt.savefig(path, **({"dpi": dpi} if dpi else {}))
    plt.close()


class _SeededPrompts:
    """Stand-in for pipe._encode_prompt (edit.py:505-522): a deterministic embedding per prompt string."""

    def __init__(self, length: int, dim: int):
        self.length, self.dim = length, dim

    def __call__(self, prompt: str) -> torch.Tensor:
        g = torch.Generator().manual_seed(zlib.crc32(prompt.encode()) if prompt else 0)
        return torch.randn(1, self.length, self.dim, generator=g)


class _EditBase(object):
    latent_scale = 1.0

    def _chunks(self, x, per_sample: int = 1):
        """Batches of at most ``memory_bound`` samples (edit.py:434-438), further bounded by the engine's own batch limit
        (``per_sample`` = 2 under classifier-free guidance, which doubles the U-Net batch).  Samples are independent, so the
        batching is not observable in the result."""
        cap = getattr(getattr(self.unet, "engine", None), "max_batch", None)
        want = max(self.memory_bound, getattr(self, "trajectory_batch", 0))      # trajectory batching (below) raises the bound on purpose
        bound = want if cap is None else max(1, min(want, cap // per_sample))
        return list(x.split(bound))

    _phase = "U-Net forward (other)"

    def _eps(self, x, t, emb=None):
        with T.phase(self._phase):
            out = self.unet(x, t) if emb is None else self.unet(x, t, encoder_hidden_states=emb)
        return out if isinstance(out, torch.Tensor) else out.sample

    def _basis_paths(self, save_dir, name):
        os.makedirs(save_dir, exist_ok=True)
        return [os.path.join(save_dir, p + name + ".pt") for p in ("u-", "s-", "vT-")]


# =================================================================== Stable Diffusion
class EditStableDiffusion(_EditBase):
    def __init__(self, args, unet=None, vae=None, prompt_encoder=None, dataset=None, scheduler=None):
        self.seed = args.seed
        self.memory_bound = getattr(args, "memory_bound", 5)
        # > 1: the 2 * vis_num_pc independent (pc, +-) edits of run_edit_local_encoder_pullback_zt advance TOGETHER -- their x-space-guidance chains
        # in one U-Net call per step, their decode trajectories in one batch -- instead of one after another as edit.py:276-307 does.  Samples are
        # independent in every kernel, so files, names and tensors are the same; the GPU sees 4x fewer, 4x fatter launches.  0 / 1: the reference's order.
        self.trajectory_batch = int(getattr(args, "trajectory_batch", 0) or 0)
        self.memory_bound_given = getattr(args, "memory_bound_given", None)      # explicit --memory_bound: bounds every U-Net call (floor: the pair of x-space guidance)
        self.unet = unet
        self.vae = vae
        self.dtype = getattr(args, "dtype", torch.float32)
        self.device = torch.device(args.device)
        self.scheduler = get_stable_diffusion_scheduler(args, scheduler)
        self.for_steps, self.inv_steps = args.for_steps, args.inv_steps
        self.use_yh_custom_scheduler = args.use_yh_custom_scheduler
        self.c_in, self.image_size = getattr(args, "c_in", 4), getattr(args, "image_size", 64)
        self.dataset = dataset
        self.dataset_name = args.dataset_name
        cfg = getattr(unet, "config", None)
        self._encode = prompt_encoder or _SeededPrompts(getattr(cfg, "ctx_len", 77), getattr(cfg, "cross_dim", 768))
        keep = lambda p: p if len(p.split(",")[0]) <= 3 else ",".join([p.split(",")[0]])      # edit.py:60-63
        self.for_prompt, self.neg_prompt, self.inv_prompt = keep(args.for_prompt), keep(args.neg_prompt), keep(args.inv_prompt)
        self.null_prompt = ""
        self.for_prompt_emb = self._get_prompt_emb(args.for_prompt)
        self.neg_prompt_emb = self._get_prompt_emb(args.neg_prompt)
        self.null_prompt_emb = self._get_prompt_emb("")
        self.inv_prompt_emb = self._get_prompt_emb(args.inv_prompt)
        self.guidance_scale = args.guidance_scale
        self.edit_prompt = args.edit_prompt
        self.edit_prompt_emb = self._get_prompt_emb(args.edit_prompt)
        self.x_edit_step_size = getattr(args, "x_edit_step_size", None)
        self.x_space_guidance_edit_step = args.x_space_guidance_edit_step
        self.x_space_guidance_scale = args.x_space_guidance_scale
        self.x_space_guidance_num_step = args.x_space_guidance_num_step
        self.x_space_guidance_use_edit_prompt = getattr(args, "x_space_guidance_use_edit_prompt", True)
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        self.edit_t = args.edit_t
        self.edit_t_idx = (self.scheduler.timesteps - self.edit_t * 1000).abs().argmin()       # edit.py:95
        self.result_folder, self.obs_folder = args.result_folder, args.obs_folder
        self.input_root = getattr(args, "input_root", "./inputs")
        self.EXP_NAME = "exp"

    def _get_prompt_emb(self, prompt):
        with T.phase("prompt encoding (CLIP text model)"):
            return self._encode(prompt).to(device=self.device, dtype=torch.float32)

    @torch.no_grad()
    def run_DDIMforward(self, num_samples=5):
        self.EXP_NAME = f"DDIMforward-for_{self.for_prompt}"
        zT = torch.randn(num_samples, self.c_in, self.image_size, self.image_size).to(device=self.device, dtype=self.dtype)
        return self.DDIMforwardsteps(zT, t_start_idx=0, t_end_idx=-1)

    @torch.no_grad()
    def run_DDIMinversion(self, idx, guidance=None, vis_traj=False):
        print("start DDIMinversion")
        self._phase = "DDIM inversion: U-Net forwards"
        self.EXP_NAME = f"DDIMinversion-{self.dataset_name}-{idx}-for_{self.for_prompt}-inv_{self.inv_prompt}"
        do_cfg = (self.guidance_scale > 1.0) & (guidance is not None)
        if not self.use_yh_custom_scheduler:
            raise ValueError("recommend to use yh custom scheduler")
        self.scheduler.set_timesteps(self.inv_steps, device=self.device, is_inversion=True)
        timesteps = self.scheduler.timesteps
        if self.dataset is not None and self.vae is not None:
            x0 = self.dataset[idx].to(self.device)
            save_image((x0 / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, f"original_x0-{self.EXP_NAME}.png"))
            z0 = self.vae.encode(x0) * 0.18215                                                  # edit.py:144-146
        else:                                                                                  # synthetic latent (no VAE offline)
            z0 = torch.randn(1, self.c_in, self.image_size, self.image_size, generator=torch.Generator().manual_seed(int(idx)))
        latents = z0.to(device=self.device, dtype=self.dtype)
        for i, t in enumerate(timesteps):
            if i == len(timesteps) - 1:
                break
            if do_cfg:
                emb = torch.cat([self.null_prompt_emb.repeat(latents.size(0), 1, 1), self.inv_prompt_emb.repeat(latents.size(0), 1, 1)], dim=0)
                e_u, e_t = self._eps(torch.cat([latents] * 2), t, emb).chunk(2)
                noise_pred = e_u + self.guidance_scale * (e_t - e_u)
            else:
                noise_pred = self._eps(latents, t, self.inv_prompt_emb.repeat(latents.size(0), 1, 1))
            latents = self.scheduler.step(noise_pred, t, latents, eta=0).prev_sample
        return latents

    @torch.no_grad()
    def DDIMforwardsteps(self, zt, t_start_idx, t_end_idx, **kwargs):
        print("start DDIMforward")
        self._phase = "DDIM forward to edit_t: U-Net forwards" if t_end_idx != -1 else "DDIM decode of the edited latents: U-Net forwards"
        do_cfg = self.guidance_scale > 1.0
        if not self.use_yh_custom_scheduler:
            raise ValueError("recommend to use yh custom scheduler")
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        latents = zt
        for t_idx, t in enumerate(self.scheduler.timesteps):
            if t_idx < t_start_idx:
                continue
            elif t_start_idx == t_idx:
                print("t_start_idx : ", t_idx)
            elif t_idx == t_end_idx:                                                            # edit.py:429-431
                print("t_end_idx : ", t_idx)
                return latents, t, t_idx
            outs = []
            for lat in self._chunks(latents, 2 if do_cfg else 1):
                if do_cfg:
                    emb = torch.cat([self.neg_prompt_emb.repeat(lat.size(0), 1, 1), self.for_prompt_emb.repeat(lat.size(0), 1, 1)], dim=0)
                    e_u, e_c = self._eps(torch.cat([lat] * 2, dim=0), t, emb).chunk(2)
                    noise_pred = e_u + self.guidance_scale * (e_c - e_u)
                else:
                    noise_pred = self._eps(lat, t, self.for_prompt_emb.repeat(lat.size(0), 1, 1))
                outs.append(self.scheduler.step(noise_pred, t, lat, eta=0).prev_sample)
            latents = torch.cat(outs, dim=0)
        if kwargs.get("finish", True) is False:                                                 # batched trajectories: the caller finishes each experiment
            return latents
        return self._finish_decode(latents, self.EXP_NAME)

    def _finish_decode(self, latents, exp_name):
        """edit.py:476-482: latents / 0.18215 -> VAE decode -> clamp -> one PNG row per experiment."""
        latents = 1 / 0.18215 * latents
        with T.phase("VAE decode"):
            x0 = self.vae.decode(latents) if self.vae is not None else latents
            x0 = (x0 / 2 + 0.5).clamp(0, 1) if self.vae is not None else x0
        with T.phase("image files (PNG encode + write)"):
            save_image(x0, os.path.join(self.result_folder, f"x0_gen-{exp_name}.png"), nrow=x0.size(0))
        return latents

    @torch.no_grad()
    def x_space_guidance(self, zt, t_idx, vk, single_edit_step, use_edit_prompt=False):
        t = self.scheduler.timesteps[t_idx]
        self._phase = "x-space guidance: batch-2 U-Net forwards"
        zt_edit = zt + single_edit_step * vk                                                    # edit.py:490
        et = self._eps(torch.cat([zt, zt_edit], dim=0), t, self.edit_prompt_emb.repeat(2, 1, 1))
        et_null, et_edit = et.chunk(2)
        return zt + self.x_space_guidance_scale * (et_edit - et_null)                           # edit.py:501

    @torch.no_grad()
    def run_edit_local_encoder_pullback_zt(self, idx, op, block_idx, vis_num, vis_num_pc=1, vis_vT=False, pca_rank=50,
                                           edit_prompt=None, edit_t=None):
        print(f"current experiment : idx : {idx}, op : {op}, block_idx : {block_idx}, vis_num : {vis_num}, vis_num_pc : {vis_num_pc}, pca_rank : {pca_rank}, edit_prompt : {edit_prompt}")
        if edit_prompt is not None:
            self.edit_prompt = edit_prompt
            self.edit_prompt_emb = self._get_prompt_emb(self.edit_prompt)
        self.scheduler.set_timesteps(self.for_steps)
        zT = self.run_DDIMinversion(idx=idx)
        zt, t, t_idx = self.DDIMforwardsteps(zT, t_start_idx=0, t_end_idx=self.edit_t_idx)
        assert t_idx == self.edit_t_idx
        name = f'local_basis-{self.dataset_name}_{idx}-{self.edit_t}T-"{self.edit_prompt}"-{op}-block_{block_idx}-seed_{self.seed}'
        save_dir = os.path.join(self.input_root, f"local_encoder_pullback_stable_diffusion-dataset_{self.dataset_name}-num_steps_{self.for_steps}-pca_rank_{pca_rank}")
        u_path, s_path, vT_path = self._basis_paths(save_dir, name)
        if os.path.exists(u_path) and os.path.exists(vT_path):
            u = torch.load(u_path, map_location=self.device).type(self.dtype)
            vT = torch.load(vT_path, map_location=self.device).type(self.dtype)
        else:
            print("!!!RUN LOCAL PULLBACK!!!")
            with T.phase("local_encoder_pullback_zt (power iteration)"):
                u, s, vT = self.unet.local_encoder_pullback_zt(
                    sample=zt, timestep=t, encoder_hidden_states=self.edit_prompt_emb, op=op, block_idx=block_idx,
                    pca_rank=pca_rank, chunk_size=5, min_iter=10, max_iter=50, convergence_threshold=1e-4)   # edit.py:236-239
            vT = vT.to(device=self.device, dtype=self.dtype)
            torch.save(u, u_path); torch.save(s, s_path); torch.save(vT, vT_path)
            save_spectrum_plot(s, os.path.join(save_dir, f"eigenvalue_spectrum-{name}.png"))        # edit.py:249-251
            # vT shown through the 3 principal channel directions of its pixels, min-max normalised (edit.py:253-263)
            pix = vT.view(-1, *zT.shape[1:]).permute(0, 2, 3, 1).reshape(-1, zT.shape[1]).float()
            _, _, basis = torch.pca_lowrank(pix, q=min(3, pix.shape[1]), center=True, niter=2)
            vis = torch.einsum("bcwh,cp->bpwh", vT.view(-1, *zT.shape[1:]).float(), basis)
            vis = vis - vis.min()
            save_image(vis / vis.max(), os.path.join(self.obs_folder, f"vT-{name}.png"))
        self.last_basis = (u, vT)
        u = u / u.norm(dim=0, keepdim=True)                                                     # edit.py:267-268
        vT = vT / vT.norm(dim=1, keepdim=True)
        original_zt = zt.clone()
        results = []
        if self.trajectory_batch > 1:
            return self._edit_trajectories_together(idx, op, block_idx, vis_num, vis_num_pc, vT, original_zt, zT.shape[1:])
        for pc_idx in range(vis_num_pc):
            for direction in [1, -1]:
                tag = "pos" if direction == 1 else "neg"
                self.EXP_NAME = f"Edit_zt-{self.dataset_name}_{idx}-edit_{self.edit_t}T-{op}-block_{block_idx}-pc_{pc_idx:0=3d}_{tag}-edit_prompt_{self.edit_prompt}"
                if os.path.exists(os.path.join(self.result_folder, self.EXP_NAME + ".png")):
                    print("!!!ALREADY DONE EXP!!!")
                    continue
                vk = direction * vT[pc_idx, :].view(-1, *zT.shape[1:])
                zt_list = [original_zt.clone()]
                for _ in range(self.x_space_guidance_num_step):
                    zt_list.append(self.x_space_guidance(zt_list[-1], t_idx=self.edit_t_idx, vk=vk,
                                                         single_edit_step=self.x_space_guidance_edit_step,
                                                         use_edit_prompt=self.x_space_guidance_use_edit_prompt))
                zt = torch.cat(zt_list, dim=0)
                zt = zt[::(zt.size(0) // vis_num)]                                              # edit.py:301-302
                results.append(self.DDIMforwardsteps(zt, t_start_idx=self.edit_t_idx, t_end_idx=-1))
        return results

    def _edit_trajectories_together(self, idx, op, block_idx, vis_num, vis_num_pc, vT, original_zt, lat_shape):
        """The loop body of edit.py:276-307 for all pending (pc, +-) experiments at once (self.trajectory_batch > 1): same experiments, names and skips;
        results equal to the sequential path up to 16-bit rounding (the GEMM tile / split-K selection depends on batch x rows, so a batch-2n call sums
        in another order than n batch-2 calls: fp32 engines agree to 2e-3, bf16 to a few 1e-2 over the chained steps -- tests/test_gpu_edit.py); the n chains of x-space guidance take ONE U-Net call of batch 2 n per step ([z_1..z_n | z_1 + s v_1 .. z_n + s v_n]) and the n * (vis_num + 1)
        decode trajectories ONE call of that batch per DDIM step (further split only by the engine's batch limit)."""
        todo = []
        for pc_idx in range(vis_num_pc):
            for direction in [1, -1]:
                tag = "pos" if direction == 1 else "neg"
                name = f"Edit_zt-{self.dataset_name}_{idx}-edit_{self.edit_t}T-{op}-block_{block_idx}-pc_{pc_idx:0=3d}_{tag}-edit_prompt_{self.edit_prompt}"
                if os.path.exists(os.path.join(self.result_folder, name + ".png")):
                    print("!!!ALREADY DONE EXP!!!")
                    continue
                todo.append((name, direction * vT[pc_idx, :].view(-1, *lat_shape)))
        if not todo:
            return []
        n = len(todo)
        t = self.scheduler.timesteps[self.edit_t_idx]
        self._phase = "x-space guidance: batch-2 U-Net forwards"
        vk = torch.cat([v for _, v in todo], dim=0)                                             # [n, C, H, W]
        z = original_zt.repeat(n, 1, 1, 1)
        chain = [z]
        cap = getattr(getattr(self.unet, "engine", None), "max_batch", None) or 2 * n
        per = max(1, min(n, cap // 2))                                                          # chains per U-Net call
        if self.memory_bound_given:
            per = max(1, min(per, self.memory_bound_given // 2))
        for _ in range(self.x_space_guidance_num_step):
            nxt = []
            for zc, vc in zip(z.split(per), vk.split(per)):
                m = zc.size(0)
                et = self._eps(torch.cat([zc, zc + self.x_space_guidance_edit_step * vc], dim=0), t, self.edit_prompt_emb.repeat(2 * m, 1, 1))   # edit.py:490
                et_null, et_edit = et.chunk(2)
                nxt.append(zc + self.x_space_guidance_scale * (et_edit - et_null))              # edit.py:501
            z = torch.cat(nxt, dim=0)
            chain.append(z)
        steps = torch.stack(chain, dim=1)                                                       # [n, num_step + 1, C, H, W]
        picked = steps[:, ::(steps.size(1) // vis_num)]                                         # edit.py:301-302, per chain
        m = picked.size(1)
        lat = self.DDIMforwardsteps(picked.reshape(n * m, *lat_shape), t_start_idx=self.edit_t_idx, t_end_idx=-1, finish=False)
        results = []
        for i, (name, _) in enumerate(todo):
            self.EXP_NAME = name
            results.append(self._finish_decode(lat[i * m:(i + 1) * m], name))
        return results


# =================================================================== unconditional (pixel space)
class EditUncondDiffusion(_EditBase):
    def __init__(self, args, unet=None, dataset=None):
        self.memory_bound = getattr(args, "memory_bound", 50)
        self.device = torch.device(args.device)
        self.dtype = getattr(args, "dtype", torch.float32)
        self.seed = args.seed
        self.unet = unet
        self.scheduler = get_custom_diffusion_scheduler(args)
        self.model_name = args.model_name
        self.image_size = getattr(args, "image_size", 256)
        self.c_in = 3
        self.dataset = dataset
        self.dataset_name = args.dataset_name
        self.for_steps, self.inv_steps = args.for_steps, args.inv_steps
        self.use_yh_custom_scheduler = args.use_yh_custom_scheduler
        self.edit_t = args.edit_t
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        self.edit_t_idx = (self.scheduler.timesteps - self.edit_t * 1000).abs().argmin()
        pb = getattr(args, "performance_boosting_t", 0.0)
        self.performance_boosting_t_idx = (self.scheduler.timesteps - pb * 1000).abs().argmin() if pb > 0 else 1000   # edit.py:584
        self.x_space_guidance_edit_step = args.x_space_guidance_edit_step
        self.x_space_guidance_scale = args.x_space_guidance_scale
        self.x_space_guidance_num_step = args.x_space_guidance_num_step
        self.result_folder, self.obs_folder = args.result_folder, args.obs_folder
        self.input_root = getattr(args, "input_root", "./inputs")
        self.EXP_NAME = "exp"

    @torch.no_grad()
    def run_DDIMforward(self, num_samples=5):
        self.EXP_NAME = "DDIMforward"
        xT = torch.randn(num_samples, self.c_in, self.image_size, self.image_size).to(device=self.device, dtype=self.dtype)
        return self.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=-1)

    @torch.no_grad()
    def run_DDIMinversion(self, idx):
        print("start DDIMinversion")
        name = f"DDIMinversion-{self.dataset_name}_{idx}"
        if not self.use_yh_custom_scheduler:
            raise ValueError("please set use_yh_custom_scheduler = True")
        self.scheduler.set_timesteps(self.inv_steps, device=self.device, is_inversion=True)
        timesteps = self.scheduler.timesteps
        if self.dataset is not None:
            x0 = self.dataset[idx]
        else:
            x0 = torch.randn(1, self.c_in, self.image_size, self.image_size, generator=torch.Generator().manual_seed(int(idx))).clamp(-1, 1)
        save_image((x0 / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, f"original_x0-{name}.png"))
        xt = x0.to(self.device, dtype=self.dtype)
        for i, t in enumerate(timesteps):
            if i == len(timesteps) - 1:
                break
            xt = self.scheduler.step(self._eps(xt, t), t, xt, eta=0).prev_sample
        save_image((xt / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, f"xT-{name}.png"))
        return xt

    @torch.no_grad()
    def DDIMforwardsteps(self, xt, t_start_idx, t_end_idx, vis_psd=False, save_image_=True, return_xt=True, performance_boosting=False):
        print("start DDIMforward")
        assert (t_start_idx < self.for_steps) & (t_end_idx <= self.for_steps)
        if not self.use_yh_custom_scheduler:
            raise ValueError("please set use_yh_custom_scheduler = True")
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        timesteps = self.scheduler.timesteps
        for i, t in enumerate(timesteps):
            if t_end_idx == i:                                                                  # edit.py:1640-1642
                print("t_end_idx : ", i)
                return xt, t, i
            elif i < t_start_idx:
                continue
            boost = performance_boosting & (self.performance_boosting_t_idx <= i) & (self.performance_boosting_t_idx != len(timesteps) - 1)
            eta = 1 if boost else 0                                                             # edit.py:1650-1653
            xt = torch.cat([self.scheduler.step(self._eps(c, t), t, c, eta=eta).prev_sample for c in self._chunks(xt)], dim=0)
        if save_image_:
            save_image((xt / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, f"x0_gen-{self.EXP_NAME}.png"), nrow=xt.size(0))
        return xt if return_xt else None

    @torch.no_grad()
    def x_space_guidance(self, xt, t_idx, vk, single_edit_step):
        t = self.scheduler.timesteps[t_idx]
        et_null, et_edit = self._eps(torch.cat([xt, xt + single_edit_step * vk], dim=0), t).chunk(2)
        return xt + self.x_space_guidance_scale * (et_edit - et_null)

    @torch.no_grad()
    def run_edit_local_encoder_pullback_zt(self, idx, vis_num, vis_num_pc=5, pca_rank=50, op="mid", block_idx=0, **kwargs):
        if self.dataset_name == "Random":
            xT = torch.randn(1, 3, self.image_size, self.image_size).to(device=self.device, dtype=self.dtype)
        else:
            xT = self.run_DDIMinversion(idx=idx)
        xt, t, t_idx = self.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=self.edit_t_idx)
        assert t_idx == self.edit_t_idx
        name = f"local_basis-{self.dataset_name}_{idx}-{self.edit_t}T-{op}-block_{block_idx}-seed_{self.seed}"
        save_dir = os.path.join(self.input_root, f"local_encoder_pullback_uncond-model_{self.model_name}-dataset_{self.dataset_name}-num_steps_{self.for_steps}-pca_rank_{pca_rank}")
        u_path, s_path, vT_path = self._basis_paths(save_dir, name)
        if os.path.exists(u_path) and os.path.exists(vT_path):
            u = torch.load(u_path, map_location=self.device).type(self.dtype)
            vT = torch.load(vT_path, map_location=self.device).type(self.dtype)
        else:
            print("!!!RUN LOCAL PULLBACK!!!")
            u, s, vT = self.unet.local_encoder_pullback_xt(x=xt.to(device=self.device, dtype=self.dtype), t=t, op=op, block_idx=block_idx,
                                                           pca_rank=pca_rank, min_iter=10, max_iter=50, convergence_threshold=1e-4)
            torch.save(u, u_path); torch.save(vT, vT_path)
            save_spectrum_plot(s, os.path.join(save_dir, f"eigenvalue_spectrum-{name}.png"), dpi=80)   # edit.py:733-735
            save_image(vT, os.path.join(self.obs_folder, f"vT-{name}.png"))                              # edit.py:737-741 (saves the raw vT)
        self.last_basis = (u, vT)
        u = u / u.norm(dim=0, keepdim=True)
        vT = vT / vT.norm(dim=1, keepdim=True)
        original_xt = xt.detach()
        for pc_idx in range(vis_num_pc):
            for direction in [1, -1]:
                tag = "pos" if direction == 1 else "neg"
                self.EXP_NAME = f"Edit_xt-{self.dataset_name}_{idx}-edit_{self.edit_t}T-{op}-block_{block_idx}-pc_{pc_idx:0=3d}_{tag}"
                if os.path.exists(os.path.join(self.result_folder, f"x0_gen-{self.EXP_NAME}.png")):
                    print("!!!ALREADY DONE!!!")
                    continue
                vk = direction * vT[pc_idx, :].view(-1, *xt.shape[1:])
                xt_list = [original_xt.clone()]
                for _ in range(self.x_space_guidance_num_step):
                    xt_list.append(self.x_space_guidance(xt_list[-1], t_idx=self.edit_t_idx, vk=vk, single_edit_step=self.x_space_guidance_edit_step))
                xt = torch.cat(xt_list, dim=0)
                xt = xt[::(xt.size(0) // vis_num)]
                self.DDIMforwardsteps(xt, t_start_idx=self.edit_t_idx, t_end_idx=-1, performance_boosting=True)
        return xt
