"""Generate the edit-loop and stop-rule golden fixtures by RUNNING THE REFERENCE'S OWN DRIVERS (build container only):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_edit.py

The reference's experiment classes cannot be constructed here (their __init__ downloads diffusers pipelines and datasets),
but their METHODS are plain Python over ``self.unet`` / ``self.scheduler`` / ``self.vae``: the instances are created with
``object.__new__`` and given the attributes __init__ would set, with
  * unet      = the vendored ``PullBackDDPM`` (reference U-Net, reduced width) for the unconditional driver;
                a toy SD-style net (oracle.unet_sd at toy width) carrying the reference's ``local_encoder_pullback_zt`` for the SD driver
  * scheduler = the reference's ``YHCustomScheduler`` / its patched ``set_timesteps`` + ``step`` (utils.py:273-315, :1171-1241)
  * vae / prompt encoder / dataset = fixed tensors (third-party pieces absent here); ``torchvision.utils.save_image`` captures what
    the driver would have written to PNG files.
Every U-Net call of the run is recorded, so the fixtures hold the whole trajectory (inversion, forward to edit_t, x-space guidance,
decode) plus the bases and file names -- rows a7, a10-a13 of SURVEY.md section 8.

Fixtures
  edit_uncond_small.pt   EditUncondDiffusion.run_edit_local_encoder_pullback_zt  (edit.py:613-779, :1601-1734)
  edit_sd_toy.pt         EditStableDiffusion.run_edit_local_encoder_pullback_zt  (edit.py:112-307, :385-502)
  pullback_history.pt    per-iteration ``dist`` printed by local_encoder_pullback_zt and its iteration count (utils.py:803-808)
"""
import contextlib
import io
import os
import re
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import torch  # noqa: E402

from make_golden import import_reference  # noqa: E402

SPEC = dict(rank=6, decay=0.7, amp=60.0, q_scale=0.1)       # spectrum shaping of the toy weights: well-conditioned top vectors


class Out:
    def __init__(self, sample):
        self.sample = sample


def _parse_history(text):
    d = [float(x) for x in re.findall(r"step convergence :\s+(?:tensor\()?([0-9.eE+-]+)", text)]
    return d, ("reach convergence threshold" in text)


def main():
    torch.set_num_threads(8)
    ru, rd = import_reference()
    import modules.edit as redit
    from diffusion_pullback_amd import configs as cf
    from oracle import unet_ddpm, unet_sd

    saved = []                                                 # (basename, tensor) of every tvu.save_image call
    redit.tvu.save_image = lambda x, path, **kw: saved.append((os.path.basename(path), x.detach().clone()))
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)                                              # the drivers write ./inputs/... relative to the cwd
    os.makedirs("res", exist_ok=True); os.makedirs("obs", exist_ok=True)
    try:
        # ================================================================ unconditional driver on the vendored DDPM
        cfgd = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), in_channels=3, out_ch=3, resolution=32)
        cfg = unet_ddpm.DDPMConfig(**cfgd)
        ns = ru.dict2namespace({"config": {"model": dict(ch=32, out_ch=3, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16],
                                                       dropout=0.0, in_channels=3, resamp_with_conv=True), "data": dict(image_size=32)}})
        ns.device = "cpu"; ns.dtype = torch.float32
        net = rd.PullBackDDPM(ns).eval()
        net.load_state_dict(cf.ddpm_init_params(cfg, seed=3, spectrum=cf.Spectrum(**SPEC)), strict=True)
        trace = []
        fwd0 = net.forward

        def traced(x, t, *a, **k):
            trace.append((float(t), x.detach().clone()))
            return fwd0(x, t, *a, **k)
        net.forward = traced

        class A:
            noise_schedule = None; device = "cpu"; dtype = torch.float32
        eu = object.__new__(redit.EditUncondDiffusion)
        x0 = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(31)).clamp(-1, 1)
        args_u = dict(for_steps=20, inv_steps=20, edit_t=0.6, x_space_guidance_edit_step=1.0, x_space_guidance_scale=0.1,
                      x_space_guidance_num_step=8, vis_num=4, vis_num_pc=1, pca_rank=2, idx=0, seed=0, rng_seed=41)
        eu.pca_device = eu.buffer_device = "cpu"; eu.memory_bound = 50; eu.device = "cpu"; eu.dtype = torch.float32; eu.seed = 0
        eu.save_result_as = "image"; eu.unet = net; eu.scheduler = ru.YHCustomScheduler(A()); eu.model_name = "CelebA_HQ_HF"
        eu.image_size = 32; eu.c_in = 3; eu.dataset = {0: x0}; eu.dataset_name = "CelebA_HQ"
        eu.for_steps, eu.inv_steps, eu.use_yh_custom_scheduler, eu.edit_t = 20, 20, True, 0.6
        eu.scheduler.set_timesteps(eu.for_steps, device="cpu")
        eu.edit_t_idx = (eu.scheduler.timesteps - eu.edit_t * 1000).abs().argmin()
        eu.performance_boosting_t_idx = 1000                   # eta = 1 tail off: its noise is a device RNG draw, not reproducible
        eu.use_x_space_guidance = True
        eu.x_space_guidance_edit_step, eu.x_space_guidance_scale, eu.x_space_guidance_num_step = 1.0, 0.1, 8
        eu.result_folder, eu.obs_folder = "res", "obs"
        torch.manual_seed(args_u["rng_seed"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            xt_last = eu.run_edit_local_encoder_pullback_zt(idx=0, vis_num=4, vis_num_pc=1, pca_rank=2, op="mid", block_idx=0)
        dists, converged = _parse_history(buf.getvalue())
        d = "inputs/local_encoder_pullback_uncond-model_CelebA_HQ_HF-dataset_CelebA_HQ-num_steps_20-pca_rank_2"
        files = sorted(os.listdir(d))
        name = "local_basis-CelebA_HQ_0-0.6T-mid-block_0-seed_0.pt"
        fix = dict(cfg=cfgd, seed=3, spectrum=SPEC, args=args_u, x0=x0, edit_t_idx=int(eu.edit_t_idx), trace_t=[t for t, _ in trace],
                   trace_x=[x for _, x in trace], u=torch.load(os.path.join(d, "u-" + name)), vT=torch.load(os.path.join(d, "vT-" + name)),
                   basis_files=files, basis_dir=d, dists=dists, converged=converged, xt_last=xt_last.clone(),
                   saved=[(n, x) for n, x in saved])
        torch.save(fix, os.path.join(HERE, "edit_uncond_small.pt"))
        print("uncond: unet calls", len(trace), "pullback iters", len(dists), "converged", converged, "saved", [n for n, _ in saved])
        saved.clear()

        # ================================================================ SD driver on a toy SD-style net
        scfgd = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1, down_attn=(True, False),
                     up_attn=(False, True), heads=(2, 2), cross_dim=16, groups=8, sample_size=8, ctx_len=5)
        scfg = unet_sd.SDConfig(**scfgd)
        sp = cf.sd_init_params(scfg, seed=9, gain=1.5, spectrum=cf.Spectrum(**SPEC))
        strace = []

        class ToyUNet:
            dtype = torch.float32

            def __call__(self, x, t, encoder_hidden_states=None, **kw):
                strace.append((float(t), x.detach().clone(), encoder_hidden_states.detach().clone()))
                return Out(unet_sd.forward(sp, scfg, x, t, encoder_hidden_states))

            def get_h(self, sample=None, timestep=None, encoder_hidden_states=None, op=None, block_idx=None, verbose=False):
                return unet_sd.forward(sp, scfg, sample, timestep, encoder_hidden_states, stop=(op, block_idx))
        toy = ToyUNet()
        toy.local_encoder_pullback_zt = types.MethodType(ru.local_encoder_pullback_zt, toy)

        g = torch.Generator().manual_seed(17)
        emb = {k: torch.randn(1, 5, 16, generator=g) for k in ("for", "inv", "neg", "null", "edit")}
        z0 = torch.randn(1, 4, 8, 8, generator=g)

        class Sched:                                           # the attributes utils.py:261-271 patches onto pipe.scheduler
            pass
        sch = Sched()
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2     # SD's scaled_linear table (diffusers, third party)
        sch.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0); sch.betas = betas
        sch.scale_model_input = lambda x, t: x

        class SA:
            use_yh_custom_scheduler = True; device = "cpu"; dtype = torch.float32
        sch = ru.get_stable_diffusion_scheduler(SA(), sch)

        class LD:
            def __init__(self, z): self.z = z
            def sample(self): return self.z

        class VAE:                                             # stand-in: encode -> the fixed latent / 0.18215, decode -> first 3 channels
            def encode(self, x): return types.SimpleNamespace(latent_dist=LD(z0 / 0.18215))
            def decode(self, l): return Out(l[:, :3])

        es = object.__new__(redit.EditStableDiffusion)
        args_s = dict(for_steps=20, inv_steps=20, edit_t=0.7, x_space_guidance_edit_step=1.0, x_space_guidance_scale=0.3,
                      x_space_guidance_num_step=8, vis_num=4, vis_num_pc=1, pca_rank=2, idx=5, seed=0, rng_seed=43, edit_prompt="tiger")
        es.seed = 0; es.pca_device = es.buffer_device = "cpu"; es.memory_bound = 5; es.vae = VAE(); es.unet = toy
        es.dtype = torch.float32; es.device = "cpu"; es.scheduler = sch; es.for_steps = es.inv_steps = 20; es.use_yh_custom_scheduler = True
        es.c_in, es.image_size = 4, 8; es.dataset = {5: torch.zeros(1, 3, 16, 16)}; es.dataset_name = "Examples"
        es.for_prompt = es.neg_prompt = es.inv_prompt = ""; es.null_prompt = ""
        es.for_prompt_emb, es.neg_prompt_emb, es.null_prompt_emb, es.inv_prompt_emb = emb["for"], emb["neg"], emb["null"], emb["inv"]
        es.guidance_scale = 0; es.edit_prompt = "sitting dog"; es.edit_prompt_emb = emb["for"]
        es._get_prompt_emb = lambda p: emb["edit"]             # run_edit re-encodes the edit prompt (edit.py:199-201)
        es.x_edit_step_size = 0
        es.x_space_guidance_edit_step, es.x_space_guidance_scale, es.x_space_guidance_num_step = 1.0, 0.3, 8
        es.x_space_guidance_use_edit_prompt = True
        es.scheduler.set_timesteps(es.for_steps, device="cpu")
        es.edit_t = 0.7
        es.edit_t_idx = (es.scheduler.timesteps - es.edit_t * 1000).abs().argmin()
        es.result_folder, es.obs_folder = "res", "obs"
        torch.manual_seed(args_s["rng_seed"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            es.run_edit_local_encoder_pullback_zt(idx=5, op="mid", block_idx=0, vis_num=4, vis_num_pc=1, pca_rank=2, edit_prompt="tiger")
        dists, converged = _parse_history(buf.getvalue())
        d = "inputs/local_encoder_pullback_stable_diffusion-dataset_Examples-num_steps_20-pca_rank_2"
        name = 'local_basis-Examples_5-0.7T-"tiger"-mid-block_0-seed_0.pt'
        fix = dict(cfg=scfgd, seed=9, gain=1.5, spectrum=SPEC, args=args_s, z0=z0, emb=emb, alphas_cumprod=sch.alphas_cumprod.clone(),
                   edit_t_idx=int(es.edit_t_idx), trace_t=[t for t, _, _ in strace], trace_x=[x for _, x, _ in strace],
                   trace_emb=[("for" if torch.equal(e[:1], emb["for"]) else "inv" if torch.equal(e[:1], emb["inv"]) else
                               "edit" if torch.equal(e[:1], emb["edit"]) else "?") for _, _, e in strace],
                   u=torch.load(os.path.join(d, "u-" + name)), s=torch.load(os.path.join(d, "s-" + name)), vT=torch.load(os.path.join(d, "vT-" + name)),
                   basis_files=sorted(os.listdir(d)), basis_dir=d, dists=dists, converged=converged, saved=[(n, x) for n, x in saved])
        torch.save(fix, os.path.join(HERE, "edit_sd_toy.pt"))
        print("sd: unet calls", len(strace), "pullback iters", len(dists), "converged", converged, "saved", [n for n, _ in saved])

        # ================================================================ stop rule: per-iteration dist and iteration count
        g = torch.Generator().manual_seed(13)
        z = torch.randn(1, 4, 8, 8, generator=g); ctx = torch.randn(1, 5, 16, generator=g); tt = torch.tensor(696.2727)
        cases = []
        for (spec, k, mn, mx, thr, rs) in [(SPEC, 3, 2, 40, 1e-3, 51), (SPEC, 2, 5, 40, 1e-4, 52), (None, 3, 1, 6, 1e-5, 53)]:
            p = cf.sd_init_params(scfg, seed=9, gain=1.5, spectrum=cf.Spectrum(**spec) if spec else None)

            class T2:
                dtype = torch.float32

                def get_h(self, sample=None, timestep=None, encoder_hidden_states=None, op=None, block_idx=None, verbose=False):
                    return unet_sd.forward(p, scfg, sample, timestep, encoder_hidden_states, stop=(op, block_idx))
            t2 = T2(); t2.local_encoder_pullback_zt = types.MethodType(ru.local_encoder_pullback_zt, t2)
            torch.manual_seed(rs)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                u, s, vT = t2.local_encoder_pullback_zt(z, tt, ctx, op="mid", block_idx=0, pca_rank=k, chunk_size=5, min_iter=mn, max_iter=mx,
                                                        convergence_threshold=thr)
            dists, converged = _parse_history(buf.getvalue())
            torch.manual_seed(rs)
            q, _ = torch.linalg.qr(torch.randn(256, k))
            cases.append(dict(spectrum=spec, k=k, min_iter=mn, max_iter=mx, thr=thr, rng_seed=rs, V0=q.T.contiguous(), dists=dists,
                              iters=len(dists), converged=converged, u=u.clone(), s=s.clone(), vT=vT.clone()))
            print("history case", k, mn, mx, thr, "iters", len(dists), "converged", converged, "last dists", [f"{x:.2e}" for x in dists[-3:]])
        torch.save(dict(cfg=scfgd, seed=9, gain=1.5, z=z, ctx=ctx, t=tt, cases=cases), os.path.join(HERE, "pullback_history.pt"))
    finally:
        os.chdir(cwd)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
