"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is pure Python; it cannot travel to the GPU box, so only the
input/output tensors it produces are committed (``*.pt`` here), never its code.
The reference's third-party imports that are absent in this image (diffusers,
torchvision, skimage) are stubbed with empty modules -- none of the stubbed names
is touched by the functions called below.

Fixtures
  scheduler.pt        YHCustomScheduler / set_timesteps / step / extract outputs
  ddpm_small.pt       vendored PullBackDDPM (reduced width) get_h at every (op, idx)
                      and full eps, weights = oracle.unet_ddpm.init_params(cfg, seed)
  pullback_xt_ddpm.pt PullBackDDPM.local_encoder_pullback_xt (diffusion.py:484-556) result
  pullback_zt_tiny.pt utils.local_encoder_pullback_zt / _xt (utils.py:722-816, :165-249)
                      bound onto a tiny seeded net (oracle.unet_sd at toy width)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Make `utils.utils` and `models.ddpm.diffusion` of the reference importable."""
    import importlib.machinery as mach
    for name in ("torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.datasets",
                 "torchvision.datasets.utils", "skimage", "lmdb"):
        if name not in sys.modules:
            m = _stub(name)
            m.__spec__ = mach.ModuleSpec(name, None)
            m.__path__ = []
    tv = sys.modules["torchvision"]
    tv.utils = sys.modules["torchvision.utils"]; tv.transforms = sys.modules["torchvision.transforms"]
    tv.datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision.datasets"].utils = sys.modules["torchvision.datasets.utils"]
    for n in ("verify_str_arg", "iterable_to_str"):
        setattr(sys.modules["torchvision.datasets.utils"], n, lambda *a, **k: None)
    sys.modules["torchvision.datasets"].VisionDataset = object
    if "diffusers" not in sys.modules:
        d = _stub("diffusers", DDIMScheduler=object, DDIMPipeline=object, StableDiffusionPipeline=object)
        d.__spec__ = mach.ModuleSpec("diffusers", None)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import utils.utils as ru                      # noqa: E402
    import models.ddpm.diffusion as rd            # noqa: E402
    return ru, rd


def main():
    import torch
    torch.set_num_threads(8)
    ru, rd = import_reference()
    from oracle import unet_ddpm, unet_sd

    # ---------------------------------------------------------------- scheduler
    class A:  # args namespace the reference scheduler reads
        noise_schedule = None; device = "cpu"; dtype = torch.float32
    sch = ru.YHCustomScheduler(A())
    out = {"alphas_cumprod": sch.alphas_cumprod.clone(), "betas": sch.betas.clone()}
    for n in (100, 50, 10):
        sch.set_timesteps(n)
        out[f"fwd{n}_t"] = sch.timesteps.clone(); out[f"fwd{n}_tn"] = sch.timesteps_next.clone()
        sch.set_timesteps(n, is_inversion=True)
        out[f"inv{n}_t"] = sch.timesteps.clone(); out[f"inv{n}_tn"] = sch.timesteps_next.clone()
    g = torch.Generator().manual_seed(7)
    xt = torch.randn(2, 3, 8, 8, generator=g); et = torch.randn(2, 3, 8, 8, generator=g)
    out["xt"], out["et"] = xt, et
    sch.set_timesteps(100)
    for i in (0, 30, 98):
        r = sch.step(et, sch.timesteps[i], xt, eta=0.0)
        out[f"step_fwd_{i}"] = r.prev_sample.clone(); out[f"x0_fwd_{i}"] = r.x0.clone()
    out["edit_idx"] = {e: int((sch.timesteps - e * 1000).abs().argmin()) for e in (1.0, 0.8, 0.7, 0.6, 0.2)}
    sch.set_timesteps(100, is_inversion=True)
    for i in (0, 50, 97):
        r = sch.step(et, sch.timesteps[i], xt, eta=0.0)
        out[f"step_inv_{i}"] = r.prev_sample.clone()
    torch.save(out, os.path.join(HERE, "scheduler.pt"))

    # ---------------------------------------------------------------- vendored DDPM, reduced width
    cfgd = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), in_channels=3, out_ch=3, resolution=32)
    cfg = unet_ddpm.DDPMConfig(**cfgd)
    ns = ru.dict2namespace({"config": {"model": dict(ch=32, out_ch=3, ch_mult=[1, 2, 2], num_res_blocks=1,
                                                   attn_resolutions=[16], dropout=0.0, in_channels=3, resamp_with_conv=True),
                                      "data": dict(image_size=32)}})
    ns.device = "cpu"; ns.dtype = torch.float32
    net = rd.PullBackDDPM(ns).eval()
    params = unet_ddpm.init_params(cfg, seed=3)
    missing = net.load_state_dict(params, strict=True)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, 32, 32, generator=g)
    t = torch.tensor(600.0)
    fix = {"cfg": cfgd, "seed": 3, "x": x, "t": t}
    with torch.no_grad():
        for op, idx in [("down", 0), ("down", 1), ("down", 2), ("mid", 0), ("up", 2), ("up", 1), ("up", 0)]:
            fix[f"h_{op}_{idx}"] = net.get_h(x, t, op=op, block_idx=idx).clone()
        fix["eps"] = net(x, t).clone()
        xb = torch.randn(2, 3, 32, 32, generator=g)
        fix["xb"] = xb; fix["eps_b"] = net(xb, t).clone()
    torch.save(fix, os.path.join(HERE, "ddpm_small.pt"))

    # pullback through the vendored class' own copy of the algorithm
    torch.manual_seed(5)
    u, s, vT = net.local_encoder_pullback_xt(x=x, t=t, op="mid", block_idx=0, pca_rank=3, chunk_size=2,
                                             min_iter=2, max_iter=6, convergence_threshold=1e-3)
    torch.manual_seed(5)
    q, _ = torch.linalg.qr(torch.randn(3 * 32 * 32, 3))
    torch.save({"cfg": cfgd, "seed": 3, "x": x, "t": t, "V0": q.T.contiguous(), "rng_seed": 5, "k": 3, "chunk_size": 2,
                "min_iter": 2, "max_iter": 6, "thr": 1e-3, "u": u.clone(), "s": s.clone(), "vT": vT.clone()},
               os.path.join(HERE, "pullback_xt_ddpm.pt"))

    # ---------------------------------------------------------------- utils.py algorithms on a toy SD-style net
    scfgd = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1, down_attn=(True, False),
                 up_attn=(False, True), heads=(2, 2), cross_dim=16, groups=8, sample_size=8, ctx_len=5)
    scfg = unet_sd.SDConfig(**scfgd)
    sp = unet_sd.init_params(scfg, seed=9, gain=1.5)

    class Toy:
        dtype = torch.float32

        def get_h(self, sample=None, timestep=None, encoder_hidden_states=None, op=None, block_idx=None, verbose=False):
            return unet_sd.forward(sp, scfg, sample, timestep, encoder_hidden_states, stop=(op, block_idx))

    class ToyU:
        dtype = torch.float32

        def get_h(self, x=None, t=None, op=None, block_idx=None, verbose=False, **kw):
            return unet_sd.forward(sp, scfg, x, t, ctx.expand(x.shape[0], -1, -1), stop=(op, block_idx))

    g = torch.Generator().manual_seed(13)
    z = torch.randn(1, 4, 8, 8, generator=g)
    ctx = torch.randn(1, 5, 16, generator=g)
    tt = torch.tensor(696.2727)
    toy = Toy(); toy.local_encoder_pullback_zt = types.MethodType(ru.local_encoder_pullback_zt, toy)
    toyu = ToyU(); toyu.local_encoder_pullback_xt = types.MethodType(ru.local_encoder_pullback_xt, toyu)
    res = {"cfg": scfgd, "seed": 9, "gain": 1.5, "z": z, "ctx": ctx, "t": tt, "cases": []}
    for (op, idx, k, chunk, mn, mx, thr, rs) in [("mid", 0, 3, 5, 2, 8, 1e-4, 21), ("mid", 0, 5, 2, 1, 5, 1e-3, 22), ("up", 0, 4, 2, 1, 4, 1e-3, 23)]:
        torch.manual_seed(rs)
        u, s, vT = toy.local_encoder_pullback_zt(z, tt, ctx, op=op, block_idx=idx, pca_rank=k, chunk_size=chunk,
                                                 min_iter=mn, max_iter=mx, convergence_threshold=thr)
        torch.manual_seed(rs)
        u2, s2, vT2 = toyu.local_encoder_pullback_xt(z, tt, op=op, block_idx=idx, pca_rank=k, chunk_size=chunk,
                                                     min_iter=mn, max_iter=mx, convergence_threshold=thr)
        torch.manual_seed(rs)
        q, _ = torch.linalg.qr(torch.randn(4 * 8 * 8, k))
        # full Jacobian SVD: what the iteration converges to
        J = torch.autograd.functional.jacobian(lambda a: toy.get_h(a, tt, ctx, op, idx).reshape(-1), z).reshape(-1, 256)
        res["cases"].append(dict(op=op, idx=idx, k=k, chunk=chunk, min_iter=mn, max_iter=mx, thr=thr, rng_seed=rs,
                                 V0=q.T.contiguous(), zt=(u.clone(), s.clone(), vT.clone()),
                                 xt=(u2.clone(), s2.clone(), vT2.clone()), svals=torch.linalg.svdvals(J)[:k].clone()))
    torch.save(res, os.path.join(HERE, "pullback_zt_tiny.pt"))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
