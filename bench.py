#!/usr/bin/env python
"""Headline benchmark: pullback top-k SVD power iterations / second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                               # BASELINE configs[2], weak scaling
    python bench.py --gpus N --samples 64 --k 10 --ctx edit --samples-per-gpu 8   # BASELINE configs[3], strong scaling

A *step* is one power iteration for k directions on one x_t sample: k JVPs + k VJPs through the U-Net prefix up to the
tap + the k x N re-orthonormalisation (reference src/utils/utils.py:756-808).  A sample's job is one primal (stash) pass
+ ITERS_PER_SAMPLE = 12 iterations (= min_iter + 2, the reference's minimum and what its Colab log ran); the primal pass
of every sample started inside the timed region is timed too.  Inputs (x_t, ctx, V0) are resident in HBM before the
clock starts.  Weights are the seeded synthetic ones with the shaped spectrum (configs.Spectrum): the top singular values
are separated, so the reference's own stop rule converges (reported as `time_to_converged_basis`).

N > 1: one process per GPU (torchrun), samples dealt to ranks, no collective inside the iterations, ONE packed RCCL
all_gather of the final bases (dist.gather_bases) inside the timed region.
  * default (weak): every rank runs the same per-GPU job on its own samples -- value = N * K / max-over-ranks time;
  * --samples T (strong): T samples in total (seeds 0..T-1), sample i on rank i mod N (dist.shard_indices), the whole job is
    T * 12 steps -- value = T * 12 / max-over-ranks time, "scaling": "strong" (the north star's 64-sample config).

Output: ONE JSON line on rank 0 with `roofline` (dominant kernel = most GPU time, measured live with HIP events around
every GEMM launch on the engine's stream in a separate instrumented iteration) and `cpu_baseline` (the CPU oracle, i.e.
the reference's algorithm with the same autodiff calls, BASELINE.md section 3 protocol: 1 warm-up + 2 timed iterations on
this host's cores, CPU model stated).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes fails with hipIpcGetMemHandle errors without it (set before HIP loads)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS_PER_SAMPLE = 12
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}      # TFLOP/s dense MFMA (MI355X_MICROARCH.md)
MAC_G = {"sd15": 130.72, "ddpm256": 67.58, "sd21": 130.72}   # GMAC of one get_h(mid) forward; SD-2.1-base: the same
                                                             # linear maps except the text K / V projections (1024- instead of 768-wide, x-independent)
TORCH_DTYPE = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--workload", default="sd15", choices=["sd15", "sd21", "ddpm256", "toy"],
                    help="sd15: BASELINE configs[2..4] (headline); sd21: stabilityai/stable-diffusion-2-1-base, the reference scripts' own default model")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--op", default="mid", choices=["down", "mid", "up"], help="feature tap (BASELINE configs[4] sweeps down/up 0..3)")
    ap.add_argument("--block-idx", type=int, default=0)
    ap.add_argument("--samples-per-gpu", type=int, default=1,
                    help="x_t samples advanced together per GPU (independent bases, shared weight stream); weak mode: steps must be a multiple")
    ap.add_argument("--samples", type=int, default=0,
                    help="strong-scaling mode (BASELINE configs[3]): total number of x_t samples sharded over the ranks; steps = samples * 12")
    ap.add_argument("--ctx", default="null", choices=["null", "edit"], help="SD conditioning: seeded null-prompt or edit-prompt embedding")
    ap.add_argument("--flat-spectrum", action="store_true", help="unshaped random-init weights (round-1 workload; the stop rule does not converge)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-k", type=int, default=0, help="directions per CPU iteration (0 = all k)")
    ap.add_argument("--cpu-iters", type=int, default=2, help="timed CPU power iterations after the warm-up (BASELINE.md section 3: 2)")
    ap.add_argument("--repeats", type=int, default=7, help="the timed region of exactly --steps steps is run this many times; value = steps / MEDIAN time")
    ap.add_argument("--no-unet-forward", action="store_true", help="skip the DDIM-loop leg (full SD-1.5 U-Net forwards at B = 1 / 2 / 5)")
    ap.add_argument("--no-sd21-leg", action="store_true", help="skip the SD-2.1-base k = 2 leg (the setting of the reference's only published timing)")
    ap.add_argument("--no-strong-leg", action="store_true", help="skip the BASELINE configs[3] leg (64 samples, k = 10, edit ctx, sharded over the ranks)")
    ap.add_argument("--strong-samples", type=int, default=64)
    ap.add_argument("--profile-run", action="store_true", help="for rocprofv3 runs: headline region only, once (no repeats, no extra legs, no CPU baseline)")
    a = ap.parse_args()
    if a.profile_run:
        a.repeats, a.no_cpu_baseline, a.no_roofline, a.no_unet_forward, a.no_strong_leg, a.no_sd21_leg = 1, True, True, True, True, True
    return a


_PARAMS = {}


def sd_params(cfg_name, cfg, only_prefix, sp):
    """Seeded synthetic SD weights, drawn ONCE per (model, spectrum) and filtered by prefix per engine: the generator walks every tensor of the
    architecture whatever the prefix (so that a prefix-restricted set has the same bits as the full one), and one draw of 860 M values costs ~15 s
    of host time -- the legs of one bench run build six engines over the same two weight sets."""
    from diffusion_pullback_amd import configs as cf
    key = (cfg_name, sp)
    if key not in _PARAMS:
        _PARAMS[key] = cf.sd_init_params(cfg, seed=0, only_prefix=None, spectrum=sp)
    full = _PARAMS[key]
    return full if only_prefix is None else {n: v for n, v in full.items() if n.startswith(only_prefix)}


def make_workload(name, dtype, device, k, spg, tap=("mid", 0), ctx_kind="null", shaped=True):
    """-> (net, get_h_oracle, sample shape, t, ctx[1,L,D] or None, V0[k,N])"""
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import configs as cf
    g = torch.Generator().manual_seed(0)
    sp = cf.Spectrum() if shaped else None
    if name in ("sd15", "sd21", "toy"):
        toy = dict(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), up_attn=(False, True), heads=(2, 2), cross_dim=64,
                   sample_size=16, ctx_len=77)
        cfg = cf.SD15 if name == "sd15" else cf.sd_config_for("stabilityai/stable-diffusion-2-1-base") if name == "sd21" else cf.SDConfig(**toy)
        enc = ("time_embedding", "conv_in", "down_blocks", "mid_block") if tap[0] != "up" else None
        params = sd_params(name, cfg, enc, sp) if name != "toy" else cf.sd_init_params(cfg, seed=0, only_prefix=enc, spectrum=sp)
        net = PullbackUNet("sd", cfg, params, dtype=dtype, device=device, max_batch=spg, max_rank=k * spg, upto=tap, verbose=False)
        t = 696.2727
        ctx = torch.randn(1, cfg.ctx_len, cfg.cross_dim, generator=g)          # fixed seeded "null" embedding
        if ctx_kind == "edit":                                                   # a different seeded embedding standing for the edit prompt
            ctx = torch.randn(1, cfg.ctx_len, cfg.cross_dim, generator=torch.Generator().manual_seed(4242))
        shape = (cfg.in_channels, cfg.sample_size, cfg.sample_size)

        def oracle_get_h(zb):      # cpu_baseline leg only; the oracle walks the seeded values with its OWN config and shape table
            from oracle import unet_sd
            ocfg = unet_sd.SD15 if name == "sd15" else unet_sd.SD21_BASE if name == "sd21" else unet_sd.SDConfig(**toy)
            return unet_sd.forward(params, ocfg, zb, torch.tensor(t), ctx.expand(zb.shape[0], -1, -1), stop=tap)
    else:
        cfg = cf.CELEBA_HQ_256
        params = cf.ddpm_init_params(cfg, seed=0, spectrum=sp)
        net = PullbackUNet("ddpm", cfg, params, dtype=dtype, device=device, max_batch=spg, max_rank=k * spg, upto=tap, verbose=False)
        t, ctx = 600.0, None
        shape = (cfg.in_channels, cfg.resolution, cfg.resolution)

        def oracle_get_h(xb):      # cpu_baseline leg only
            from oracle import unet_ddpm
            return unet_ddpm.forward(params, cfg, xb, torch.tensor(t), stop=tap)
    n_in = shape[0] * shape[1] * shape[2]
    V0 = torch.linalg.qr(torch.randn(n_in, k, generator=g))[0].T.contiguous()
    return net, oracle_get_h, shape, t, ctx, V0


def workload_name(a, strong, tap):
    """config.workload: the BASELINE.json config this run measures (configs[4] = every --op down/up run, named with its tap)."""
    if a.workload == "ddpm256":
        base = "BASELINE configs[1]: CelebA-HQ DDPM 256x256 x[3,256,256], t=600"
        return base + (", mid-block h[512,8,8]" if tap == ("mid", 0) else f", tap {tap[0]}{tap[1]} (not a BASELINE config)")
    if a.workload == "toy":
        return "toy SD-style net (plumbing check)"
    if a.workload == "sd21":
        return (f"stabilityai/stable-diffusion-2-1-base (the reference scripts' default model, not a BASELINE config): 4x64x64 latent, seeded ctx[1,77,1024], "
                f"tap {tap[0]}_block_{tap[1]}, k={a.k}, t=696.27")
    if tap != ("mid", 0):
        return (f"BASELINE configs[4]: SD-v1.5 down/up-block sweep, tap {tap[0]}_block_{tap[1]}, 4x64x64 latent, seeded null ctx[1,77,768], t=696.27"
                + ("" if a.k == 5 else f" (k={a.k}: BASELINE names k=5)"))
    if strong or a.k == 10 or a.ctx == "edit":
        return ("BASELINE configs[3]: SD-v1.5 4x64x64 latents, edit-prompt ctx[1,77,768], mid-block, k=10, samples sharded over the GPUs"
                + ("" if (a.k == 10 and a.ctx == "edit") else f" (run with k={a.k}, ctx={a.ctx})"))
    return "BASELINE configs[2]: SD-v1.5 4x64x64 latent, no edit prompt (seeded null ctx[1,77,768]), mid-block h[1280,8,8], t=696.27"


def pmc_traffic(dom, enabled):
    """HBM bytes per launch of kernel `dom` from the committed PMC passes -- only if they were collected on THIS build of libdpb.so
    (the PMC file records the source hash of the library it profiled); otherwise None with the reason."""
    if not enabled:
        return None, "no PMC passes are committed for this workload (profiles/ holds them for the headline config only)"
    from diffusion_pullback_amd import lib as L
    cur = L._built_hash()
    cands = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic_sd15_mid_k5_bf16.json")), reverse=True)
    for f in cands:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
        except (OSError, ValueError):
            continue
        if d.get("_src_hash") != cur:
            continue
        ent = d["kernels"].get(dom)
        if ent and ent.get("write_kb_per_launch") is not None:
            return (2.0 * ent["fetch_kb_per_launch"] + ent["write_kb_per_launch"]) * 1024.0, \
                f"HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (profiles/{f}, same source hash {cur[:12]} as the running libdpb.so)"
    return None, (f"null: no committed PMC file was collected on this build (running libdpb.so source hash {cur[:12]}; candidates: {cands[:3]}) -- "
                  "re-run tools/pmc_mfma.sh + tools/collect_profiles.py after the last kernel change")


def trace_time_by_kind(enabled):
    """Kernel time per power iteration of every GEMM kernel kind from the committed rocprofv3 kernel trace of THIS build (tools/collect_profiles.py
    stores it next to the PMC traffic, under the same source-hash rule); None when no trace of this build is committed."""
    if not enabled:
        return None, "no kernel trace is committed for this workload"
    from diffusion_pullback_amd import lib as L
    cur = L._built_hash()
    for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic_sd15_mid_k5_bf16.json")), reverse=True):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
        except (OSError, ValueError):
            continue
        if d.get("_src_hash") == cur and d.get("_trace"):
            return d["_trace"]["by_kind"], f"rocprofv3 --kernel-trace --stats, {d['_trace']['source']} (same source hash {cur[:12]} as the running libdpb.so)"
    return None, f"null: no committed kernel trace was collected on this build ({cur[:12]}): the fraction falls back to the raw HIP-event figure"


def pmc_step_bytes(enabled):
    """HBM bytes of ONE whole power iteration summed over every kernel of the committed PMC passes (same source-hash rule as pmc_traffic)."""
    if not enabled:
        return None, "no PMC passes are committed for this workload"
    from diffusion_pullback_amd import lib as L
    cur = L._built_hash()
    for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic_sd15_mid_k5_bf16.json")), reverse=True):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
        except (OSError, ValueError):
            continue
        if d.get("_src_hash") == cur and d.get("_whole_step_hbm_bytes"):
            return float(d["_whole_step_hbm_bytes"]), (f"sum over all kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024 per iteration, profiles/{f} "
                                                       f"(same source hash {cur[:12]}); frac = bytes / ms_per_step / 8 TB/s")
    return None, f"null: no committed PMC file with a whole-step sum was collected on this build ({cur[:12]})"


def strong_leg(a, dev, dtype, dist, pdist, rank, world):
    """The whole BASELINE configs[3] job on `world` GPUs: T samples, k = 10, edit ctx, 12 iterations each; -> dict for the JSON line."""
    T, K3, S3 = a.strong_samples, 10, 8
    net3, _, shape, t, ctx, V0 = make_workload("sd15", dtype, dev, K3, S3, ("mid", 0), "edit", True)
    eng3 = net3.engine
    n_in, n_h = eng3.n_in, eng3.tap_numel(("mid", 0))
    mine = pdist.shard_indices(T, rank, world)
    xs = torch.stack([torch.randn(*shape, generator=torch.Generator().manual_seed(1000 + i)) for i in mine]).to(dev) if mine else None
    groups = [list(range(j, min(j + S3, len(mine)))) for j in range(0, len(mine), S3)]
    ctx_d = ctx.to(dev).expand(S3, -1, -1).contiguous()
    V0_d = V0.to(dev).repeat(S3, 1).contiguous()

    def job():
        out = {}
        for grp in groups:
            b = len(grp)
            eng3.primal(xs[grp[0]:grp[0] + b], t, ctx_d[:b], ("mid", 0))
            V, U, s, _ = eng3.iterate(("mid", 0), V0_d[:b * K3].clone(), ITERS_PER_SAMPLE)
            for j, li in enumerate(grp):
                out[mine[li]] = (U[j * K3:(j + 1) * K3].T, s[j * K3:(j + 1) * K3], V[j * K3:(j + 1) * K3])
        return out
    if groups:                                             # untimed: kernels of the k = 10 / 8-sample shapes, allocator
        eng3.primal(xs[:len(groups[0])], t, ctx_d[:len(groups[0])], ("mid", 0))
        eng3.iterate(("mid", 0), V0_d[:len(groups[0]) * K3].clone(), 2)
    if dist:                                               # untimed: RCCL channels for this packed size / these layouts
        zu, zs, zv = torch.zeros(K3, n_h, device=dev).T, torch.zeros(K3, device=dev), torch.zeros(K3, n_in, device=dev)
        pdist.gather_bases({i: (zu, zs, zv) for i in mine}, T, shape=(n_h, K3, n_in))
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    local = job()
    torch.cuda.synchronize(dev)
    tg = time.perf_counter()
    allres = pdist.gather_bases(local, T, shape=(n_h, K3, n_in)) if dist else local
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gms = 1e3 * (time.perf_counter() - tg)
    counts = [len(mine)]
    if dist:
        td = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = td.item()
        cl = [None] * world
        dist.all_gather_object(cl, len(mine))
        counts = cl
    assert len(allres) == T
    ok = all(bool(torch.isfinite(v[1]).all()) for v in allres.values())
    del net3, eng3
    torch.cuda.empty_cache()
    return {"workload": f"BASELINE configs[3]: {T} x_t samples in total, SD-v1.5 mid-block, k=10, edit-prompt ctx, 12 iterations each, {S3} advanced together per GPU",
            "value": T * ITERS_PER_SAMPLE / dt, "unit": "iters/s (sample-iterations, whole job, max over ranks)", "seconds": dt, "n_gpus": world, "scaling": "strong",
            "samples_per_rank": counts, "gather_and_barrier_ms": gms if dist else None, "finite": ok,
            "flops_per_sample_iteration": 2 * K3 * 2 * MAC_G["sd15"] * 1e9,
            "note": "strong scaling = this value at N GPUs / this value at 1 GPU (the driver's SCALE runs); the headline `value` above is the weak configs[2] rate"}


def unet_forward_leg(a, dev, dtype, dname, t, ctx, time_cpu):
    """Full SD-1.5 U-Net forwards (eps prediction) at B = 1 / 2 / 5 / 20 through dpb_forward (no stash), forwards/s and fraction of the MFMA peak
    (1: DDIM inversion / forward, 2: one x-space-guidance step, 5: the reference's decode chunk, 20: the CLI's decode of all edited latents in one call)."""
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import configs as cf
    cfg = cf.SD15
    tw = time.perf_counter()
    params = sd_params("sd15", cfg, None, cf.Spectrum())
    net = PullbackUNet("sd", cfg, params, dtype=dtype, device=dev, max_batch=20, max_rank=5, upto=None, verbose=False)
    build_s = time.perf_counter() - tw
    eng = net.engine
    g = torch.Generator().manual_seed(5)
    out = {"model": "SD-v1.5 UNet2DConditionModel, 859.5 M parameters, z[B,4,64,64], ctx[B,77,768], t=696.27, eps output", "dtype": dname,
           "mode": "dpb_forward: forward only, no tangent / adjoint stash", "engine_build_s": round(build_s, 1), "batches": {}}
    for B in (1, 2, 5, 20):
        z = torch.randn(B, 4, 64, 64, generator=g).to(dev)
        c = ctx.to(dev).expand(B, -1, -1).contiguous()
        for _ in range(3):
            eng.forward(z, t, c, "eps")
        fl = eng.stats()[1]                                                    # algorithmic GEMM + attention flops of one pass
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            e_ = eng.forward(z, t, c, "eps")
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / n
        # the same forward WITH the stash (dpb_primal + read), what the loop paid before dpb_forward existed (not at B = 20: the loop never stashed there)
        dp = float("nan")
        if B <= 5:
            for _ in range(2):
                eng.primal(z, t, c, "eps"); eng.read("eps")
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            for _ in range(n):
                eng.primal(z, t, c, "eps"); eng.read("eps")
            torch.cuda.synchronize(dev)
            dp = (time.perf_counter() - t1) / n
        out["batches"][str(B)] = {"ms_per_forward": 1e3 * dt, "forwards_per_s": 1.0 / dt, "samples_per_s": B / dt, "tflops": fl / dt / 1e12,
                                  "frac_of_mfma_peak": fl / dt / 1e12 / PEAK[dname], "flops_per_forward": fl, "launches": eng.stats()[0],
                                  "ms_per_forward_with_stash": (1e3 * dp if dp == dp else None), "finite": bool(torch.isfinite(e_).all())}
    out["x_space_guidance_steps_per_s"] = out["batches"]["2"]["forwards_per_s"]
    out["reference_note"] = ("reference Colab log (T4, SD-2.1-base fp32): 2.13 it/s for x-space guidance = one batch-2 U-Net forward + axpy per step "
                             "(example-code.ipynb:146-164; src/modules/edit.py:484-502) -- context, not a same-node comparison")
    if time_cpu:
        from oracle import unet_sd
        z1 = torch.randn(1, 4, 64, 64, generator=g)
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        with torch.no_grad():
            unet_sd.forward(params, unet_sd.SD15, z1, torch.tensor(t), ctx)
            tc = time.perf_counter()
            for _ in range(2):
                unet_sd.forward(params, unet_sd.SD15, z1, torch.tensor(t), ctx)
            tcpu = (time.perf_counter() - tc) / 2
        out["cpu_oracle_forward"] = {"ms_per_forward": 1e3 * tcpu, "threads": torch.get_num_threads(), "kind": "port (oracle/unet_sd.py, fp32)", "batch": 1}
        out["speedup_vs_cpu_b1"] = tcpu / (out["batches"]["1"]["ms_per_forward"] * 1e-3)
    del net, eng
    torch.cuda.empty_cache()
    return out


def sd21_leg(dev):
    """The reference's only published timing (example-code.ipynb:123-145): local_encoder_pullback_zt on stabilityai/stable-diffusion-2-1-base, mid block,
    pca_rank 2, fp32, 12 iterations, 14.31 s on a Colab T4.  The same call here (synthetic weights at the exact shapes): wall seconds of primal + 12 iterations
    through the reference-signature method (host loop, stop rule evaluated every iteration), fp32 and bf16 engines.  Context, not a same-node comparison."""
    out = {"workload": "stabilityai/stable-diffusion-2-1-base U-Net, z[1,4,64,64], ctx[1,77,1024], mid block, pca_rank 2, 12 iterations "
                       "(reference src/scripts/main_various_local_encoder_pullback_with_edit_prompt.sh:11; example-code.ipynb:123-145)",
           "reference_published": {"seconds": 14.31, "hardware": "Colab T4, fp32 (example-code.ipynb:145 'power method runtime')", "iterations": 12}}
    for dname in ("fp32", "bf16"):
        net, _, shape, t, ctx, V0 = make_workload("sd21", TORCH_DTYPE[dname], dev, 2, 1, ("mid", 0), "null", True)
        z = torch.randn(1, *shape, generator=torch.Generator().manual_seed(1000)).to(dev)
        kw = dict(sample=z, timestep=t, encoder_hidden_states=ctx, op="mid", block_idx=0, pca_rank=2, min_iter=10, max_iter=12, convergence_threshold=0.0, V0=V0)
        net.local_encoder_pullback_zt(**kw)                       # untimed: code objects, allocator
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        u, s_, vT = net.local_encoder_pullback_zt(**kw)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out["sd21_mid_k2_" + dname] = {"seconds": dt, "iterations": net.last_iters, "ms_per_iteration": 1e3 * dt / net.last_iters, "s": [round(v, 3) for v in s_.cpu().tolist()],
                                       "finite": bool(torch.isfinite(vT).all()), "speedup_vs_published_t4": 14.31 / dt}
        del net
        torch.cuda.empty_cache()
    return out


def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, logical, (len(phys) or logical)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:        # launched by torchrun (also at N = 1: the RCCL path runs at world_size 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from diffusion_pullback_amd import dist as pdist
    dname = a.dtype or ("fp32" if a.workload == "ddpm256" else "bf16")
    dtype = TORCH_DTYPE[dname]
    k = a.k
    S = a.samples_per_gpu
    strong = a.samples > 0
    tap = (a.op, a.block_idx)
    net, oracle_get_h, shape, t, ctx, V0 = make_workload(a.workload, dtype, dev, k, S, tap, a.ctx, not a.flat_spectrum)
    eng = net.engine
    n_in, n_h = eng.n_in, eng.tap_numel(tap)

    if strong:
        mine = pdist.shard_indices(a.samples, rank, world)          # sample i -> rank i mod world
        steps = a.samples * ITERS_PER_SAMPLE
        xs = torch.stack([torch.randn(*shape, generator=torch.Generator().manual_seed(1000 + i)) for i in mine]).to(dev) if mine else None
        groups = [list(range(j, min(j + S, len(mine)))) for j in range(0, len(mine), S)]     # ragged last group allowed
        n_samples = len(mine)
    else:
        assert a.steps % S == 0 and a.warmup % S == 0, "--steps and --warmup must be multiples of --samples-per-gpu"
        steps = a.steps
        group = S * ITERS_PER_SAMPLE                      # steps per group of S concurrently advanced samples
        n_groups = (a.steps + group - 1) // group
        n_warm = (a.warmup + group - 1) // group if a.warmup > 0 else 0
        n_samples = n_groups * S
        gx = torch.Generator().manual_seed(1000 + rank)
        xs = torch.randn(max(n_groups, n_warm, 1) * S, *shape, generator=gx).to(dev)   # synthetic latents, resident in HBM
    ctx_d = ctx.to(dev).expand(S, -1, -1).contiguous() if ctx is not None else None
    V0_d = V0.to(dev).repeat(S, 1).contiguous()           # same seeded V0 for every sample

    def run_weak(nsteps, xs_):
        done, gi, res = 0, 0, {}
        while done < nsteps:
            n = min(ITERS_PER_SAMPLE, (nsteps - done) // S)
            eng.primal(xs_[gi * S:(gi + 1) * S], t, ctx_d, tap)
            V, U, s, _ = eng.iterate(tap, V0_d.clone(), n)
            for j in range(S):                            # global sample id: this rank's gi-th group
                res[(gi * S + j) * world + rank] = (U[j * k:(j + 1) * k].T, s[j * k:(j + 1) * k], V[j * k:(j + 1) * k])
            done += n * S
            gi += 1
        return res

    def run_strong():
        res = {}
        for grp in groups:
            b = len(grp)
            eng.primal(xs[grp[0]:grp[0] + b], t, None if ctx_d is None else ctx_d[:b], tap)
            V, U, s, _ = eng.iterate(tap, V0_d[:b * k].clone(), ITERS_PER_SAMPLE)
            for j, li in enumerate(grp):
                res[mine[li]] = (U[j * k:(j + 1) * k].T, s[j * k:(j + 1) * k], V[j * k:(j + 1) * k])
        return res

    if a.warmup > 0:                                      # untimed: page in the kernels, weights and the allocator
        if strong:
            if groups:
                eng.primal(xs[:len(groups[0])], t, None if ctx_d is None else ctx_d[:len(groups[0])], tap)
                eng.iterate(tap, V0_d[:len(groups[0]) * k].clone(), min(ITERS_PER_SAMPLE, max(1, a.warmup)))
        else:
            run_weak(a.warmup, xs)
    if dist and a.warmup > 0:                             # untimed: RCCL sets up its channels on the first all_gather of a size class
        n_warm_total = a.samples if strong else n_samples * world      # the SAME packed size as the timed gather (a new size class sets up anew)
        # ... and the same tensor layouts (u is a transposed view: its packing copy is another kernel, loaded lazily on first use)
        zu, zs, zv = torch.zeros(k, n_h, device=dev).T, torch.zeros(k, device=dev), torch.zeros(k, n_in, device=dev)
        pdist.gather_bases({i: (zu, zs, zv) for i in pdist.shard_indices(n_warm_total, rank, world)}, n_warm_total, shape=(n_h, k, n_in))
    n_total = a.samples if strong else n_samples * world
    times, gathers = [], []
    for _ in range(max(1, a.repeats)):                    # every repeat is the SAME region of exactly `steps` steps (same latents, same V0)
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        local_res = run_strong() if strong else run_weak(steps, xs)
        if dist:   # final basis gather: the only collective of the path, ONE packed all_gather (RCCL over xGMI)
            torch.cuda.synchronize(dev)                   # (splits the timed region into compute | collective for the report; no extra cost:
            tg = time.perf_counter()                      #  the gather needs the finished bases anyway)
            allres = pdist.gather_bases(local_res, n_total, shape=(n_h, k, n_in))
        else:
            allres = local_res
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt_i = time.perf_counter() - t0
        if dist:
            gathers.append(1e3 * (time.perf_counter() - tg))
            td = torch.tensor([dt_i], device=dev, dtype=torch.float64)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)     # max over ranks, per repeat
            dt_i = td.item()
        times.append(dt_i)
    st_ = sorted(times)
    dt = st_[len(st_) // 2] if len(st_) % 2 else 0.5 * (st_[len(st_) // 2 - 1] + st_[len(st_) // 2])     # median over the repeats
    gather_ms = sorted(gathers)[len(gathers) // 2] if gathers else None
    assert len(allres) == n_total, (len(allres), n_total)
    s0 = allres[0][1]
    finite = all(bool(torch.isfinite(v[1]).all() and torch.isfinite(v[2]).all()) for v in allres.values())

    total_steps = steps if strong else world * steps
    res = {
        "metric": "pullback top-k SVD iters/sec (SD-v1.5 mid-block, 4x64x64)" if a.workload == "sd15" else "pullback top-k SVD iters/sec (SD-2.1-base mid-block, 4x64x64)" if a.workload == "sd21" else f"pullback top-k SVD iters/sec ({a.workload} mid-block)",
        "value": total_steps / dt, "unit": "iters/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / (steps if not strong else max(1, len(mine) * ITERS_PER_SAMPLE)), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": dname, "data": "synthetic (seeded random-init weights at the exact architecture shapes with a shaped spectrum, randn latents)",
        "config": {"workload": workload_name(a, strong, tap),
                   "tap": list(tap), "pca_rank": k, "ctx": a.ctx, "iters_per_sample": ITERS_PER_SAMPLE, "samples_total": n_total,
                   "samples_this_rank": n_samples, "samples_advanced_together": S, "spectrum": "flat" if a.flat_spectrum else "shaped (configs.Spectrum())",
                   "parallelism": (f"{world} ranks (RCCL world_size {world}), sample i on rank i mod {world}, one packed all_gather of (u,s,vT)" if dist
                                   else "single process, no collective"),
                   "rccl_world_size": world if dist else 0,
                   "gather_and_barrier_ms": gather_ms,           # part of the timed region: the packed all_gather + the closing barrier (this rank, median)
                   "repeats": len(times), "repeat_seconds": [round(x, 6) for x in times], "value_min": total_steps / max(times), "value_max": total_steps / min(times),
                   "timed_gpu_seconds_total": round(sum(times), 4), "value_is": "steps / median over the repeats of the max-over-ranks time of one `steps`-step region"},
        "finite": finite, "s_top": [round(v, 4) for v in s0.cpu().tolist()[:k]],
    }


    if a.workload == "sd15" and not strong and not a.no_strong_leg and tap == ("mid", 0) and not a.flat_spectrum:
        # ---- BASELINE configs[3] / north-star ">= 6x strong scaling at 8 GPUs" leg, in EVERY run (so `bench.py --gpus N` alone reports it): 64 seeded
        # x_t samples in total, edit-prompt ctx, k = 10; sample i on rank i mod N (dist.shard_indices), 8 advanced together, 12 iterations each, one
        # packed all_gather of the bases at the end (reference: one process per sample, src/scripts/main_celeba_hf_local_encoder_pullback.sh:2-9).
        # Not part of `value` (which stays the weak headline so that N = 1 agrees with BENCH).
        try:
            res["strong_scaling"] = strong_leg(a, dev, dtype, dist, pdist, rank, world)
        except Exception as ex:                              # never lose the headline line to the extra leg
            res["strong_scaling"] = {"error": repr(ex)[:300]}
            if dist:
                raise

    if rank == 0 and not a.no_roofline:
        # ---- roofline leg: one instrumented iteration, HIP events around every GEMM launch on the engine stream
        x1 = xs[0:S] if xs is not None and xs.shape[0] >= S else torch.randn(S, *shape).to(dev)
        eng.primal(x1, t, ctx_d, tap)
        eng.profile(True)
        eng.iterate(tap, V0_d.clone(), 1)
        if os.environ.get("DPB_PROFILE_CSV"):
            eng.profile_dump(os.environ["DPB_PROFILE_CSV"])
        tname = "float" if dname == "fp32" else dname.replace("fp16", "f16")
        kinds = {"gemm_kernel<%s,64,64,4>" % tname: eng.profile_read(0), "gemm_kernel<%s,128,128,4>" % tname: eng.profile_read(1),
                 "gemm_dma_kernel<128,128,3> / <256,128,3>": eng.profile_read(2), "gemm_dma_kernel<64,64,4>": eng.profile_read(3),
                 "gemm_ring64_kernel<128,128,2>": eng.profile_read(4), "conv_halo_kernel": eng.profile_read(5),
                 "gemm_ring64_kernel<256,256,2> (8 waves)": eng.profile_read(6), "gemm_p8_kernel (256x256, 8 waves, 8-phase)": eng.profile_read(11),
                 "gemm_wres_kernel (weights-resident streaming, K = 320)": eng.profile_read(12)}
        attn = {"attention forward (flash)": eng.profile_read(7), "attention tangent (attn_jvp_kernel)": eng.profile_read(8),
                "attention adjoint (query-major + key-major launches)": eng.profile_read(9), "cross-attention tangent / adjoint (attn_cross_kernel)": eng.profile_read(10)}
        ovh_ms = eng.profile_overhead_ms()
        KIND_OF = dict(zip(kinds, (0, 1, 2, 3, 4, 5, 6, 11, 12)))
        ATTN_OF = dict(zip(attn, (7, 8, 9, 10)))
        raw_ms = {n: eng.profile_read(1000 + c)[1] for n, c in list(KIND_OF.items()) + list(ATTN_OF.items())}   # unclamped raw bracket sums (kind + 1000)
        eng.profile(False)
        dom = max(kinds, key=lambda n: kinds[n][1])                      # dominant = most GPU time
        n_d, ms_c, fl_d = kinds[dom]                                     # ms_c: bracket times minus the calibrated empty bracket
        ms_d = raw_ms[dom]                                               # RAW event time (unclamped bracket sum): what the event-based fraction is computed from
        ach = fl_d / (ms_d * 1e-3) / 1e12 if ms_d > 0 else 0.0
        ach_corr = fl_d / (ms_c * 1e-3) / 1e12 if ms_c > 0 else 0.0
        mac = MAC_G.get(a.workload) if tap == ("mid", 0) else None
        gemm_ms = sum(v[1] for v in kinds.values())
        headline_cfg = a.workload == "sd15" and dname == "bf16" and S == 1 and k == 5 and tap == ("mid", 0)
        traffic, tnote = pmc_traffic(dom, headline_cfg)
        trace, trnote = trace_time_by_kind(headline_cfg)
        tr = trace.get(str(KIND_OF[dom])) if trace else None
        # `achieved` / `frac` are THIS run's measurement: algorithmic flops of the dominant kind's launches over their HIP-event bracket times minus the empty-bracket
        # time calibrated in this run (dpb_engine_profile_overhead).  The committed rocprofv3 kernel trace is a cross-check under its own keys (`*_trace`), and only when
        # it was taken on this build, with the same dispatch (launch count of the dominant kind agrees, no dispatch switch set in the environment).
        switches = sorted(v for v in os.environ if v.startswith("DPB_") and v not in ("DPB_PROFILE_CSV", "DPB_LIB"))
        ach_tr = frac_tr = None
        if not tr or tr["ms_per_iter"] <= 0:
            tr_why = trnote
        elif switches:
            tr_why = "not compared: dispatch switches set in the environment (%s)" % ", ".join(switches)
        elif abs(tr["launches_per_iter"] - n_d) > 0.05 * max(n_d, 1):      # (the trace averages over iterations AND the run's primal passes: ~2 % more launches)
            tr_why = "not compared: the committed trace has %.1f launches of this kind per iteration, this run %d" % (tr["launches_per_iter"], n_d)
        else:
            ach_tr = fl_d / (tr["ms_per_iter"] * 1e-3) / 1e12
            frac_tr, tr_why = ach_tr / PEAK[dname], trnote
        frac_main = ach_corr / PEAK[dname]
        frac_src = ("HIP events of this run around every launch of the kind on the engine stream, minus the empty-bracket time calibrated in this run "
                    "(raw brackets: frac_events_raw; committed rocprofv3 kernel trace of the same build: frac_trace)")
        step_bytes, snote = pmc_step_bytes(headline_cfg)
        ms_step = 1e3 * dt / (steps if not strong else max(1, len(mine) * ITERS_PER_SAMPLE))
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach_corr, "peak": PEAK[dname], "unit": "TFLOP/s", "frac": frac_main,
                           "frac_source": frac_src, "achieved_events_raw": ach, "frac_events_raw": ach / PEAK[dname],
                           "achieved_trace": ach_tr, "frac_trace": frac_tr, "trace_note": tr_why,
                           "trace_agrees_with_events": (abs(ach_tr - ach_corr) <= 0.1 * ach_tr) if ach_tr else None,
                           "trace_ms_per_iteration": tr["ms_per_iter"] if tr else None, "trace_launches_per_iteration": tr["launches_per_iter"] if tr else None,
                           "traffic": traffic, "traffic_note": tnote,
                           "launches_per_pass": n_d, "avg_launch_us": 1e3 * ms_d / max(n_d, 1), "flops_per_pass": fl_d,
                           "event_bracket_overhead_us": 1e3 * ovh_ms,
                           "achieved_bracket_corrected": ach_corr, "frac_bracket_corrected": ach_corr / PEAK[dname],
                           "avg_launch_us_bracket_corrected": 1e3 * ms_c / max(n_d, 1),
                           "timing_note": "`achieved` / `frac` (= `*_bracket_corrected`): algorithmic flops of the dominant kernel kind's launches over their HIP-event "
                                          "bracket times on the engine stream minus a calibrated empty-bracket time per launch (event_bracket_overhead_us), all measured in "
                                          "this run; `*_events_raw`: the same without the correction (an event pair costs about as much as a small kernel); `*_trace`: the "
                                          "same flops over the kind's kernel time in the committed rocprofv3 kernel trace of this build under profiles/ (cross-check only, "
                                          "null when build or dispatch differ: trace_note)",
                           "all_gemm_kernels": {n: {"launches": v[0], "avg_launch_us": 1e3 * raw_ms[n] / max(v[0], 1),
                                                    "achieved": v[2] / (raw_ms[n] * 1e-3) / 1e12 if v[0] > 0 else 0.0,
                                                    "achieved_bracket_corrected": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0} for n, v in kinds.items()},
                           "attention_kernels": {n: {"brackets": v[0], "avg_bracket_us": 1e3 * raw_ms[n] / max(v[0], 1), "algorithmic_flops": v[2],
                                                     "achieved": v[2] / (raw_ms[n] * 1e-3) / 1e12 if v[0] > 0 else 0.0,
                                                     "frac": v[2] / (raw_ms[n] * 1e-3) / 1e12 / PEAK[dname] if v[0] > 0 else 0.0} for n, v in attn.items()},
                           "attention_flops_note": "algorithmic L x L x d products per head: forward 2, tangent 5 per tangent, adjoint 7 per cotangent, cross-attention 2 "
                                                   "(L x 77 x d); the shared-probability kernels do fewer MFMAs than that (P computed once per sample)",
                           "whole_step_hbm_bytes": step_bytes, "whole_step_hbm_frac": (step_bytes / (ms_step * 1e-3) / 8e12) if step_bytes else None,
                           "whole_step_hbm_note": snote,
                           "gemm_time_share_of_step": gemm_ms / (ms_step * S),
                           "algorithmic_flops_per_step": 2 * k * 2 * mac * 1e9 if mac else None,
                           "whole_step_frac_of_peak": (2 * k * 2 * mac * 1e9 / (ms_step * 1e-3) / 1e12 / PEAK[dname]) if mac else None,
                           "launches_per_step": eng.stats()[0]}

    if rank == 0 and not a.no_roofline and S == 1:
        # ---- time to a converged basis under the reference's own stop rule (utils.py:803-808: allclose(V_prev, V, atol=1e-3) and
        # i > min_iter=10, at most 100 iterations), host loop with one 8-byte read-back per iteration; not part of `value`
        x1 = xs[0:1] if xs is not None and xs.shape[0] >= 1 else torch.randn(1, *shape).to(dev)
        torch.cuda.synchronize(dev); tc = time.perf_counter()
        if a.workload == "ddpm256":
            net.local_encoder_pullback_xt(x=x1, t=t, op=tap[0], block_idx=tap[1], pca_rank=k, V0=V0)
        else:
            net.local_encoder_pullback_zt(sample=x1, timestep=t, encoder_hidden_states=ctx, op=tap[0], block_idx=tap[1], pca_rank=k, V0=V0)
        torch.cuda.synchronize(dev)
        res["time_to_converged_basis"] = {"ms": 1e3 * (time.perf_counter() - tc), "iters": net.last_iters, "final_dist": net.last_dist,
                                          "converged": net.last_iters < 100,
                                          "rule": "reference stop rule: allclose(V_prev, V, atol=1e-3) and i > 10, max_iter 100"}

    if rank == 0 and not a.no_roofline and not strong and S == 1 and a.workload == "sd15":
        # ---- throughput with several x_t samples advanced together (what a multi-sample job such as BASELINE configs[3] runs per GPU): the
        # weight stream and every launch are shared by the samples.  Reported next to `value` (which stays the one-sample-at-a-time rate).
        SB = 4
        try:
            netb = make_workload(a.workload, dtype, dev, k, SB, tap, a.ctx, not a.flat_spectrum)[0]
            xb = torch.randn(SB, *shape, generator=torch.Generator().manual_seed(77)).to(dev)
            cb = ctx.to(dev).expand(SB, -1, -1).contiguous()
            Vb = V0.to(dev).repeat(SB, 1).contiguous()
            netb.engine.primal(xb, t, cb, tap); netb.engine.iterate(tap, Vb.clone(), 2)
            torch.cuda.synchronize(dev); tb = time.perf_counter()
            for _ in range(2):
                netb.engine.primal(xb, t, cb, tap); netb.engine.iterate(tap, Vb.clone(), ITERS_PER_SAMPLE)
            torch.cuda.synchronize(dev)
            res["batched_throughput"] = {"samples_advanced_together": SB, "value": 2 * SB * ITERS_PER_SAMPLE / (time.perf_counter() - tb), "unit": "iters/s",
                                         "note": "2 x (primal + 12 iterations) of 4 independent samples in one batch; not the headline value"}
            del netb
        except Exception as ex:                              # never lose the headline line to the extra leg
            res["batched_throughput"] = {"error": str(ex)[:200]}


    if rank == 0 and a.workload == "sd15" and not a.no_unet_forward and not strong and tap == ("mid", 0) and S == 1 and k == 5:
        # ---- DDIM / guidance-loop leg: the loop is one full U-Net forward per step (edit.py:454-458; x-space guidance: one
        # batch-2 forward per step, edit.py:484-502; the reference's Colab log shows 2.13 it/s for it on a T4, example-code.ipynb:146-164)
        try:
            res["unet_forward"] = unet_forward_leg(a, dev, dtype, dname, t, ctx, not a.no_cpu_baseline and world == 1)
        except Exception as ex:
            res["unet_forward"] = {"error": repr(ex)[:300]}

    if rank == 0 and a.workload == "sd15" and not a.no_sd21_leg and not strong and tap == ("mid", 0) and S == 1 and k == 5:
        try:
            res["sd21_reference_setting"] = sd21_leg(dev)
        except Exception as ex:
            res["sd21_reference_setting"] = {"error": repr(ex)[:300]}

    if rank == 0 and not a.no_roofline and not strong and S == 1 and a.workload == "sd15":
        # ---- two independent samples in flight on two HIP streams (two engines over the SAME weights, own workspaces): what a per-sample job
        # scheduler -- the reference launches one process per sample -- gets from one GPU when the one-sample pass leaves it latency-bound.
        # Reported next to `value` (one sample at a time, one stream), never instead of it.
        try:
            from diffusion_pullback_amd.engine import Engine
            cfgm = net.config
            eB = Engine(eng.tape, cfgm.block_out_channels[0], True, False, cfgm.in_channels, 1, k)
            sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            xa = torch.randn(2, *shape, generator=torch.Generator().manual_seed(91)).to(dev)
            c1 = ctx.to(dev)
            torch.cuda.synchronize(dev)

            def pair(n_it):
                Vs = []
                for e_, st_, j in ((eng, sA, 0), (eB, sB, 1)):
                    with torch.cuda.stream(st_):
                        e_.primal(xa[j:j + 1], t, c1, tap)
                        Vs.append(V0.to(dev).clone())
                for _ in range(n_it):
                    for e_, st_, j in ((eng, sA, 0), (eB, sB, 1)):
                        with torch.cuda.stream(st_):
                            e_.iterate(tap, Vs[j], 1)
                return Vs
            pair(2)
            torch.cuda.synchronize(dev); t2 = time.perf_counter()
            for _ in range(3):
                Vs = pair(ITERS_PER_SAMPLE)
            torch.cuda.synchronize(dev)
            res["two_stream_throughput"] = {"value": 3 * 2 * ITERS_PER_SAMPLE / (time.perf_counter() - t2), "unit": "iters/s", "streams": 2,
                                            "finite": bool(torch.isfinite(Vs[0]).all() and torch.isfinite(Vs[1]).all()),
                                            "note": "3 x (primal + 12 iterations) of 2 independent samples, one per HIP stream, launches interleaved per iteration; not the headline value"}
            del eB
        except Exception as ex:
            res["two_stream_throughput"] = {"error": repr(ex)[:200]}

    if rank == 0 and not a.no_cpu_baseline and world == 1:
        # ---- CPU baseline: the oracle (same jacfwd / functional.jacobian / svd calls as the reference), fp32, same
        # inputs, all physical cores, 1 warm-up + `cpu_iters` timed power iterations
        from oracle import pullback as opb
        ck = min(a.cpu_k, k) if a.cpu_k else k
        model, logical, physical = cpu_info()
        x_cpu = (xs[0:1] if xs is not None and xs.shape[0] >= 1 else torch.randn(1, *shape)).cpu()
        # thread count: the fastest of a short calibration (one get_h forward each) -- torch's CPU kernels stop scaling well below the
        # 128 cores of the GPU boxes' hosts, and the baseline should be the CPU path at its best, not at its most oversubscribed
        cand = sorted({c for c in (8, 16, 32, 64, physical) if 1 <= c <= logical})
        calib = {}
        chunk, variant = (5, "zt") if a.workload != "ddpm256" else (25, "xt")
        if "DPB_CPU_THREADS" in os.environ:
            cores = int(os.environ["DPB_CPU_THREADS"])
        else:
            # calibrated on what is timed -- ONE direction of JVP (jacfwd) + VJP (reverse mode) -- at every candidate count (no early stop:
            # round 3's first-slow-sample rule picked 8 threads on one box and 32 on the next); a count is skipped only once the calibration
            # itself has used up its budget
            v1 = V0[:1].reshape(1, *shape)
            t_cal = time.perf_counter()
            for c in cand:
                torch.set_num_threads(c)
                tq = time.perf_counter()
                u_c = opb.jvp_step(oracle_get_h, x_cpu, v1, 1, chunk, variant)
                opb.vjp_step(oracle_get_h, x_cpu, u_c)
                calib[c] = time.perf_counter() - tq
                if time.perf_counter() - t_cal > 60.0:
                    break
            cores = min(calib, key=calib.get)
        torch.set_num_threads(cores)

        def cpu_iteration(Vc):
            u_c = opb.jvp_step(oracle_get_h, x_cpu, Vc, Vc.shape[0], chunk, variant)
            w_c = opb.vjp_step(oracle_get_h, x_cpu, u_c)
            return torch.linalg.svd(w_c, full_matrices=False)[2].reshape(Vc.shape)
        tw0 = time.perf_counter()
        cpu_iteration(V0[:1].reshape(1, *shape))                       # warm-up: thread pool, allocator, one direction through every op
        tw = time.perf_counter() - tw0
        Vc = V0[:ck].reshape(ck, *shape)
        times = []
        for _ in range(max(1, a.cpu_iters)):
            tc0 = time.perf_counter()
            Vc = cpu_iteration(Vc)
            times.append(time.perf_counter() - tc0)
        tc = sum(times) / len(times)
        res["cpu_baseline"] = {"value": (ck / k) / tc, "unit": "iters/s", "cores": cores, "threads": cores, "kind": "port", "cpu_model": model,
                               "logical_cpus": logical, "physical_cores": physical,
                               "cores_note": "`cores` = `threads` = torch intra-op THREADS used (the fastest of the calibration table), not a core reservation",
                               "thread_calibration_s_per_jvp_vjp_direction": {str(c): round(v, 3) for c, v in calib.items()},
                               "sample": f"1 warm-up (1 direction, {tw:.1f}s) + {len(times)} timed power iterations (k={ck} of {k} directions; "
                                         f"JVP+VJP+SVD, fp32, {cores} threads) of the same workload: " + ", ".join(f"{x:.1f}s" for x in times)
                                         + ("" if ck == k else f", scaled by {ck}/{k}")}
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
