#!/usr/bin/env python
"""Headline benchmark: pullback top-k SVD power iterations / second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A *step* is one power iteration for k directions on one x_t sample: k JVPs + k VJPs through the
U-Net prefix up to the tap + the k x N re-orthonormalisation (reference src/utils/utils.py:756-808).
A sample's job is one primal (stash) pass + ITERS_PER_SAMPLE=12 iterations (= min_iter+2, the
reference's minimum and what its Colab log ran); the primal pass of every sample started inside the
timed region is timed too.  Inputs (x_t, ctx, V0) are resident in HBM before the clock starts.

N>1: one process per GPU (torchrun), independent samples per rank (weak scaling, no data-path
collective), one RCCL all_gather of the final bases (u, s, vT) inside the timed region.

Output: ONE JSON line on rank 0 with `roofline` (dominant kernel = the 128x128 MFMA GEMM, measured live
with HIP events around every launch in a separate instrumented iteration) and `cpu_baseline` (the CPU
oracle, i.e. the reference's algorithm with the same autodiff calls, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS_PER_SAMPLE = 12
PEAK = {"bf16": 2500.0, "fp32": 157.3}      # TFLOP/s dense MFMA (MI355X_MICROARCH.md)
MAC_G = {"sd15": 130.72, "ddpm256": 67.58}  # GMAC of one get_h(mid) forward


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--workload", default="sd15", choices=["sd15", "ddpm256", "toy"])
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"])
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--op", default="mid", choices=["down", "mid", "up"], help="feature tap (BASELINE config 5 sweeps down/up 0..3)")
    ap.add_argument("--block-idx", type=int, default=0)
    ap.add_argument("--samples-per-gpu", type=int, default=1,
                    help="x_t samples advanced together per GPU (independent bases, shared weight stream); steps must be a multiple")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-k", type=int, default=0, help="directions in the bounded CPU sample (0 = all k: one full power iteration, ~20 s)")
    return ap.parse_args()


def make_workload(name, dtype, device, k, spg, tap=("mid", 0)):
    """-> (net, get_h_oracle, samples x [S,...], t, ctx, V0[k,N], tap)"""
    from diffusion_pullback_amd import PullbackUNet
    g = torch.Generator().manual_seed(0)
    if name == "sd15" or name == "toy":
        from diffusion_pullback_amd import configs as cf
        cfg = cf.SD15 if name == "sd15" else cf.SDConfig(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False),
                                                                  up_attn=(False, True), heads=(2, 2), cross_dim=64, sample_size=16, ctx_len=77)
        enc = ("time_embedding", "conv_in", "down_blocks", "mid_block") if tap[0] != "up" else None
        params = cf.sd_init_params(cfg, seed=0, only_prefix=enc)
        net = PullbackUNet("sd", cfg, params, dtype=dtype, device=device, max_batch=spg, max_rank=k * spg, upto=tap, verbose=False)
        t = 696.2727
        ctx = torch.randn(1, cfg.ctx_len, cfg.cross_dim, generator=g)          # fixed seeded "null" embedding
        shape = (cfg.in_channels, cfg.sample_size, cfg.sample_size)
        def oracle_get_h(zb):      # cpu_baseline leg only
            from oracle import unet_sd
            return unet_sd.forward(params, cfg, zb, torch.tensor(t), ctx.expand(zb.shape[0], -1, -1), stop=tap)
    else:
        from diffusion_pullback_amd import configs as cf
        cfg = cf.CELEBA_HQ_256
        params = cf.ddpm_init_params(cfg, seed=0)
        net = PullbackUNet("ddpm", cfg, params, dtype=dtype, device=device, max_batch=spg, max_rank=k * spg, upto=tap, verbose=False)
        t, ctx = 600.0, None
        shape = (cfg.in_channels, cfg.resolution, cfg.resolution)
        def oracle_get_h(xb):      # cpu_baseline leg only
            from oracle import unet_ddpm
            return unet_ddpm.forward(params, cfg, xb, torch.tensor(t), stop=tap)
    n_in = shape[0] * shape[1] * shape[2]
    V0 = torch.linalg.qr(torch.randn(n_in, k, generator=g))[0].T.contiguous()
    return net, oracle_get_h, shape, t, ctx, V0


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    dname = a.dtype or ("fp32" if a.workload == "ddpm256" else "bf16")
    dtype = torch.float32 if dname == "fp32" else torch.bfloat16
    k = a.k
    S = a.samples_per_gpu
    assert a.steps % S == 0 and a.warmup % S == 0, "--steps and --warmup must be multiples of --samples-per-gpu"
    tap = (a.op, a.block_idx)
    net, oracle_get_h, shape, t, ctx, V0 = make_workload(a.workload, dtype, dev, k, S, tap)
    eng = net.engine

    group = S * ITERS_PER_SAMPLE                      # steps per group of S concurrently advanced samples
    n_groups = (a.steps + group - 1) // group
    n_warm = (a.warmup + group - 1) // group if a.warmup > 0 else 0
    n_samples = n_groups * S
    gx = torch.Generator().manual_seed(1000 + rank)
    xs = torch.randn(max(n_groups, n_warm, 1) * S, *shape, generator=gx).to(dev)   # synthetic latents, resident in HBM
    ctx_d = ctx.to(dev).expand(S, -1, -1).contiguous() if ctx is not None else None
    V0_d = V0.to(dev).repeat(S, 1).contiguous()           # same seeded V0 for every sample

    def run(steps, xs_):
        out = None
        done = 0
        gi = 0
        while done < steps:
            n = min(ITERS_PER_SAMPLE, (steps - done) // S)
            eng.primal(xs_[gi * S:(gi + 1) * S], t, ctx_d, tap)
            V = V0_d.clone()
            out = eng.iterate(tap, V, n)
            done += n * S
            gi += 1
        return out

    if a.warmup > 0:
        run(a.warmup, xs)
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    V, U, s, conv = run(a.steps, xs)
    if dist:   # final basis gather (the only collective of the path)
        for ten in (U, s, V):
            buf = [torch.empty_like(ten) for _ in range(world)]
            dist.all_gather(buf, ten.contiguous())
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist:
        td = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = td.item()
    finite = bool(torch.isfinite(s).all() and torch.isfinite(V).all())

    res = {
        "metric": "pullback top-k SVD iters/sec (SD-v1.5 mid-block, 4x64x64)" if a.workload == "sd15" else f"pullback top-k SVD iters/sec ({a.workload} mid-block)",
        "value": world * a.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dname, "data": "synthetic (seeded random-init weights at the exact architecture shapes, randn latents)",
        "config": {"workload": {"sd15": "BASELINE configs[2]: SD-v1.5 4x64x64 latent, no edit prompt (seeded null ctx[1,77,768]), mid-block h[1280,8,8], t=696.27",
                                "ddpm256": "BASELINE configs[1]: CelebA-HQ DDPM 256x256 x[3,256,256], mid-block h[512,8,8], t=600",
                                "toy": "toy SD-style net (plumbing check)"}[a.workload],
                   "tap": list(tap), "pca_rank": k, "iters_per_sample": ITERS_PER_SAMPLE, "samples_timed_per_gpu": n_samples, "samples_advanced_together": S,
                   "parallelism": f"{world} x independent samples, final all_gather of (u,s,vT)" if world > 1 else "single GPU"},
        "finite": finite, "s_top": [round(v, 5) for v in s.cpu().tolist()[:k]],
    }

    if rank == 0 and not a.no_roofline:
        # ---- roofline leg: one instrumented iteration, HIP events around every GEMM launch on the engine stream
        eng.primal(xs[0:S], t, ctx_d, tap)
        eng.profile(True)
        eng.iterate(tap, V0_d.clone(), 1)
        if os.environ.get("DPB_PROFILE_CSV"):
            eng.profile_dump(os.environ["DPB_PROFILE_CSV"])
        tname = "float" if dname == "fp32" else "bf16"
        kinds = {"gemm_kernel<%s,64,64,4>" % tname: eng.profile_read(0), "gemm_kernel<%s,128,128,4>" % tname: eng.profile_read(1),
                 "gemm_dma_kernel<128,128,3> / <256,128,3>": eng.profile_read(2), "gemm_dma_kernel<64,64,4>": eng.profile_read(3),
                 "gemm_ring64_kernel<128,128,2>": eng.profile_read(4), "conv_halo_kernel": eng.profile_read(5)}
        eng.profile(False)
        dom = max(kinds, key=lambda n: kinds[n][1])                      # dominant = most GPU time
        n_d, ms_d, fl_d = kinds[dom]
        ach = fl_d / (ms_d * 1e-3) / 1e12 if ms_d > 0 else 0.0
        mac = MAC_G.get(a.workload) if tap == ("mid", 0) else None
        gemm_ms = sum(v[1] for v in kinds.values())
        traffic = None                                                  # HBM bytes per launch of the dominant kernel from the committed PMC passes
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic_sd15_mid_k5_bf16.json")
        if a.workload == "sd15" and dname == "bf16" and S == 1 and os.path.exists(pmc):
            ent = json.load(open(pmc))["kernels"].get(dom)
            if ent and ent.get("write_kb_per_launch") is not None:
                traffic = (2.0 * ent["fetch_kb_per_launch"] + ent["write_kb_per_launch"]) * 1024.0
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK[dname], "unit": "TFLOP/s", "frac": ach / PEAK[dname],
                           "traffic": traffic, "traffic_note": "HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (profiles/r01_pmc_traffic_*.json)", "launches_per_pass": n_d, "avg_launch_us": 1e3 * ms_d / max(n_d, 1), "flops_per_pass": fl_d,
                           "all_gemm_kernels": {n: {"launches": v[0], "avg_launch_us": 1e3 * v[1] / max(v[0], 1),
                                                    "achieved": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0} for n, v in kinds.items()},
                           "gemm_time_share_of_step": gemm_ms / (res["ms_per_step"] * S),
                           "algorithmic_flops_per_step": 2 * k * 2 * mac * 1e9 if mac else None,
                           "whole_step_frac_of_peak": (2 * k * 2 * mac * 1e9 / (res["ms_per_step"] * 1e-3) / 1e12 / PEAK[dname]) if mac else None}

    if rank == 0 and not a.no_roofline and S == 1:
        # ---- time to a converged basis under the reference's own stop rule (utils.py:803-808: allclose(V_prev, V, atol=1e-3) and
        # i > min_iter=10, at most 100 iterations), host loop with one 8-byte read-back per iteration; not part of `value`
        torch.cuda.synchronize(dev); tc = time.perf_counter()
        if a.workload == "ddpm256":
            net.local_encoder_pullback_xt(x=xs[0:1], t=t, op=tap[0], block_idx=tap[1], pca_rank=k, V0=V0)
        else:
            net.local_encoder_pullback_zt(sample=xs[0:1], timestep=t, encoder_hidden_states=ctx, op=tap[0], block_idx=tap[1], pca_rank=k, V0=V0)
        torch.cuda.synchronize(dev)
        res["time_to_converged_basis"] = {"ms": 1e3 * (time.perf_counter() - tc), "iters": net.last_iters, "final_dist": net.last_dist,
                                          "rule": "reference stop rule: allclose(V_prev, V, atol=1e-3) and i > 10, max_iter 100"}

    if rank == 0 and not a.no_cpu_baseline and world == 1:
        # ---- CPU baseline: the oracle (same jacfwd / functional.jacobian / svd calls as the reference) on the host cores
        from oracle import pullback as opb
        ck = min(a.cpu_k, k) if a.cpu_k else k
        cores = min(os.cpu_count() or 1, 32)      # torch CPU kernels stop scaling (and oversubscribe) far below 256 threads
        torch.set_num_threads(cores)
        x_cpu = xs[0:1].cpu()
        Vc = V0[:ck].reshape(ck, *shape)
        tc0 = time.perf_counter()
        u_c = opb.jvp_step(oracle_get_h, x_cpu, Vc, ck, 5 if a.workload != "ddpm256" else 25, "zt" if a.workload != "ddpm256" else "xt")
        w_c = opb.vjp_step(oracle_get_h, x_cpu, u_c)
        torch.linalg.svd(w_c, full_matrices=False)
        tc = time.perf_counter() - tc0
        res["cpu_baseline"] = {"value": (ck / k) / tc, "unit": "iters/s", "cores": cores, "kind": "port",
                               "sample": f"1 power iteration (k={ck} of {k} directions; JVP+VJP+SVD, fp32) of the same workload, no warm-up, {tc:.1f}s"
                                         + ("" if ck == k else f", scaled by {ck}/{k}")}
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
