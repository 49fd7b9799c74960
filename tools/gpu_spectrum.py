"""Not a test: spectrum of the pullback Jacobian at full size for candidate Spectrum(...) shapings of the synthetic weights,
fp32 engine (k=14, 30 iterations) and bf16-vs-fp32 top-k agreement.  python tools/gpu_spectrum.py > gpurun_out/spectrum.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import PullbackUNet, configs as cf

DEV = "cuda:0"


def cosines(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a * b).sum(-1).abs() / (a.norm(dim=-1) * b.norm(dim=-1))).tolist()


def sd(sp, k=14, iters=30, k16=(5, 10)):
    enc = ("time_embedding", "conv_in", "down_blocks", "mid_block")
    params = cf.sd_init_params(cf.SD15, seed=0, only_prefix=enc, spectrum=sp)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, 77, 768, generator=g); z = torch.randn(1, 4, 64, 64, generator=g); t = 696.2727
    net32 = PullbackUNet("sd", cf.SD15, params, dtype=torch.float32, device=DEV, max_batch=1, max_rank=k, upto=("mid", 0), verbose=False)
    V0 = torch.linalg.qr(torch.randn(16384, k, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
    _, s32, v32, conv = net32.pullback_fixed(z, t, ctx, "mid", 0, k, iters, V0)
    h = net32.engine.read(("mid", 0))
    print(f"SD15 {sp}\n  sigma fp32 k={k}: {[round(x, 2) for x in s32.tolist()]}\n  ratios {[round((s32[i+1]/s32[i]).item(), 3) for i in range(k-1)]}  conv {conv.tolist()} |h|max {h.abs().max().item():.1f} rms {h.pow(2).mean().sqrt().item():.2f}", flush=True)
    del net32
    net16 = PullbackUNet("sd", cf.SD15, params, dtype=torch.bfloat16, device=DEV, max_batch=1, max_rank=max(k16), upto=("mid", 0), verbose=False)
    for kk in k16:
        net32 = PullbackUNet("sd", cf.SD15, params, dtype=torch.float32, device=DEV, max_batch=1, max_rank=kk, upto=("mid", 0), verbose=False)
        V0k = torch.linalg.qr(torch.randn(16384, kk, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
        _, sa, va, _ = net32.pullback_fixed(z, t, ctx, "mid", 0, kk, 12, V0k)
        _, sb, vb, _ = net16.pullback_fixed(z, t, ctx, "mid", 0, kk, 12, V0k)
        print(f"  k={kk} 12 iters: |cos| bf16 vs fp32 {[round(c, 4) for c in cosines(va, vb)]}  s rel {[round(x, 4) for x in ((sb - sa) / sa).tolist()]}", flush=True)
        u, s, vT = net16.local_encoder_pullback_zt(z, torch.tensor(t), ctx, op="mid", block_idx=0, pca_rank=kk, V0=V0k)
        print(f"  k={kk} reference stop rule (bf16): iters {net16.last_iters} dist {net16.last_dist:.2e}")
        u, s, vT = net32.local_encoder_pullback_zt(z, torch.tensor(t), ctx, op="mid", block_idx=0, pca_rank=kk, V0=V0k)
        print(f"  k={kk} reference stop rule (fp32): iters {net32.last_iters} dist {net32.last_dist:.2e}", flush=True)
        del net32


def ddpm(sp, k=8, iters=30):
    cfg = cf.CELEBA_HQ_256
    params = cf.ddpm_init_params(cfg, seed=0, spectrum=sp)
    net = PullbackUNet("ddpm", cfg, params, dtype=torch.float32, device=DEV, max_batch=1, max_rank=k, upto=("mid", 0), verbose=False)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 256, 256, generator=g)
    V0 = torch.linalg.qr(torch.randn(196608, k, generator=g))[0].T.contiguous()
    _, s, v, conv = net.pullback_fixed(x, 600.0, None, "mid", 0, k, iters, V0)
    print(f"DDPM256 {sp}\n  sigma fp32 k={k}: {[round(x, 2) for x in s.tolist()]}\n  ratios {[round((s[i+1]/s[i]).item(), 3) for i in range(k-1)]} conv {conv.tolist()}", flush=True)
    u, s, vT = net.local_encoder_pullback_xt(x, torch.tensor(600.0), op="mid", block_idx=0, pca_rank=5, V0=V0[:5])
    print(f"  k=5 reference stop rule (fp32): iters {net.last_iters} dist {net.last_dist:.2e}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["sd", "ddpm"]
    if "sd" in which:
        for sp in (cf.Spectrum(amp=100), cf.Spectrum(amp=400), cf.Spectrum(amp=1000)):
            sd(sp)
    if "ddpm" in which:
        for sp in (cf.Spectrum(amp=100), cf.Spectrum(amp=400)):
            ddpm(sp)
