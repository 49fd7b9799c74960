"""Not a test: leading singular values at a down / up tap of SD-1.5 for candidate Spectrum shapings (fp32, k = 8, 16 iterations) and whether the
fp16 engine stays finite there.   python tools/gpu_spectrum_tap.py up3 down1 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import PullbackUNet, configs as cf

DEV = "cuda:0"
ENC = ("time_embedding", "conv_in", "down_blocks", "mid_block")


def run(tap, sp, k=8, iters=16):
    params = cf.sd_init_params(cf.SD15, seed=0, only_prefix=ENC if tap[0] == "down" else None, spectrum=sp)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, 77, 768, generator=g); z = torch.randn(1, 4, 64, 64, generator=g)
    V0 = torch.linalg.qr(torch.randn(16384, k, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
    out = {}
    for dt in (torch.float32, torch.float16):
        net = PullbackUNet("sd", cf.SD15, params, dtype=dt, device=DEV, max_batch=1, max_rank=k, upto=tap, verbose=False)
        _, s, v, _ = net.pullback_fixed(z, 696.2727, ctx, tap[0], tap[1], k, iters, V0)
        out[dt] = (s.cpu(), v.cpu())
        del net
        torch.cuda.empty_cache()
    s32, s16 = out[torch.float32][0], out[torch.float16][0]
    a, b = out[torch.float32][1].double(), out[torch.float16][1].double()
    cos = ((a * b).sum(-1).abs() / (a.norm(dim=-1) * b.norm(dim=-1) + 1e-30)).tolist()
    print(f"{tap} {sp}\n  sigma fp32 {[round(x, 1) for x in s32.tolist()]}\n  ratios {[round((s32[i + 1] / s32[i]).item(), 3) for i in range(k - 1)]}\n"
          f"  sigma fp16 {[round(x, 1) for x in s16.tolist()]}  |cos| {[round(c, 4) for c in cos]}", flush=True)


if __name__ == "__main__":
    for name in sys.argv[1:] or ["up3"]:
        tap = (name[:-1], int(name[-1]))
        last = cf.Spectrum.for_tap(*tap).also
        cands = [cf.Spectrum(amp=100), cf.Spectrum(amp=40), cf.Spectrum(amp=100, also=last), cf.Spectrum(amp=40, also=last), cf.Spectrum(amp=100, decay=0.7, also=last)]
        for sp in cands:
            try:
                run(tap, sp)
            except Exception as ex:
                print(tap, sp, "FAILED", repr(ex)[:200], flush=True)
