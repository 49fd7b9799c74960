"""Not a test: full SD-1.5 U-Net forwards (eps) through dpb_forward, for rocprofv3 kernel-trace runs of the DDIM / guidance loop's step.
    python tools/gpu_unet_bench.py [bf16|fp16|fp32] [batch=2] [forwards=20]       (batch 2 = one x-space-guidance step, edit.py:484-502)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import PullbackUNet, configs as cf

dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
params = cf.sd_init_params(cf.SD15, seed=0, spectrum=cf.Spectrum())
net = PullbackUNet("sd", cf.SD15, params, dtype=dt, device="cuda:0", max_batch=max(B, 1), max_rank=1, verbose=False)
g = torch.Generator().manual_seed(0)
ctx = torch.randn(1, 77, 768, generator=g).cuda().expand(B, -1, -1).contiguous()
x = torch.randn(B, 4, 64, 64, generator=g).cuda()
for _ in range(3):
    net.engine.forward(x, 696.27, ctx, "eps")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    net.engine.forward(x, 696.27, ctx, "eps")
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
launches, flops, _ = net.engine.stats()
print(f"{dt} forward batch {B}: {ms:.2f} ms  ({1e3 / ms:.1f} forwards/s, {flops / ms / 1e9:.0f} TFLOP/s, {launches} launches)", flush=True)
