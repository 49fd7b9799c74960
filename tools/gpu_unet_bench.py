"""Not a test: times the primal passes (get_h to mid, full eps forward at batch 1/2/5) of the SD-1.5-shaped engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import PullbackUNet, configs as cf

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
params = cf.sd_init_params(cf.SD15, seed=0)
net = PullbackUNet("sd", cf.SD15, params, dtype=dt, device="cuda:0", max_batch=5, max_rank=5, verbose=False)
g = torch.Generator().manual_seed(0)
ctx = torch.randn(1, 77, 768, generator=g).cuda()
for tap, B in ((("mid", 0), 1), ("eps", 1), ("eps", 2), ("eps", 5)):
    x = torch.randn(B, 4, 64, 64, generator=g).cuda()
    c = ctx.expand(B, -1, -1).contiguous()
    for _ in range(3):
        net.engine.primal(x, 696.27, c, tap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        net.engine.primal(x, 696.27, c, tap)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{dt} primal {tap} batch {B}: {ms:.2f} ms  ({B / ms * 1e3:.1f} samples/s)", flush=True)
