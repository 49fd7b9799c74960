"""Not a test: prints per-tap (primal, jvp, vjp) relative errors of the HIP engine vs the oracle.
Usage on the GPU box:  python tools/gpu_diag.py > gpurun_out/diag.txt 2>&1"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_parity import _toy_sd, _small_ddpm, check_passes
from diffusion_pullback_amd import PullbackUNet
from oracle import unet_sd, unet_ddpm


def run(name, fn):
    rep = []
    try:
        fn(rep)
        status = "OK"
    except AssertionError as e:
        status = "TOL"
    except Exception:
        status = "EXC\n" + traceback.format_exc()
    print(f"== {name}: {status}")
    for tap, e in rep:
        print(f"   {str(tap):14s} primal {e[0]:.3e}  jvp {e[1]:.3e}  vjp {e[2]:.3e}")
    sys.stdout.flush()


for dtype in (torch.float32, torch.bfloat16):
    f, cfg, p = _toy_sd()
    def sd(rep, dtype=dtype):
        net = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=2, max_rank=8, verbose=False)
        fwd = lambda a, tap: unet_sd.forward(p, cfg, a, f["t"], f["ctx"].expand(a.shape[0], -1, -1), stop=None if tap == "eps" else tap)
        check_passes(net, fwd, f["z"], float(f["t"]), f["ctx"], [("down", 0), ("down", 1), ("mid", 0), ("up", 0), ("up", 1), "eps"], dtype, report=rep)
    run(f"toy_sd {dtype}", sd)
    fd, cfgd, pd = _small_ddpm()
    def dd(rep, dtype=dtype):
        net = PullbackUNet("ddpm", cfgd, pd, dtype=dtype, device="cuda:0", max_batch=2, max_rank=8, verbose=False)
        fwd = lambda a, tap: unet_ddpm.forward(pd, cfgd, a, fd["t"], stop=None if tap == "eps" else tap)
        check_passes(net, fwd, fd["x"], float(fd["t"]), None, [("down", 0), ("down", 1), ("down", 2), ("mid", 0), ("up", 2), ("up", 1), ("up", 0), "eps"], dtype, report=rep)
    run(f"small_ddpm {dtype}", dd)
