"""Not a test: the A-stationary short-K kernel (forced, tile code 530) against the 128x128 BK=64 ring (515) on 1x1-convolution products of given sizes:
bitwise equality (same K16 MFMA order) and the micro-benchmark of both.   python tools/gpu_astat_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gpu_gemm_bench import conv_engine, run, lib, L, DEV

for (H, cin, cout, b, dt) in [(64, 320, 320, 5, torch.bfloat16), (64, 320, 960, 5, torch.bfloat16), (64, 320, 2560, 5, torch.float16), (40, 192, 640, 3, torch.bfloat16),
                              (8, 64, 320, 5, torch.float16), (24, 256, 320, 2, torch.bfloat16), (64, 320, 320, 40, torch.bfloat16)]:
    e = conv_engine(H, cin, cout, 1, dt, b)
    x = torch.randn(b, cin, H, H, device=DEV)
    outs = {}
    for tile in (515, 530):
        L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
        e.primal(x, 1.0, None, "o")
        torch.cuda.synchronize()
        outs[tile] = e.read("o").clone()
    L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
    d = (outs[515].float() - outs[530].float()).abs().max().item()
    print(f"M={b*H*H} N={cout} K={cin} {dt}: equal={torch.equal(outs[515], outs[530])} finite={torch.isfinite(outs[530]).all().item()} maxdiff={d:.3g} ref_absmax={outs[515].float().abs().max().item():.3g}", flush=True)
    del e
V = ((0, 0, 4), (515, 1, 4), (530, 1, 4))
run("lin 64^2 320->320 b5", 64, 320, 320, 1, 5, variants=V)
run("lin 64^2 320->960 b5", 64, 320, 960, 1, 5, variants=V)
run("lin 64^2 320->1280 b5", 64, 320, 1280, 1, 5, variants=V)
run("lin 64^2 320->2560 b5", 64, 320, 2560, 1, 5, variants=V)
run("lin 64^2 320->320 b40", 64, 320, 320, 1, 40, variants=V)
