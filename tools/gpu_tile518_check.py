"""Not a test: the 256x256 ring tile (forced, code 518) against the 128x128 ring (515) on plain-row products of given sizes, bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gpu_gemm_bench import conv_engine, lib, L, DEV

for (H, cin, cout, b) in [(8, 1280, 10240, 40), (8, 1280, 5120, 40), (16, 1280, 1280, 40), (16, 1280, 3840, 40), (16, 5120, 1280, 40),
                          (32, 640, 1920, 40), (32, 640, 5120, 40), (32, 640, 2560, 40), (16, 1280, 10240, 40), (16, 1280, 5120, 40)]:
    e = conv_engine(H, cin, cout, 1, torch.bfloat16, b)
    x = torch.randn(b, cin, H, H, device=DEV)
    outs = {}
    for tile in (515, 518):
        L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
        e.primal(x, 1.0, None, "o")
        torch.cuda.synchronize()
        outs[tile] = e.read("o").clone()
    L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
    print(f"M={b*H*H} N={cout} K={cin}: equal={torch.equal(outs[515], outs[518])} finite={torch.isfinite(outs[518]).all().item()}", flush=True)
    del e
