"""Not a test: micro-benchmark of the GEMM/implicit-GEMM kernel on the pullback path's layer shapes through 1-op tapes.
python tests/gpu_gemm_bench.py > gpurun_out/gemm_bench.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import lib as L
from diffusion_pullback_amd.engine import Engine
from diffusion_pullback_amd.tape import Tape

DEV = "cuda:0"
lib = L.load()


def conv_engine(H, cin, cout, ks, dtype, batch):
    p = {"c.weight": torch.randn(cout, cin, ks, ks) * 0.02, "c.bias": torch.zeros(cout), "t.weight": torch.randn(8, 8) * 0.1}
    t = Tape(p, dtype, DEV)
    t.temb_in = t.buf(1, 8, L.BUF_SHARED)
    t.x = t.buf(H * H, cin)
    o = t.conv("c", t.x, (H, H), cout, ks=ks)
    t.tap("o", o, cout, H, H)
    return Engine(t, 8, False, True, cin, max_batch=batch, max_tangents=batch)


def run(name, H, cin, cout, ks, batch, dtype=torch.bfloat16, variants=((0, 0, 4), (64, 0, 4), (65, 0, 4))):
    e = conv_engine(H, cin, cout, ks, dtype, batch)
    x = torch.randn(batch, cin, H, H, device=DEV)
    M, N, K = batch * H * H, cout, ks * ks * cin
    out = []
    for tile, sk, kch in variants:
        L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", sk)); L.check(lib.dpb_debug_set(b"gemm_kch", kch))
        for _ in range(3):
            e.primal(x, 1.0, None, "o")
        e.profile(True)
        for _ in range(10):
            e.primal(x, 1.0, None, "o")
        msb = sum(e.profile_read(kind)[1] for kind in (0, 1, 2, 3)); mss = 0.0
        e.profile(False)
        ms = (msb + mss) / 10
        out.append(f"t{tile}/s{sk}/k{kch}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.0f}TF")
    L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0)); L.check(lib.dpb_debug_set(b"gemm_kch", 0))
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:6d} | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    run("conv3x3 64^2 320->320 b5", 64, 320, 320, 3, 5)
    run("conv3x3 32^2 640->640 b5", 32, 640, 640, 3, 5)
    run("conv3x3 16^2 1280->1280 b5", 16, 1280, 1280, 3, 5)
    run("conv3x3 8^2 1280->1280 b5", 8, 1280, 1280, 3, 5)
    run("lin 64^2 320->320 b5", 64, 320, 320, 1, 5)
    run("lin 32^2 640->640 b5", 32, 640, 640, 1, 5)
    run("lin 16^2 1280->1280 b5", 16, 1280, 1280, 1, 5)
    run("lin 64^2 320->2560 b5", 64, 320, 2560, 1, 5)
    run("lin 32^2 640->5120 b5", 32, 640, 5120, 1, 5)
    run("conv3x3 256^2 128->128 f32 b5", 256, 128, 128, 3, 5, torch.float32, ((64, 0, 4), (128, 1, 4)))
    run("conv3x3 64^2 256->256 f32 b5", 64, 256, 256, 3, 5, torch.float32, ((64, 0, 4), (128, 1, 4)))
