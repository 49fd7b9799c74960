"""Not a test: micro-benchmark of the GEMM/implicit-GEMM kernel on the pullback path's layer shapes through 1-op tapes.
python tools/gpu_gemm_bench.py > gpurun_out/gemm_bench.txt"""
import os, sys, time
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
        for _ in range(10):
            e.primal(x, 1.0, None, "o")
        e.profile(True)
        for _ in range(10):
            e.primal(x, 1.0, None, "o")
        msb = sum(e.profile_read(kind)[1] for kind in (0, 1, 2, 3, 4, 5, 6, 11)); mss = 0.0
        e.profile(False)
        ms = (msb + mss) / 10
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            e.primal(x, 1.0, None, "o")
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e6      # whole op incl. split-K reduce and layout kernels
        out.append(f"t{tile}/s{sk}: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF (op {wall:6.1f}us)")
    L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0)); L.check(lib.dpb_debug_set(b"gemm_kch", 0))
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:6d} | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    V = ((0, 0, 4), (515, 4, 4), (515, 8, 4), (515, 16, 4), (600, 2, 4), (600, 4, 4), (600, 5, 4), (600, 10, 4))
    run("conv3x3 8^2 1280->1280 b5", 8, 1280, 1280, 3, 5, variants=V)
    run("conv3x3 16^2 1280->1280 b5", 16, 1280, 1280, 3, 5, variants=V)
    run("conv3x3 16^2 640->1280 b5", 16, 640, 1280, 3, 5, variants=V)
    run("lin 8^2 1280->1280 b5", 8, 1280, 1280, 1, 5, variants=V)
    run("lin 16^2 1280->1280 b5", 16, 1280, 1280, 1, 5, variants=V)
    run("lin 16^2 1280->10240 b5", 16, 1280, 10240, 1, 5, variants=V)
    run("lin 16^2 5120->1280 b5", 16, 5120, 1280, 1, 5, variants=V)
    V2 = ((0, 0, 4), (515, 1, 4), (600, 1, 4), (600, 2, 4), (600, 3, 4), (600, 5, 4))
    run("conv3x3 64^2 320->320 b5", 64, 320, 320, 3, 5, variants=V2)
    run("conv3x3 32^2 640->640 b5", 32, 640, 640, 3, 5, variants=V2)
    run("conv3x3 32^2 320->640 b5", 32, 320, 640, 3, 5, variants=V2)
    run("lin 64^2 320->320 b5", 64, 320, 320, 1, 5, variants=V2)
    run("lin 32^2 640->640 b5", 32, 640, 640, 1, 5, variants=V2)
    run("lin 64^2 320->2560 b5", 64, 320, 2560, 1, 5, variants=V2)
    run("lin 64^2 1280->320 b5", 64, 1280, 320, 1, 5, variants=V2)
    run("lin 32^2 640->5120 b5", 32, 640, 5120, 1, 5, variants=V2)
    run("lin 32^2 2560->640 b5", 32, 2560, 640, 1, 5, variants=V2)
    run("lin 64^2 2560->2560 b5", 64, 2560, 2560, 1, 5, variants=V2)
