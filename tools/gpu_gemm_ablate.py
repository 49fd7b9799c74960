"""Not a test: ablation of the LDS-DMA ring GEMM (which side bounds the K loop?).  Each variant is a separate build of
gemm_dma.hip with -DDPB_ABLATE=bits (1 = no MFMA, 2 = no DMA refills, 4 = no LDS fragment reads) linked into
csrc/build/abl<bits>/libdpb.so;  run as  `for ab in 0 1 2 3 4 5 6 7; do python tools/gpu_gemm_ablate.py $ab; done`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pullback_amd import lib as L
ab = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if ab:
    L.LIB_PATH = os.path.join(L.CSRC, "build", f"abl{ab}", "libdpb.so")
from gpu_gemm_bench import *  # noqa


def run_ab(name, H, cin, cout, ks, batch, tiles=(515,)):
    e = conv_engine(H, cin, cout, ks, torch.bfloat16, batch)
    x = torch.randn(batch, cin, H, H, device=DEV)
    M, N, K = batch * H * H, cout, ks * ks * cin
    out = []
    for tile in tiles:
        L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
        for _ in range(3):
            e.primal(x, 1.0, None, "o")
        e.profile(True)
        for _ in range(10):
            e.primal(x, 1.0, None, "o")
        ms = sum(e.profile_read(kind)[1] for kind in (0, 1, 2, 3, 4, 5, 6)) / 10
        e.profile(False)
        out.append(f"t{tile}:{ms*1e3:6.1f}us")
    L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
    print(f"ab{ab} {name:26s} M={M:6d} N={N:5d} K={K:6d} " + " ".join(out) + f"  (full = {2*M*N*K/1e6:.0f} MF)", flush=True)


if __name__ == "__main__":
    run_ab("conv3x3 64^2 320->320 b5", 64, 320, 320, 3, 5)
    run_ab("conv3x3 32^2 640->640 b5", 32, 640, 640, 3, 5)
    run_ab("lin 64^2 320->320 b5", 64, 320, 320, 1, 5)
    run_ab("lin 64^2 320->2560 b5", 64, 320, 2560, 1, 5)
    run_ab("lin 64^2 2560->2560 b5", 64, 2560, 2560, 1, 5)
