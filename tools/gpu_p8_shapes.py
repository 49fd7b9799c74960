"""Not a test: the dispatch WITH the 8-phase tile (dpb_debug_set("p8", 1), default) against the round-4 dispatch (p8 = 0: rings / halo kernel) and the
forced 8-phase tile on the product shapes of the path at many tangents (BASELINE configs[3]: 80 tangents; trajectory batches of the editing CLI),
plain rows and 3x3 convolutions.  One engine per shape, the three arms interleaved twice after a warm-up.  -> profiles/r05_p8_shapes.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gpu_gemm_bench import conv_engine, lib, L, DEV

KINDS = (0, 1, 2, 3, 4, 5, 6, 11)


def time_arm(e, x, p8, tile):
    L.check(lib.dpb_debug_set(b"p8", p8)); L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1 if tile else 0))
    for _ in range(3):
        e.primal(x, 1.0, None, "o")
    e.profile(True)
    for _ in range(10):
        e.primal(x, 1.0, None, "o")
    reads = [e.profile_read(k) for k in KINDS]
    e.profile(False)
    ms = sum(r[1] for r in reads) / 10
    kind = [k for k, r in zip(KINDS, reads) if r[0]]
    return ms * 1e3, kind


def shape(name, H, cin, cout, ks, b):
    e = conv_engine(H, cin, cout, ks, torch.bfloat16, b)
    x = torch.randn(b, cin, H, H, device=DEV)
    M, N, K = b * H * H, cout, ks * ks * cin
    arms = (("r04 dispatch", 0, 0), ("r05 dispatch", 1, 0), ("8-phase forced", 1, 530))
    best = {}
    for rep in range(3):
        for (nm, p8, tile) in arms:
            us, kind = time_arm(e, x, p8, tile)
            if rep:
                best[nm] = (min(best.get(nm, (1e9,))[0], us), kind)
    fl = 2.0 * M * N * K
    print(f"{name:30s} M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(f"{nm}: {v[0]:7.1f} us {fl / v[0] / 1e6:5.0f} TF kinds {v[1]}" for nm, v in best.items()), flush=True)
    L.check(lib.dpb_debug_set(b"p8", 1)); L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
    del e


for b in (5, 20, 80):
    shape(f"lin 64^2 320->1280 b{b}", 64, 320, 1280, 1, b)
    shape(f"lin 64^2 1280->320 b{b}", 64, 1280, 320, 1, b)
    shape(f"lin 64^2 320->2560 b{b}", 64, 320, 2560, 1, b)
    shape(f"lin 32^2 640->5120 b{b}", 32, 640, 5120, 1, b)
    shape(f"lin 32^2 2560->640 b{b}", 32, 2560, 640, 1, b)
    shape(f"lin 32^2 640->640 b{b}", 32, 640, 640, 1, b)
    shape(f"lin 32^2 640->1920 b{b}", 32, 640, 1920, 1, b)
    shape(f"lin 16^2 1280->10240 b{b}", 16, 1280, 10240, 1, b)
    shape(f"lin 16^2 5120->1280 b{b}", 16, 5120, 1280, 1, b)
    shape(f"lin 16^2 1280->1280 b{b}", 16, 1280, 1280, 1, b)
    shape(f"lin 16^2 1280->3840 b{b}", 16, 1280, 3840, 1, b)
    shape(f"lin 8^2 1280->10240 b{b}", 8, 1280, 10240, 1, b)
    shape(f"conv3x3 64^2 320->320 b{b}", 64, 320, 320, 3, b)
    shape(f"conv3x3 64^2 640->320 b{b}", 64, 640, 320, 3, b)
    shape(f"conv3x3 32^2 640->640 b{b}", 32, 640, 640, 3, b)
    shape(f"conv3x3 32^2 1280->640 b{b}", 32, 1280, 640, 3, b)
    shape(f"conv3x3 16^2 1280->1280 b{b}", 16, 1280, 1280, 3, b)
    shape(f"conv3x3 16^2 2560->1280 b{b}", 16, 2560, 1280, 3, b)
    shape(f"conv3x3 8^2 1280->1280 b{b}", 8, 1280, 1280, 3, b)
