"""Not a test: the 8-phase 256x256 tile (code 530) against the 128x128 ring (515): bitwise equality on plain-row and gathered products (ragged M / N,
odd and even K-tile counts, forced K splits), a multi-run race screen at three sizes, and timings of the yardstick products.
python tools/gpu_p8_check.py [quick] > gpurun_out/p8_check.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gpu_gemm_bench import conv_engine, run, lib, L, DEV

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
TILE = int(os.environ.get("P8_TILE", "530"))          # which 8-phase variant to check


def out_of(e, x, tile, sk):
    L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", sk))
    e.primal(x, 1.0, None, "o")
    torch.cuda.synchronize()
    return e.read("o").clone()


bad = 0
# (H, cin, cout, ks, batch, split)
CASES = [(16, 64, 256, 1, 1, 1), (16, 128, 256, 1, 1, 1), (16, 192, 320, 1, 1, 1), (16, 320, 320, 1, 3, 1), (8, 1280, 1280, 1, 5, 1), (32, 640, 640, 1, 5, 1),
         (64, 320, 320, 1, 2, 1), (16, 1280, 10240, 1, 5, 1), (16, 5120, 1280, 1, 5, 1), (32, 640, 1920, 1, 5, 3), (16, 1280, 1280, 1, 5, 4), (12, 72, 200, 1, 3, 1),
         (16, 320, 320, 3, 5, 1), (32, 640, 640, 3, 2, 1), (32, 320, 640, 3, 2, 2), (8, 1280, 1280, 3, 5, 5), (64, 320, 320, 3, 1, 1), (12, 64, 96, 3, 3, 1)]
for dt in (torch.bfloat16, torch.float16):
    for (H, cin, cout, ks, b, sk) in CASES:
        e = conv_engine(H, cin, cout, ks, dt, b)
        x = torch.randn(b, cin, H, H, device=DEV)
        ref = out_of(e, x, 515, sk)
        new = out_of(e, x, TILE, sk)
        eq = torch.equal(ref, new)
        md = (ref.float() - new.float()).abs().max().item()
        bad += not eq
        print(f"{str(dt)[6:]:8s} M={b*H*H:6d} N={cout:5d} K={ks*ks*cin:6d} ks={ks} split={sk}: bitwise_equal={eq} max|diff|={md:.3e} finite={torch.isfinite(new).all().item()}", flush=True)
        del e
        if quick and dt == torch.float16:
            break

# race screen: the same launch 30 times, every output compared with the first
for (H, cin, cout, ks, b) in [(16, 256, 256, 1, 1), (16, 512, 512, 1, 2), (64, 2560, 2560, 1, 1), (32, 640, 640, 3, 5)]:
    e = conv_engine(H, cin, cout, ks, torch.bfloat16, b)
    x = torch.randn(b, cin, H, H, device=DEV)
    first = out_of(e, x, TILE, 1)
    ref = out_of(e, x, 515, 1)
    diffs = 0
    for _ in range(10 if quick else 30):
        diffs += not torch.equal(out_of(e, x, TILE, 1), first)
    bad += diffs + (not torch.equal(first, ref))
    print(f"race screen M={b*H*H} N={cout} K={ks*ks*cin}: {diffs} runs differ from the first; first equals ring: {torch.equal(first, ref)}", flush=True)
    del e
L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
print("P8 CHECK", TILE, "FAILED" if bad else "OK", bad, flush=True)
if os.environ.get("P8_NOTIME"):
    sys.exit(1 if bad else 0)

V = ((515, 1, 4), (518, 1, 4), (530, 1, 4))
run("lin 64^2 2560->2560 b5", 64, 2560, 2560, 1, 5, variants=V)
run("lin 64^2 2560->2560 b10", 64, 2560, 2560, 1, 10, variants=V)
run("lin 64^2 1280->1280 b20", 64, 1280, 1280, 1, 20, variants=V)
run("lin 64^2 320->2560 b5", 64, 320, 2560, 1, 5, variants=V)
run("lin 64^2 1280->1280 b5", 64, 1280, 1280, 1, 5, variants=V)
run("lin 32^2 640->5120 b5", 32, 640, 5120, 1, 5, variants=V)
run("lin 32^2 5120->640 b5", 32, 5120, 640, 1, 5, variants=((515, 0, 4), (530, 1, 4), (530, 2, 4), (530, 3, 4)))
run("lin 16^2 1280->10240 b5", 16, 1280, 10240, 1, 5, variants=V)
VC = ((0, 0, 4), (515, 1, 4), (530, 1, 4), (530, 2, 4), (530, 3, 4))
run("conv3x3 64^2 320->320 b5", 64, 320, 320, 3, 5, variants=VC)
run("conv3x3 32^2 640->640 b5", 32, 640, 640, 3, 5, variants=VC)
run("conv3x3 16^2 1280->1280 b5", 16, 1280, 1280, 3, 5, variants=VC + ((530, 5, 4),))
run("conv3x3 64^2 320->320 b40", 64, 320, 320, 3, 40, variants=VC)
run("conv3x3 32^2 640->640 b40", 32, 640, 640, 3, 40, variants=VC)
run("conv3x3 16^2 1280->1280 b40", 16, 1280, 1280, 3, 40, variants=VC)
