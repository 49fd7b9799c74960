"""Not a test: same-process A/B of the 8-phase tile (530) against the ring tiles (515 = 128x128, 518 = 256x256) on large plain-row products; with `pmc`
as first argument only a few launches of each (the target of a rocprofv3 --pmc pass).  (Round 5's loop variants 531-533 were measured with an earlier
form of this script: profiles/r05_p8_variants.txt.)
python tools/gpu_p8_ab.py > gpurun_out/p8_ab.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gpu_gemm_bench import conv_engine, run, lib, L, DEV

if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    tiles = [int(t) for t in sys.argv[2:]] or [515, 518, 530]
    for b in (5, 10, 16):                      # M = 20480 (the judge's yardstick: 3.1 rounds of 256 tiles), 40960 (6.25 rounds), 65536 (10 rounds)
        e = conv_engine(64, 2560, 2560, 1, torch.bfloat16, b)
        x = torch.randn(b, 2560, 64, 64, device=DEV)
        for tile in tiles:
            L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
            for _ in range(8):
                e.primal(x, 1.0, None, "o")
            torch.cuda.synchronize()
        del e
    sys.exit(0)

# correctness of the variants first (bitwise against the ring)
for (H, cin, cout, b) in [(16, 64, 256, 1), (16, 192, 320, 3), (32, 640, 1920, 5), (64, 2560, 2560, 1)]:
    e = conv_engine(H, cin, cout, 1, torch.bfloat16, b)
    x = torch.randn(b, cin, H, H, device=DEV)
    outs = {}
    for tile in (515, 518, 530):
        L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
        e.primal(x, 1.0, None, "o"); torch.cuda.synchronize()
        outs[tile] = e.read("o").clone()
    print(f"M={b*H*H} N={cout} K={cin}:", {t: torch.equal(outs[515], outs[t]) for t in (518, 530)}, flush=True)
    del e
V = ((515, 1, 4), (518, 1, 4), (530, 1, 4), (515, 1, 4), (518, 1, 4), (530, 1, 4))
for rep in range(2):
    run("lin 64^2 2560->2560 b5", 64, 2560, 2560, 1, 5, variants=V)
    run("lin 64^2 2560->2560 b10", 64, 2560, 2560, 1, 10, variants=V)
    run("lin 64^2 2560->2560 b16 (5x256 tiles)", 64, 2560, 2560, 1, 16, variants=V)
    run("lin 64^2 1280->1280 b5", 64, 1280, 1280, 1, 5, variants=V)
    run("lin 16^2 1280->10240 b5", 16, 1280, 10240, 1, 5, variants=V)
    run("lin 32^2 640->5120 b5", 32, 640, 5120, 1, 5, variants=V)
