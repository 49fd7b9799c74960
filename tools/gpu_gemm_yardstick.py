"""Not a test: the 20480 x 2560 x 2560 plain-row product on the 128x128 (515) and 256x256 (518) ring tiles, a few launches each -- the
target of a `rocprofv3 --pmc ... --kernel-trace` pass (profiles/r02_pmc_gemm_yardstick.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gpu_gemm_bench import conv_engine, lib, L, DEV

e = conv_engine(64, 2560, 2560, 1, torch.bfloat16, 5)
x = torch.randn(5, 2560, 64, 64, device=DEV)
for tile in (515, 518):
    L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
    for _ in range(6):
        e.primal(x, 1.0, None, "o")
    torch.cuda.synchronize()
