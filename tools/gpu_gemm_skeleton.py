"""Not a test: true (kernel-trace) durations of short-K ring GEMMs for an ablation build (argv[1] = DPB_ABLATE bits, csrc `make ablate`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pullback_amd import lib as L
ab = int(sys.argv[1])
if ab: L.LIB_PATH = os.path.join(L.CSRC, "build", f"abl{ab}", "libdpb.so")
from gpu_gemm_bench import *
for (H, cin, cout, ks) in ((64, 320, 320, 1), (32, 640, 640, 1), (64, 320, 320, 3)):
    e = conv_engine(H, cin, cout, ks, torch.bfloat16, 5)
    x = torch.randn(5, cin, H, H, device=DEV)
    L.check(lib.dpb_debug_set(b"gemm_tile", 515)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1))
    for _ in range(8): e.primal(x, 1.0, None, "o")
torch.cuda.synchronize()
