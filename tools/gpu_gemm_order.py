"""Not a test: one weight-heavy conv (16^2, 1280->1280, 5 tangents) under a forced block order (argv[1]: 0 A-major, 1 B-major),
for `rocprofv3 --pmc FETCH_SIZE` comparisons of the HBM read traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_gemm_bench import *  # noqa

order = int(sys.argv[1]) if len(sys.argv) > 1 else -1
H, cin, cout = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16, 1280, 1280)
e = conv_engine(H, cin, cout, 3, torch.bfloat16, 5)
x = torch.randn(5, cin, H, H, device=DEV)
L.check(lib.dpb_debug_set(b"gemm_order", order))
for _ in range(6):
    e.primal(x, 1.0, None, "o")
torch.cuda.synchronize()
