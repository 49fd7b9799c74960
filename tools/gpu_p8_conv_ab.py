"""Not a test: the 8-phase tile forced (tile 530) on the 3x3 convolutions of the path at many tangents -- run once per build of libdpb.so (DPB_LIB=...),
to compare the gather variants before / after a kernel change in one session.  -> one line per shape: us, TF/s."""
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
    return sum(r[1] for r in reads) / 10 * 1e3, [k for k, r in zip(KINDS, reads) if r[0]]


for (name, H, cin, cout, b) in (("conv3x3 16^2 1280->1280", 16, 1280, 1280, 80), ("conv3x3 16^2 2560->1280", 16, 2560, 1280, 80), ("conv3x3 32^2 640->640", 32, 640, 640, 20),
                                ("conv3x3 32^2 1280->640", 32, 1280, 640, 20), ("conv3x3 64^2 320->320", 64, 320, 320, 80), ("conv3x3 32^2 640->640", 32, 640, 640, 80)):
    e = conv_engine(H, cin, cout, 3, torch.bfloat16, b)
    x = torch.randn(b, cin, H, H, device=DEV)
    best = 1e9
    for rep in range(3):
        us, kind = time_arm(e, x, 1, 530)
        best = min(best, us)
    fl = 2.0 * b * H * H * cout * 9 * cin
    print(f"{os.environ.get('DPB_LIB', 'tree')[-28:]:28s} {name:26s} b{b:<3d} {best:8.1f} us {fl / best / 1e6:6.0f} TF/s kinds {kind}", flush=True)
    del e
