"""Not a test: the weights-resident streaming kernel (gemm_wres.hip, tile 540) against the round-5 dispatch (dpb_debug_set("wres", 0): BK = 64 rings / 8-phase
tile) on the K = 320 linear layers of the 64 x 64 level, TANGENT and ADJOINT products (plain epilogue, with and without a row operand), 1 .. 80 tangents.
Event brackets around the product launches only (dpb_engine_profile), bracket-corrected, best of 3 x 10.  -> profiles/r06_wres_shapes.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import lib as L
from diffusion_pullback_amd.engine import Engine
from diffusion_pullback_amd.tape import Tape

DEV = "cuda:0"
lib = L.load()
KINDS = (0, 1, 2, 3, 4, 5, 6, 11, 12)


def engine(H, cin, cout, batch, res):
    p = {"c.weight": torch.randn(cout, cin, 1, 1) * 0.02, "c.bias": torch.zeros(cout)}
    t = Tape(p, torch.bfloat16, DEV)
    t.temb_in = t.buf(1, 8, L.BUF_SHARED)
    t.x = t.buf(H * H, cin)
    o = t.conv("c", t.x, (H, H), cout, ks=1, stride=1, pad=0, res=t.x if res else -1)
    t.tap("o", o, cout, H, H)
    return Engine(t, 8, False, True, cin, max_batch=batch, max_tangents=batch)


def bracket_us(e, fn):
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        e.profile(True)
        for _ in range(10):
            fn()
        reads = [e.profile_read(k) for k in KINDS]
        e.profile(False)
        best = min(best, sum(r[1] for r in reads) / 10 * 1e3)
        kinds = [k for k, r in zip(KINDS, reads) if r[0]]
    return best, kinds


for cout in (320, 960):
    for res in (False, True):
        for batch in (1, 2, 5, 10, 20, 80):
            H = 64
            e = engine(H, 320, cout, batch, res)
            x = torch.randn(batch, 320, H, H, device=DEV)
            V = torch.randn(batch, 320 * H * H, device=DEV)
            U = torch.randn(batch, cout * H * H, device=DEV)
            e.primal(x, 1.0, None, "o")
            row = []
            for name, fn in (("jvp", lambda: e.jvp("o", V)), ("vjp", lambda: e.vjp("o", U))):
                if name == "vjp" and cout != 320:
                    continue                      # the adjoint of a 320 -> 960 layer has K = 960: not this kernel's
                res_ = {}
                for arm, (wres, tile) in (("r05 dispatch", (0, 0)), ("wres", (1, 540))):
                    L.check(lib.dpb_debug_set(b"wres", wres)); L.check(lib.dpb_debug_set(b"gemm_tile", tile)); L.check(lib.dpb_debug_set(b"gemm_splitk", 1 if tile else 0))
                    res_[arm] = bracket_us(e, fn)
                L.check(lib.dpb_debug_set(b"wres", 1)); L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
                a, b = res_["r05 dispatch"], res_["wres"]
                M = batch * H * H
                gb = (M * 320 + M * cout * (2 if res else 1)) * 2 / 1e9
                row.append(f"{name}: r05 {a[0]:7.1f} us (kinds {a[1]}) | wres {b[0]:7.1f} us (kinds {b[1]}) {gb / (b[0] * 1e-6) / 1e3:5.2f} TB/s  x{a[0] / b[0]:.2f}")
            print(f"M={batch * H * H:6d} N={cout:4d} K=320 operand={int(res)} | " + " || ".join(row), flush=True)
            del e
