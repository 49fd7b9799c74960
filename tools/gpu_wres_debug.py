"""Not a test: where does the weights-resident kernel (tile 540) differ from the ring (515)?  Prints mismatch coordinates (row m, column n) per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pullback_amd import lib as L
from diffusion_pullback_amd.engine import Engine
from diffusion_pullback_amd.tape import Tape
lib = L.load()
g = torch.Generator().manual_seed(11)


def engine(H, cin, cout, dtype, batch, res=False):
    p = {"c.weight": torch.randn(cout, cin, 1, 1, generator=g) * 0.05, "c.bias": torch.randn(cout, generator=g)}
    t = Tape(p, dtype, "cuda:0")
    t.temb_in = t.buf(1, 8, L.BUF_SHARED)
    t.x = t.buf(H * H, cin)
    o = t.conv("c", t.x, (H, H), cout, ks=1, stride=1, pad=0, res=t.x if res else -1)
    t.tap("o", o, cout, H, H)
    return Engine(t, 8, False, True, cin, max_batch=batch, max_tangents=batch)


for (H, cout, batch, res) in [(64, 320, 5, False), (64, 320, 5, True), (12, 320, 1, False), (20, 320, 3, True), (64, 960, 2, False)]:
    e = engine(H, 320, cout, torch.bfloat16, batch, res)
    x = torch.randn(batch, 320, H, H, generator=g).cuda()
    V = torch.randn(batch, 320 * H * H, generator=g).cuda()
    U = torch.randn(batch, cout * H * H, generator=g).cuda()
    outs = {}
    for tile in (515, 540, 540):
        L.check(lib.dpb_debug_set(b"gemm_splitk", 1)); L.check(lib.dpb_debug_set(b"gemm_tile", tile))
        e.primal(x, 1.0, None, "o")
        outs.setdefault(tile, []).append((e.jvp("o", V).clone(), e.vjp("o", U).clone()))
    for name, idx, C in (("jvp", 0, cout), ("vjp", 1, 320)):
        a, b, b2 = outs[515][0][idx], outs[540][0][idx], outs[540][1][idx]
        d = (a != b).view(batch, C, H * H)
        nz = d.nonzero()
        rows = sorted(set((int(t) * H * H + int(p)) for t, c, p in nz.tolist()))
        cols = sorted(set(int(c) for t, c, p in nz.tolist()))
        print(f"case {(H, cout, batch, res)} {name}: {int(d.sum())} of {d.numel()} differ; rows {rows[:12]}{'...' if len(rows) > 12 else ''} ({len(rows)}) cols {cols[:12]}{'...' if len(cols) > 12 else ''} ({len(cols)}); "
              f"max |d| {(a - b).abs().max().item():.3e}; second run equal to first: {torch.equal(b, b2)}", flush=True)
    del e
L.check(lib.dpb_debug_set(b"gemm_tile", 0)); L.check(lib.dpb_debug_set(b"gemm_splitk", 0))
