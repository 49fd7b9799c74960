"""Not a test: the power iteration replayed as a captured hipGraph (dpb_debug_set("graph_iterate", 1)) against eager launches, same
process, same buffers, on a non-default stream (the legacy default stream cannot be captured).
    python tools/gpu_graph_iterate.py [k] [samples]  > gpurun_out/graph_iterate.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffusion_pullback_amd import lib as L

k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
net, _, shape, t, ctx, V0 = bench.make_workload("sd15", torch.bfloat16, dev, k, S)
eng = net.engine
tap = ("mid", 0)
lib = L.load()
x = torch.randn(S, *shape, generator=torch.Generator().manual_seed(1)).to(dev)
ctx_d = ctx.to(dev).expand(S, -1, -1).contiguous()
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    eng.primal(x, t, ctx_d, tap)
    V = V0.to(dev).repeat(S, 1).contiguous()
    # fixed output buffers: call the C ABI directly so that every call sees the same pointers
    import ctypes as C
    U = torch.empty(k * S, eng.tap_numel(tap), device=dev); s = torch.empty(k * S, device=dev); conv = torch.empty(S, 2, device=dev)
    eng._set_stream()
    buf = eng.tape.taps[tap]

    def run(n):
        L.check(lib.dpb_pullback_iterate(eng.h, buf, V.data_ptr(), U.data_ptr(), s.data_ptr(), conv.data_ptr(), k, n))

    res = {}
    for rep in range(3):
        for mode in (0, 1):
            L.check(lib.dpb_debug_set(b"graph_iterate", mode))
            V.copy_(V0.to(dev).repeat(S, 1))
            run(14)                                   # warm (and capture)
            st.synchronize(); t0 = time.perf_counter()
            run(96)
            st.synchronize(); dt = (time.perf_counter() - t0) / 96 * 1e3
            res.setdefault(mode, []).append(dt)
            print(f"rep {rep} {'hipGraph replay' if mode else 'eager launches '}: {dt:.3f} ms / iteration   s[:3] = {s[:3].tolist()}", flush=True)
    L.check(lib.dpb_debug_set(b"graph_iterate", 0))
    print("eager", min(res[0]), "graph", min(res[1]), "ratio graph/eager", min(res[1]) / min(res[0]))
