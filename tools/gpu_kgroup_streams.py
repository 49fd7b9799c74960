"""Prototype measurement for direction-group pipelining: the k directions of ONE sample split into G groups, each
group's JVP -> VJP chain on its own HIP stream (own engine = own scratch, same weights), one event join in front of the re-orthonormalisation.

    python tools/gpu_kgroup_streams.py [--k 5] [--iters 12] [--splits 5 3+2 2+2+1 1+1+1+1+1]

Prints ms per power iteration for every split, and the max |s| difference against the one-stream result (the per-direction maps are the
same kernels on fewer rows, so s agrees to rounding).  Built entirely on the public C ABI (dpb_jvp / dpb_vjp / dpb_orth)."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffusion_pullback_amd import lib as L  # noqa: E402
from diffusion_pullback_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--op", default="mid")
    ap.add_argument("--block-idx", type=int, default=0)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--splits", nargs="*", default=["5", "3+2", "2+2+1", "1+1+1+1+1"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tap = (a.op, a.block_idx)
    k = a.k
    net, _, shape, t, ctx, V0 = bench.make_workload("sd15", bench.TORCH_DTYPE[a.dtype], dev, k, 1, tap, "null", True)
    lib = L.load()
    cfg = net.config
    eng0 = net.engine
    n_in, n_h = eng0.n_in, eng0.tap_numel(tap)
    buf = eng0.tape.taps[tap]
    x = torch.randn(1, *shape, generator=torch.Generator().manual_seed(1000)).to(dev)
    c1 = ctx.to(dev)
    maxg = max(len(s.split("+")) for s in a.splits)
    engines = [eng0] + [Engine(eng0.tape, cfg.block_out_channels[0], True, False, cfg.in_channels, 1, k) for _ in range(maxg - 1)]
    streams = [torch.cuda.Stream(dev) for _ in range(maxg)]
    main_s = torch.cuda.Stream(dev)
    V = torch.empty(k, n_in, device=dev); U = torch.empty(k, n_h, device=dev); W = torch.empty(k, n_in, device=dev)
    s = torch.empty(k, device=dev); conv = torch.empty(2, device=dev)
    scratch = torch.empty(int(lib.dpb_orth_scratch_bytes(k, n_in)) // 8 + 1, dtype=torch.float64, device=dev)
    P = lambda t_, off=0: C.c_void_p(t_.data_ptr() + off)
    ref_s = None
    for split in a.splits:
        sizes = [int(v) for v in split.split("+")]
        assert sum(sizes) == k
        G = len(sizes)
        offs = [sum(sizes[:i]) for i in range(G)]
        for g in range(G):                     # the same primal on every group's engine
            with torch.cuda.stream(streams[g]):
                engines[g].primal(x, t, c1, tap)
        torch.cuda.synchronize(dev)

        def run(n_it):
            V.copy_(V0.to(dev))
            ev_main = torch.cuda.Event()
            for _ in range(n_it):
                with torch.cuda.stream(main_s):
                    ev_main.record(main_s)
                evs = []
                for g in range(G):
                    st = streams[g]
                    st.wait_event(ev_main)
                    e = engines[g]
                    L.check(lib.dpb_engine_set_stream(e.h, C.c_void_p(st.cuda_stream)))
                    L.check(lib.dpb_jvp(e.h, buf, P(V, offs[g] * n_in * 4), sizes[g], P(U, offs[g] * n_h * 4)))
                    L.check(lib.dpb_vjp(e.h, buf, P(U, offs[g] * n_h * 4), sizes[g], P(W, offs[g] * n_in * 4)))
                    ev = torch.cuda.Event(); ev.record(st); evs.append(ev)
                for ev in evs:
                    main_s.wait_event(ev)
                L.check(lib.dpb_orth(P(W), P(V), P(V), P(s), P(conv), P(scratch), k, n_in, C.c_void_p(main_s.cuda_stream)))
                ev_main = torch.cuda.Event()
        run(2)
        torch.cuda.synchronize(dev)
        times = []
        for _ in range(a.reps):
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            run(a.iters)
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) / a.iters * 1e3)
        sv = s.cpu()
        if ref_s is None:
            ref_s = sv.clone()
        times.sort()
        print(f"split {split:>12}: {times[len(times) // 2]:.3f} ms/iter (min {times[0]:.3f}, max {times[-1]:.3f})  s={[round(v, 2) for v in sv.tolist()]}  "
              f"max|ds|/s={float(((sv - ref_s).abs() / ref_s).max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
