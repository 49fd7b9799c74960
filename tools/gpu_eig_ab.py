"""Re-orthonormalisation (dpb_orth) timing by rank: the one-wave cyclic Jacobi eigen-solve of rounds 1-5 (DPB_EIG_PAR=0) against the sixteen-wave
round-robin one (default since round 6), each in its own process (the switch is read once), on the SD latent size N = 16384.
    python tools/gpu_eig_ab.py            -> table on stdout (profiles/r06_eig_parallel.txt)"""
import ctypes as C, json, os, subprocess, sys

import torch


def run(ks):
    from diffusion_pullback_amd import lib as L
    lib = L.load()
    out = {}
    g = torch.Generator().manual_seed(0)
    for k in ks:
        n = 16384
        scale = torch.logspace(0, -2, k)[:, None]
        W = (torch.linalg.qr(torch.randn(k, k, generator=g))[0] @ (scale * torch.linalg.qr(torch.randn(n, k, generator=g))[0].T)).float().cuda()
        Vp = torch.linalg.qr(torch.randn(n, k, generator=g))[0].T.contiguous().float().cuda()
        V = torch.empty_like(W); s = torch.empty(k, device="cuda"); conv = torch.empty(2, device="cuda")
        scratch = torch.empty(int(lib.dpb_orth_scratch_bytes(k, n)) // 8 + 1, dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: L.check(lib.dpb_orth(W.data_ptr(), Vp.data_ptr(), V.data_ptr(), s.data_ptr(), conv.data_ptr(), scratch.data_ptr(), k, n, st))
        for _ in range(3): call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): call()
        b.record(); torch.cuda.synchronize()
        _, s_ref, V_ref = torch.linalg.svd(W.double().cpu(), full_matrices=False)
        cos = (V.double().cpu() * V_ref).sum(-1).abs().min().item()
        out[k] = {"us": a.elapsed_time(b) * 50.0, "min_cos_vs_fp64_svd": cos, "s_rel": ((s.double().cpu() - s_ref.sqrt()).abs() / s_ref.sqrt()).max().item(),
                  "s": s.cpu().tolist(), "V0": V[:, :8].cpu().tolist()}
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(json.dumps(run([int(x) for x in sys.argv[1:]])))
        sys.exit(0)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ks_old, ks_new = [5, 10, 16, 17, 25, 32, 50, 56], [5, 10, 16, 17, 25, 32, 50, 56, 64, 80, 96, 97, 112, 128]
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    get = lambda par, ks: json.loads(subprocess.run([sys.executable, __file__] + [str(k) for k in ks], env=dict(env, DPB_EIG_PAR=str(par)),
                                                    capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    old, new = get(0, ks_old), get(1, ks_new)
    small = get(2, [2, 3, 5, 8, 10, 16])
    ref = get(1, [2, 3, 5, 8, 10, 16])
    print('small ranks, cyclic one wave vs round-robin four waves (DPB_EIG_PAR=2), us per dpb_orth:', {k: (round(ref[k]['us'], 1), round(small[k]['us'], 1)) for k in small})
    print("dpb_orth (gram + eigen-solve + apply + finish, 4 launches) per call, N = 16384, 20 calls back to back; k <= 5 is the same kernel in both columns")
    print(f"{'k':>4} {'cyclic, 1 wave (us)':>20} {'default: round-robin (us)':>28} {'min |cos| vs fp64 svd':>22} {'max rel err of s':>18} {'max |V_new - V_old|':>20}")
    for k in ks_new:
        o, n = old.get(str(k)), new[str(k)]
        dv = max(abs(x - y) for r, q in zip(n["V0"], o["V0"]) for x, y in zip(r, q)) if o else float("nan")
        print(f"{k:>4} {(o['us'] if o else float('nan')):>20.1f} {n['us']:>28.1f} {n['min_cos_vs_fp64_svd']:>22.9f} {n['s_rel']:>18.2e} {dv:>20.2e}")
