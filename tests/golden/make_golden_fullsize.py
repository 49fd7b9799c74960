"""Full-size golden vectors of the HEADLINE configs, produced by the REFERENCE's own functions.

Run in the build container only (needs /root/reference; ~30-40 min on 8 vCPU):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fullsize.py [sd15] [ddpm256] [sd21]

  pullback_sd15_mid_k5.pt    utils.local_encoder_pullback_zt (reference src/utils/utils.py:722-816) bound with
                             types.MethodType onto an object whose get_h is the full-size SD-v1.5 oracle net
                             (oracle.unet_sd, shaped-spectrum weights configs.sd_init_params(SD15, seed 0, Spectrum())):
                             z_t[1,4,64,64], ctx[1,77,768], t = 696.2727, op 'mid', pca_rank 5, chunk_size 5, the
                             reference's default stop rule (min_iter 10, atol 1e-3) capped at max_iter 12  (BASELINE configs[2])
  pullback_sd21_mid_k2.pt    the same function on the full-size SD-2.1-base oracle net (the reference scripts' default model id; ctx[1,77,1024],
                             64-wide heads, Linear proj_in / proj_out), pca_rank 2, 12 iterations -- the settings of the reference's only
                             published timing (example-code.ipynb:123-145)
  pullback_ddpm256_mid_k5.pt the vendored PullBackDDPM.local_encoder_pullback_xt (src/models/ddpm/diffusion.py:484-556)
                             of the full-size CelebA-HQ-256 net (config src/configs/custom_celeba_ddpm.yml) with
                             configs.ddpm_init_params(CELEBA_HQ_256, seed 0, Spectrum()): x[1,3,256,256], t = 600, k = 5,
                             same stop rule / cap  (BASELINE configs[1])

Only (s, vT), the seed under which the reference function drew its V0 (tests re-draw it: qr(randn(N, k)) after manual_seed), the per-iteration `dist` prints and small probes
of u are committed (u itself is k x 81920 / 32768 floats and is implied by vT: u_i = J v_i / |.|).  Inputs are re-created
from seeds by the tests (tests/test_gpu_fullsize.py::_sd15_inputs and the DDPM test), not stored.
"""
import contextlib
import io
import os
import re
import sys
import time
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

import torch  # noqa: E402

from make_golden import import_reference  # noqa: E402

T_SD = 696.2727
RNG_SEED = 31          # torch.manual_seed before the reference call: it draws V0 = qr(randn(N, k)) itself (utils.py:750-751)
K = 5
MAX_ITER = 12          # the reference's default rule cannot stop before i = 11 (i > min_iter = 10): 12 iterations


def _history(text):
    return [float(x) for x in re.findall(r"power method : \d+-th step convergence :\s+(?:tensor\()?([0-9.eE+-]+)", text)]


def _run(fn):
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        out = fn()
    return out, buf.getvalue(), time.time() - t0


def sd15(ru):
    from diffusion_pullback_amd import configs as cf
    from oracle import unet_sd
    enc = ("time_embedding", "conv_in", "down_blocks", "mid_block")
    params = cf.sd_init_params(cf.SD15, seed=0, only_prefix=enc, spectrum=cf.Spectrum())
    g = torch.Generator().manual_seed(0)                       # == tests/test_gpu_fullsize.py::_sd15_inputs
    ctx = torch.randn(1, 77, 768, generator=g)
    z = torch.randn(1, 4, 64, 64, generator=g)

    class Net:
        dtype = torch.float32

        def get_h(self, sample=None, timestep=None, encoder_hidden_states=None, op=None, block_idx=None, verbose=False):
            return unet_sd.forward(params, unet_sd.SD15, sample, timestep, encoder_hidden_states, stop=(op, block_idx))   # the oracle's own config

    net = Net()
    net.local_encoder_pullback_zt = types.MethodType(ru.local_encoder_pullback_zt, net)
    torch.manual_seed(RNG_SEED)
    (u, s, vT), log, dt = _run(lambda: net.local_encoder_pullback_zt(z, torch.tensor(T_SD), ctx, op="mid", block_idx=0, pca_rank=K, chunk_size=5,
                                                                     min_iter=10, max_iter=MAX_ITER, convergence_threshold=1e-3))
    hist = _history(log)
    fix = {"workload": "sd15 mid k5", "weights": "configs.sd_init_params(SD15, seed=0, only_prefix=ENC, spectrum=Spectrum())",
           "inputs": "Generator(0): ctx = randn(1,77,768); z = randn(1,4,64,64)", "t": T_SD, "rng_seed": RNG_SEED, "k": K, "chunk_size": 5,
           "min_iter": 10, "max_iter": MAX_ITER, "thr": 1e-3, "iters": len(hist), "dist_history": hist,
           "s": s.clone(), "vT": vT.clone(), "u_norms": u.norm(dim=0).clone(), "u_head": u[:256].clone(), "seconds": dt}
    torch.save(fix, os.path.join(HERE, "pullback_sd15_mid_k5.pt"))
    print("sd15: %d iterations in %.0f s; s = %s; dist = %s" % (len(hist), dt, s.tolist(), hist), flush=True)


def sd21(ru):
    """The reference scripts' own default model (stabilityai/stable-diffusion-2-1-base, src/scripts/main_various_local_encoder_pullback_with_edit_prompt.sh:11)
    at the settings of its only published timing (example-code.ipynb:123-145: pca_rank 2, fp32, 12 iterations)."""
    from diffusion_pullback_amd import configs as cf
    from oracle import unet_sd
    enc = ("time_embedding", "conv_in", "down_blocks", "mid_block")
    params = cf.sd_init_params(cf.SD21_BASE, seed=0, only_prefix=enc, spectrum=cf.Spectrum())
    g = torch.Generator().manual_seed(0)                       # == tests/test_gpu_fullsize.py::_sd21_inputs
    ctx = torch.randn(1, 77, 1024, generator=g)
    z = torch.randn(1, 4, 64, 64, generator=g)

    class Net:
        dtype = torch.float32

        def get_h(self, sample=None, timestep=None, encoder_hidden_states=None, op=None, block_idx=None, verbose=False):
            return unet_sd.forward(params, unet_sd.SD21_BASE, sample, timestep, encoder_hidden_states, stop=(op, block_idx))   # the oracle's own config

    net = Net()
    net.local_encoder_pullback_zt = types.MethodType(ru.local_encoder_pullback_zt, net)
    k = 2
    torch.manual_seed(RNG_SEED)
    (u, s, vT), log, dt = _run(lambda: net.local_encoder_pullback_zt(z, torch.tensor(T_SD), ctx, op="mid", block_idx=0, pca_rank=k, chunk_size=5,
                                                                     min_iter=10, max_iter=MAX_ITER, convergence_threshold=1e-3))
    hist = _history(log)
    fix = {"workload": "sd21-base mid k2", "weights": "configs.sd_init_params(SD21_BASE, seed=0, only_prefix=ENC, spectrum=Spectrum())",
           "inputs": "Generator(0): ctx = randn(1,77,1024); z = randn(1,4,64,64)", "t": T_SD, "rng_seed": RNG_SEED, "k": k, "chunk_size": 5,
           "min_iter": 10, "max_iter": MAX_ITER, "thr": 1e-3, "iters": len(hist), "dist_history": hist,
           "s": s.clone(), "vT": vT.clone(), "u_norms": u.norm(dim=0).clone(), "u_head": u[:256].clone(), "seconds": dt}
    torch.save(fix, os.path.join(HERE, "pullback_sd21_mid_k2.pt"))
    print("sd21: %d iterations in %.0f s; s = %s; dist = %s" % (len(hist), dt, s.tolist(), hist), flush=True)


def ddpm256(ru, rd):
    from diffusion_pullback_amd import configs as cf
    cfg = cf.CELEBA_HQ_256
    ns = ru.dict2namespace({"config": {"model": dict(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                                                   attn_resolutions=list(cfg.attn_resolutions), dropout=0.0, in_channels=cfg.in_channels,
                                                   resamp_with_conv=True),
                                      "data": dict(image_size=cfg.resolution)}})
    ns.device = "cpu"; ns.dtype = torch.float32
    net = rd.PullBackDDPM(ns).eval()
    net.load_state_dict(cf.ddpm_init_params(cfg, seed=0, spectrum=cf.Spectrum()), strict=True)
    g = torch.Generator().manual_seed(0)                       # == tests/test_gpu_fullsize.py::test_ddpm256_*: first draw of Generator(0)
    x = torch.randn(1, 3, 256, 256, generator=g)
    t = torch.tensor(600.0)
    torch.manual_seed(RNG_SEED)
    (u, s, vT), log, dt = _run(lambda: net.local_encoder_pullback_xt(x=x, t=t, op="mid", block_idx=0, pca_rank=K, chunk_size=25,
                                                                     min_iter=10, max_iter=MAX_ITER, convergence_threshold=1e-3))
    hist = _history(log)
    fix = {"workload": "ddpm256 mid k5", "weights": "configs.ddpm_init_params(CELEBA_HQ_256, seed=0, spectrum=Spectrum())",
           "inputs": "Generator(0): x = randn(1,3,256,256)", "t": 600.0, "rng_seed": RNG_SEED, "k": K, "chunk_size": 25,
           "min_iter": 10, "max_iter": MAX_ITER, "thr": 1e-3, "iters": len(hist), "dist_history": hist,
           "s": s.clone(), "vT": vT.clone(), "u_norms": u.norm(dim=0).clone(), "u_head": u[:256].clone(), "seconds": dt}
    torch.save(fix, os.path.join(HERE, "pullback_ddpm256_mid_k5.pt"))
    print("ddpm256: %d iterations in %.0f s; s = %s; dist = %s" % (len(hist), dt, s.tolist(), hist), flush=True)


def main():
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    which = sys.argv[1:] or ["ddpm256", "sd15"]
    ru, rd = import_reference()
    if "ddpm256" in which:
        ddpm256(ru, rd)
    if "sd15" in which:
        sd15(ru)
    if "sd21" in which:
        sd21(ru)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
