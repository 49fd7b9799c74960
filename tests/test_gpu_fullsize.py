"""GPU parity at BASELINE.json's full sizes (SD-v1.5 4x64x64 -> mid 1280x8x8 and every down / up tap; CelebA-HQ DDPM 256x256).

Synthetic weights are spectrum-shaped (configs.Spectrum: sigma_1..12 of the mid-block Jacobian separated by >= 9 % each, above a
flat bulk), so single singular vectors are well conditioned and the north-star criterion -- top-5 |cos| >= 0.99 of the 16-bit
paths against fp32 on identical seeded inputs -- is asserted per vector with NO escape hatch.  Full-size oracle runs are
expensive on the CPU, so they are computed once per module: one direction of primal / JVP / VJP at every tap (all three engine
dtypes are compared against the same oracle vectors) and one 2-iteration k=5 oracle pullback.  The rest are size-independent
properties of the operator pair (J, J^T): adjointness, linearity, orthonormality, convergence under the reference's stop rule.

Tolerances (relative Frobenius error per pass vs the fp32 oracle): fp32 engine 5e-4, bf16 4e-2, fp16 1e-2.
"""
import functools
import os

import pytest
import torch

from _util import abs_cos, oracle_jvp, oracle_vjp, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: 5e-4, torch.bfloat16: 4e-2, torch.float16: 1e-2}
T_SD = 696.2727
ENC = ("time_embedding", "conv_in", "down_blocks", "mid_block")
TAPS = [("down", 0), ("down", 1), ("down", 2), ("down", 3), ("mid", 0), ("up", 0), ("up", 1), ("up", 2), ("up", 3)]


def _threads():
    torch.set_num_threads(min(64, os.cpu_count() or 8))


@functools.lru_cache(maxsize=None)
def _sd15_params(full: bool):
    from diffusion_pullback_amd import configs as cf
    return cf.sd_init_params(cf.SD15, seed=0, only_prefix=None if full else ENC, spectrum=cf.Spectrum())


@functools.lru_cache(maxsize=None)
def _sd15_inputs():
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, 77, 768, generator=g)
    z = torch.randn(1, 4, 64, 64, generator=g)
    return z, ctx


def _sd15(dtype, k=5, max_batch=1, full=False):
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    return PullbackUNet("sd", cf.SD15, _sd15_params(full), dtype=dtype, device=DEV, max_batch=max_batch, max_rank=k * max_batch,
                        upto=("up", 3) if full else ("mid", 0), verbose=False)


def _oracle_f(full, tap, ctx):
    from oracle import unet_sd
    p = _sd15_params(full)                      # seeded VALUES only; the oracle walks them with its OWN config and shape table
    return lambda a: unet_sd.forward(p, unet_sd.SD15, a, torch.tensor(T_SD), ctx.expand(a.shape[0], -1, -1), stop=tap)


# ---- SD-2.1-base: the model id of the reference's own SD scripts (src/scripts/main_various_local_encoder_pullback_with_edit_prompt.sh:11)
@functools.lru_cache(maxsize=None)
def _sd21_params(full: bool):
    from diffusion_pullback_amd import configs as cf
    return cf.sd_init_params(cf.SD21_BASE, seed=0, only_prefix=None if full else ENC, spectrum=cf.Spectrum())


@functools.lru_cache(maxsize=None)
def _sd21_inputs():
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, 77, 1024, generator=g)
    z = torch.randn(1, 4, 64, 64, generator=g)
    return z, ctx


def _sd21(dtype, k=2, full=False):
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    return PullbackUNet("sd", cf.sd_config_for("stabilityai/stable-diffusion-2-1-base"), _sd21_params(full), dtype=dtype, device=DEV, max_batch=1,
                        max_rank=k, upto=("up", 3) if full else ("mid", 0), verbose=False)


def _oracle_f21(full, tap, ctx):
    from oracle import unet_sd
    p = _sd21_params(full)
    return lambda a: unet_sd.forward(p, unet_sd.SD21_BASE, a, torch.tensor(T_SD), ctx.expand(a.shape[0], -1, -1), stop=tap)


@pytest.fixture(scope="module")
def sd15_fp32():
    return _sd15(torch.float32)


@pytest.fixture(scope="module")
def oracle_taps():
    """One direction of primal / JVP / VJP of the full SD-1.5 U-Net at every tap, fp32 CPU oracle (computed once)."""
    _threads()
    z, ctx = _sd15_inputs()
    g = torch.Generator().manual_seed(2)
    V = torch.randn(1, 16384, generator=g)
    out = {}
    for tap in TAPS:
        f = _oracle_f(True, tap, ctx)
        with torch.no_grad():
            h = f(z)
        U = torch.randn(1, h.numel(), generator=g)
        out[tap] = dict(h=h, V=V, JV=oracle_jvp(f, z, V), U=U, JTU=oracle_vjp(f, z, U))
    return out


@pytest.fixture(scope="module")
def oracle_taps_sd21():
    """One direction of primal / JVP / VJP of the full SD-2.1-base U-Net at every tap, fp32 CPU oracle with the oracle's own SD21_BASE."""
    _threads()
    z, ctx = _sd21_inputs()
    g = torch.Generator().manual_seed(3)
    V = torch.randn(1, 16384, generator=g)
    out = {}
    for tap in TAPS:
        f = _oracle_f21(True, tap, ctx)
        with torch.no_grad():
            h = f(z)
        U = torch.randn(1, h.numel(), generator=g)
        out[tap] = dict(h=h, V=V, JV=oracle_jvp(f, z, V), U=U, JTU=oracle_vjp(f, z, U))
    return out


@pytest.fixture(scope="module")
def oracle_pullback_k5():
    """2 iterations of the reference algorithm (oracle.pullback, same autodiff calls as utils.py:766-799), k = 5, mid block."""
    from oracle import pullback as opb
    _threads()
    z, ctx = _sd15_inputs()
    k = 5
    V0 = torch.linalg.qr(torch.randn(16384, k, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
    u, s, vT = opb.pullback(_oracle_f(False, ("mid", 0), ctx), z, pca_rank=k, chunk_size=5, min_iter=0, max_iter=2, convergence_threshold=1e-9,
                            variant="zt", V0=V0)
    return dict(V0=V0, u=u, s=s, vT=vT)


def _relfro_sv(s, vT, s_ref, vT_ref):
    """relative Frobenius error of diag(s) vT after aligning each row's sign (singular vectors are defined up to sign)"""
    a = (s[:, None] * vT).double().cpu()
    b = (s_ref[:, None] * vT_ref).double().cpu()
    sign = torch.sign((a * b).sum(-1, keepdim=True))
    return ((a * sign - b).norm() / b.norm()).item()


# ------------------------------------------------------------------------------------------------ operator properties
def test_sd15_operator_properties_fp32(sd15_fp32):
    z, ctx = _sd15_inputs()
    tap = ("mid", 0)
    e = sd15_fp32.engine
    e.primal(z, T_SD, ctx, tap)
    g = torch.Generator().manual_seed(1)
    V = torch.randn(3, 16384, generator=g).to(DEV)
    U = torch.randn(3, 81920, generator=g).to(DEV)
    JV, JTU = e.jvp(tap, V), e.vjp(tap, U)
    lhs, rhs = JV @ U.T, V @ JTU.T                                   # <J v_i, u_j> vs <v_i, J^T u_j>
    assert rel(lhs, rhs) < 2e-4, rel(lhs, rhs)
    comb = e.jvp(tap, (0.3 * V[0] - 1.7 * V[1])[None])
    assert rel(comb[0], 0.3 * JV[0] - 1.7 * JV[1]) < 2e-4            # linearity of the tangent pass
    comb = e.vjp(tap, (0.3 * U[0] - 1.7 * U[1])[None])
    assert rel(comb[0], 0.3 * JTU[0] - 1.7 * JTU[1]) < 2e-4          # linearity of the adjoint pass


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: every tap, every dtype
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_sd15_every_tap_one_direction_vs_oracle(dtype, oracle_taps):
    """configs[4] (down / up block_0..3 sweep) + the mid tap: primal, JVP and VJP of one direction at FULL size against the fp32 CPU
    oracle, in the engine's three dtypes.  In 16 bit this runs the fused L=4096 / 1024 / 256 attention kernels (head dim 40 / 80 / 160),
    the halo-tile 3x3 convolutions at 64^2 / 32^2 / 16^2, the BK=64 ring GEMMs and the up-path UPCONV / concat ops at their real shapes."""
    z, ctx = _sd15_inputs()
    net = _sd15(dtype, k=1, full=True)
    e = net.engine
    tol = TOL[dtype]
    errs = {}
    for tap in TAPS:
        o = oracle_taps[tap]
        e.primal(z, T_SD, ctx, tap)
        errs[tap] = (rel(e.read(tap), o["h"]), rel(e.jvp(tap, o["V"].to(DEV)), o["JV"]), rel(e.vjp(tap, o["U"].to(DEV)), o["JTU"]))
    print(dtype, {k: tuple(round(x, 5) for x in v) for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not all(x < tol for x in v)}
    assert not bad, f"(primal, jvp, vjp) relative errors over {tol}: {bad}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_sd21_base_every_tap_one_direction_vs_oracle(dtype, oracle_taps_sd21):
    """The reference scripts' default model (stabilityai/stable-diffusion-2-1-base) at FULL size: primal, JVP and VJP of one direction at every
    tap against the fp32 CPU oracle walking the same seeded values with its own SD21_BASE (1024-wide context, 5 / 10 / 20 / 20 heads of 64,
    Linear proj_in / proj_out).  Runs the d = 64 fused attention kernels at L = 4096 / 1024 / 256 / 64 and the 1024-wide text K / V products."""
    z, ctx = _sd21_inputs()
    net = _sd21(dtype, k=1, full=True)
    e = net.engine
    tol = TOL[dtype]
    errs = {}
    for tap in TAPS:
        o = oracle_taps_sd21[tap]
        e.primal(z, T_SD, ctx, tap)
        errs[tap] = (rel(e.read(tap), o["h"]), rel(e.jvp(tap, o["V"].to(DEV)), o["JV"]), rel(e.vjp(tap, o["U"].to(DEV)), o["JTU"]))
    print("sd21", dtype, {k: tuple(round(x, 5) for x in v) for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not all(x < tol for x in v)}
    assert not bad, f"(primal, jvp, vjp) relative errors over {tol}: {bad}"


# ------------------------------------------------------------------------------------------------ the algorithm vs the oracle
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_sd15_pullback_two_iterations_vs_oracle(dtype, oracle_pullback_k5):
    """Same V0, same two iterations as the CPU oracle: per-vector |cos|, singular values and diag(s) vT."""
    o = oracle_pullback_k5
    z, ctx = _sd15_inputs()
    net = _sd15(dtype)
    u, s, vT = net.local_encoder_pullback_zt(z, torch.tensor(T_SD), ctx, op="mid", block_idx=0, pca_rank=5, chunk_size=5, min_iter=0, max_iter=2,
                                             convergence_threshold=1e-9, V0=o["V0"])
    cos_v, cos_u = abs_cos(vT, o["vT"]), abs_cos(u.T, o["u"].T)
    fro = _relfro_sv(s.cpu(), vT.cpu(), o["s"], o["vT"])
    print(dtype, "s", s.cpu().tolist(), "oracle", o["s"].tolist(), "|cos v|", cos_v.tolist(), "|cos u|", cos_u.tolist(), "relfro", fro)
    if dtype == torch.float32:
        assert torch.allclose(s.cpu(), o["s"], rtol=1e-3) and (cos_v > 0.9999).all() and (cos_u > 0.9999).all() and fro < 5e-3
    else:
        assert torch.allclose(s.cpu(), o["s"], rtol=2e-2), (s.cpu(), o["s"])
        assert (cos_v > 0.99).all() and (cos_u > 0.99).all(), (cos_v, cos_u)
        assert fro < 1e-1, fro


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_sd15_16bit_top5_vs_fp32(dtype, sd15_fp32):
    """NORTH STAR: top-5 singular vectors of the 16-bit path vs the fp32 path on identical seeded inputs after the reference's 12
    iterations: |cos| >= 0.99 for EVERY vector, singular values to 1 %, diag(s) vT to 5 % -- on a spectrum whose gaps make
    the per-vector comparison meaningful (asserted)."""
    z, ctx = _sd15_inputs()
    net16 = _sd15(dtype)
    k = 5
    V0 = torch.linalg.qr(torch.randn(16384, k, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
    _, s32, v32, _ = sd15_fp32.pullback_fixed(z, T_SD, ctx, "mid", 0, k, 12, V0)
    _, s16, v16, _ = net16.pullback_fixed(z, T_SD, ctx, "mid", 0, k, 12, V0)
    ratios = (s32[1:] / s32[:-1]).cpu()
    cos = abs_cos(v16, v32)
    fro = _relfro_sv(s16.cpu(), v16.cpu(), s32.cpu(), v32.cpu())
    print(dtype, "sigma fp32", s32.cpu().tolist(), "16-bit", s16.cpu().tolist(), "|cos|", cos.tolist(), "relfro", fro)
    assert (ratios < 0.95).all(), f"spectrum not separated: {s32.cpu().tolist()}"        # the criterion below is well conditioned
    assert (cos > 0.99).all(), cos
    assert torch.allclose(s16, s32, rtol=1e-2), (s16, s32)
    assert fro < 5e-2, fro
    assert torch.allclose((v16 @ v16.T).cpu(), torch.eye(k), atol=1e-3)                   # orthonormal rows


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_sd15_converges_under_reference_stop_rule(dtype):
    """utils.py:803-808 with the reference's defaults (min_iter 10, max_iter 100, atol 1e-3): the shaped spectrum converges at the
    earliest iteration the rule allows (i > min_iter -> 12 iterations), in fp32 and with bf16 rounding noise."""
    z, ctx = _sd15_inputs()
    net = _sd15(dtype)
    V0 = torch.linalg.qr(torch.randn(16384, 5, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
    net.local_encoder_pullback_zt(z, torch.tensor(T_SD), ctx, op="mid", block_idx=0, pca_rank=5, V0=V0)
    print(dtype, "iterations", net.last_iters, "dist", net.last_dist)
    assert net.last_iters == 12, (net.last_iters, net.last_dist)


# ------------------------------------------------------------------------------------------------ BASELINE configs[3]
@pytest.mark.parametrize("S", [4, 8], ids=["S4", "S8"])
def test_sd15_config3_k10_samples_advanced_together(S):
    """configs[3]: edit-prompt context, k = 10, several x_t samples advanced together on one GPU (one rank's share of the 64-sample
    job), the bench's own 12 iterations: the batched run equals one-at-a-time runs in ALL ten vectors (|cos| >= 0.99, s to 2 %), and one of
    the ten tangents / cotangents of the LAST sample's block of the batched pass equals the oracle JVP / VJP of that direction.  S = 8 is the
    batch bench.py's strong leg runs (nt = 80: the launches that go to the 256 x 256 tile); S = 4 is the shape that overflowed the split-K slabs in round 2."""
    _threads()
    k, iters = 10, 12
    g = torch.Generator().manual_seed(77)
    ctx = torch.randn(1, 77, 768, generator=g)                                   # seeded "edit prompt" embedding (not the null ctx)
    zs = torch.cat([torch.randn(4, 4, 64, 64, generator=g), torch.randn(4, 4, 64, 64, generator=torch.Generator().manual_seed(78))])[:S]
    V0 = torch.linalg.qr(torch.randn(16384, k, generator=g))[0].T.contiguous()
    net = _sd15(torch.bfloat16, k=k, max_batch=S)
    _, s_b, V_b, _ = net.pullback_fixed(zs, T_SD, ctx.expand(S, -1, -1), "mid", 0, k, iters, V0)
    s_b, V_b = s_b.clone(), V_b.clone()
    for i in range(S):
        _, s_i, V_i, _ = net.pullback_fixed(zs[i:i + 1], T_SD, ctx, "mid", 0, k, iters, V0)
        sb, Vb = s_b[k * i:k * (i + 1)], V_b[k * i:k * (i + 1)]
        cos = abs_cos(Vb, V_i)
        # the batched and the single launch pick different tiles / split-K factors (16-bit rounding differs); after the 12 iterations of the
        # bench the slowest direction has contracted by (sigma_11 / sigma_10)^24 ~ 0.8^24 = 0.005: every vector is held to the north-star bar
        print(i, "s batched", sb.cpu().tolist(), "single", s_i.cpu().tolist(), "|cos|", cos.tolist())
        assert torch.allclose(sb, s_i, rtol=2e-2), (i, sb, s_i)
        assert (cos > 0.99).all(), (i, cos)
    # distinct samples have distinct bases (the batch is not one sample repeated)
    assert abs_cos(V_b[0:1], V_b[k:k + 1]).item() < 0.99
    # one oracle direction out of the LAST sample's block of the BATCHED tangent / adjoint pass (nt = S * k rows in every launch)
    e = net.engine
    last = S - 1
    e.primal(zs, T_SD, ctx.expand(S, -1, -1), ("mid", 0))
    U = e.jvp(("mid", 0), V0.repeat(S, 1).to(DEV))
    f = _oracle_f(False, ("mid", 0), ctx)
    assert rel(U[last * k + 7:last * k + 8], oracle_jvp(f, zs[last:last + 1], V0[7:8])) < TOL[torch.bfloat16]
    Uc = torch.randn(k, 81920, generator=g)
    W = e.vjp(("mid", 0), Uc.repeat(S, 1).to(DEV))
    assert rel(W[last * k + 3:last * k + 4], oracle_vjp(f, zs[last:last + 1], Uc[3:4])) < TOL[torch.bfloat16]


# ------------------------------------------------------------------------------------------------ BASELINE configs[1]
def test_ddpm256_operator_and_oracle_fp32():
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    from oracle import unet_ddpm
    _threads()
    cfg = cf.CELEBA_HQ_256
    params = cf.ddpm_init_params(cfg, seed=0, spectrum=cf.Spectrum())
    net = PullbackUNet("ddpm", cfg, params, dtype=torch.float32, device=DEV, max_batch=1, max_rank=5, upto=("mid", 0), verbose=False)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 256, 256, generator=g)
    t = 600.0
    tap = ("mid", 0)
    e = net.engine
    e.primal(x, t, None, tap)
    V = torch.randn(2, 196608, generator=g).to(DEV)
    U = torch.randn(2, 32768, generator=g).to(DEV)
    JV, JTU = e.jvp(tap, V), e.vjp(tap, U)
    assert rel(JV @ U.T, V @ JTU.T) < 2e-4
    f = lambda a: unet_ddpm.forward(params, cfg, a, torch.tensor(t), stop=tap)
    with torch.no_grad():
        assert rel(e.read(tap), f(x)) < 2e-4
    assert rel(JV[:1], oracle_jvp(f, x, V[:1].cpu())) < 5e-4
    assert rel(JTU[:1], oracle_vjp(f, x, U[:1].cpu())) < 5e-4
    # configs[1]: k = 5 under the reference's stop rule converges at the earliest allowed iteration on the shaped spectrum
    V0 = torch.linalg.qr(torch.randn(196608, 5, generator=g))[0].T.contiguous()
    u, s, vT = net.local_encoder_pullback_xt(x, torch.tensor(t), op="mid", block_idx=0, pca_rank=5, V0=V0)
    assert net.last_iters == 12 and ((s[1:] / s[:-1]) < 0.95).all(), (net.last_iters, s)


# ------------------------------------------------------------------------------------------------ the HEADLINE configs vs the REFERENCE's own function
def _check_vs_reference(fix, s, vT, dtype):
    s, vT = s.float().cpu(), vT.float().cpu()
    cos = abs_cos(vT, fix["vT"])
    fro = _relfro_sv(s, vT, fix["s"], fix["vT"])
    print(dtype, "s", s.tolist(), "reference", fix["s"].tolist(), "|cos|", cos.tolist(), "relfro", fro)
    assert ((fix["s"][1:] / fix["s"][:-1]) < 0.95).all()                       # the per-vector criterion is well conditioned
    assert (cos > (0.9999 if dtype == torch.float32 else 0.99)).all(), cos       # north star: top-5 |cos| >= 0.99
    assert torch.allclose(s, fix["s"], rtol=1e-3 if dtype == torch.float32 else 2e-2), (s, fix["s"])
    assert fro < (5e-3 if dtype == torch.float32 else 5e-2), fro


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_sd15_headline_vs_reference_function_golden(dtype):
    """BASELINE configs[2] / north star, no transitivity: the fixture holds (s, vT) RETURNED BY THE REFERENCE'S OWN
    utils.local_encoder_pullback_zt (src/utils/utils.py:722-816) run on the CPU over the full-size SD-v1.5 net with these weights and inputs,
    k = 5, its default stop rule capped at 12 iterations (tests/golden/make_golden_fullsize.py).  The product draws the same V0 (V0=None under
    the recorded seed, like the reference: utils.py:750-751) and must match every one of the five vectors / values."""
    from _util import load_golden
    fix = load_golden("pullback_sd15_mid_k5.pt")
    assert fix["iters"] == 12 and fix["k"] == 5
    z, ctx = _sd15_inputs()
    net = _sd15(dtype)
    torch.manual_seed(fix["rng_seed"])
    u, s, vT = net.local_encoder_pullback_zt(z, torch.tensor(T_SD), ctx, op="mid", block_idx=0, pca_rank=5, chunk_size=fix["chunk_size"],
                                             min_iter=fix["min_iter"], max_iter=fix["max_iter"], convergence_threshold=fix["thr"])
    assert net.last_iters == 12
    _check_vs_reference(fix, s, vT, dtype)
    # u = J V_prev of the last iteration (un-normalised, utils.py:810): column norms and a probe of the leading rows, sign-aligned per column
    un = u.float().cpu().norm(dim=0)
    assert torch.allclose(un, fix["u_norms"], rtol=1e-3 if dtype == torch.float32 else 3e-2), (un, fix["u_norms"])
    uh, rh = u[:256].float().cpu(), fix["u_head"]
    sign = torch.sign((uh * rh).sum(0, keepdim=True))
    assert rel(uh * sign, rh) < (2e-3 if dtype == torch.float32 else 1e-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_sd21_base_k2_vs_reference_function_golden(dtype):
    """The reference scripts' default model at the settings of its only published timing (example-code.ipynb:123-145: pca_rank 2, 12 iterations):
    (s, vT) RETURNED BY THE REFERENCE'S OWN utils.local_encoder_pullback_zt bound onto the full-size SD-2.1-base oracle net
    (tests/golden/make_golden_fullsize.py sd21).  The product draws the same V0 under the recorded seed."""
    from _util import load_golden
    fix = load_golden("pullback_sd21_mid_k2.pt")
    assert fix["iters"] == 12 and fix["k"] == 2
    z, ctx = _sd21_inputs()
    net = _sd21(dtype)
    torch.manual_seed(fix["rng_seed"])
    u, s, vT = net.local_encoder_pullback_zt(z, torch.tensor(T_SD), ctx, op="mid", block_idx=0, pca_rank=2, chunk_size=fix["chunk_size"],
                                             min_iter=fix["min_iter"], max_iter=fix["max_iter"], convergence_threshold=fix["thr"])
    assert net.last_iters == 12
    _check_vs_reference(fix, s, vT, dtype)
    un = u.float().cpu().norm(dim=0)
    assert torch.allclose(un, fix["u_norms"], rtol=1e-3 if dtype == torch.float32 else 3e-2), (un, fix["u_norms"])


def test_ddpm256_headline_vs_reference_function_golden():
    """BASELINE configs[1]: (s, vT) returned by the vendored PullBackDDPM.local_encoder_pullback_xt (src/models/ddpm/diffusion.py:484-556) on the
    full-size CelebA-HQ-256 net, k = 5, fp32, 12 iterations."""
    from _util import load_golden
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    fix = load_golden("pullback_ddpm256_mid_k5.pt")
    assert fix["iters"] == 12 and fix["k"] == 5
    cfg = cf.CELEBA_HQ_256
    params = cf.ddpm_init_params(cfg, seed=0, spectrum=cf.Spectrum())
    net = PullbackUNet("ddpm", cfg, params, dtype=torch.float32, device=DEV, max_batch=1, max_rank=5, upto=("mid", 0), verbose=False)
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(fix["rng_seed"])
    u, s, vT = net.local_encoder_pullback_xt(x, torch.tensor(600.0), op="mid", block_idx=0, pca_rank=5, chunk_size=fix["chunk_size"],
                                             min_iter=fix["min_iter"], max_iter=fix["max_iter"], convergence_threshold=fix["thr"])
    assert net.last_iters == 12
    _check_vs_reference(fix, s, vT, torch.float32)


# ------------------------------------------------------------------------------------------------ the TIMED loop (dpb_pullback_iterate) vs the REFERENCE's own function
def _reference_v0(fix, n_in, k):
    """The V0 the reference draws under the fixture's seed (utils.py:750-753: qr of a CPU randn(n_in, k))."""
    torch.manual_seed(fix["rng_seed"])
    q, _ = torch.linalg.qr(torch.randn(n_in, k, dtype=torch.float))
    return q.T.contiguous()


def _check_u(fix, u, dtype, head=True):
    un = u.float().cpu().norm(dim=0)
    assert torch.allclose(un, fix["u_norms"], rtol=1e-3 if dtype == torch.float32 else 3e-2), (un, fix["u_norms"])
    if head and "u_head" in fix:
        uh, rh = u[:256].float().cpu(), fix["u_head"]
        sign = torch.sign((uh * rh).sum(0, keepdim=True))
        assert rel(uh * sign, rh) < (2e-3 if dtype == torch.float32 else 1e-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_sd15_fused_iterate_vs_reference_function_golden(dtype):
    """The path bench.py times (dpb_primal + dpb_pullback_iterate = pullback_fixed: JVP, VJP and re-orthonormalisation of all 12 iterations chained
    on the device, tangent aliasing included) against (u, s, vT) RETURNED BY THE REFERENCE'S OWN utils.local_encoder_pullback_zt (utils.py:756-810)
    on the full-size SD-v1.5 net -- same fixture, same bars as the host-driven loop above, no transitivity through the toy nets."""
    from _util import load_golden
    fix = load_golden("pullback_sd15_mid_k5.pt")
    assert fix["iters"] == 12 and fix["k"] == 5
    z, ctx = _sd15_inputs()
    net = _sd15(dtype)
    u, s, vT, _ = net.pullback_fixed(z, T_SD, ctx, "mid", 0, 5, 12, _reference_v0(fix, 16384, 5))
    _check_vs_reference(fix, s, vT, dtype)
    _check_u(fix, u, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_sd21_base_k2_fused_iterate_vs_reference_function_golden(dtype):
    from _util import load_golden
    fix = load_golden("pullback_sd21_mid_k2.pt")
    z, ctx = _sd21_inputs()
    net = _sd21(dtype)
    u, s, vT, _ = net.pullback_fixed(z, T_SD, ctx, "mid", 0, 2, 12, _reference_v0(fix, 16384, 2))
    _check_vs_reference(fix, s, vT, dtype)
    _check_u(fix, u, dtype, head=False)


def test_ddpm256_fused_iterate_vs_reference_function_golden():
    from _util import load_golden
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    fix = load_golden("pullback_ddpm256_mid_k5.pt")
    cfg = cf.CELEBA_HQ_256
    params = cf.ddpm_init_params(cfg, seed=0, spectrum=cf.Spectrum())
    net = PullbackUNet("ddpm", cfg, params, dtype=torch.float32, device=DEV, max_batch=1, max_rank=5, upto=("mid", 0), verbose=False)
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    u, s, vT, _ = net.pullback_fixed(x, 600.0, None, "mid", 0, 5, 12, _reference_v0(fix, 196608, 5))
    _check_vs_reference(fix, s, vT, torch.float32)


# ------------------------------------------------------------------------------------------------ the batched forward-only pass (the DDIM / edit loop's U-Net call)
@pytest.mark.parametrize("dtype,B", [(torch.bfloat16, 2), (torch.bfloat16, 5), (torch.float16, 2), (torch.float16, 5)], ids=["bf16-B2", "bf16-B5", "fp16-B2", "fp16-B5"])
def test_sd15_forward_only_eps_vs_oracle(dtype, B):
    """engine.forward(z, t, ctx, "eps") -- what DDIMforwardsteps / x_space_guidance call per step (/root/reference/src/modules/edit.py:454-458,
    :484-502), with the forward-only shortcuts (fused temb / context projections, one-launch cross-attention, GEGLU epilogue, output head) -- at the
    CLI's batch sizes and full SD-1.5 size, per sample against the CPU oracle's whole U-Net (stop=None)."""
    from oracle import unet_sd
    _threads()
    g = torch.Generator().manual_seed(11)
    z = torch.randn(B, 4, 64, 64, generator=g)
    ctx = torch.randn(B, 77, 768, generator=g)
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    net = PullbackUNet("sd", cf.SD15, _sd15_params(True), dtype=dtype, device=DEV, max_batch=B, max_rank=B, upto=None, verbose=False)   # whole U-Net, output head included
    eps = net.engine.forward(z, T_SD, ctx, "eps").float().cpu()
    assert eps.shape == (B, 4, 64, 64)
    p = _sd15_params(True)
    picks = [0, B - 1] if B > 2 else [0, 1]
    for i in picks:                                           # first and last sample of the batch (distinct inputs and contexts)
        with torch.no_grad():
            ref = unet_sd.forward(p, unet_sd.SD15, z[i:i + 1], torch.tensor(T_SD), ctx[i:i + 1], stop=None)
        assert rel(eps[i:i + 1], ref) < TOL[dtype], (i, rel(eps[i:i + 1], ref))
    # the batch is not one sample repeated, and rows do not leak across samples: sample 0 alone gives the same eps as inside the batch (16-bit rounding)
    solo = net.engine.forward(z[:1], T_SD, ctx[:1], "eps").float().cpu()
    assert rel(solo, eps[:1]) < TOL[dtype]
    assert rel(eps[:1], eps[1:2]) > 0.1


def test_sd21_forward_only_eps_vs_oracle_fp16_b2():
    from oracle import unet_sd
    _threads()
    g = torch.Generator().manual_seed(12)
    z = torch.randn(2, 4, 64, 64, generator=g)
    ctx = torch.randn(2, 77, 1024, generator=g)
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    net = PullbackUNet("sd", cf.sd_config_for("stabilityai/stable-diffusion-2-1-base"), _sd21_params(True), dtype=torch.float16, device=DEV, max_batch=2,
                       max_rank=2, upto=None, verbose=False)
    eps = net.engine.forward(z, T_SD, ctx, "eps").float().cpu()
    p = _sd21_params(True)
    for i in (0, 1):
        with torch.no_grad():
            ref = unet_sd.forward(p, unet_sd.SD21_BASE, z[i:i + 1], torch.tensor(T_SD), ctx[i:i + 1], stop=None)
        assert rel(eps[i:i + 1], ref) < TOL[torch.float16], (i, rel(eps[i:i + 1], ref))


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: the criterion at down / up taps
@pytest.mark.parametrize("tap", [("down", 1), ("up", 3)], ids=["down1", "up3"])
def test_sd15_config4_top5_fp16_vs_fp32_at_down_and_up_taps(tap):
    """configs[4] (down / up sweep, fp16): the mid-block shaping is not in the prefix of a down tap, so Spectrum.for_tap also shapes the last
    self-attention INSIDE the tap's prefix; the top-5 criterion (|cos| >= 0.99 per vector, fp16 vs fp32, 12 iterations, identical seeded
    inputs) is then well conditioned at the tap (asserted) and asserted."""
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    z, ctx = _sd15_inputs()
    k = 5
    sp = cf.Spectrum.for_tap(*tap)
    params = cf.sd_init_params(cf.SD15, seed=0, only_prefix=ENC if tap[0] == "down" else None, spectrum=sp)
    V0 = torch.linalg.qr(torch.randn(16384, k, generator=torch.Generator().manual_seed(0)))[0].T.contiguous()
    out = {}
    for dtype in (torch.float32, torch.float16):
        net = PullbackUNet("sd", cf.SD15, params, dtype=dtype, device=DEV, max_batch=1, max_rank=k, upto=tap, verbose=False)
        _, s, v, _ = net.pullback_fixed(z, T_SD, ctx, tap[0], tap[1], k, 12, V0)
        out[dtype] = (s.clone().cpu(), v.clone().cpu())
        del net
        torch.cuda.empty_cache()
    (s32, v32), (s16, v16) = out[torch.float32], out[torch.float16]
    cos = abs_cos(v16, v32)
    print(tap, "sigma fp32", s32.tolist(), "fp16", s16.tolist(), "|cos|", cos.tolist())
    assert ((s32[1:] / s32[:-1]) < 0.95).all(), f"spectrum not separated at {tap}: {s32.tolist()}"
    assert (cos > 0.99).all(), cos
    assert torch.allclose(s16, s32, rtol=1e-2), (s16, s32)
    assert _relfro_sv(s16, v16, s32, v32) < 5e-2
