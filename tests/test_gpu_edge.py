"""GPU edge cases of the drop-in surface: rank 1 and the maximum rank, error behaviour, timestep forms, chunking."""
import pytest
import torch

from _util import abs_cos, load_golden, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def toy():
    from diffusion_pullback_amd import PullbackUNet
    from oracle import unet_sd
    f = load_golden("pullback_zt_tiny.pt")
    cfg = unet_sd.SDConfig(**f["cfg"])
    p = unet_sd.init_params(cfg, seed=f["seed"], gain=f["gain"])
    return f, cfg, p, PullbackUNet("sd", cfg, p, dtype=torch.float32, device=DEV, max_batch=2, max_rank=50, verbose=False)


def test_rank_one_and_max_rank_match_full_jacobian_svd(toy):
    f, cfg, p, net = toy
    from oracle import unet_sd
    J = torch.autograd.functional.jacobian(
        lambda a: unet_sd.forward(p, cfg, a, f["t"], f["ctx"], stop=("mid", 0)).reshape(-1), f["z"]).reshape(-1, 256)
    sv = torch.linalg.svdvals(J)
    _, _, Vh = torch.linalg.svd(J, full_matrices=False)
    for k in (1, 16, 50):                            # 50 = the reference signature's default pca_rank
        V0 = torch.linalg.qr(torch.randn(256, k, generator=torch.Generator().manual_seed(k)))[0].T.contiguous()
        u, s, vT = net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=k, chunk_size=5,
                                                 min_iter=10, max_iter=60, convergence_threshold=1e-5, V0=V0)
        assert u.shape == (J.shape[0], k) and s.shape == (k,) and vT.shape == (k, 256)
        n = 1 if k == 1 else 8                      # leading part of the spectrum has converged
        if k == 50:
            assert torch.allclose(s.cpu()[:20], sv[:20], rtol=5e-3)
        assert torch.allclose(s.cpu()[:n], sv[:n], rtol=2e-3), (k, s.cpu()[:n], sv[:n])
        assert (abs_cos(vT[:n], Vh[:n]) > 0.99).all()
        assert torch.allclose((vT @ vT.T).cpu(), torch.eye(k), atol=1e-3)


def test_ranks_above_56_match_full_jacobian_svd(toy):
    """pca_rank 57..128 (round 6: round-robin Jacobi eigen-solve on sixteen waves, row-tiled apply kernel): the reference takes any
    pca_rank (utils.py:722, default 50); 96 is the largest rank whose eigenvector matrix lives in LDS, 128 the library's limit."""
    from diffusion_pullback_amd import PullbackUNet
    from oracle import unet_sd
    f, cfg, p, _ = toy
    net = PullbackUNet("sd", cfg, p, dtype=torch.float32, device=DEV, max_batch=1, max_rank=128, verbose=False)
    J = torch.autograd.functional.jacobian(
        lambda a: unet_sd.forward(p, cfg, a, f["t"], f["ctx"], stop=("mid", 0)).reshape(-1), f["z"]).reshape(-1, 256)
    sv = torch.linalg.svdvals(J)
    _, _, Vh = torch.linalg.svd(J, full_matrices=False)
    for k in (57, 96, 128):
        V0 = torch.linalg.qr(torch.randn(256, k, generator=torch.Generator().manual_seed(k)))[0].T.contiguous()
        u, s, vT = net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=k, chunk_size=25,
                                                 min_iter=10, max_iter=60, convergence_threshold=1e-5, V0=V0)
        assert u.shape == (J.shape[0], k) and s.shape == (k,) and vT.shape == (k, 256)
        assert torch.allclose(s.cpu()[:20], sv[:20], rtol=5e-3), (k, s.cpu()[:20], sv[:20])
        assert (abs_cos(vT[:8], Vh[:8]) > 0.99).all()
        assert torch.allclose((vT @ vT.T).cpu(), torch.eye(k), atol=1e-3)
        assert (s[:-1] >= s[1:]).all()              # descending
    with pytest.raises(ValueError):                 # the library's limit, not the engine's capacity
        PullbackUNet("sd", cfg, p, dtype=torch.float32, device=DEV, max_batch=1, max_rank=200, verbose=False).local_encoder_pullback_zt(
            f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=129)


def test_error_behaviour(toy):
    f, cfg, p, net = toy
    from diffusion_pullback_amd import DpbError
    with pytest.raises(ValueError):                 # reference message path, utils.py:527
        net.get_h(f["z"], f["t"], f["ctx"], op="down", block_idx=9)
    with pytest.raises(ValueError):
        net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=57)
    with pytest.raises(ValueError):
        net.local_encoder_pullback_zt(f["z"].repeat(2, 1, 1, 1), f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=2)
    with pytest.raises(DpbError):                   # wrong latent size
        net.get_h(torch.zeros(1, 4, 9, 9), f["t"], f["ctx"], op="mid", block_idx=0)
    with pytest.raises(DpbError):                   # conditional net without encoder_hidden_states
        net.get_h(f["z"], f["t"], None, op="mid", block_idx=0)
    with pytest.raises(DpbError):                   # batch above max_batch
        net(f["z"].repeat(3, 1, 1, 1), f["t"], f["ctx"].repeat(3, 1, 1))


def test_timestep_forms_and_chunking_agree(toy):
    f, cfg, p, net = toy
    h0 = net.get_h(f["z"], f["t"], f["ctx"], op="mid", block_idx=0)
    h1 = net.get_h(f["z"], float(f["t"]), f["ctx"], op="mid", block_idx=0)
    h2 = net.get_h(f["z"], f["t"].reshape(1), f["ctx"], op="mid", block_idx=0)
    assert torch.equal(h0, h1) and torch.equal(h0, h2)          # the default path is bitwise reproducible (fixed-order GroupNorm statistics)
    V0 = torch.linalg.qr(torch.randn(256, 6, generator=torch.Generator().manual_seed(1)))[0].T.contiguous()
    outs = [net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=6, chunk_size=c, min_iter=1, max_iter=3,
                                          convergence_threshold=1e-9, V0=V0) for c in (25, 2, 3)]     # 1, 3 and 2 chunks
    for u, s, vT in outs[1:]:
        assert torch.allclose(s, outs[0][1], rtol=1e-4) and torch.allclose(vT, outs[0][2], atol=1e-4)


def test_engine_is_bitwise_reproducible_by_default():
    """The DEFAULT path adds no floating-point numbers in an order decided at run time: GroupNorm statistics of the two-pass kernels are per-block
    partials added in block order by the apply launch, the one-launch kernel and split-K slabs are ordered by construction, the Gram matrix of the
    re-orthonormalisation is reduced from per-block partials in block order: repeated runs of the 16-bit engine give identical bits.  The round-1
    atomic statistics (dpb_debug_set("gn_deterministic", 0), A/B only) agree with them to 16-bit round-off."""
    import torch
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import lib as L
    from oracle import unet_sd
    cfg = unet_sd.SDConfig(block_out_channels=(320, 640), layers_per_block=1, down_attn=(True, True), up_attn=(True, True),
                           heads=(8, 8), cross_dim=768, sample_size=64, ctx_len=77)     # 64x64: two-pass GroupNorm; 32x32: one-launch kernel
    p = unet_sd.init_params(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(2, 4, 64, 64, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
    net = PullbackUNet("sd", cfg, p, dtype=torch.bfloat16, device="cuda:0", max_batch=2, max_rank=6, upto=("mid", 0), verbose=False)
    V0 = torch.linalg.qr(torch.randn(4 * 64 * 64, 3, generator=g))[0].T.contiguous()
    runs = []
    for _ in range(3):
        _, s, V, _ = net.pullback_fixed(z, 696.2727, ctx, "mid", 0, 3, 3, V0)
        runs.append((s.clone(), V.clone(), net.engine.read(("mid", 0)).clone()))
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))
    L.check(L.load().dpb_debug_set(b"gn_deterministic", 0))
    try:
        _, s, V, _ = net.pullback_fixed(z, 696.2727, ctx, "mid", 0, 3, 3, V0)
        assert torch.allclose(s, runs[0][0], rtol=2e-2) and rel(net.engine.read(("mid", 0)), runs[0][2]) < 2e-2
    finally:
        L.check(L.load().dpb_debug_set(b"gn_deterministic", 1))


def test_iterate_keeps_the_tap_tangent_on_the_device_bitwise():
    """Inside dpb_pullback_iterate the tap's tangent J V goes from the tangent pass straight into the adjoint pass (its buffer becomes the cotangent
    seed) and U is converted out by the LAST iteration only: bitwise the results of the fp32 NCHW round trip through U in every iteration
    (dpb_debug_set("iter_alias", 0)), in bf16 and fp32, one and two samples, and U is still J V_prev of the last iteration."""
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import lib as L
    from oracle import unet_sd
    lib = L.load()
    cfg = unet_sd.SDConfig(block_out_channels=(320, 640), layers_per_block=1, down_attn=(True, True), up_attn=(True, True),
                           heads=(8, 8), cross_dim=768, sample_size=32, ctx_len=77)
    p = unet_sd.init_params(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(2, 4, 32, 32, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
    V0 = torch.linalg.qr(torch.randn(4096, 3, generator=g))[0].T.contiguous()
    try:
        for dtype in (torch.bfloat16, torch.float32):
            net = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=2, max_rank=6, upto=("mid", 0), verbose=False)
            for B in (1, 2):
                out = {}
                for alias in (0, 1):
                    L.check(lib.dpb_debug_set(b"iter_alias", alias))
                    u, s_, v, _ = net.pullback_fixed(z[:B], 696.2727, ctx[:B], "mid", 0, 3, 4, V0)
                    out[alias] = (u.clone(), s_.clone(), v.clone())
                assert all(torch.equal(a, b) for a, b in zip(out[0], out[1])), (dtype, B)
                assert torch.isfinite(out[1][0]).all() and out[1][0].abs().sum() > 0
            del net
    finally:
        L.check(lib.dpb_debug_set(b"iter_alias", 1))


def test_graph_replay_of_the_power_iteration_matches_eager_launches():
    """dpb_debug_set("graph_iterate", 1): dpb_pullback_iterate captures one iteration as a hipGraph (after an eager one) and replays it --
    the launch sequence of an iteration is fixed for fixed buffers.  Same bits as eager launches (the default path is bitwise reproducible); capture needs a
    non-default stream (the legacy default stream cannot be captured: there the option is ignored)."""
    import torch
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import lib as L
    from oracle import unet_sd
    lib = L.load()
    f = load_golden("pullback_zt_tiny.pt")
    cfg = unet_sd.SDConfig(**f["cfg"])
    p = unet_sd.init_params(cfg, seed=f["seed"], gain=f["gain"])
    net = PullbackUNet("sd", cfg, p, dtype=torch.bfloat16, device="cuda:0", max_batch=1, max_rank=3, verbose=False)
    eng = net.engine
    tap = ("mid", 0)
    V0 = torch.linalg.qr(torch.randn(256, 3, generator=torch.Generator().manual_seed(1)))[0].T.contiguous().cuda()
    try:
        st = torch.cuda.Stream("cuda:0")
        out = {}
        with torch.cuda.stream(st):
            eng.primal(f["z"], f["t"], f["ctx"], tap)
            V = V0.clone(); U = torch.empty(3, eng.tap_numel(tap), device="cuda:0"); s = torch.empty(3, device="cuda:0"); conv = torch.empty(1, 2, device="cuda:0")
            eng._set_stream()
            for mode in (0, 1, 1):                       # eager, capture + replay, replay of the cached graph
                L.check(lib.dpb_debug_set(b"graph_iterate", mode))
                V.copy_(V0)
                L.check(lib.dpb_pullback_iterate(eng.h, eng.tape.taps[tap], V.data_ptr(), U.data_ptr(), s.data_ptr(), conv.data_ptr(), 3, 6))
                st.synchronize()
                out.setdefault(mode, []).append((V.clone(), U.clone(), s.clone()))
        for got in out[1]:
            for a, b in zip(out[0][0], got):
                assert torch.isfinite(a).all() and torch.equal(a, b)
    finally:
        L.check(lib.dpb_debug_set(b"graph_iterate", 0))


@pytest.mark.parametrize("heads", [8, 5], ids=["d40", "d64"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_shared_probability_attention_kernels_match_per_tangent_kernels(dtype, heads):
    """attn_adj_kv_shared_kernel (head dim 40: one block carries all cotangents of a sample, P computed once by producer waves) against the
    per-cotangent kernel of round 2 (dpb_debug_set("attn_shared", 0)) on the same inputs: full and ragged cotangent groups (k = 5, 4, 7 = 5 + 2,
    10), two samples, accumulate flags as the tape sets them.  The two round P / gS to 16 bit at slightly different places.  Head dim 40 (SD-1.x, the
    default route) and 64 (SD-2.x: 5 heads at C = 320; measured no faster there, so opt-in through bit 2 of the switch, which also moves the query-major
    pass to its multi-cotangent kernel -- both instantiations are held to the per-cotangent kernels here)."""
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import lib as L
    from oracle import unet_sd
    lib = L.load()
    cfg = unet_sd.SDConfig(block_out_channels=(320, 640), layers_per_block=1, down_attn=(True, True), up_attn=(True, True),
                           heads=(heads, heads), cross_dim=768, sample_size=32, ctx_len=77)     # 32x32 tokens = 1024; 8 heads of 40 or 5 of 64
    p = unet_sd.init_params(cfg, seed=3)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(2, 4, 32, 32, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
    net = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=2, max_rank=20, upto=("down", 0), verbose=False)
    e = net.engine
    tap = ("down", 0)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    try:
        for B, k in [(1, 5), (1, 4), (1, 7), (2, 5), (1, 10)]:
            V = torch.randn(B * k, 4 * 32 * 32, generator=g)
            U = torch.randn(B * k, e.tap_numel(tap), generator=g)
            out = {}
            on = 2 if heads == 8 else 6
            for bits in (0, on):
                L.check(lib.dpb_debug_set(b"attn_shared", bits))
                e.primal(z[:B], 696.2727, ctx[:B], tap)
                out[bits] = (e.jvp(tap, V).clone(), e.vjp(tap, U).clone())
            for a, b in zip(out[0], out[on]):
                assert torch.isfinite(b).all()
                assert rel(b, a) < tol, (B, k, rel(b, a))
                for i in range(B * k):                                  # every tangent / cotangent, not just the norm of the stack
                    assert rel(b[i], a[i]) < 2 * tol, (B, k, i, rel(b[i], a[i]))
    finally:
        L.check(lib.dpb_debug_set(b"attn_shared", 2))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_short_sequence_fused_attention_matches_the_materialised_path(dtype, monkeypatch):
    """L = 64 keys at head dim 160 (the 8x8 level of SD-1.x) runs the flash-style kernels with one 64-key stage and half of each block's waves
    past the end of the sequence (clamped rows, no stores): primal, tangent and adjoint against the materialised path (scores in HBM) that
    DPB_FUSED_ATTN_MIN_L=256 selects, one and two samples, k = 5 and a ragged k = 3."""
    from diffusion_pullback_amd import PullbackUNet
    from oracle import unet_sd
    cfg = unet_sd.SDConfig(block_out_channels=(1280,), layers_per_block=1, down_attn=(True,), up_attn=(True,), heads=(8,), cross_dim=768,
                           sample_size=8, ctx_len=77)                     # 8x8 tokens = 64, 8 heads of 160
    p = unet_sd.init_params(cfg, seed=5)
    g = torch.Generator().manual_seed(6)
    z = torch.randn(2, 4, 8, 8, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
    tap = ("mid", 0)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    nets = {}
    for name, min_l in (("materialised", "256"), ("fused", "64")):
        monkeypatch.setenv("DPB_FUSED_ATTN_MIN_L", min_l)
        nets[name] = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=2, max_rank=10, upto=tap, verbose=False)
    monkeypatch.delenv("DPB_FUSED_ATTN_MIN_L")
    for B, k in [(1, 5), (2, 5), (1, 3)]:
        V = torch.randn(B * k, 4 * 8 * 8, generator=g)
        U = torch.randn(B * k, nets["fused"].engine.tap_numel(tap), generator=g)
        out = {}
        for name, net in nets.items():
            e = net.engine
            e.primal(z[:B], 696.2727, ctx[:B], tap)
            out[name] = (e.read(tap).clone(), e.jvp(tap, V).clone(), e.vjp(tap, U).clone())
        for a, b in zip(out["materialised"], out["fused"]):
            assert torch.isfinite(b).all()
            assert rel(b, a) < tol, (B, k, rel(b, a))
    n_fused = nets["fused"].engine.stats()[0]
    n_mat = nets["materialised"].engine.stats()[0]
    assert n_fused < n_mat, (n_fused, n_mat)                              # launches of the last pass (the adjoint)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_flash_forward_deferred_rescale_under_peaked_scores(dtype, monkeypatch):
    """attn_fwd_kernel keeps a stale running max until a query of the wave outgrows it by 2^4 (round 6).  Here the q / k projections of a d = 40,
    L = 1024 layer (eight 128-key stages) are scaled up so that the scores spread over many log2 units: rows whose maximum sits in a late stage take
    the rescale branch long after the first stage, others never.  Primal, tangent and adjoint (which consume the (stale max, 1 / l) statistics) against
    the fp32 CPU oracle; the yardstick is the materialised path (three-pass softmax over scores stored in 16 bits, which is what limits IT at this score
    range): the fused kernels, whose scores stay in fp32 registers, have to be at least as close to the oracle."""
    from diffusion_pullback_amd import PullbackUNet
    from oracle import unet_sd
    from _util import oracle_jvp, oracle_vjp
    cfg = unet_sd.SDConfig(block_out_channels=(320,), layers_per_block=1, down_attn=(True,), up_attn=(True,), heads=(8,), cross_dim=768,
                           sample_size=32, ctx_len=77)                    # 32x32 tokens = 1024, 8 heads of 40
    p = unet_sd.init_params(cfg, seed=7)
    for n in list(p):
        if n.endswith("attn1.to_q.weight") or n.endswith("attn1.to_k.weight"):
            p[n] = p[n] * 3.0                                             # scores x 9
    g = torch.Generator().manual_seed(8)
    z = torch.randn(1, 4, 32, 32, generator=g); ctx = torch.randn(1, 77, 768, generator=g)
    tap = ("mid", 0)
    f = lambda a: unet_sd.forward(p, cfg, a, torch.tensor(696.2727), ctx, stop=tap)
    V = torch.randn(2, 4 * 32 * 32, generator=g)
    with torch.no_grad():
        h_ref = f(z)
    U = torch.randn(2, h_ref.numel(), generator=g)
    ref = (h_ref.reshape(-1), oracle_jvp(f, z, V), oracle_vjp(f, z, U))
    errs = {}
    launches = {}
    for name, min_l in (("materialised", "4096"), ("fused", "256")):
        monkeypatch.setenv("DPB_FUSED_ATTN_MIN_L", min_l)
        net = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=1, max_rank=2, upto=tap, verbose=False)
        e = net.engine
        e.primal(z, 696.2727, ctx, tap)
        out = (e.read(tap).reshape(-1).cpu(), e.jvp(tap, V.cuda()).cpu(), e.vjp(tap, U.cuda()).cpu())
        assert all(torch.isfinite(o).all() for o in out), name
        errs[name] = [rel(o, r) for o, r in zip(out, ref)]
        launches[name] = e.stats()[0]
        del net, e
    monkeypatch.delenv("DPB_FUSED_ATTN_MIN_L")
    print("relative errors vs the fp32 oracle (primal, tangent, adjoint):", errs)
    assert launches["fused"] < launches["materialised"], launches                  # the fused kernels did run
    for ef, em in zip(errs["fused"], errs["materialised"]):
        assert ef <= 1.25 * em + 2e-3, errs
    assert max(errs["fused"]) < (0.15 if dtype == torch.bfloat16 else 0.03), errs     # and stay sane in absolute terms at a 9x score range


def test_deferred_split_k_reduction_is_bitwise_the_separate_reduce_kernel():
    """Split-K products whose consumer is a one-launch GroupNorm or a LayerNorm leave their fp32 slabs to that kernel (no splitk_reduce_kernel launch,
    no 16-bit round trip of the tensor unless another op reads it).  The consumer adds the slabs in slab order, adds the residual and rounds exactly
    like the reduce kernel: tangent and adjoint passes at full SD-1.5 width (8x8 / 16x16 levels: every split-K shape of the headline) are bitwise
    identical with dpb_debug_set("lazy_reduce", 0 | 1), and fewer kernels are launched."""
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    from diffusion_pullback_amd import lib as L
    lib = L.load()
    enc = ("time_embedding", "conv_in", "down_blocks", "mid_block")
    params = cf.sd_init_params(cf.SD15, seed=0, only_prefix=enc, spectrum=cf.Spectrum())
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(2, 77, 768, generator=g); z = torch.randn(2, 4, 64, 64, generator=g)
    tap = ("mid", 0)
    for dtype in (torch.bfloat16, torch.float32):
        net = PullbackUNet("sd", cf.SD15, params, dtype=dtype, device="cuda:0", max_batch=2, max_rank=10, upto=tap, verbose=False)
        e = net.engine
        for B, k in [(1, 5), (2, 5), (1, 1)]:
            V = torch.randn(B * k, 16384, generator=g)
            U = torch.randn(B * k, e.tap_numel(tap), generator=g)
            out, launches = {}, {}
            try:
                for lazy in (0, 1):
                    L.check(lib.dpb_debug_set(b"lazy_reduce", lazy))
                    fw = e.forward(z[:B], 696.2727, ctx[:B], tap).clone()      # forward-only pass (primal products never defer)
                    e.primal(z[:B], 696.2727, ctx[:B], tap)
                    hp = e.read(tap).clone()
                    jv = e.jvp(tap, V).clone(); lj = e.stats()[0]
                    vj = e.vjp(tap, U).clone(); lv = e.stats()[0]
                    out[lazy], launches[lazy] = (jv, vj, hp, fw), (lj, lv)
            finally:
                L.check(lib.dpb_debug_set(b"lazy_reduce", 1))
            assert all(torch.equal(a, b) for a, b in zip(out[0], out[1])), (dtype, B, k)
            assert torch.equal(out[1][2], out[1][3])
            print(dtype, B, k, "launches (jvp, vjp): separate reduce", launches[0], "deferred", launches[1])
            if dtype == torch.bfloat16 and k == 5:
                assert launches[1][0] < launches[0][0] and launches[1][1] < launches[0][1]
        del net
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_layernorm_fused_into_product_epilogue_matches_separate_kernels(dtype):
    """The 320-wide products next to a LayerNorm run on a row-complete 128 x 320 ring tile whose epilogue applies the LayerNorm tangent (proj_in /
    to_out products: h and LN'(h) leave one launch) or adjoint (adjoints of the q / k / v, cross-attention q and FF-in products: the cotangent of the
    LayerNorm output is never stored).  Against the separate GEMM + LayerNorm kernels (dpb_debug_set("ln_fuse", 0)): same values up to the 16-bit
    rounding of different fp32 summation orders; one to k tangents, two samples, fewer launches."""
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import lib as L
    from oracle import unet_sd
    lib = L.load()
    cfg = unet_sd.SDConfig(block_out_channels=(320, 640), layers_per_block=1, down_attn=(True, True), up_attn=(True, True),
                           heads=(8, 8), cross_dim=768, sample_size=32, ctx_len=77)     # 32x32 tokens, C = 320 on the first level
    p = unet_sd.init_params(cfg, seed=5)
    g = torch.Generator().manual_seed(6)
    z = torch.randn(2, 4, 32, 32, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
    net = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=2, max_rank=10, upto=("down", 0), verbose=False)
    e = net.engine
    tap = ("down", 0)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    try:
        for B, k in [(1, 1), (1, 5), (2, 5)]:
            V = torch.randn(B * k, 4 * 32 * 32, generator=g)
            U = torch.randn(B * k, e.tap_numel(tap), generator=g)
            out, launches = {}, {}
            for fuse in (0, 1):
                L.check(lib.dpb_debug_set(b"ln_fuse", fuse))
                e.primal(z[:B], 696.2727, ctx[:B], tap)
                jv = e.jvp(tap, V).clone(); lj = e.stats()[0]
                vj = e.vjp(tap, U).clone(); lv = e.stats()[0]
                out[fuse], launches[fuse] = (jv, vj), (lj, lv)
            for a, b in zip(out[0], out[1]):
                assert torch.isfinite(b).all()
                for i in range(B * k):
                    assert rel(b[i], a[i]) < tol, (B, k, i, rel(b[i], a[i]))
            # three LayerNorms per transformer block: all three tangents fused; of the adjoints the two behind K <= 1024 products (q/k/v and
            # cross-attention q), the one behind the FF-in adjoint (K = 8 C) stays a separate kernel
            assert launches[1][0] == launches[0][0] - 3 and launches[1][1] == launches[0][1] - 2, launches
    finally:
        L.check(lib.dpb_debug_set(b"ln_fuse", 0))      # the default since round 6


def test_forward_pass_shortcuts_match_the_unfused_kernels():
    """Round 4, dpb_forward (the DDIM / guidance loop's U-Net call) at full SD-1.5 width, batch 2: (i) GEGLU in the epilogue of the unsplit FF-in
    products is BITWISE the product + GEGLU kernel (dpb_debug_set("geglu_fwd", 0)); (ii) the one-launch forward of the text-conditioned attention
    layers (P recomputed in fp32 registers, V^T built in LDS) agrees with the materialised GEMM + softmax + transpose + GEMM path
    (dpb_debug_set("cross_primal", 0)) to 16-bit rounding; both launch fewer kernels.  The net-wide fused temb / context projections are part of
    the tape in every arm (their parity is the oracle comparison of tests/test_gpu_fullsize.py)."""
    from diffusion_pullback_amd import PullbackUNet, configs as cf
    from diffusion_pullback_amd import lib as L
    lib = L.load()
    params = cf.sd_init_params(cf.SD15, seed=0, only_prefix=("time_embedding", "conv_in", "down_blocks"))
    g = torch.Generator().manual_seed(8)
    ctx = torch.randn(2, 77, 768, generator=g); z = torch.randn(2, 4, 64, 64, generator=g)
    tap = ("down", 2)
    net = PullbackUNet("sd", cf.SD15, params, dtype=torch.bfloat16, device="cuda:0", max_batch=2, max_rank=2, upto=tap, verbose=False)
    e = net.engine
    out = {}
    try:
        for key, (gf, cp) in {"both": (1, 1), "no_geglu": (0, 1), "no_cross": (1, 0)}.items():
            L.check(lib.dpb_debug_set(b"geglu_fwd", gf)); L.check(lib.dpb_debug_set(b"cross_primal", cp))
            out[key] = (e.forward(z, 696.2727, ctx, tap).clone(), e.stats()[0])
    finally:
        L.check(lib.dpb_debug_set(b"geglu_fwd", 1)); L.check(lib.dpb_debug_set(b"cross_primal", 1))
    assert torch.isfinite(out["both"][0]).all()
    assert torch.equal(out["both"][0], out["no_geglu"][0])
    assert rel(out["both"][0], out["no_cross"][0]) < 2e-2
    print("launches: all shortcuts", out["both"][1], "without the GEGLU epilogue", out["no_geglu"][1], "without the one-launch cross-attention", out["no_cross"][1])
    assert out["both"][1] < out["no_geglu"][1] and out["both"][1] < out["no_cross"][1]


def test_forward_only_pass_matches_the_stashing_pass_and_invalidates_it():
    """dpb_forward (the U-Net calls of the DDIM / guidance loop, edit.py:454-458): same output bits as dpb_primal + dpb_read_buffer, GEGLU inputs
    left untouched, and no stash -- dpb_jvp / dpb_vjp / dpb_pullback_iterate refuse until the next dpb_primal (no silent use of stale state)."""
    from diffusion_pullback_amd import PullbackUNet
    from diffusion_pullback_amd import lib as L
    from oracle import unet_sd
    f = load_golden("pullback_zt_tiny.pt")
    cfg = unet_sd.SDConfig(**f["cfg"])
    p = unet_sd.init_params(cfg, seed=f["seed"], gain=f["gain"])
    for dtype in (torch.float32, torch.bfloat16):
        net = PullbackUNet("sd", cfg, p, dtype=dtype, device="cuda:0", max_batch=2, max_rank=3, verbose=False)
        e = net.engine
        z = torch.cat([f["z"], f["z"].flip(-1)]); ctx = f["ctx"].expand(2, -1, -1)
        e.primal(z, float(f["t"]), ctx, "eps")
        a = e.read("eps").clone()
        b = e.forward(z, float(f["t"]), ctx, "eps")
        assert torch.equal(a, b)
        with pytest.raises(L.DpbError):
            e.jvp("eps", torch.randn(2, e.n_in))
        with pytest.raises(L.DpbError):
            e.vjp("eps", torch.randn(2, e.tap_numel("eps")))
        e.primal(z, float(f["t"]), ctx, ("mid", 0))                 # a stashing pass makes the engine differentiable again
        assert torch.isfinite(e.jvp(("mid", 0), torch.randn(2, e.n_in))).all()
        # forward() twice in a row: the first one must not have clobbered anything the second one reads (GEGLU factors are NOT written)
        assert torch.equal(e.forward(z, float(f["t"]), ctx, "eps"), b)
