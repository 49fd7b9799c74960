"""Host-side mirror of the reference's U-Net plugin surface for the pullback path.

The reference attaches ``get_h`` / ``local_encoder_pullback_zt`` (Stable Diffusion) and
``get_h`` / ``local_encoder_pullback_xt`` (unconditional) onto a diffusers U-Net with
``types.MethodType`` (reference src/utils/utils.py:103-104, :326-337).  ``PullbackUNet`` is a
U-Net object exposing exactly those methods (same names, argument meaning, return layout and
error behaviour) on top of the HIP engine; ``bind`` attaches them onto an existing module.

Semantics kept from the reference (src/utils/utils.py:722-816 and :165-249):
  * returns ``u`` [N_h, k] as a transposed (non-contiguous) view whose columns are J V_prev of
    the last iteration (un-normalised), ``s`` = sqrt(singular values of J^T J V_prev), ``vT`` [k, N_in];
  * stop iff allclose(V_prev, V, atol=thr, rtol=1e-5) and i > min_iter, at most max_iter iterations;
  * prints the per-iteration ``torch.dist(V_prev, V)`` and the runtime like the reference;
  * invalid (op, block_idx) raises ValueError with the reference's message (utils.py:527).
Deliberate differences:
  * the primal forward runs once per call (x_t, t fixed) instead of inside every JVP/VJP;
  * all k tangents ride one batched pass; ``chunk_size`` is accepted and only bounds the batch;
  * V0 is drawn on the CPU generator (reproducible; the reference's device RNG draw is not) or injected;
  * each singular vector is signed for non-negative overlap with the previous iterate (LAPACK's
    sign is arbitrary), which only makes the reference's stop rule well defined;
  * pca_rank <= 128 (RANK_LIMIT: the k x k eigen-solve of the re-orthonormalisation lives in LDS; the reference's signature default is 50, its
    call sites use 2..10: main.py:33, BASELINE configs); an engine is built for `max_rank` tangents (default 56).
"""
from __future__ import annotations

import time
import types
from typing import Optional, Tuple

import torch

from . import lib as L
from .engine import Engine
from .tape import build_ddpm, build_sd

MAX_RANK = 56          # default tangent capacity of an engine (workspace sizing)
RANK_LIMIT = 128       # largest pca_rank of the library (csrc/kernels.h ORTH_MAX_RANK)


class UNetOutput:
    """Mirror of diffusers' UNet2DConditionOutput: the reference reads ``.sample`` (edit.py:454-458)."""

    def __init__(self, sample):
        self.sample = sample


def _t_float(t) -> float:
    if torch.is_tensor(t):
        return float(t.reshape(-1)[0].to(torch.float32).item())
    return float(torch.tensor(t, dtype=torch.float32).item())


class PullbackUNet:
    def __init__(self, kind: str, cfg, params, dtype=torch.float32, device="cuda:0", max_batch: int = 5,
                 max_rank: int = MAX_RANK, upto: Optional[Tuple[str, int]] = None, verbose: bool = True):
        assert kind in ("sd", "ddpm")
        self.kind, self.config, self.dtype_compute = kind, cfg, dtype
        self.device = torch.device(device)
        self.dtype = torch.float32                       # boundary dtype (what callers see)
        self.verbose = verbose
        if kind == "sd":
            tape = build_sd(cfg, params, dtype, self.device, upto)
            self.engine = Engine(tape, cfg.block_out_channels[0], True, False, cfg.in_channels, max_batch, max_rank)
            self.in_shape = (cfg.in_channels, cfg.sample_size, cfg.sample_size)
        else:
            tape = build_ddpm(cfg, params, dtype, self.device, upto)
            self.engine = Engine(tape, cfg.ch, False, True, cfg.in_channels, max_batch, max_rank)
            self.in_shape = (cfg.in_channels, cfg.resolution, cfg.resolution)
        self.max_rank = max_rank
        # False: local_encoder_pullback_zt / _xt run all k directions here (default).  None or a torch.distributed group: the directions of the
        # ONE sample are dealt to that group's ranks (every rank must make the same call; see _pullback and dist.k_sharded_power_iteration)
        self.k_shard_group = False

    # ------------------------------------------------------------------ feature map
    def _tap(self, op, block_idx):
        key = (op, block_idx)
        if key not in self.engine.tape.taps:
            raise ValueError(f"(op, block_idx) = ({op, block_idx}) is not valid")
        return key

    def get_h(self, sample=None, timestep=None, encoder_hidden_states=None, op=None, block_idx=None, verbose=False,
              x=None, t=None, **kwargs):
        """SD: get_h(sample, timestep, encoder_hidden_states, op, block_idx) (utils.py:438-441);
        uncond: get_h(x, t, op, block_idx) (utils.py:114-116).  Returns [B, C, H, W] features."""
        sample = x if sample is None else sample
        timestep = t if timestep is None else timestep
        key = self._tap(op, block_idx)
        h = self.engine.forward(sample, _t_float(timestep), encoder_hidden_states, key)
        if verbose:
            print(f"op : {op}, block_idx : {block_idx}, return h.shape : {h.shape}")
        return h.to(sample.dtype)

    def __call__(self, sample, timestep, encoder_hidden_states=None, **kwargs):
        eps = self.engine.forward(sample, _t_float(timestep), encoder_hidden_states, "eps").to(sample.dtype)
        return UNetOutput(eps) if self.kind == "sd" else eps

    # ------------------------------------------------------------------ power iteration
    def _pullback(self, x, t, ctx, op, block_idx, k, chunks, min_iter, max_iter, thr, V0):
        if x.shape[0] != 1:
            raise ValueError("local_encoder_pullback expects a single sample (batch 1), as the reference does")
        if not (1 <= k <= min(self.max_rank, RANK_LIMIT)):
            raise ValueError(f"pca_rank={k} outside [1, {min(self.max_rank, RANK_LIMIT)}] supported by the HIP engine (built for max_rank={self.max_rank} "
                             f"tangents; the library's limit is {RANK_LIMIT})")
        key = self._tap(op, block_idx)
        eng = self.engine
        n_in = eng.n_in
        # One sample on several GPUs (self.k_shard_group set, process group initialised, every rank called with the same inputs): this rank
        # runs the JVP / VJP of its slice of the k directions, one all_gather of W precedes the re-orthonormalisation (dist.py).  W -- and
        # with it V, s and the stop decision -- is identical on every rank.
        import torch.distributed as tdist
        from . import dist as pdist
        shard = self.k_shard_group is not False and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size(self.k_shard_group) > 1
        drawn = V0 is None
        if drawn:
            q, _ = torch.linalg.qr(torch.randn(n_in, k, dtype=torch.float))     # utils.py:750-753 (CPU generator)
            V0 = q.T
        V = V0.reshape(k, n_in).to(device=self.device, dtype=torch.float32).contiguous()
        if shard and drawn:
            # every rank drew from its OWN CPU generator: the ranks must start from one V0, or the row signs orth() aligns to V_prev -- and with
            # them the gathered u_i = +-J v_i -- are rank-specific.  Rank 0 of the group decides.
            Vc = V if tdist.get_backend(self.k_shard_group) == "nccl" else V.cpu()
            tdist.broadcast(Vc, src=tdist.get_global_rank(self.k_shard_group, 0) if self.k_shard_group is not None else 0, group=self.k_shard_group)
            V = Vc.to(self.device)
        time_s = time.time()
        eng.primal(x, _t_float(t), ctx, key)
        U = s = None
        self.last_history = []                                 # per-iteration ||V_prev - V||_2 (what the reference prints, utils.py:804)
        lo, hi = pdist.k_shard(k, tdist.get_rank(self.k_shard_group), tdist.get_world_size(self.k_shard_group)) if shard else (0, k)
        for i in range(max_iter):
            V_prev = V
            if hi > lo:
                U = torch.cat([eng.jvp(key, vi) for vi in V[lo:hi].chunk(chunks)], dim=0)
                W = torch.cat([eng.vjp(key, ui) for ui in U.chunk(chunks)], dim=0)
            else:
                U, W = V.new_zeros(0, eng.tap_numel(key)), V.new_zeros(0, n_in)
            if shard:
                W = pdist._all_gather_rows(W, k, self.k_shard_group)
            V, s, conv = eng.orth(W, V_prev)
            dist, viol = conv.tolist()                                            # the only host sync per iteration
            self.last_history.append(dist)
            if self.verbose:
                print(f"power method : {i}-th step convergence : ", dist)
            if viol <= thr and i > min_iter:
                if self.verbose:
                    print("reach convergence threshold : ", dist)
                break
        self.last_iters, self.last_dist = i + 1, dist          # introspection for bench.py's time-to-converged-basis leg
        if self.verbose:
            print("power method runtime ==", time.time() - time_s)
        if shard:
            U = pdist._all_gather_rows(U, k, self.k_shard_group)
        dt = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float32
        return U.T.to(dt), s.to(dt), V.to(dt)

    def local_encoder_pullback_zt(self, sample, timestep, encoder_hidden_states=None, op=None, block_idx=None,
                                  pca_rank=50, chunk_size=25, min_iter=10, max_iter=100, convergence_threshold=1e-3,
                                  V0=None):
        """Reference: src/utils/utils.py:722-816."""
        chunks = max(1, pca_rank // chunk_size)                                   # utils.py:761-764
        return self._pullback(sample, timestep, encoder_hidden_states, op, block_idx, pca_rank, chunks, min_iter, max_iter,
                              convergence_threshold, V0)

    def local_encoder_pullback_xt(self, x, t, op=None, block_idx=None, pca_rank=50, chunk_size=25, min_iter=10,
                                  max_iter=100, convergence_threshold=1e-3, V0=None):
        """Reference: src/utils/utils.py:165-249."""
        chunks = pca_rank // chunk_size if pca_rank % chunk_size == 0 else pca_rank // chunk_size + 1   # utils.py:178
        return self._pullback(x, t, None, op, block_idx, pca_rank, chunks, min_iter, max_iter, convergence_threshold, V0)

    def pullback_fixed(self, x, t, ctx, op, block_idx, pca_rank, n_iters, V0):
        """Fixed-iteration variant with no host synchronisation (what bench.py times)."""
        key = self._tap(op, block_idx)
        self.engine.primal(x, _t_float(t), ctx, key)
        b = x.shape[0]                                  # b samples advance together; V0 is [b*k, N] or [k, N] (shared)
        V = V0.reshape(-1, self.engine.n_in).to(device=self.device, dtype=torch.float32)
        if V.shape[0] == pca_rank and b > 1:
            V = V.repeat(b, 1)
        V, U, s, conv = self.engine.iterate(key, V.contiguous().clone(), n_iters)
        return U.T, s, V, conv


    def pullback_k_sharded(self, x, t, ctx, op, block_idx, pca_rank, n_iters, V0, group=None):
        """ONE sample on several GPUs: the pca_rank directions are dealt to the ranks of `group` (dist.k_sharded_power_iteration); every
        rank calls this with the same x, t, ctx, V0 and gets the same (u [N_h, k], s [k], vT [k, N_in], conv).  For the case the
        sample-sharded path cannot use all GPUs (fewer samples than ranks: the editing CLI's single image)."""
        from . import dist as pdist
        if x.shape[0] != 1:
            raise ValueError("pullback_k_sharded works on a single sample")
        key = self._tap(op, block_idx)
        eng = self.engine
        eng.primal(x, _t_float(t), ctx, key)
        V = V0.reshape(pca_rank, eng.n_in).to(device=self.device, dtype=torch.float32).contiguous()

        def jtj(Vl):
            U = eng.jvp(key, Vl)
            return U, eng.vjp(key, U)
        U, s, V, conv = pdist.k_sharded_power_iteration(jtj, eng.orth, V, n_iters, group)
        return U.T, s, V, conv


def bind(unet, kind: str, cfg, dtype=torch.float32, device="cuda:0", **kw) -> PullbackUNet:
    """Attach the HIP-backed methods onto an existing U-Net module, like the reference's
    ``types.MethodType`` injection (utils.py:103-104, :326-337).  ``unet.state_dict()`` uses diffusers' keys: ``UNet2DConditionModel`` for "sd"
    (consumed as is by tape.build_sd), ``UNet2DModel`` for "ddpm" (renamed by weights.ddpm_hf_to_vendored_names; the vendored
    naming of src/models/ddpm/diffusion.py is accepted unchanged)."""
    sd = {k: v.detach().cpu() for k, v in unet.state_dict().items()}
    if kind == "ddpm":                    # diffusers UNet2DModel keys (the reference's live path, utils.py:101-104) -> builder names
        from .weights import ddpm_hf_to_vendored_names
        sd = ddpm_hf_to_vendored_names(sd, cfg)
    impl = PullbackUNet(kind, cfg, sd, dtype, device, **kw)
    unet._dpb = impl
    unet.get_h = types.MethodType(lambda self, *a, **k: self._dpb.get_h(*a, **k), unet)
    if kind == "sd":
        unet.local_encoder_pullback_zt = types.MethodType(lambda self, *a, **k: self._dpb.local_encoder_pullback_zt(*a, **k), unet)
    else:
        unet.local_encoder_pullback_xt = types.MethodType(lambda self, *a, **k: self._dpb.local_encoder_pullback_xt(*a, **k), unet)
    return impl
