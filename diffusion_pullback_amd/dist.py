"""Multi-GPU sharding of the pullback path: one process per GPU, independent samples, one final gather.

Each x_t sample's power iteration is independent (the reference itself launches one process per
sample pinned with --device cuda:N, src/scripts/main_celeba_hf_local_encoder_pullback.sh:2-9), so
samples are dealt round-robin to ranks, weights are replicated, and nothing is exchanged inside the
iterations.  The only collective is an ``all_gather`` of the final bases (u, s, vT) (plus a 24-byte one of their shapes when the
caller does not pass them) -- with the
``nccl`` backend that is RCCL over xGMI; <= 4 MB per sample, latency-bound, so one flat gather of a
packed tensor per call (not one per sample, and never a ring all-reduce inside the loop).
The same code runs under ``gloo`` on CPU tensors (tests/test_dist.py, world_size 2).

Fewer samples than GPUs (the editing CLI works on ONE image): ``k_sharded_power_iteration`` deals the k directions of one
sample to the ranks instead -- w_i = J^T J v_i is independent per direction -- with one all_gather of W [k, N_in] (<= 655 KB at
k = 10, SD latents) per iteration in front of the re-orthonormalisation, which every rank repeats on the full W.  Never a ring
all-reduce inside the loop.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_indices(n_samples: int, rank: int, world: int) -> List[int]:
    """sample i -> rank i mod world"""
    return list(range(rank, n_samples, world))


def gather_bases(local: Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], n_samples: int, group=None, shape=None):
    """local: {sample_idx: (u [N_h,k], s [k], vT [k,N_in])} for this rank's samples.
    Returns the full {idx: (u, s, vT)} on every rank with ONE all_gather of a packed buffer.  ``shape`` = (N_h, k, N_in) when the caller knows
    it (bench.py, the CLI); without it a second, tiny all_gather of the shapes precedes the payload (a rank that holds no sample cannot know them)."""
    if not dist.is_initialized():          # plain single-process use: nothing to exchange
        return dict(local)
    world = dist.get_world_size(group)      # an initialised group runs the collective even at world_size 1 (RCCL path testable on 1 GPU)
    rank = dist.get_rank(group)
    any_item = next(iter(local.values())) if local else None
    if any_item is not None:
        dev, dt = any_item[0].device, torch.float32
    else:
        dev, dt = _coll_device(), torch.float32
    if shape is not None:
        n_h, k, n_in = (int(v) for v in shape)
    else:
        meta = torch.zeros(3, dtype=torch.int64)
        if any_item is not None:
            u, s, vT = any_item
            meta = torch.tensor([u.shape[0], s.shape[0], vT.shape[1]], dtype=torch.int64)
        metas = [_to_coll(torch.zeros_like(meta)) for _ in range(world)]
        dist.all_gather(metas, _to_coll(meta), group=group)
        n_h, k, n_in = [int(v) for v in max(metas, key=lambda m: int(m.sum())).tolist()]
    per = (n_samples + world - 1) // world                  # slots per rank
    stride = n_h * k + k + k * n_in
    buf = torch.zeros(per * stride, dtype=dt, device=dev)
    for slot, idx in enumerate(shard_indices(n_samples, rank, world)):
        u, s, vT = local[idx]
        o = slot * stride
        buf[o:o + n_h * k] = u.to(dt).reshape(-1)
        buf[o + n_h * k:o + n_h * k + k] = s.to(dt)
        buf[o + n_h * k + k:o + stride] = vT.to(dt).reshape(-1)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    res = {}
    for r in range(world):
        for slot, idx in enumerate(shard_indices(n_samples, r, world)):
            o = slot * stride
            b = out[r]
            res[idx] = (b[o:o + n_h * k].reshape(n_h, k), b[o + n_h * k:o + n_h * k + k].clone(), b[o + n_h * k + k:o + stride].reshape(k, n_in))
    return res


def _coll_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _to_coll(t):
    return t.to(_coll_device())


def sharded_pullback(compute: Callable[[int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], n_samples: int, group=None):
    """Run ``compute(sample_idx) -> (u, s, vT)`` for this rank's samples, then gather all bases everywhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    local = {i: compute(i) for i in shard_indices(n_samples, rank, world)}
    return gather_bases(local, n_samples, group)


def k_shard(k: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous direction slice [lo, hi) of rank `rank`: ceil(k / world) per rank, the tail ranks may hold fewer or none."""
    per = (k + world - 1) // world
    lo = min(k, rank * per)
    return lo, min(k, lo + per)


def _all_gather_rows(x: torch.Tensor, k: int, group=None) -> torch.Tensor:
    """x: this rank's rows [hi - lo, n] of a [k, n] matrix split by k_shard -> the full [k, n] on every rank (ONE all_gather of equal,
    zero-padded pieces; the padding sits at the tail of the flattened result and is cut off)."""
    if not dist.is_initialized():
        return x
    world = dist.get_world_size(group)
    per = (k + world - 1) // world
    piece = torch.zeros(per, x.shape[1], dtype=x.dtype, device=x.device)
    piece[:x.shape[0]] = x
    out = torch.empty(world * per, x.shape[1], dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, piece, group=group)
    return out[:k]


def k_sharded_power_iteration(jtj: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
                              orth: Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                              V0: torch.Tensor, n_iters: int, group=None):
    """Subspace iteration on J^T J for ONE sample with the k directions dealt to the ranks.

    jtj(V_local [m, N_in]) -> (U_local [m, N_h], W_local [m, N_in]) = (J V, J^T J V) for this rank's m >= 1 directions (the engine's
    jvp + vjp on a primal every rank has run for the same sample); orth(W [k, N_in], V_prev [k, N_in]) -> (V, s, conv) is the
    re-orthonormalisation (engine.orth).  V0 [k, N_in] is the same on every rank.  Returns (U [k, N_h], s [k], V [k, N_in], conv),
    identical on every rank: per iteration one all_gather of W, at the end one of U.  (Reference loop: src/utils/utils.py:756-808.)"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    k = V0.shape[0]
    lo, hi = k_shard(k, rank, world)
    V = V0
    U_loc = s = conv = None
    n_h = None
    for _ in range(n_iters):
        if hi > lo:
            U_loc, W_loc = jtj(V[lo:hi].contiguous())
            n_h = U_loc.shape[1]
        else:                                              # more ranks than directions: this rank only takes part in the collectives
            W_loc = V.new_zeros(0, V.shape[1])
        W = _all_gather_rows(W_loc, k, group)
        V, s, conv = orth(W, V)
    if world > 1:                                          # J V of the last iteration, all directions
        nh = torch.tensor([n_h or 0], dtype=torch.int64, device=V.device)
        dist.all_reduce(nh, op=dist.ReduceOp.MAX, group=group)
        if U_loc is None or hi == lo:
            U_loc = V.new_zeros(0, int(nh.item()))
        U = _all_gather_rows(U_loc, k, group)
    else:
        U = U_loc
    return U, s, V, conv
