"""world_size-2 gloo test (CPU) of the multi-GPU path: round-robin sample sharding + one all_gather of the bases.
The per-sample compute is the CPU oracle here (the product compute needs a GPU); the sharding/gather code is the product's."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_pullback_amd.dist import gather_bases, shard_indices, sharded_pullback


def _compute(i):
    g = torch.Generator().manual_seed(100 + i)
    k, n_h, n_in = 3, 20, 12
    return torch.randn(k, n_h, generator=g).T, torch.rand(k, generator=g), torch.randn(k, n_in, generator=g)


def _worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = sharded_pullback(_compute, n_samples)
    ok = sorted(res) == list(range(n_samples))
    for i in range(n_samples):
        u, s, vT = _compute(i)
        ok = ok and torch.equal(res[i][0], u.contiguous()) and torch.equal(res[i][1], s) and torch.equal(res[i][2], vT)
    q.put((rank, ok))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_indices():
    assert shard_indices(5, 0, 2) == [0, 2, 4] and shard_indices(5, 1, 2) == [1, 3]
    assert shard_indices(1, 3, 8) == [] and sum((shard_indices(64, r, 8) for r in range(8)), []).__len__() == 64


def test_single_process_gather_is_identity():
    local = {0: _compute(0)}
    assert gather_bases(local, 1)[0][1].equal(local[0][1])


def test_two_rank_gloo_gather():
    for n_samples in (5, 1):            # ragged (3+2) and fewer samples than ranks
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, n_samples, q)) for r in range(2)]
        [p.start() for p in ps]
        out = [q.get(timeout=120) for _ in ps]
        [p.join(timeout=60) for p in ps]
        assert all(ok for _, ok in out), out
