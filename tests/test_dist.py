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


# ---- one sample, k directions dealt to the ranks (dist.k_sharded_power_iteration): the collective logic with a small dense J on CPU
def _fake_problem():
    g = torch.Generator().manual_seed(7)
    J = torch.randn(30, 20, generator=g, dtype=torch.float64)

    def jtj(V):
        U = V @ J.T
        return U, U @ J

    def orth(W, Vp):
        q, r = torch.linalg.qr(W.T)
        sign = torch.sign(torch.diagonal(r))
        V = (q * sign).T.contiguous()
        return V, torch.linalg.norm(W, dim=1), torch.stack([(V - Vp).norm(), (V - Vp).abs().max()])
    return J, jtj, orth, g


def _k_worker(rank, world, port, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_pullback_amd.dist import k_sharded_power_iteration
    J, jtj, orth, g = _fake_problem()
    V0 = torch.linalg.qr(torch.randn(20, k, generator=g, dtype=torch.float64))[0].T.contiguous()
    calls = []

    def counted(V):
        calls.append(V.shape[0])
        return jtj(V)
    U, s, V, conv = k_sharded_power_iteration(counted, orth, V0, 4)
    q.put((rank, U.tolist(), s.tolist(), V.tolist(), calls))      # plain lists: a tensor's shared-memory handle can die with the worker
    dist.destroy_process_group()


def test_k_shard_slices():
    from diffusion_pullback_amd.dist import k_shard
    assert [k_shard(5, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 5), (5, 5)]
    assert [k_shard(10, r, 8) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 10), (10, 10), (10, 10)]
    assert k_shard(3, 0, 1) == (0, 3)


def test_two_rank_gloo_k_sharded_iteration_matches_single_process():
    from diffusion_pullback_amd.dist import k_sharded_power_iteration
    for k in (5, 1):                    # ragged (3 + 2 directions) and fewer directions than ranks (rank 1 only joins the collectives)
        J, jtj, orth, g = _fake_problem()
        V0 = torch.linalg.qr(torch.randn(20, k, generator=g, dtype=torch.float64))[0].T.contiguous()
        U1, s1, V1, _ = k_sharded_power_iteration(jtj, orth, V0, 4)          # no process group: plain loop
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_k_worker, args=(r, 2, port, k, q)) for r in range(2)]
        [p.start() for p in ps]
        out = [q.get(timeout=120) for _ in ps]
        [p.join(timeout=60) for p in ps]
        for rank, U, s, V, calls in out:
            U, s, V = (torch.tensor(v, dtype=torch.float64) for v in (U, s, V))
            assert torch.allclose(U, U1, atol=1e-10) and torch.allclose(s, s1, atol=1e-10) and torch.allclose(V, V1, atol=1e-10), (k, rank)
            lo, hi = (0, (k + 1) // 2) if rank == 0 else ((k + 1) // 2, k)
            assert calls == ([hi - lo] * 4 if hi > lo else []), (k, rank, calls)     # each rank only ran its own directions


class _FakeEngine:
    """The slice of Engine that PullbackUNet._pullback uses, over a small dense J on CPU (float64)."""
    def __init__(self):
        self.J, self._jtj, self._orth, _ = _fake_problem()
        self.n_in = 20

    def tap_numel(self, key):
        return 30

    def primal(self, *a):
        pass

    def jvp(self, key, V):                                  # float32 in and out, like the device engine
        return (V.double() @ self.J.T).float()

    def vjp(self, key, U):
        return (U.double() @ self.J).float()

    def orth(self, W, Vp):
        V, s, conv = self._orth(W.double(), Vp.double())
        return V.float(), s.float(), conv.float()


def _pullback_worker(rank, world, port, k, shard, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_pullback_amd.pullback import PullbackUNet
    net = object.__new__(PullbackUNet)                      # the loop only: no device engine on this machine
    net.engine, net.device, net.max_rank, net.verbose, net.k_shard_group = _FakeEngine(), torch.device("cpu"), 8, False, (None if shard else False)
    net._tap = lambda op, idx: "mid"
    g = torch.Generator().manual_seed(11)
    V0 = torch.linalg.qr(torch.randn(20, k, generator=g, dtype=torch.float64))[0].T.contiguous()
    u, s, vT = net._pullback(torch.zeros(1, 1), 1.0, None, "mid", 0, k, 1, 2, 6, 1e-30, V0.float())
    q.put((rank, u.tolist(), s.tolist(), vT.tolist(), net.last_iters))
    dist.destroy_process_group()


def test_reference_api_loop_with_k_sharding_matches_unsharded_on_two_ranks():
    """PullbackUNet._pullback (the local_encoder_pullback_zt / _xt loop with the reference's stop rule) with k_shard_group set, 2 gloo ranks,
    over a stand-in engine: same (u, s, vT) and the same iteration count on both ranks as the unsharded loop."""
    for k in (5, 1):
        res = {}
        for shard in (False, True):
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            port = _free_port()
            ps = [ctx.Process(target=_pullback_worker, args=(r, 2, port, k, shard, q)) for r in range(2)]
            [p.start() for p in ps]
            res[shard] = sorted([q.get(timeout=120) for _ in ps], key=lambda o: o[0])
            [p.join(timeout=60) for p in ps]
        ref = res[False][0]
        for out in res[True] + res[False]:
            assert out[4] == ref[4]
            assert all(torch.allclose(torch.tensor(a), torch.tensor(b), rtol=1e-4, atol=1e-4) for a, b in zip(out[1:4], ref[1:4])), (k, out[0])


def _pullback_v0_worker(rank, world, port, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_pullback_amd.pullback import PullbackUNet
    net = object.__new__(PullbackUNet)
    net.engine, net.device, net.max_rank, net.verbose, net.k_shard_group = _FakeEngine(), torch.device("cpu"), 8, False, None
    net._tap = lambda op, idx: "mid"
    torch.manual_seed(1000 + 17 * rank)                     # the ranks are NOT seeded alike: their own V0 draws differ
    u, s, vT = net._pullback(torch.zeros(1, 1), 1.0, None, "mid", 0, k, 1, 2, 5, 1e-30, None)
    q.put((rank, u.tolist(), s.tolist(), vT.tolist()))
    dist.destroy_process_group()


def test_k_sharded_loop_with_drawn_v0_is_identical_on_differently_seeded_ranks():
    """with k_shard_group set and V0=None every rank drew its own V0; rank 0's draw is broadcast now, so (u, s, vT) agree on
    all ranks and u stays sign-consistent with vT (u_i = J v_i of the previous iterate, checked through the converged basis)."""
    k = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_pullback_v0_worker, args=(r, 2, port, k, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=120) for _ in ps], key=lambda o: o[0])
    [p.join(timeout=60) for p in ps]
    (_, u0, s0, v0), (_, u1, s1, v1) = out
    for a, b in ((u0, u1), (s0, s1), (v0, v1)):
        assert torch.equal(torch.tensor(a), torch.tensor(b))
