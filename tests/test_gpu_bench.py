"""GPU tests of the measurement entry point and of the multi-GPU collective on the RCCL ("nccl") backend at world_size 1 (the 8-GPU runs
are the driver's): the packed basis gather on device tensors, bench.py under torchrun vs plain, and the strong-scaling
(BASELINE configs[3]) mode with a ragged last group."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(args, torchrun=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if torchrun:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gather_bases_on_nccl_world_size_1():
    import torch.distributed as dist
    from diffusion_pullback_amd.dist import gather_bases
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        g = torch.Generator().manual_seed(0)
        local = {i: (torch.randn(3, 40, generator=g).to(dev).T, torch.rand(3, generator=g).to(dev), torch.randn(3, 24, generator=g).to(dev)) for i in range(3)}
        res = gather_bases(local, 3)
        assert sorted(res) == [0, 1, 2]
        for i in range(3):
            assert res[i][0].is_cuda and all(torch.equal(a, b.contiguous()) for a, b in zip(res[i], local[i]))
    finally:
        dist.destroy_process_group()


def test_k_sharded_single_sample_pullback_on_nccl_world_size_1():
    """PullbackUNet.pullback_k_sharded (one sample, directions dealt to the ranks, one all_gather of W per iteration) on the RCCL backend at
    world_size 1 equals the fused on-device loop (pullback_fixed); the 2-rank split of the directions is covered on CPU (tests/test_dist.py)."""
    import torch.distributed as dist
    from diffusion_pullback_amd import PullbackUNet
    from oracle import unet_sd
    from _util import abs_cos, load_golden
    f = load_golden("pullback_zt_tiny.pt")
    cfg = unet_sd.SDConfig(**f["cfg"])
    p = unet_sd.init_params(cfg, seed=f["seed"], gain=f["gain"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    net = PullbackUNet("sd", cfg, p, dtype=torch.float32, device="cuda:0", max_batch=1, max_rank=4, verbose=False)
    V0 = torch.linalg.qr(torch.randn(256, 4, generator=torch.Generator().manual_seed(3)))[0].T.contiguous()
    u0, s0, V_0, _ = net.pullback_fixed(f["z"], f["t"], f["ctx"], "mid", 0, 4, 5, V0)
    u0, s0, V_0 = u0.clone(), s0.clone(), V_0.clone()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        u1, s1, V_1, _ = net.pullback_k_sharded(f["z"], f["t"], f["ctx"], "mid", 0, 4, 5, V0)
        net.k_shard_group = None                                            # the reference-API loop with sharding switched on (world 1: plain loop)
        u2, s2, V_2 = net.local_encoder_pullback_zt(f["z"], f["t"], f["ctx"], op="mid", block_idx=0, pca_rank=4, min_iter=1, max_iter=5,
                                                    convergence_threshold=1e-9, V0=V0)
        net.k_shard_group = False
    finally:
        dist.destroy_process_group()
    assert torch.allclose(s1, s0, rtol=1e-4) and (abs_cos(V_1, V_0) > 0.9999).all() and (abs_cos(u1.T, u0.T) > 0.9999).all()
    assert torch.allclose(s2.cpu(), s0.cpu(), rtol=1e-4) and (abs_cos(V_2, V_0) > 0.9999).all()


def test_bench_under_torchrun_matches_plain_run():
    common = ["--gpus", "1", "--steps", "240", "--warmup", "12", "--workload", "toy", "--k", "3", "--no-cpu-baseline", "--no-roofline"]
    plain = _bench(common)
    tr = _bench(common, torchrun=True)
    assert plain["config"]["rccl_world_size"] == 0 and tr["config"]["rccl_world_size"] == 1
    assert plain["scaling"] == tr["scaling"] == "weak" and plain["finite"] and tr["finite"] and plain["n_gpus"] == tr["n_gpus"] == 1
    assert all(abs(a - b) <= 3e-2 * abs(a) for a, b in zip(plain["s_top"], tr["s_top"])), (plain["s_top"], tr["s_top"])
    assert 0.5 < tr["value"] / plain["value"] < 2.0, (tr["value"], plain["value"])     # same 20-sample job + one world_size-1 packed all_gather


def test_bench_strong_scaling_mode_shards_samples():
    r = _bench(["--gpus", "1", "--samples", "5", "--samples-per-gpu", "2", "--k", "3", "--ctx", "edit", "--workload", "toy", "--warmup", "12",
                "--no-cpu-baseline", "--no-roofline"], torchrun=True)
    assert r["scaling"] == "strong" and r["steps"] == 60 and r["config"]["samples_total"] == 5 and r["config"]["samples_this_rank"] == 5
    assert r["config"]["ctx"] == "edit" and r["finite"] and r["value"] > 0
