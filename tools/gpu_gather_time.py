"""Not a test: cost of the final basis gather (dist.gather_bases) and of the closing barrier on the nccl (= RCCL) backend, per call.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29519 tools/gpu_gather_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from diffusion_pullback_amd import dist as pdist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ["LOCAL_RANK"])
dev = torch.device("cuda", local); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
n_h, k, n_in, per_rank = 81920, 5, 16384, 2
n_total = per_rank * world
loc = {i: (torch.randn(k, n_h, device=dev).T, torch.randn(k, device=dev), torch.randn(k, n_in, device=dev)) for i in pdist.shard_indices(n_total, rank, world)}
for it in range(6):
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    res = pdist.gather_bases(loc, n_total)
    torch.cuda.synchronize(dev); t1 = time.perf_counter()
    dist.barrier()
    torch.cuda.synchronize(dev); t2 = time.perf_counter()
    if rank == 0:
        print(f"call {it}: gather_bases {1e3 * (t1 - t0):7.2f} ms   barrier {1e3 * (t2 - t1):7.2f} ms", flush=True)
assert len(res) == n_total
dist.destroy_process_group()
