"""Multi-GPU use of the engine: independent environment replicas, one process per GPU.

One environment is a single coupled grid, so the hot path does not shard: N GPUs run N environments.
The only cross-GPU traffic the north star asks for is an optional gather of the batched observation tensor, so that
one policy batch can see every replica.  `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI inside a node); the
same code runs on "gloo" CPU tensors, which is how the CPU tests cover it.
"""
import os

import torch
import torch.distributed as dist


def replica_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: a single replica)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def replica_seed(base_seed, rank):
    """engine seed of replica `rank`: replicas must not be clones of each other"""
    return base_seed + rank


def gather_counts(n, device=None):
    """all-gather the per-replica agent counts (they differ after deaths); returns a python list"""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [int(n)]
    mine = torch.tensor([int(n)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [int(t.item()) for t in out]


def gather_observations(view, n, capacity=None):
    """All-gather the first `n` rows of this replica's observation tensor.

    view     : [capacity_local, ...] tensor whose first n rows are valid (device tensor for nccl, CPU for gloo)
    capacity : common row capacity of the exchange buffers (default: max n over replicas).  RCCL all_gather needs
               equal shapes, so shards are exchanged padded and trimmed afterwards; on the 8 x MI355X full mesh
               every pair has its own xGMI link, so the all-gather is one hop per peer.
    returns  : (list of per-replica tensors trimmed to their own n, list of counts)
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    counts = gather_counts(n, device=view.device)
    if world == 1:
        return [view[:n]], counts
    cap = int(capacity) if capacity is not None else max(counts)
    assert cap >= max(counts), "capacity smaller than a replica's agent count"
    if view.shape[0] >= cap:
        send = view[:cap].contiguous()
    else:
        send = torch.zeros((cap,) + tuple(view.shape[1:]), dtype=view.dtype, device=view.device)
        send[:n] = view[:n]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    return [r[:c] for r, c in zip(recv, counts)], counts


def max_over_replicas(seconds, device=None):
    """wall time of the slowest replica (the bench's timing rule)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_replicas(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
