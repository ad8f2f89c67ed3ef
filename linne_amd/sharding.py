"""Multi-GPU plumbing: one process per GPU, frames (or whole tracks) sharded with NO data-path collective.

Frames are independent on the prediction path: frame f goes to rank f mod world, each rank
analyses its shard on its own GPU, and the only exchanges are control-plane ones -- a barrier around the timed
region, a MAX over the ranks' elapsed times, and (for whole streams) a gather of the serialised blocks to rank 0,
which interleaves them back into stream order.  torch.distributed is used for exactly that (backend "nccl" = RCCL
on the GPU box, "gloo" in CPU tests).
"""
import time


def shard_round_robin(num_items, rank, world):
    """indices of the items rank `rank` owns"""
    return list(range(rank, num_items, world))


def barrier_and_sync(dist=None, cuda_sync=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if cuda_sync is not None:
        cuda_sync()


def timed_steps(step_fn, steps, dist=None, cuda_sync=None, tensor_factory=None):
    """barrier + sync, `steps` calls of step_fn, barrier + sync; returns the MAX elapsed seconds over ranks"""
    barrier_and_sync(dist, cuda_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier_and_sync(dist, cuda_sync)
    dt = time.perf_counter() - t0
    return reduce_max(dt, dist, tensor_factory)


def reduce_max(value, dist=None, tensor_factory=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = tensor_factory([value]) if tensor_factory else torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_stream(local_blocks, num_items, dist):
    """rank 0 receives every rank's serialised blocks (python objects over the control plane) and returns them in item
    order; other ranks return None"""
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local_blocks, gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * num_items
    for r in range(world):
        for k, idx in enumerate(shard_round_robin(num_items, r, world)):
            out[idx] = gathered[r][k]
    return out
