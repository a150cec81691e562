"""Multi-GPU sharding of a batch of independent pairs (one process per GPU).

The reference has no distributed layer: pairs are independent and its only parallel story is user
threads sharing a read-only profile (/root/reference/tests/test_parasail.rs:689-723).  So the path
shards with no data-path collective: rank g aligns a contiguous block of pairs.  The one exchange
step is the gather of the fixed-width result records to rank 0 (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).
"""
import numpy as np


def shard_bounds_uniform(n, world):
    """Contiguous split of n pairs: rank g gets [b[g], b[g+1])."""
    return [(n * g) // world for g in range(world + 1)]


def shard_bounds_by_cells(qlens, rlens, world):
    """Contiguous split that balances sum(qlen*rlen) (mixed-length batches).  Pairs keep their
    input order, so concatenating the per-rank records restores it."""
    cells = np.asarray(qlens, dtype=np.int64) * np.asarray(rlens, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(cells)])
    total = int(cum[-1])
    bounds = [0]
    for g in range(1, world):
        target = total * g // world
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(max(k, bounds[-1]), len(cells))
        bounds.append(k)
    bounds.append(len(cells))
    return bounds


def gather_records(local, counts, dst=0, group=None, async_op=False):
    """Gather per-rank record tensors ([n_g, 4] int32) to `dst` in rank order.

    counts[g] = rows held by rank g (known to every rank from the shard plan).  Equal counts use
    a single gather; ragged counts are padded to the maximum.  Returns (tensor_or_None, work)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mx = max(counts)
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    work = dist.gather(send, bufs, dst=dst, group=group, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        return torch.cat([bufs[g][: counts[g]] for g in range(world)], dim=0)
    if async_op:
        return finish, work
    return finish(), None
