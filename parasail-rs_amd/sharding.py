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


def _recv_buffers(cache, key, like, world, rows=None):
    """`world` receive tensors of `rows` rows (default: all of `like`'s) with `like`'s trailing shape, dtype and device.  With a
    cache (a dict the caller keeps, e.g. one per double-buffer slot) they are views of tensors allocated ONCE at `like`'s full
    size and reused from step to step: a step of the 8-GPU cfg-4 exchange would otherwise take 8 x 20 MB from the allocator
    on the communication stream every time."""
    import torch
    need = like.shape[0] if rows is None else rows
    tail = tuple(like.shape[1:])
    if cache is None:
        return [torch.empty((need,) + tail, dtype=like.dtype, device=like.device) for _ in range(world)]
    have = cache.get(key)
    if (have is None or len(have) != world or have[0].shape[0] < need or tuple(have[0].shape[1:]) != tail
            or have[0].dtype != like.dtype or have[0].device != like.device):
        cap = max(need, like.shape[0])
        have = [torch.empty((cap,) + tail, dtype=like.dtype, device=like.device) for _ in range(world)]
        cache[key] = have
    return [b[:need] for b in have]


def gather_records(local, counts, dst=0, group=None, async_op=False, cache=None):
    """Gather per-rank record tensors ([n_g, 4] int32) to `dst` in rank order.

    counts[g] = rows held by rank g (known to every rank from the shard plan).  Equal counts use
    a single gather; ragged counts are padded to the maximum.  `cache`: see _recv_buffers (the caller must not reuse a
    cache while a gather into it is still in flight).  Returns (tensor_or_None, work)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mx = max(counts)
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    bufs = _recv_buffers(cache, "records", send, world) if rank == dst else None
    work = dist.gather(send, bufs, dst=dst, group=group, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        return torch.cat([bufs[g][: counts[g]] for g in range(world)], dim=0)
    if async_op:
        return finish, work
    return finish(), None


def gather_text(text, toff, counts, dst=0, group=None, async_op=False, cache=None):
    """Gather packed CIGAR text (the device-entry layout: one uint8 buffer + int64 offsets[n_g + 1] per rank) to `dst`.

    Two phases, like gather_strings: every rank learns every rank's byte count (one tiny all_gather + one host read, the
    only synchronisation), then the offsets and the text travel padded to the longest shard.  Returns (finish, works):
    finish() on `dst` gives (text of all ranks back to back, offsets[sum(counts) + 1]) in rank order, None elsewhere;
    with async_op the two gathers are in flight until every work in `works` has been waited for."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = counts[rank]
    total = toff[n_local:n_local + 1].clone()
    sizes = [torch.empty_like(total) for _ in range(world)]
    dist.all_gather(sizes, total, group=group)
    nbytes = [int(x.item()) for x in sizes]
    mx_b, mx_n = max(max(nbytes), 1), max(counts) + 1
    body = text[:mx_b] if text.shape[0] >= mx_b else torch.cat([text, text.new_zeros(mx_b - text.shape[0])])
    offs = toff[:mx_n] if toff.shape[0] >= mx_n else torch.cat([toff, toff.new_zeros(mx_n - toff.shape[0])])
    body, offs = body.contiguous(), offs.contiguous()
    # (receive buffers from the caller's cache, sized once for the sender's whole text capacity: the bytes differ per step)
    body_all = _recv_buffers(cache, "text", text if text.shape[0] >= mx_b else body, world, rows=mx_b) if rank == dst else None
    offs_all = _recv_buffers(cache, "text_off", offs, world) if rank == dst else None
    works = [dist.gather(offs, offs_all, dst=dst, group=group, async_op=async_op),
             dist.gather(body, body_all, dst=dst, group=group, async_op=async_op)]

    def finish():
        if rank != dst:
            return None
        parts, offsets, base = [], [offs_all[0].new_zeros(1)], 0
        for g in range(world):
            parts.append(body_all[g][: nbytes[g]])
            offsets.append(offs_all[g][1: counts[g] + 1] + base)
            base += nbytes[g]
        return torch.cat(parts), torch.cat(offsets)
    return finish, ([w for w in works if w is not None] if async_op else [])


def gather_strings(local_strings, dst=0, group=None):
    """Gather variable-length byte strings (CIGARs) to `dst` in rank order, two phases:
    lengths first (fixed width), then the concatenated bytes padded to the longest shard.
    Returns the full list on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    enc = [x if isinstance(x, (bytes, bytearray)) else x.encode() for x in local_strings]
    lens = torch.tensor([len(x) for x in enc], dtype=torch.int64, device=dev)
    meta = torch.tensor([len(enc), int(lens.sum().item()) if len(enc) else 0], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0].item()) for m in metas]
    nbytes = [int(m[1].item()) for m in metas]
    max_n, max_b = max(counts), max(nbytes)
    lens_p = torch.zeros(max_n, dtype=torch.int64, device=dev)
    lens_p[: len(enc)] = lens
    body = torch.zeros(max(max_b, 1), dtype=torch.uint8, device=dev)
    if nbytes[rank]:
        body[: nbytes[rank]] = torch.frombuffer(bytearray(b"".join(enc)), dtype=torch.uint8).to(dev)
    lens_all = [torch.empty_like(lens_p) for _ in range(world)] if rank == dst else None
    body_all = [torch.empty_like(body) for _ in range(world)] if rank == dst else None
    dist.gather(lens_p, lens_all, dst=dst, group=group)
    dist.gather(body, body_all, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for g in range(world):
        ls = lens_all[g][: counts[g]].cpu().tolist()
        raw = body_all[g][: nbytes[g]].cpu().numpy().tobytes()
        pos = 0
        for l in ls:
            out.append(raw[pos:pos + l].decode())
            pos += l
    return out
