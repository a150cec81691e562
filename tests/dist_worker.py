"""Worker for tests/test_dist.py: world_size ranks over gloo on CPU.  Exercises the N>1 path of
the product -- the shard plan and the gather of result records (parasail-rs_amd/sharding.py) --
with the CPU oracle standing in for the GPU kernel (there is no GPU in the CPU test tier)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g                       # noqa: E402
from util import random_seqs, mutate              # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g.load_pkg()
    from importlib import import_module
    sharding = import_module("parasail_rs_amd.sharding")
    orc = g.load_oracle()
    om = orc.Matrix.create("ACGT", 2, -3)

    rng = np.random.default_rng(99)               # same inputs on every rank
    qs = random_seqs(rng, 301, 20, 200)
    rs = [mutate(rng, q, 0.1, 0.05) if i % 3 else random_seqs(rng, 1, 20, 600)[0] for i, q in enumerate(qs)]
    ql, rl = [len(x) for x in qs], [len(x) for x in rs]

    for plan in ("cells", "uniform"):
        bounds = sharding.shard_bounds_by_cells(ql, rl, world) if plan == "cells" \
            else sharding.shard_bounds_uniform(len(qs), world)
        assert bounds[0] == 0 and bounds[-1] == len(qs) and all(a <= b for a, b in zip(bounds, bounds[1:]))
        lo, hi = bounds[rank], bounds[rank + 1]
        qb, qo = orc.pack(qs[lo:hi]); rb, ro = orc.pack(rs[lo:hi])
        local = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
        rec = torch.from_numpy(np.concatenate([local, np.zeros((hi - lo, 1), np.int32)], axis=1))
        counts = [bounds[k + 1] - bounds[k] for k in range(world)]
        for async_op in (False, True):
            out, work = sharding.gather_records(rec, counts, dst=0, async_op=async_op)
            if async_op:
                work.wait()
                out = out()
            if rank == 0:
                qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
                want = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
                assert out.shape == (len(qs), 4)
                assert (out[:, :3].numpy() == want).all(), plan
            else:
                assert out is None
        if plan == "cells" and world > 1:
            cells = np.array(ql, dtype=np.int64) * np.array(rl, dtype=np.int64)
            per = [int(cells[bounds[k]:bounds[k + 1]].sum()) for k in range(world)]
            assert max(per) - min(per) <= 2 * int(cells.max()), per      # balanced to within two pairs
    # variable-length results (CIGAR strings): two-phase gather restores input order
    bounds = sharding.shard_bounds_by_cells(ql, rl, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    local = []
    for k in range(lo, min(hi, lo + 40)):
        local.append(orc.cigar(orc.align(orc.SG, qs[k], rs[k], 5, 2, om, trace=True)))
    got = sharding.gather_strings(local, dst=0)
    if rank == 0:
        want = []
        for rk in range(world):
            for k in range(bounds[rk], min(bounds[rk + 1], bounds[rk] + 40)):
                want.append(orc.cigar(orc.align(orc.SG, qs[k], rs[k], 5, 2, om, trace=True)))
        assert got == want
    else:
        assert got is None
    assert sharding.gather_strings([], dst=0) in ([], None)
    # packed text + offsets (the device entry's layout): ragged shards, text buffers larger than what was written
    counts = [min(bounds[rk + 1], bounds[rk] + 40) - bounds[rk] + 3 * rk for rk in range(world)]     # known to every rank from the plan
    mine = [("%d:%s" % (rank, c)).encode() for c in (local + ["1=", "2X", "3="] * world)[: counts[rank]]]
    body = np.frombuffer(b"".join(mine), dtype=np.uint8)
    toff = np.zeros(counts[rank] + 1, dtype=np.int64); np.cumsum([len(x) for x in mine], out=toff[1:])
    text = torch.from_numpy(np.concatenate([body, np.full(100 + rank, 0x55, dtype=np.uint8)]))       # capacity > bytes written
    for async_op in (False, True):
        fin, works = sharding.gather_text(text, torch.from_numpy(toff), counts, dst=0, async_op=async_op)
        for w in works:
            w.wait()
        res = fin()
        if rank == 0:
            all_text, all_off = res
            raw = all_text.numpy().tobytes()
            items = [raw[int(all_off[k]):int(all_off[k + 1])].decode() for k in range(sum(counts))]
            assert int(all_off[0]) == 0 and int(all_off[-1]) == len(raw)
            pos = 0
            for rk in range(world):
                assert all(x.startswith("%d:" % rk) for x in items[pos:pos + counts[rk]]), rk
                pos += counts[rk]
            assert items[: counts[0]] == [x.decode() for x in mine]
        else:
            assert res is None
    dist.barrier()
    if rank == 0:
        print("dist ok world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
