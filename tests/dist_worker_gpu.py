"""Worker for tests/test_gpu_fullshape.py::test_two_ranks_share_one_gpu_hip_path: world_size ranks, gloo rendezvous,
every rank on GPU (local_rank % device_count).  Each rank aligns its shard with the HIP kernels through the C ABI
(device-resident entry), the 16-byte records are gathered to rank 0 (parasail-rs_amd/sharding.py) and compared with
the single-process result of the same library and with the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g                       # noqa: E402
import workloads as wl                            # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    pkg = g.load_pkg()
    from importlib import import_module
    sharding = import_module("parasail_rs_amd.sharding")

    # (a) uniform lengths, contiguous split: config 2's shape, 40 000 pairs
    n = 40000
    qbuf, qoff, rbuf, roff = wl.make_cfg2(n)                       # same inputs on every rank
    pm = pkg.Matrix.create(b"ACGT", 2, -3)
    cfg = pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 16, 0, pm.inner)
    bounds = sharding.shard_bounds_uniform(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]

    def run_shard(cfg, qbuf, qoff, rbuf, roff, lo, hi, mq, mr):
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in
             (qbuf[qoff[lo]:qoff[hi]], qoff[lo:hi + 1] - qoff[lo], rbuf[roff[lo]:roff[hi]], roff[lo:hi + 1] - roff[lo])]
        out = torch.zeros((hi - lo, 4), dtype=torch.int32, device=dev)
        pkg.align_batch_device(cfg, hi - lo, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), mq, mr,
                               out.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        return out.cpu()

    rec = run_shard(cfg, qbuf, qoff, rbuf, roff, lo, hi, 150, 150)
    assert "pmx_sw16_kernel" in pkg.lib.pmx_last_kernel().decode()
    counts = [bounds[k + 1] - bounds[k] for k in range(world)]
    got, _ = sharding.gather_records(rec, counts, dst=0)
    if rank == 0:
        whole = run_shard(cfg, qbuf, qoff, rbuf, roff, 0, n, 150, 150)
        assert got.shape == (n, 4) and (got == whole).all()
        orc = g.load_oracle()
        om = orc.Matrix.create("ACGT", 2, -3)
        want = orc.align_batch(orc.SW, qbuf[:150 * 4000], qoff[:4001], rbuf[:150 * 4000], roff[:4001], 5, 2, om)
        assert (got[:4000, :3].numpy() == want).all()

    # (b) mixed lengths, cell-balanced split: config 5's shape (per-pair queries here), 3 000 pairs
    q, rb, ro, _ = wl.make_cfg5(3000)
    qa = np.frombuffer(q, dtype=np.uint8)
    qb = np.tile(qa, 3000); qo = wl.uniform_offsets(3000, 1000)
    rl = ro[1:] - ro[:-1]
    bounds = sharding.shard_bounds_by_cells(np.full(3000, 1000), rl, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    cfg5 = pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 0, 4, pm.inner)          # sat, length-sorted processing order
    rec = run_shard(cfg5, qb, qo, rb, ro, lo, hi, 1000, 5000)
    counts = [bounds[k + 1] - bounds[k] for k in range(world)]
    got, _ = sharding.gather_records(rec, counts, dst=0)
    if rank == 0:
        whole = run_shard(cfg5, qb, qo, rb, ro, 0, 3000, 1000, 5000)
        assert (got == whole).all()
        idx = np.arange(0, 3000, 41)
        want = orc.align_stats_sample(orc.SW, idx, qb, qo, rb, ro, 5, 2, om)
        assert (got[idx, :3].numpy() == want[:, :3]).all()
    # (c) config 4's shape: records AND packed CIGAR text gathered to rank 0 (sharding.gather_text), 6 000 pairs, uneven split
    n4 = 6000
    qbuf, qoff, rbuf, roff = wl.make_cfg4(n4)
    cfg4 = pkg.pmx_config_t(pkg.MODE_SG, pkg.SG_ALL, 5, 2, 16, pkg.WANT_CIGAR, pm.inner)
    bounds = [0] + [n4 * (k + 1) // world + (7 if k + 1 < world else 0) for k in range(world)]
    lo, hi = bounds[rank], bounds[rank + 1]

    def run_cigar(lo, hi):
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in
             (qbuf[qoff[lo]:qoff[hi]], qoff[lo:hi + 1] - qoff[lo], rbuf[roff[lo]:roff[hi]], roff[lo:hi + 1] - roff[lo])]
        m = hi - lo
        out = torch.zeros((m, 4), dtype=torch.int32, device=dev)
        cap = 320 * m
        text = torch.zeros(cap, dtype=torch.uint8, device=dev)
        toff = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        pkg.align_batch_cigar_device(cfg4, m, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 250, 250,
                                     out.data_ptr(), text.data_ptr(), cap, toff.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        return out.cpu(), text.cpu(), toff.cpu()

    rec, text, toff = run_cigar(lo, hi)
    counts = [bounds[k + 1] - bounds[k] for k in range(world)]
    got, _ = sharding.gather_records(rec, counts, dst=0)
    fin, works = sharding.gather_text(text, toff, counts, dst=0, async_op=True)
    for w_ in works:
        w_.wait()
    if rank == 0:
        all_text, all_off = fin()
        wrec, wtext, wtoff = run_cigar(0, n4)
        assert (got == wrec).all()
        assert all_off.shape[0] == n4 + 1 and (all_off == wtoff).all()
        total = int(wtoff[-1])
        assert all_text.shape[0] == total and (all_text == wtext[:total]).all()
        raw = all_text.numpy().tobytes()
        idx = np.arange(0, n4, 97)
        texts, _ = orc.cigar_sample(orc.SG, idx, qbuf, qoff, rbuf, roff, 5, 2, om)
        assert all(raw[int(all_off[k]):int(all_off[k + 1])].decode() == texts[t] for t, k in enumerate(idx))
    dist.barrier()
    if rank == 0:
        print("dist gpu ok world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
