"""N>1 path on CPU: two ranks over gloo (one process per rank, launched like the driver launches
bench.py), shard plan + gather of records checked against the single-process oracle result."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_properties(pkg):
    from importlib import import_module
    sh = import_module("parasail_rs_amd.sharding")
    assert sh.shard_bounds_uniform(10, 4) == [0, 2, 5, 7, 10]
    assert sh.shard_bounds_uniform(3, 8)[-1] == 3
    rng = np.random.default_rng(1)
    ql = rng.integers(1, 1000, size=500); rl = rng.integers(500, 5000, size=500)
    for world in (1, 2, 4, 8):
        b = sh.shard_bounds_by_cells(ql, rl, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == 500 and sorted(b) == b
        cells = ql.astype(np.int64) * rl
        per = [cells[b[k]:b[k + 1]].sum() for k in range(world)]
        assert max(per) - min(per) <= 2 * cells.max()
    b = sh.shard_bounds_by_cells([5], [5], 4)
    assert b[0] == 0 and b[-1] == 1 and sorted(b) == b


def test_c_planner_matches_the_python_planner(pkg):
    """pmx_shard_bounds_by_cells (the planner behind pmx_align_batch_multi; pure host arithmetic, no GPU) cuts where
    parasail-rs_amd/sharding.py cuts: contiguous blocks, input order, balanced to within two pairs' cells"""
    from importlib import import_module
    sh = import_module("parasail_rs_amd.sharding")
    rng = np.random.default_rng(2)
    for n in (1, 7, 500, 4001):
        ql = rng.integers(1, 1000, size=n); rl = rng.integers(500, 5000, size=n)
        qoff = np.zeros(n + 1, dtype=np.int64); np.cumsum(ql, out=qoff[1:])
        roff = np.zeros(n + 1, dtype=np.int64); np.cumsum(rl, out=roff[1:])
        for parts in (1, 2, 3, 8):
            b = pkg.shard_bounds_by_cells(qoff, roff, parts)
            assert b == sh.shard_bounds_by_cells(ql, rl, parts), (n, parts)
            assert b[0] == 0 and b[-1] == n and sorted(b) == b
            cells = ql.astype(np.int64) * rl
            per = [int(cells[b[k]:b[k + 1]].sum()) for k in range(parts)]
            assert max(per) - min(per) <= 2 * int(cells.max())
            # a shared query: the reference lengths alone decide
            bs = pkg.shard_bounds_by_cells(None, roff, parts)
            assert bs == sh.shard_bounds_by_cells(np.ones(n, dtype=np.int64), rl, parts)
    # ordering: concatenating the blocks restores the input order
    b = pkg.shard_bounds_by_cells(qoff, roff, 5)
    assert np.concatenate([np.arange(b[k], b[k + 1]) for k in range(5)]).tolist() == list(range(n))


def test_two_ranks_gloo(orc, pkg):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "dist ok world=2" in p.stdout
