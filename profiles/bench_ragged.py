#!/usr/bin/env python3
"""Ragged short-read batch through the host API: query lengths uniform 30..150, reference windows = query length + 0..40."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import DNA
pkg = g.load_pkg()
rng = np.random.default_rng(11)
n = 1000000
ql = rng.integers(30, 151, size=n); rl = ql + rng.integers(0, 41, size=n)
qoff = np.zeros(n + 1, dtype=np.int64); np.cumsum(ql, out=qoff[1:])
roff = np.zeros(n + 1, dtype=np.int64); np.cumsum(rl, out=roff[1:])
qbuf = DNA[rng.integers(0, 4, size=int(qoff[-1]))]; rbuf = DNA[rng.integers(0, 4, size=int(roff[-1]))]
cells = int((ql * rl).sum())
al = pkg.Aligner.new().local().matrix(pkg.Matrix.create(b"ACGT", 2, -3)).gap_open(5).gap_extend(2).solution_width(16).build()
al.align_batch_packed(qbuf, qoff, rbuf, roff)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); out = al.align_batch_packed(qbuf, qoff, rbuf, roff); ts.append(time.perf_counter() - t0)
print("ragged 30..150 bp, 1M pairs, %.2e cells: %.1f ms -> %.0f GCUPS host API (%s), checksum %d" %
      (cells, min(ts) * 1e3, cells / min(ts) / 1e9, pkg.lib.pmx_last_kernel().decode(), int(out["score"].sum())))
