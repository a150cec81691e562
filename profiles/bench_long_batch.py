#!/usr/bin/env python3
"""Batches of long pairs (queries beyond the packed kernels' 2048 rows): the one-wave-per-pair general kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(7)
m = pkg.Matrix.create(b"ACGT", 2, -3)
for L, n in ((3000, 1024), (5000, 512)):
    qs = random_seqs(rng, n, L, L); rs = [mutate(rng, q, 0.08, 0.03) for q in qs]
    for name, b in (("sw_striped_sat", pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2)),
                    ("nw_striped_sat", pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2))):
        al = b.build()
        al.align_batch(qs[:70], rs[:70])
        t0 = time.perf_counter(); got = al.align_batch(qs, rs); t = time.perf_counter() - t0
        cells = sum(len(q) * len(r) for q, r in zip(qs, rs))
        print("%-16s %5d pairs of %d x ~%d: %8.1f ms  %7.1f GCUPS (%s)" % (name, n, L, L, t * 1e3, cells / t / 1e9, pkg.lib.pmx_last_kernel().decode()), flush=True)
