#!/usr/bin/env python3
"""Banded global alignment with a band wider than the band-only kernels take (k > 63): the general kernel, which sweeps only the
columns a 64-row band of the query can reach."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(5)
m = pkg.Matrix.create(b"ACGT", 2, -3)
for L, k, n in ((5000, 100, 64), (20000, 200, 16), (1000, 100, 2000)):
    qs = random_seqs(rng, n, L, L); rs = [mutate(rng, q, 0.08, 0.03) for q in qs]
    al = pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2).build()
    al.align_batch_banded(qs[:2], rs[:2], k)
    t0 = time.perf_counter(); got = al.align_batch_banded(qs, rs, k); t = time.perf_counter() - t0
    print("nw banded k=%d, %d pairs of %d x ~%d: %.1f ms (%s)" % (k, n, L, L, t * 1e3, pkg.lib.pmx_last_kernel().decode()))
