#!/usr/bin/env python3
"""One long pair per call (queries beyond the packed kernels' 2048 rows run in the general kernel, one wave per pair)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(6)
m = pkg.Matrix.create(b"ACGT", 2, -3)
for L in (1000, 2000, 3000, 5000, 10000, 20000):
    q = random_seqs(rng, 1, L, L)[0]; r = mutate(rng, q, 0.08, 0.03)
    for name, b in (("sw_striped_sat", pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2)),
                    ("nw_striped_sat", pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2))):
        al = b.build()
        al.align(q, r)
        t0 = time.perf_counter(); res = al.align(q, r); t = time.perf_counter() - t0
        print("%-16s %6d x %6d: %9.2f ms  %7.2f GCUPS" % (name, L, len(r), t * 1e3, L * len(r) / t / 1e9), flush=True)
