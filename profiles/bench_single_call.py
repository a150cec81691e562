#!/usr/bin/env python3
"""Latency of the one-pair entry (`Aligner::align()` semantics, parasail_* function pointer)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs
pkg = g.load_pkg()
rng = np.random.default_rng(3)
q, r = random_seqs(rng, 2, 150, 150)
m = pkg.Matrix.create(b"ACGT", 2, -3)
for name, b in (("sw_striped_16", pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2).solution_width(16)),
                ("nw_striped_16", pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2).solution_width(16)),
                ("sg_striped_sat", pkg.Aligner.new().semi_global().matrix(m).gap_open(5).gap_extend(2)),
                ("nw_stats_striped_sat", pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2).use_stats()),
                ("sg_trace_striped_sat", pkg.Aligner.new().semi_global().matrix(m).gap_open(5).gap_extend(2).use_trace()),
                ("sg_table_striped_sat", pkg.Aligner.new().semi_global().matrix(m).gap_open(5).gap_extend(2).use_table()),
                ("sw_rowcol_striped_sat", pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2).use_last_rowcol())):
    al = b.build()
    for _ in range(20): al.align(q, r)
    t0 = time.perf_counter()
    for _ in range(300): res = al.align(q, r)
    dt = (time.perf_counter() - t0) / 300
    print("%-24s %.1f us per align() of 150x150" % (name, dt * 1e6))
