#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (pmx_align_batch) on BASELINE config 2:
H2D of 316 MB + kernel + D2H of 16 MB.  Never the bench.py `value` (that one has inputs in HBM)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g, bench
pkg = g.load_pkg()
qbuf, qoff, rbuf, roff = bench.make_cfg2_inputs()
al = pkg.Aligner.new().local().matrix(pkg.Matrix.create(b"ACGT", 2, -3)).gap_open(5).gap_extend(2).solution_width(16).build()
al.align_batch_packed(qbuf, qoff, rbuf, roff)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = al.align_batch_packed(qbuf, qoff, rbuf, roff); ts.append(time.perf_counter() - t0)
t = min(ts)
print("cfg2 through pmx_align_batch (host buffers): %.1f ms -> %.0f GCUPS PCIe-inclusive" % (t * 1e3, 1e6 * 22500 / t / 1e9))
