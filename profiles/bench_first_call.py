#!/usr/bin/env python3
"""Cold-start cost: library load, first single-pair call, first batch call in a fresh process."""
import os, sys, time
t0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_pkg()
t1 = time.perf_counter()
al = pkg.Aligner.new().local().matrix(pkg.Matrix.create(b"ACGT", 2, -3)).gap_open(5).gap_extend(2).solution_width(16).build()
r = al.align(b"ACGTACGTACGT" * 10, b"ACGTTCGTACGT" * 10)
t2 = time.perf_counter()
r = al.align(b"ACGTACGTACGT" * 10, b"ACGTTCGTACGT" * 10)
t3 = time.perf_counter()
got = al.align_batch([b"ACGTACGTACGT" * 10] * 5000, [b"ACGTTCGTACGT" * 10] * 5000)
t4 = time.perf_counter()
print("load %.0f ms, first align() %.0f ms, second align() %.2f ms, first 5000-pair batch %.1f ms" %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
