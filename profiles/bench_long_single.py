#!/usr/bin/env python3
"""One long pair per `Aligner::align()` call: pmx_long32_kernel (the query's 256-row bands spread over the chip), with the time
of the CPU for the same pair beside it: local -- the striped int16 CPU port (oracle/pmx_striped_cpu.c, one core: one pair is one
thread's work in the reference too); global -- the scalar oracle (no vectorised CPU port of nw exists in this repo)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
orc = g.load_oracle()
rng = np.random.default_rng(6)
m = pkg.Matrix.create(b"ACGT", 2, -3)
om = orc.Matrix.create("ACGT", 2, -3)
for L in (1000, 2000, 5000, 20000, 100000) if os.environ.get("PMX_LONG_QUICK") else (1000, 2000, 3000, 5000, 10000, 20000, 50000, 100000):
    q = random_seqs(rng, 1, L, L)[0]; r = mutate(rng, q, 0.08, 0.03)
    for name, mode, b in (("sw_striped_sat", 2, pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2)),
                          ("nw_striped_sat", 0, pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2))):
        al = b.build()
        al.align(q, r)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); res = al.align(q, r); ts.append(time.perf_counter() - t0)
        t = min(ts)
        kernel = pkg.lib.pmx_last_kernel().decode().split("/")[0]
        cpu = ""
        if L <= 20000:
            qb, qo = orc.pack([q]); rb, ro = orc.pack([r])
            t0 = time.perf_counter()
            if mode == 2 and L * 2 < 32000:
                out, _ = orc.cpu_sw_striped16_batch(qb, qo, rb, ro, 5, 2, om, threads=1)
                what = "striped int16 CPU port, 1 core"
            else:
                out = orc.align_batch(mode, qb, qo, rb, ro, 5, 2, om)
                what = "scalar oracle, 1 core"
            tc = time.perf_counter() - t0
            ok = (int(out[0][0]), int(out[0][1]), int(out[0][2])) == (res.get_score(), res.get_end_query(), res.get_end_ref())
            cpu = " | CPU %9.2f ms (%s) %s" % (tc * 1e3, what, "same result" if ok else "RESULT DIFFERS")
        print("%-16s %6d x %6d: %9.3f ms  %8.2f GCUPS  %-22s%s" % (name, L, len(r), t * 1e3, L * len(r) / t / 1e9, kernel, cpu), flush=True)
