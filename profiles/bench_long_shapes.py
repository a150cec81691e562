#!/usr/bin/env python3
"""Time per reference column of the long-pair kernel against the number of bands (query length), one align() call on a
50 kbp reference, per kernel form: {} = what the dispatcher's time model picks (pmx_api.hip, long_batch)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(6)
m = pkg.Matrix.create(b"ACGT", 2, -3)
out = {}
RL = 50000
r = random_seqs(rng, 1, RL, RL)[0]
for QL in (512, 768, 1024, 2048, 4096, 8192, 16384):
    q = random_seqs(rng, 1, QL, QL)[0]
    for name, b in (("sw", pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2)), ("nw", pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2))):
        al = b.build(); al.align(q, r)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); res = al.align(q, r); ts.append(time.perf_counter() - t0)
        out["%%s %%d" %% (name, QL)] = [min(ts) * 1e3, pkg.lib.pmx_last_kernel().decode().split("/")[0]]
print(json.dumps(out))
''' % (ROOT, ROOT)
for env in ({}, {"PMX_LONG_ONE_COLUMN": "1"}, {"PMX_LONG_TWO_COLUMNS": "1", "PMX_LONG_ROWS_PER_LANE": "4"}, {"PMX_LONG_TWO_COLUMNS": "1", "PMX_LONG_ROWS_PER_LANE": "2"}):
    e = dict(os.environ); e.update(env); pass
    p = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
    if p.returncode:
        print(env, "FAILED", p.stderr[-400:]); continue
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print(env)
    for k in d: print("   %-10s %9.3f ms  %7.1f ns per column  %s" % (k, d[k][0], d[k][0] * 1e6 / 50000, d[k][1]))
