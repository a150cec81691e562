#!/usr/bin/env python3
"""A batch of long pairs (512 x 5 kbp x 5 kbp, queries beyond the packed kernels' 2 048 rows) through the long-pair kernel, per form.
Wall time of the host entry (host buffers in, records out).  On a shared GPU host the runtime's synchronisation behind such a 4 ms batch
was seen to take 20-30 ms in steps of 10 ms in some processes (the first process of a session never): trust the fastest run -- the
kernels under rocprofv3 take 3.5 ms (global, skewed form) and 5.2 ms (local) whatever the wall clock says."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(8)
m = pkg.Matrix.create(b"ACGT", 2, -3)
qs = random_seqs(rng, 512, 4800, 5200); rs = [mutate(rng, q, 0.08, 0.03) for q in qs]
cells = sum(len(q) * len(r) for q, r in zip(qs, rs))
out = {}
for name, b in (("sw", pkg.Aligner.new().local().matrix(m).gap_open(5).gap_extend(2)), ("nw", pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2))):
    al = b.build(); al.align_batch(qs, rs)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); got = al.align_batch(qs, rs); ts.append(time.perf_counter() - t0)
    out[name] = [min(ts) * 1e3, cells / min(ts) / 1e9, int(got["score"].sum()), pkg.lib.pmx_last_kernel().decode().split("/")[0]]
print(json.dumps(out))
''' % (ROOT, ROOT)
base = None
for env in ({}, {"PMX_LONG_ONE_COLUMN": "1"}, {"PMX_LONG_TWO_COLUMNS": "1"}, {"PMX_LONG_TWO_COLUMNS": "1", "PMX_LONG_ROWS_PER_LANE": "2"}):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
    if p.returncode:
        print(env, "FAILED", p.stderr[-400:]); continue
    d = json.loads(p.stdout.strip().splitlines()[-1])
    if base is None: base = d
    print(env, "same results" if all(d[k][2] == base[k][2] for k in d) else "DIFFERS")
    for k in d: print("   %s  %8.3f ms (host buffers in, records out)  %8.1f GCUPS  %s" % (k, d[k][0], d[k][1], d[k][3]))
