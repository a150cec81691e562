import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(7)
m = pkg.Matrix.create(b"ACGT", 2, -3)
L = 3000
qs0 = random_seqs(rng, 256, L, L); rs0 = [mutate(rng, q, 0.08, 0.03) for q in qs0]
al = pkg.Aligner.new().matrix(m).gap_open(5).gap_extend(2).build()
for n in (256, 1024, 2048, 3072, 4096):
    qs = (qs0 * (n // 256 + 1))[:n]; rs = (rs0 * (n // 256 + 1))[:n]
    for one in (0, 1):
        if one: os.environ["PMX_GENERAL_ONE_WAVE"] = "1"
        else: os.environ.pop("PMX_GENERAL_ONE_WAVE", None)
        al.align_batch(qs[:70], rs[:70])
        t0 = time.perf_counter(); al.align_batch(qs, rs); t = time.perf_counter() - t0
        print("n=%5d one_wave=%d  %8.1f ms  %7.1f GCUPS" % (n, one, t * 1e3, n * L * L / t / 1e9), flush=True)
