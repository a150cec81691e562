import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg()
rng = np.random.default_rng(1)
dm = pkg.Matrix.create(b"ACGT", 2, -3)
N = int(os.environ.get("CFG4_N", "20000"))
qs = random_seqs(rng, N, 250, 250)
rs = [mutate(rng, x, 0.1, 0.02) for x in qs]
MODE = os.environ.get("CFG4_MODE", "semi_global")
al = getattr(pkg.Aligner.new(), MODE)().matrix(dm).gap_open(5).gap_extend(2).solution_width(16).use_trace().build()
for _ in range(3):
    t0 = time.perf_counter(); al.align_batch_cigar(qs, rs); print("cfg4 %.4f s" % (time.perf_counter() - t0))
