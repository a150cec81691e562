#!/usr/bin/env python3
"""Secondary timings (not the headline bench): reduced BASELINE configs 3, 4, 5 through the host
batch API (H2D + kernels + D2H included).  Prints GCUPS per config."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate, DNA, AA
pkg = g.load_pkg()
rng = np.random.default_rng(1)

def timeit(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)

# cfg3: 300 aa profile vs ~5 kaa refs, nw + stats
pm = pkg.Matrix.from_name("blosum62")
q = random_seqs(rng, 1, 300, 300, AA)[0]
refs = random_seqs(rng, 2000, 4500, 5000, AA)
al = pkg.Aligner.new().profile(pkg.Profile.new(q, True, pm)).matrix(pm).gap_open(11).gap_extend(1).solution_width(16).build()
cells = sum(300 * len(r) for r in refs)
t = timeit(lambda: al.align_batch([], refs))
print("cfg3 nw_stats_striped_profile_16  n=%d  %.1f GCUPS (%.3f s)" % (len(refs), cells / t / 1e9, t))
# cfg3b: same without stats
al = pkg.Aligner.new().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(11).gap_extend(1).solution_width(16).build()
t = timeit(lambda: al.align_batch([], refs))
print("cfg3b nw_striped_profile_16       n=%d  %.1f GCUPS (%.3f s)" % (len(refs), cells / t / 1e9, t))

# cfg4: 250x250 related DNA, sg + trace + CIGAR
dm = pkg.Matrix.create(b"ACGT", 2, -3)
qs = random_seqs(rng, 20000, 250, 250)
rs = [mutate(rng, x, 0.1, 0.02) for x in qs]
al = pkg.Aligner.new().semi_global().matrix(dm).gap_open(5).gap_extend(2).solution_width(16).use_trace().build()
cells = sum(len(a) * len(b) for a, b in zip(qs, rs))
t = timeit(lambda: al.align_batch_cigar(qs, rs))
print("cfg4 sg_trace_striped_16 + CIGAR  n=%d  %.1f GCUPS (%.3f s)" % (len(qs), cells / t / 1e9, t))
al = pkg.Aligner.new().semi_global().matrix(dm).gap_open(5).gap_extend(2).solution_width(16).build()
t = timeit(lambda: al.align_batch(qs, rs))
print("cfg4b sg_striped_16 (score only)  n=%d  %.1f GCUPS (%.3f s)" % (len(qs), cells / t / 1e9, t))

# cfg5: 1 kbp query vs 0.5-5 kbp refs, sw sat
q = random_seqs(rng, 1, 1000, 1000)[0]
lens = np.exp(rng.uniform(np.log(500), np.log(5000), size=4000)).astype(int)
refs = [DNA[rng.integers(0, 4, size=int(l))].tobytes() for l in lens]
al = pkg.Aligner.new().local().profile(pkg.Profile.new(q, False, dm)).matrix(dm).gap_open(5).gap_extend(2).build()
cells = sum(1000 * len(r) for r in refs)
t = timeit(lambda: al.align_batch([], refs))
print("cfg5 sw_striped_profile_sat       n=%d  %.1f GCUPS (%.3f s)" % (len(refs), cells / t / 1e9, t))
al2 = pkg.Aligner.new().local().matrix(dm).gap_open(5).gap_extend(2).build()
t = timeit(lambda: al2.align_batch([q] * len(refs), refs))
print("cfg5b sw_striped_sat (one-off)    n=%d  %.1f GCUPS (%.3f s)" % (len(refs), cells / t / 1e9, t))
