#!/usr/bin/env python3
"""Kernel-level timings of the reduced BASELINE configs 3, 4b, 5 with inputs resident in HBM
(torch tensors + pmx_align_batch_device, HIP events on the launch stream).  Secondary numbers,
not the headline bench."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import DNA, AA
pkg = g.load_pkg()
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)

def randbatch(n, lo, hi, alpha):
    lens = rng.integers(lo, hi + 1, size=n)
    off = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
    buf = alpha[rng.integers(0, len(alpha), size=int(off[-1]))]
    return buf, off

def run(name, cfg, qbuf, qoff, rbuf, roff, mq, mr, stats=False, reps=5, shared_q=None):
    n = len(roff) - 1
    d_r = torch.from_numpy(rbuf).to(dev); d_ro = torch.from_numpy(roff).to(dev)
    if shared_q is None:
        d_q = torch.from_numpy(qbuf).to(dev); d_qo = torch.from_numpy(qoff).to(dev)
        cells = int(((qoff[1:] - qoff[:-1]) * (roff[1:] - roff[:-1])).sum())
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    st = torch.zeros((n, 3), dtype=torch.int32, device=dev) if stats else None
    stream = torch.cuda.current_stream(dev)
    def once():
        pkg.align_batch_device(cfg, n, d_q.data_ptr(), d_qo.data_ptr(), d_r.data_ptr(), d_ro.data_ptr(), mq, mr,
                               out.data_ptr(), st.data_ptr() if stats else None, stream.cuda_stream)
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): once()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-44s n=%-7d %8.1f GCUPS  (%.3f ms, kernel %s)" % (name, n, cells / ms / 1e6, ms,
          pkg.lib.pmx_kernel_for(C.byref(cfg), mq, mr).decode()))

b62 = pkg.Matrix.from_name("blosum62"); dna = pkg.Matrix.create(b"ACGT", 2, -3)
# cfg3 (one-off form: same 300-aa query repeated), nw + stats and nw score-only, refs 4.5-5 kaa
n = 20000
q = AA[rng.integers(0, 20, size=300)]
qbuf = np.tile(q, n); qoff = np.arange(n + 1, dtype=np.int64) * 300
rbuf, roff = randbatch(n, 4500, 5000, AA)
run("cfg3 nw_stats_striped_16 300aa x ~4.75kaa", pkg.pmx_config_t(pkg.MODE_NW, 0, 11, 1, 16, pkg.WANT_STATS, b62.inner), qbuf, qoff, rbuf, roff, 300, 5000, stats=True, reps=3)
# the same through the profile arm: one shared query profile per 4-wave workgroup
prof = pkg.Profile.new(q.tobytes(), True, b62)
cfgp = pkg.pmx_config_t(pkg.MODE_NW, 0, 11, 1, 16, pkg.WANT_STATS, b62.inner)
d_r = torch.from_numpy(rbuf).to(dev); d_ro = torch.from_numpy(roff).to(dev)
outp = torch.zeros((n, 4), dtype=torch.int32, device=dev); stp = torch.zeros((n, 3), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
def oncep():
    pkg.align_profile_batch_device(cfgp, prof, n, d_r.data_ptr(), d_ro.data_ptr(), 5000, outp.data_ptr(), stp.data_ptr(), stream.cuda_stream)
oncep(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(3): oncep()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%-44s n=%-7d %8.1f GCUPS  (%.3f ms)" % ("cfg3 nw_stats_striped_profile_16 (shared profile)", n, 300 * int((roff[1:] - roff[:-1]).sum()) / ms / 1e6, ms))
run("cfg3b nw_striped_16 (score only)", pkg.pmx_config_t(pkg.MODE_NW, 0, 11, 1, 16, 0, b62.inner), qbuf, qoff, rbuf, roff, 300, 5000, reps=3)
# cfg4b sg score-only 250x250
n = 1000000
qbuf = DNA[rng.integers(0, 4, size=n * 250)]; qoff = np.arange(n + 1, dtype=np.int64) * 250
rbuf = DNA[rng.integers(0, 4, size=n * 250)]; roff = qoff.copy()
run("cfg4b sg_striped_16 250x250 (score only)", pkg.pmx_config_t(pkg.MODE_SG, 15, 5, 2, 16, 0, dna.inner), qbuf, qoff, rbuf, roff, 250, 250)
run("      sg_stats_striped_16 250x250", pkg.pmx_config_t(pkg.MODE_SG, 15, 5, 2, 16, pkg.WANT_STATS, dna.inner), qbuf, qoff, rbuf, roff, 250, 250, stats=True)
run("      sw_striped_16 250x250", pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 16, 0, dna.inner), qbuf, qoff, rbuf, roff, 250, 250)
# cfg5 1 kbp x 0.5-5 kbp sw sat
n = 20000
q = DNA[rng.integers(0, 4, size=1000)]
qbuf = np.tile(q, n); qoff = np.arange(n + 1, dtype=np.int64) * 1000
lens = np.exp(rng.uniform(np.log(500), np.log(5000), size=n)).astype(np.int64)
roff = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=roff[1:])
rbuf = DNA[rng.integers(0, 4, size=int(roff[-1]))]
run("cfg5 sw_striped_sat 1kbp x 0.5-5kbp (length-sorted)", pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 0, 4, dna.inner), qbuf, qoff, rbuf, roff, 1000, 5000, reps=3)
# short reads, global / semi-global score only (8-lane shapes of the second-generation kernel)
n = 1000000
qbuf = DNA[rng.integers(0, 4, size=n * 150)]; qoff = np.arange(n + 1, dtype=np.int64) * 150
rbuf = DNA[rng.integers(0, 4, size=n * 150)]; roff = qoff.copy()
run("sg_striped_16 150x150 (score only)", pkg.pmx_config_t(pkg.MODE_SG, 15, 5, 2, 16, 0, dna.inner), qbuf, qoff, rbuf, roff, 150, 150)
run("nw_striped_16 150x150 (score only)", pkg.pmx_config_t(pkg.MODE_NW, 0, 5, 2, 16, 0, dna.inner), qbuf, qoff, rbuf, roff, 150, 150)
# classic protein database search: one 300-aa query (reused profile) against 20k references of 4.5-5 kaa, local, BLOSUM62 11/1
n = 20000
q = AA[rng.integers(0, 20, size=300)]
rbuf, roff = randbatch(n, 4500, 5000, AA)
prof = pkg.Profile.new(q.tobytes(), False, b62)
cfgp = pkg.pmx_config_t(pkg.MODE_SW, 0, 11, 1, 16, 0, b62.inner)
d_r = torch.from_numpy(rbuf).to(dev); d_ro = torch.from_numpy(roff).to(dev)
outp = torch.zeros((n, 4), dtype=torch.int32, device=dev)
def oncesw():
    pkg.align_profile_batch_device(cfgp, prof, n, d_r.data_ptr(), d_ro.data_ptr(), 5000, outp.data_ptr(), None, stream.cuda_stream)
oncesw(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(3): oncesw()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%-44s n=%-7d %8.1f GCUPS  (%.3f ms, %s)" % ("sw_striped_profile_16 300aa x ~4.75kaa BLOSUM62", n, 300 * int((roff[1:] - roff[:-1]).sum()) / ms / 1e6, ms, pkg.lib.pmx_last_kernel().decode()))
cfgn = pkg.pmx_config_t(pkg.MODE_NW, 0, 11, 1, 16, 0, b62.inner)
def oncenw():
    pkg.align_profile_batch_device(cfgn, prof, n, d_r.data_ptr(), d_ro.data_ptr(), 5000, outp.data_ptr(), None, stream.cuda_stream)
oncenw(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(3): oncenw()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%-44s n=%-7d %8.1f GCUPS  (%.3f ms, %s)" % ("nw_striped_profile_16 300aa x ~4.75kaa BLOSUM62", n, 300 * int((roff[1:] - roff[:-1]).sum()) / ms / 1e6, ms, pkg.lib.pmx_last_kernel().decode()))
# the same shapes with per-pair queries (no shared profile): LDS profiles per pair, reference symbols from HBM
qt = np.tile(q, n); qto = np.arange(n + 1, dtype=np.int64) * 300
run("sw_striped_16 300aa x ~4.75kaa per-pair queries", pkg.pmx_config_t(pkg.MODE_SW, 0, 11, 1, 16, 0, b62.inner), qt, qto, rbuf, roff, 300, 5000, reps=3)
# shorter reads, local
for L in (50, 75, 100):
    n = 1000000
    qbuf = DNA[rng.integers(0, 4, size=n * L)]; qoff = np.arange(n + 1, dtype=np.int64) * L
    rbuf = DNA[rng.integers(0, 4, size=n * L)]; roff = qoff.copy()
    run("sw_striped_16 %dx%d" % (L, L), pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 16, 0, dna.inner), qbuf, qoff, rbuf, roff, L, L)
