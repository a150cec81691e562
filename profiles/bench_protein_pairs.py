#!/usr/bin/env python3
"""Per-pair protein batches (all-vs-all style): 200k pairs of ~300 aa x ~300 aa, BLOSUM62 11/1, device-resident."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import AA
pkg = g.load_pkg()
rng = np.random.default_rng(3)
dev = torch.device("cuda", 0)
n = 200000
def randbatch(lo, hi):
    lens = rng.integers(lo, hi + 1, size=n)
    off = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
    return AA[rng.integers(0, 20, size=int(off[-1]))], off
qbuf, qoff = randbatch(250, 320); rbuf, roff = randbatch(250, 320)
cells = int(((qoff[1:] - qoff[:-1]) * (roff[1:] - roff[:-1])).sum())
b62 = pkg.Matrix.from_name("blosum62")
d = [torch.from_numpy(x).to(dev) for x in (qbuf, qoff, rbuf, roff)]
out = torch.zeros((n, 4), dtype=torch.int32, device=dev); st = torch.zeros((n, 3), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
for name, mode, sg, want in (("sw_striped_16", pkg.MODE_SW, 0, 0), ("nw_striped_16", pkg.MODE_NW, 0, 0), ("sg_striped_16", pkg.MODE_SG, 15, 0),
                             ("nw_stats_striped_16", pkg.MODE_NW, 0, pkg.WANT_STATS), ("sw_stats_striped_16", pkg.MODE_SW, 0, pkg.WANT_STATS)):
    cfg = pkg.pmx_config_t(mode, sg, 11, 1, 16, want, b62.inner)
    def once():
        pkg.align_batch_device(cfg, n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 320, 320,
                               out.data_ptr(), st.data_ptr() if want else None, stream.cuda_stream)
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(3): once()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-24s 200k x ~285x285 aa  %8.1f GCUPS  (%.2f ms, %s)" % (name, cells / ms / 1e6, ms, pkg.lib.pmx_last_kernel().decode()), flush=True)
