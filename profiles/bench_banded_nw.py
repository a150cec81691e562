#!/usr/bin/env python3
"""Band-only kernels on config 4's shape (1.25 M related pairs of 250 x 250, device-resident), band k around the main diagonal --
the reference's `Aligner::banded_nw` (src/aligner/mod.rs:454-489) over a batch -- global, semi-global and local; the full-matrix
kernel of the same mode beside it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import __graft_entry__ as g
import workloads as wl
pkg = g.load_pkg()
dev = torch.device("cuda", 0)
n = 1_250_000
qbuf, qoff, rbuf, roff = wl.make_cfg4(n)
d = [torch.from_numpy(x).to(dev) for x in (qbuf, qoff, rbuf, roff)]
out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
m = pkg.Matrix.create(b"ACGT", 2, -3)
stream = torch.cuda.current_stream(dev)
def timed(run, reps=5):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mode, name in ((pkg.MODE_NW, "nw"), (pkg.MODE_SG, "sg"), (pkg.MODE_SW, "sw")):
    cfg = pkg.pmx_config_t(mode, pkg.SG_ALL, 5, 2, 16, 0, m.inner)
    ms = timed(lambda: pkg.align_batch_device(cfg, n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 250, 250,
                                              out.data_ptr(), None, stream.cuda_stream))
    full = out.clone()
    print("%s full matrix            %8.3f ms  %8.1f GCUPS  %s" % (name, ms, n * 62500 / ms / 1e6, pkg.lib.pmx_last_kernel().decode()), flush=True)
    for band in (15, 31, 48):
        for env in (None, "PMX_BANDED_NO_STRIP"):
            if env:
                os.environ[env] = "1"
            def run():
                rc = pkg.lib.pmx_align_batch_banded_device(C.byref(cfg), n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                           250, 250, band, None, out.data_ptr(), stream.cuda_stream)
                assert rc == 0, pkg.lib.pmx_last_error()
            ms = timed(run)
            same = float((out[:, 0] == full[:, 0]).float().mean().item())
            cells = n * 250 * (2 * band + 1)
            print("%s band %2d %-22s %8.3f ms  %8.1f GCUPS of band cells  same score as full %.3f  %s" % (
                name, band, env or "", ms, cells / ms / 1e6, same, pkg.lib.pmx_last_kernel().decode()), flush=True)
            if env:
                del os.environ[env]
