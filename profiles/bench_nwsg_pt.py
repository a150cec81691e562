#!/usr/bin/env python3
"""Perm-table (top-aligned, no LDS profile) form of the global / semi-global kernels against the LDS-profile form, score-only,
device-resident, equal-length DNA reads."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
import workloads as wl
pkg = g.load_pkg()
dev = torch.device("cuda", 0)
m = pkg.Matrix.create(b"ACGT", 2, -3)
stream = torch.cuda.current_stream(dev)
for L, n in ((100, 1_000_000), (150, 1_000_000), (250, 400_000)):
    rng = np.random.default_rng(L)
    q = wl.DNA[rng.integers(0, 4, size=(n, L), dtype=np.uint8)].reshape(-1)
    r = wl.DNA[rng.integers(0, 4, size=(n, L), dtype=np.uint8)].reshape(-1)
    off = wl.uniform_offsets(n, L)
    d = [torch.from_numpy(x).to(dev) for x in (q, off, r, off)]
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    for mode, sg, name in ((pkg.MODE_NW, 0, "nw"), (pkg.MODE_SG, pkg.SG_ALL, "sg"), (pkg.MODE_SG, 1 | 4, "sg_qb_db")):
        for env in (None, "PMX_NWSG16_NO_PERMTABLE"):
            if env:
                os.environ[env] = "1"
            cfg = pkg.pmx_config_t(mode, sg, 5, 2, 16, 0, m.inner)
            run = lambda: pkg.align_batch_device(cfg, n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), L, L,
                                                 out.data_ptr(), None, stream.cuda_stream)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                run()
            e1.record(stream); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("%-9s %4d x %4d  n %8d  %-24s %8.3f ms %8.1f GCUPS  %s" % (name, L, L, n, env or "", ms, n * L * L / ms / 1e6,
                                                                          pkg.lib.pmx_last_kernel().decode()), flush=True)
            if env:
                del os.environ[env]
