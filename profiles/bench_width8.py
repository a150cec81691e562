#!/usr/bin/env python3
"""`nw_striped_8` / `sg_striped_8` against their 16-bit forms on config 2's shape (1 M pairs of 150 x 150, device-resident): the
reference treats width 8 as its FASTEST width (src/aligner/mod.rs:125-130); here it runs the same packed int16 kernel plus the
range tracking that yields the saturation flag (round 2: the general int32 kernel, a 20-50x cliff)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
import workloads as wl
pkg = g.load_pkg()
dev = torch.device("cuda", 0)
n = 1_000_000
qbuf, qoff, rbuf, roff = wl.make_cfg2(n)
d = [torch.from_numpy(x).to(dev) for x in (qbuf, qoff, rbuf, roff)]
out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
m = pkg.Matrix.create(b"ACGT", 2, -3)
stream = torch.cuda.current_stream(dev)
# a gap model with open < extend (no packed kernel serves global / semi-global there): the 32-bit band kernel, against the general kernel
for mode, name in ((pkg.MODE_NW, "nw"), (pkg.MODE_SG, "sg")):
    for env in (None, "PMX_NO_LONG_KERNEL"):
        if env:
            os.environ[env] = "1"
        cfg = pkg.pmx_config_t(mode, pkg.SG_ALL, 2, 5, 32, 0, m.inner)
        run = lambda: pkg.align_batch_device(cfg, n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 150, 150,
                                             out.data_ptr(), None, stream.cuda_stream)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(3):
            run()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("%s_striped_32 open 2 < extend 5 %-18s %8.3f ms  %8.1f GCUPS  %s" % (name, env or "", ms, n * 22500 / ms / 1e6,
                                                                             pkg.lib.pmx_last_kernel().decode()), flush=True)
        if env:
            del os.environ[env]
for mode, name in ((pkg.MODE_NW, "nw"), (pkg.MODE_SG, "sg"), (pkg.MODE_SW, "sw")):
    for width in (16, 8):
        for env in ((None,) if width == 16 or mode == pkg.MODE_SW else (None, "PMX_NWSG8_GENERAL")):
            if env:
                os.environ[env] = "1"
            cfg = pkg.pmx_config_t(mode, pkg.SG_ALL, 5, 2, width, 0, m.inner)
            run = lambda: pkg.align_batch_device(cfg, n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 150, 150,
                                                 out.data_ptr(), None, stream.cuda_stream)
            run(); torch.cuda.synchronize()
            reps = 3 if env else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                run()
            e1.record(stream); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            sat = float((out[:, 3] & 1).float().mean().item())
            print("%s_striped_%-2d %-18s %8.3f ms  %8.1f GCUPS  saturated %.3f  %s" % (name, width, env or "", ms, n * 22500 / ms / 1e6, sat,
                                                                                  pkg.lib.pmx_last_kernel().decode()), flush=True)
            if env:
                del os.environ[env]
