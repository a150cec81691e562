#!/usr/bin/env python3
"""One banded configuration on config 4's shape (1.25 M related pairs of 250 x 250, device-resident), for profiling:
   python profiles/bench_band_one.py <mode nw|sg|sw> <band> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import __graft_entry__ as g
import workloads as wl
pkg = g.load_pkg()
dev = torch.device("cuda", 0)
mode_name, band = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1_250_000
qbuf, qoff, rbuf, roff = wl.make_cfg4(n)
d = [torch.from_numpy(x).to(dev) for x in (qbuf, qoff, rbuf, roff)]
out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
m = pkg.Matrix.create(b"ACGT", 2, -3)
stream = torch.cuda.current_stream(dev)
mode = {"nw": pkg.MODE_NW, "sg": pkg.MODE_SG, "sw": pkg.MODE_SW}[mode_name]
cfg = pkg.pmx_config_t(mode, pkg.SG_ALL, 5, 2, 16, 0, m.inner)
def run():
    rc = pkg.lib.pmx_align_batch_banded_device(C.byref(cfg), n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                               250, 250, band, None, out.data_ptr(), stream.cuda_stream)
    assert rc == 0, pkg.lib.pmx_last_error()
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(reps):
    run()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("%s band %d  %.3f ms  %.1f GCUPS of band cells  %s" % (mode_name, band, ms, n * 250 * (2 * band + 1) / ms / 1e6, pkg.lib.pmx_last_kernel().decode()))
