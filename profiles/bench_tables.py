#!/usr/bin/env python3
"""Score-table throughput of the row-by-row kernel (pmx_table.hip) against the HBM write roof: 4 bytes per cell.
Prints one JSON line (kept under profiles/r02/).  Secondary measurement -- not a BASELINE config."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import workloads as wl
pkg = g.load_pkg()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
pm = pkg.Matrix.create(b"ACGT", 2, -3)
out_lines = []
for (n, ql, rl) in ((40000, 250, 250), (8000, 500, 1000), (4000, 1000, 1000)):
    qb = wl.DNA[rng.integers(0, 4, size=n * ql)]; rb = wl.DNA[rng.integers(0, 4, size=n * rl)]
    qo = wl.uniform_offsets(n, ql); ro = wl.uniform_offsets(n, rl); to = wl.uniform_offsets(n, ql * rl)
    d = [torch.from_numpy(x).to(dev) for x in (qb, qo, rb, ro, to)]
    table = torch.empty(n * ql * rl, dtype=torch.int32, device=dev)
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for mode, name in ((pkg.MODE_NW, "nw_table_striped_32"), (pkg.MODE_SW, "sw_table_striped_32")):
        cfg = pkg.pmx_config_t(mode, 0, 5, 2, 32, 0, pm.inner)
        def once():
            rc = pkg.lib.pmx_align_batch_table_device(C.byref(cfg), n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                      ql, rl, d[4].data_ptr(), table.data_ptr(), None, None, out.data_ptr(), stream.cuda_stream)
            assert rc == 0, pkg.lib.pmx_last_error().decode()
        once(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5): once()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        cells = n * ql * rl
        gbs = 4.0 * cells / (ms * 1e-3) / 1e9
        out_lines.append({"workload": "%d pairs x (%d x %d) DNA, %s, score table [qlen][rlen] int32" % (n, ql, rl, name),
                          "kernel": pkg.lib.pmx_last_kernel().decode(), "ms": round(ms, 3), "gcups": round(cells / (ms * 1e-3) / 1e9, 1),
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                                       "algorithmic_bytes": 4 * cells + n * (ql + rl + 12),
                                       "note": "4 B per cell written once; achievable write bandwidth on this chip is below the 8 TB/s spec"}})
print(json.dumps({"metric": "score-table throughput (secondary)", "lines": out_lines}))
