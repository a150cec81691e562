#!/usr/bin/env python3
"""One GPU's share of BASELINE config 5: one 1 kbp query (reused profile) against 1.25M references of
0.5-5 kbp (log-uniform), `sw_striped_profile_sat`, host API (references in host memory), plus the
device-resident kernel time.  A noisy copy of the query is planted in 1 % of the references."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_pkg()
N = int(os.environ.get("CFG5_N", "1250000"))
rng = np.random.default_rng(20260005)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
q = lut[rng.integers(0, 4, size=1000, dtype=np.uint8)]
lens = np.exp(rng.uniform(np.log(500), np.log(5000), size=N)).astype(np.int64)
roff = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=roff[1:])
rbuf = lut[rng.integers(0, 4, size=int(roff[-1]), dtype=np.uint8)]
for k in rng.choice(N, size=N // 100, replace=False):
    if lens[k] >= 1000:
        pos = int(roff[k] + rng.integers(0, lens[k] - 999))
        cp = q.copy(); m = rng.random(1000) < 0.05
        cp[m] = lut[rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)]
        rbuf[pos:pos + 1000] = cp
cells = 1000 * int(roff[-1])
dm = pkg.Matrix.create(b"ACGT", 2, -3)
prof = pkg.Profile.new(q.tobytes(), False, dm)
cfg = pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 0, 0, dm.inner)
out = np.zeros(N, dtype=pkg.RECORD_DTYPE)
for _ in range(2):
    t0 = time.perf_counter()
    rc = pkg.lib.pmx_align_profile_batch(C.byref(cfg), prof.inner, N, rbuf.ctypes.data, roff.ctypes.data, out.ctypes.data, None)
    t = time.perf_counter() - t0
    assert rc == 0
    print("cfg5 share, host API: %d refs, %.2e cells in %.3f s -> %.0f GCUPS (%s); max score %d, %d scores > 1000" %
          (N, cells, t, cells / t / 1e9, pkg.lib.pmx_last_kernel().decode(), out["score"].max(), int((out["score"] > 1000).sum())), flush=True)
dev = torch.device("cuda", 0)
d_r = torch.from_numpy(rbuf).to(dev); d_ro = torch.from_numpy(roff).to(dev)
d_out = torch.zeros((N, 4), dtype=torch.int32, device=dev)
cfgs = pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 0, pkg.WANT_SORTED if hasattr(pkg, "WANT_SORTED") else 4, dm.inner)
st = torch.cuda.current_stream(dev)
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    pkg.align_profile_batch_device(cfgs, prof, N, d_r.data_ptr(), d_ro.data_ptr(), 5000, d_out.data_ptr(), None, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("cfg5 share, device-resident (length-sorted): %.1f ms -> %.0f GCUPS" % (ms, cells / ms / 1e6), flush=True)
assert (d_out[:, 0].cpu().numpy() == out["score"]).all()
