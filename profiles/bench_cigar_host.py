#!/usr/bin/env python3
"""Host-API timing of the batch CIGAR entry (BASELINE config 4 shape): the C call alone (packed inputs
in, records + CIGAR text out), separated from Python-side packing / string splitting."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_pkg()
N = int(os.environ.get("CFG4_N", "1000000"))
rng = np.random.default_rng(20260004)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
q = lut[rng.integers(0, 4, size=(N, 250), dtype=np.uint8)]
r = q.copy()
sub = rng.random((N, 250)) < 0.10
r[sub] = lut[rng.integers(0, 4, size=int(sub.sum()), dtype=np.uint8)]
qbuf = q.reshape(-1); rbuf = r.reshape(-1)
off = np.arange(N + 1, dtype=np.int64) * 250
dm = pkg.Matrix.create(b"ACGT", 2, -3)
al = pkg.Aligner.new().semi_global().matrix(dm).gap_open(5).gap_extend(2).solution_width(16).use_trace().build()
cfg = al._config(); cfg.want &= ~pkg.WANT_STATS
out = np.zeros(N, dtype=pkg.RECORD_DTYPE); coff = np.zeros(N + 1, dtype=np.int64)
for _ in range(3):
    cbuf = C.c_void_p()
    t0 = time.perf_counter()
    rc = pkg.lib.pmx_align_batch_cigar(C.byref(cfg), N, qbuf.ctypes.data, off.ctypes.data, rbuf.ctypes.data, off.ctypes.data,
                                       out.ctypes.data, C.byref(cbuf), coff.ctypes.data)
    t = time.perf_counter() - t0
    assert rc == 0
    print("pmx_align_batch_cigar: %d pairs 250x250 in %.3f s -> %.0f GCUPS end to end (%d CIGAR bytes)" %
          (N, t, N * 62500 / t / 1e9, coff[-1]), flush=True)
    pkg.lib.pmx_free(cbuf)
