#!/usr/bin/env python3
"""Development check of the band-strip kernel against the banded oracle (all modes, free-end sets, bands, band centres)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from util import random_seqs, mutate
pkg = g.load_pkg(); orc = g.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
tot = bad_tot = 0
names = {}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    match, mis = [(2, -3), (5, -4), (1, -1), (3, -2)][int(rng.integers(0, 4))]
    open_ = int(rng.choice([1, 2, 3, 5, 11, 20])); ext = int(rng.choice([0, 1, 2, 3]))
    if open_ < ext: open_, ext = ext, open_
    k = int(rng.choice([0, 1, 2, 3, 5, 7, 8, 11, 15, 16, 23, 31, 32, 48, 50, 63]))
    lo, hi = [(1, 12), (5, 60), (40, 300), (200, 600)][int(rng.integers(0, 4))]
    mode = int(rng.integers(0, 3))
    n = int(rng.choice([1, 3, 64, 257, 700]))
    pm, om = pkg.Matrix.create(b"ACGT", match, mis), orc.Matrix.create("ACGT", match, mis)
    qs = random_seqs(rng, n, lo, hi)
    rs, diag = [], np.zeros(n, dtype=np.int32)
    for t, q in enumerate(qs):
        body = mutate(rng, q, 0.1, 0.05) if rng.random() < 0.8 else random_seqs(rng, 1, lo, hi)[0]
        pre = random_seqs(rng, 1, 0, 50)[0] if rng.random() < 0.5 else b""
        post = random_seqs(rng, 1, 0, 50)[0] if rng.random() < 0.3 else b""
        rs.append((pre + body + post) or b"A")
        diag[t] = len(pre) + int(rng.integers(-6, 7)) if rng.random() < 0.85 else int(rng.integers(-hi - 5, hi + 5))
    if rng.random() < 0.3:      # a wildcard somewhere
        t = int(rng.integers(0, n)); r = bytearray(rs[t]); r[int(rng.integers(0, len(r)))] = ord("N"); rs[t] = bytes(r)
        t = int(rng.integers(0, n)); q = bytearray(qs[t]); q[int(rng.integers(0, len(q)))] = ord("N"); qs[t] = bytes(q)
    b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
    [b.global_, b.semi_global, b.local][mode]()
    sg = orc.SG_ALL
    if mode == 1:
        sg = int(rng.integers(1, 16))
        qg = [t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
        dg = [t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
        b.allow_query_gaps(qg).allow_ref_gaps(dg)
    al = b.build()
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for dg_ in (None, diag):
        got = al.align_batch_banded(qs, rs, k, dg_)
        kn = pkg.lib.pmx_last_kernel().decode()
        names[kn] = names.get(kn, 0) + 1
        want = orc.align_banded_batch(mode, qb, qo, rb, ro, open_, ext, om, k, dg_, sg_flags=sg)
        bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]) | (got["flags"] != 0))[0]
        tot += n; bad_tot += len(bad)
        if len(bad):
            print("FAIL it", it, "mode", mode, "sg", sg, "mat", match, mis, "gap", open_, ext, "k", k, "n", n, "len", lo, hi, "diag", dg_ is not None, kn,
                  "nbad", len(bad), [(int(x), tuple(int(v) for v in (got["score"][x], got["end_query"][x], got["end_ref"][x], got["flags"][x])), tuple(int(v) for v in want[x]),
                                      len(qs[x]), len(rs[x]), int(diag[x]) if dg_ is not None else 0) for x in bad[:4]], flush=True)
print("pairs", tot, "bad", bad_tot, names)
