#!/usr/bin/env python3
"""Generates tests/golden/cfg1_protein_pair.json: BASELINE config 1 (one protein pair, BLOSUM62, gaps 11/1).
The reference cannot run here, so the expected values come from the scalar oracle
(oracle/pmx_oracle.c) and pin it against regressions; the GPU tests replay the pair through the mirror.
Run from the repo root:  python tests/golden/make_cfg1.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

orc = g.load_oracle()
rng = np.random.default_rng(20260006)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
q = AA[rng.integers(0, 20, size=120)].tobytes()
core = bytearray(q[10:100])
for k in range(0, len(core), 9):
    core[k] = AA[rng.integers(0, 20)]
del core[40:43]
r = AA[rng.integers(0, 20, size=15)].tobytes() + bytes(core) + AA[rng.integers(0, 20, size=12)].tobytes()
m = orc.Matrix.from_file(os.path.join(ROOT, "tests", "golden", "blosum62.txt"))
out = {"query": q.decode(), "ref": r.decode(), "matrix": "blosum62", "open": 11, "extend": 1, "cases": {}}
for name, mode in (("sw", orc.SW), ("nw", orc.NW), ("sg", orc.SG)):
    w = orc.align(mode, q, r, 11, 1, m, stats=True, trace=True)
    out["cases"][name] = {"score": int(w.score), "end_query": int(w.end_query), "end_ref": int(w.end_ref),
                          "matches": int(w.matches), "similar": int(w.similar), "length": int(w.length),
                          "cigar": orc.cigar(w)}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "cfg1_protein_pair.json"), "w"), indent=1)
print(json.dumps(out["cases"], indent=1))
