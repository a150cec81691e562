"""CPU tier: the band-strip kernel's arithmetic model (tests/bstrip_model.py) against the banded oracle.

The GPU kernel (pmx_bstrip.hip) follows the model cell for cell; what this test pins without a GPU is the DESIGN: boundary
conditions produced by virtual columns / the edge input / the initial state, guarded cells beyond the band, the captures and
their tie rules, and that every intermediate stays inside the int16 window whenever the host predicate admits the batch."""
import numpy as np
import pytest

import bstrip_model as bm
from util import random_seqs, mutate


def _mat5(om):
    return [[int(om.scores[a, b]) for b in range(5)] for a in range(5)]


def _idx(seq):
    return [b"ACGT".index(bytes([c])) for c in seq]


CAPS = [8, 12, 16, 24, 32, 48, 64, 96, 104, 128]


def _cap_for(k):
    for c in CAPS:
        if c >= 2 * k + 1:
            return c
    return 2 * k + 2


def _run_case(orc, rng, mode, sg, match, mis, open_, ext, k, lo, hi, n_pairs, double_skew):
    om = orc.Matrix.create("ACGT", match, mis)
    mat = _mat5(om)
    qs = random_seqs(rng, n_pairs, lo, hi)
    rs, diag = [], np.zeros(n_pairs, dtype=np.int32)
    for t, q in enumerate(qs):
        body = mutate(rng, q, 0.1, 0.05) if rng.random() < 0.8 else random_seqs(rng, 1, lo, hi)[0]
        pre = random_seqs(rng, 1, 0, 30)[0] if rng.random() < 0.5 else b""
        post = random_seqs(rng, 1, 0, 30)[0] if rng.random() < 0.3 else b""
        rs.append((pre + body + post) or b"A")
        diag[t] = len(pre) + int(rng.integers(-6, 7)) if rng.random() < 0.8 else int(rng.integers(-hi - 5, hi + 5))
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_banded_batch(mode, qb, qo, rb, ro, open_, ext, om, k, diag, sg_flags=sg)
    cap = _cap_for(k)
    n_checked = 0
    for t in range(n_pairs):
        q, r = _idx(qs[t]), _idx(rs[t])
        try:
            got = bm.align(mode, sg, q, r, open_, ext, mat, k, int(diag[t]), cap, double_skew)
        except bm.WindowError as e:
            assert "does not hold" in str(e), (str(e), mode, sg, match, mis, open_, ext, k, len(q), len(r), int(diag[t]))
            continue
        n_checked += 1
        assert tuple(int(x) for x in got) == tuple(int(x) for x in want[t]), \
            (mode, sg, match, mis, open_, ext, k, int(diag[t]), len(q), len(r), got, want[t], double_skew)
    return n_checked


@pytest.mark.parametrize("double_skew", [False, True])
@pytest.mark.parametrize("mode", [bm.NW, bm.SG, bm.SW])
def test_model_matches_the_banded_oracle(orc, mode, double_skew):
    if mode == bm.SW and double_skew:
        pytest.skip("local alignment has one skew only")
    rng = np.random.default_rng(9100 + mode + (7 if double_skew else 0))
    total = 0
    for it in range(60):
        match, mis = [(2, -3), (5, -4), (1, -1), (3, -2)][int(rng.integers(0, 4))]
        open_ = int(rng.choice([1, 2, 3, 5, 11, 20])); ext = int(rng.choice([0, 1, 2, 3]))
        if open_ < ext:
            open_, ext = ext, open_
        k = int(rng.choice([0, 1, 2, 3, 5, 7, 15, 16, 31]))
        lo, hi = [(1, 6), (1, 12), (5, 40), (30, 90)][int(rng.integers(0, 4))]
        sg = int(rng.integers(0, 16)) if mode == bm.SG else 0
        total += _run_case(orc, rng, mode, sg, match, mis, open_, ext, k, lo, hi, 12, double_skew)
    assert total > 300
