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


def test_library_window_predicate_equals_the_model(pkg):
    """the host's admissibility predicate of the band-strip kernel (pmx_bstrip_window, C ABI test hook) against the model's
    `bias_and_low` on a lattice that straddles every edge: score range, gap constants, lengths where the window closes"""
    import ctypes as C
    lib = pkg.lib
    lib.pmx_bstrip_window.restype = C.c_int
    n_ok = n_no = 0
    for mode in (bm.NW, bm.SG, bm.SW):
        for ds in (0, 1):
            if mode == bm.SW and ds:
                continue
            for (smin, smax) in ((-3, 2), (-4, 5), (-1, 1), (-2, 3), (-30, 20), (-4, 11), (0, 250), (-128, 127)):
                for open_ in (0, 1, 2, 3, 5, 11, 20, 60, 121, 122, 130):
                    for ext in (0, 1, 2, 3, 5, 11, 20):
                        for (m, n) in ((1, 1), (150, 150), (250, 250), (1000, 5000), (2000, 2000), (3000, 3000), (5000, 5000), (10000, 300), (12000, 12000), (30000, 30000)):
                            for cap in (8, 32, 104, 128):
                                rows = min(m, n + cap)
                                want = bm.bias_and_low(mode, 0, m, n, open_, ext, smin, smax, (cap - 2) // 2, cap, rows, bool(ds))
                                b, l = C.c_int(0), C.c_int(0)
                                ok = lib.pmx_bstrip_window(mode, m, n, open_, ext, smin, smax, cap, rows, ds, C.byref(b), C.byref(l))
                                assert bool(ok) == (want is not None), (mode, ds, smin, smax, open_, ext, m, n, cap)
                                if want is not None:
                                    assert (b.value, l.value) == want, (mode, ds, smin, smax, open_, ext, m, n, cap, b.value, l.value, want)
                                    n_ok += 1
                                else:
                                    n_no += 1
    assert n_ok > 2000 and n_no > 2000, (n_ok, n_no)


@pytest.mark.parametrize("mode", [bm.NW, bm.SG, bm.SW])
def test_model_stays_inside_the_window_at_the_predicates_edge(orc, mode):
    """Scoring schemes the predicate only just admits (the next larger extension penalty or match score is rejected), on the
    inputs that stretch the value range: identical sequences (highest scores), no match at all (lowest), a long sequence against a
    one-letter one and bands far off the diagonal (longest boundary gaps), poly-A (every cell ties).  Every intermediate of the
    model is checked against the window (`check=True`), and the result against the banded oracle."""
    rng = np.random.default_rng(9900 + mode)
    tried = 0
    for L, k in ((120, 31), (90, 15), (200, 48), (60, 3)):
        cap = _cap_for(k)
        for double_skew in ((False,) if mode == bm.SW else (False, True)):
            for open_ in (2, 11, 60, 120):
                for mis in (-1, -4, -30):
                    # the largest match score / extension penalty the predicate admits for this shape
                    ext = 0
                    while ext + 1 <= open_ and bm.window_ok(mode, 15, L, L, open_, ext + 1, mis, 2, k, cap, L, double_skew):
                        ext += 1
                    match = 1
                    while bm.window_ok(mode, 15, L, L, open_, ext, mis, match + 1, k, cap, L, double_skew):
                        match += 1
                    if not bm.window_ok(mode, 15, L, L, open_, ext, mis, match, k, cap, L, double_skew):
                        continue
                    om = orc.Matrix.create("ACGT", match, mis)
                    mat = _mat5(om)
                    a = random_seqs(rng, 1, L, L)[0]
                    pairs = [(a, a, 0), (b"A" * L, b"C" * L, 0), (a, b"G", 0), (b"T", a, 3), (b"A" * L, b"A" * (L - 7), -2),
                             (a, mutate(rng, a, 0.2, 0.1), 1), (a, a[L // 2:], -(L // 2)), (a[L // 2:], a, L // 2), (a, a, k + L // 3)]
                    for sg in ((0, 5, 10, 15) if mode == bm.SG else (0,)):
                        B_LOW = bm.bias_and_low(mode, sg, L, L, open_, ext, mis, match, k, cap, L, double_skew)
                        for q, r, d in pairs:
                            qb, qo = orc.pack([q]); rb, ro = orc.pack([r])
                            want = orc.align_banded_batch(mode, qb, qo, rb, ro, open_, ext, om, k, np.array([d], dtype=np.int32), sg_flags=sg)[0]
                            got = bm.align(mode, sg, _idx(q), _idx(r), open_, ext, mat, k, d, cap, double_skew, B_LOW=B_LOW, check=True)
                            assert tuple(int(x) for x in got) == tuple(int(x) for x in want), (mode, sg, L, k, open_, ext, mis, match, d, got, want)
                            tried += 1
    assert tried > 300, tried
