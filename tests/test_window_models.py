"""CPU tier: the stored-arithmetic model of the packed global / semi-global kernels (tests/nwsgv_model.c) at the corners of the host's
range proof (`pmx_nwsgv_bias`, exported as the test hook `pmx_window_nwsgv`).

pmx_nwsg16.hip has no promotion pass: a batch the predicate admits must keep EVERY stored 16-bit pattern inside [1024, 31743].  The
round-3 soak found a window bug (fuzz seed 5018: the last-column capture left the window) that no CPU-tier test could have caught.
The model replays the kernel's stored form lane for lane -- bias, column skew, row offsets, virtual rows and columns, captures -- and
counts what leaves its domain; here it runs on the inputs that stretch the range (all matches, no match, a copy behind a long gap,
poly-A) at the LONGEST reference the predicate admits per scoring scheme and shape, and one column short of it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from util import random_seqs

HERE = os.path.dirname(os.path.abspath(__file__))


class Out(C.Structure):
    _fields_ = [("score", C.c_int), ("end_query", C.c_int), ("end_ref", C.c_int), ("lo", C.c_int), ("hi", C.c_int),
                ("violations", C.c_int), ("first_violation_kind", C.c_int)]


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("nwsgv_model") / "nwsgv_model.so")
    subprocess.run(["gcc", "-O3", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "nwsgv_model.c")], check=True)
    lib = C.CDLL(so)
    lib.nwsgv_model.restype = C.c_int
    return lib


def _window(pkg, qlen, rlen, smin, smax, open_, ext, rowx, rows):
    pkg.lib.pmx_window_nwsgv.restype = C.c_int
    return pkg.lib.pmx_window_nwsgv(qlen, rlen, 5, smin, smax, open_, ext, rowx, rows)


def _run(model, G, R, rowx, top, legacy, q, r, max_rlen, om, open_, ext, sg, nb):
    mat = np.ascontiguousarray(om.scores[:5, :5].astype(np.int32))
    qi = np.ascontiguousarray(_LUT[np.frombuffer(q, dtype=np.uint8)])
    ri = np.ascontiguousarray(_LUT[np.frombuffer(r, dtype=np.uint8)])
    out = Out()
    col_pen, row_pen = int(not (sg & 1)), int(not (sg & 4))
    rc = model.nwsgv_model(G, R, rowx, top, legacy, qi.ctypes.data_as(C.c_void_p), len(q), ri.ctypes.data_as(C.c_void_p), len(r), max_rlen,
                           mat.ctypes.data_as(C.c_void_p), 5, open_, ext, col_pen, row_pen, int(bool(sg & 2)), int(bool(sg & 8)), nb, C.byref(out))
    assert rc == 0
    return out


_LUT = np.zeros(256, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _LUT[_c] = _i
_ROT = bytes.maketrans(b"ACGT", b"CGTA")


def _oracle(orc, q, r, om, open_, ext, sg):
    qb, qo = orc.pack([q]); rb, ro = orc.pack([r])
    mode = orc.NW if sg == 0 else orc.SG
    return tuple(int(x) for x in orc.align_batch(mode, qb, qo, rb, ro, open_, ext, om, sg_flags=sg if sg else orc.SG_ALL)[0])


SHAPES = [(8, 7), (8, 16), (16, 10), (16, 16), (32, 10), (64, 2), (64, 16), (8, 20)]


def test_model_matches_the_oracle_on_small_pairs(orc, pkg, model):
    """both alignments of the query in the shape (virtual rows on top / padding rows below), with and without the row offset, every
    free-end set: same score and end positions as the oracle, nothing outside its domain"""
    rng = np.random.default_rng(4100)
    n = 0
    for it in range(160):
        G, R = SHAPES[int(rng.integers(0, len(SHAPES)))]
        match, mis = [(2, -3), (5, -4), (1, -1), (3, -2), (9, -9)][int(rng.integers(0, 5))]
        open_, ext = [(5, 2), (1, 1), (10, 1), (4, 4), (3, 0), (0, 0), (20, 3)][int(rng.integers(0, 7))]
        om = orc.Matrix.create("ACGT", match, mis)
        qlen = int(rng.choice([1, 2, G * R, G * R - 1, int(rng.integers(1, G * R + 1))]))
        rlen = int(rng.choice([1, 2, 17, int(rng.integers(1, 200))]))
        q, r = random_seqs(rng, 1, qlen, qlen)[0], random_seqs(rng, 1, rlen, rlen)[0]
        if rng.random() < 0.5:
            r = (q * (rlen // qlen + 1))[:rlen]
        sg = int(rng.integers(0, 16))
        for rowx in (1, 0):
            nb = _window(pkg, qlen, rlen, mis, match, open_, ext, rowx, G * R)
            if not nb:
                continue
            for top in (0, 1):
                out = _run(model, G, R, rowx, top, 0, q, r, rlen + int(rng.integers(0, 40)), om, open_, ext, sg, nb)
                want = _oracle(orc, q, r, om, open_, ext, sg)
                assert (out.score, out.end_query, out.end_ref) == want and out.violations == 0, \
                    (G, R, rowx, top, match, mis, open_, ext, sg, qlen, rlen, (out.score, out.end_query, out.end_ref), want, out.violations, out.lo, out.hi)
                n += 1
    assert n > 400, n


@pytest.mark.parametrize("rowx", [1, 0])
def test_nothing_leaves_the_window_at_the_longest_reference_the_proof_admits(orc, pkg, model, rowx):
    rng = np.random.default_rng(4200 + rowx)
    n = 0
    tightest = 0
    for match, mis, open_, ext in ((2, -3, 5, 2), (1, -1, 1, 1), (5, -4, 10, 1), (3, -2, 4, 4), (9, -9, 20, 3), (1, -30, 60, 1), (2, -3, 120, 5)):
        om = orc.Matrix.create("ACGT", match, mis)
        for G, R in ((8, 7), (16, 16), (64, 2)) + (((64, 16),) if ext == 2 else ()):
            for qlen in sorted({1, min(50, G * R), G * R}):
                lo, hi = 0, 30000
                if not _window(pkg, qlen, 1, mis, match, open_, ext, rowx, G * R):
                    continue
                lo = 1
                while hi - lo > 0:                                  # the longest reference the predicate admits for this shape
                    mid = (lo + hi + 1) // 2
                    lo, hi = (mid, hi) if _window(pkg, qlen, mid, mis, match, open_, ext, rowx, G * R) else (lo, mid - 1)
                for rlen in sorted({lo, max(1, lo - 1)}):
                    if qlen * rlen > 9_000_000:
                        continue                                    # (kept to sizes the scalar oracle does in a blink)
                    nb = _window(pkg, qlen, rlen, mis, match, open_, ext, rowx, G * R)
                    assert nb
                    q0 = random_seqs(rng, 1, qlen, qlen)[0]
                    far = random_seqs(rng, 1, rlen, rlen)[0]
                    refs = [(q0 * (rlen // qlen + 1))[:rlen],
                            (q0 * (rlen // qlen + 1))[:rlen].translate(_ROT),
                            (far[:rlen - qlen] + q0) if rlen > qlen else far, (q0 + far[:rlen - qlen]) if rlen > qlen else far]
                    for r, q in [(x, q0) for x in (refs if rlen == lo else refs[::2])] + [(b"A" * rlen, b"A" * qlen)]:
                        for sg in ((0, 15, 10, 5) if rlen == lo else (2, 8)):
                            for top in (0, 1):
                                out = _run(model, G, R, rowx, top, 0, q, r, rlen, om, open_, ext, sg, nb)
                                assert out.violations == 0, ("outside its domain", out.first_violation_kind, out.lo, out.hi, G, R, top, match, mis, open_, ext, sg, qlen, rlen)
                                tightest = max(tightest, out.hi)
                                if top == 0 and n % 16 == 0:         # (the oracle on a sample: the values are what this test is about)
                                    want = _oracle(orc, q, r, om, open_, ext, sg)
                                    assert (out.score, out.end_query, out.end_ref) == want, (G, R, match, mis, open_, ext, sg, qlen, rlen, want)
                                n += 1
    assert n > 1000, n
    assert tightest > 24000, tightest                               # the corners really are close to the top of the window


def test_the_model_flags_the_capture_form_the_round_3_soak_caught(orc, pkg, model):
    """(fuzz seed 5018) the last-column capture once compared the rows with their offsets taken off and the capture bias added: a short
    query against a reference of tens of kilobases under extend = 1 with a free query end left the window.  The model, run with that
    form, flags exactly that operand; with the form the kernels use now it is clean on the same input."""
    om = orc.Matrix.create("ACGT", 2, -3)
    rng = np.random.default_rng(5018)
    G, R, qlen, open_, ext = 8, 7, 50, 5, 1
    flagged = clean = 0
    for rlen in (18000, 22000, 26000, 28000):
        nb = _window(pkg, qlen, rlen, -3, 2, open_, ext, 1, G * R)
        if not nb:
            continue
        q = random_seqs(rng, 1, qlen, qlen)[0]
        r = random_seqs(rng, 1, rlen - qlen, rlen - qlen)[0] + q
        for sg in (2, 10, 15):
            old = _run(model, G, R, 1, 0, 1, q, r, rlen, om, open_, ext, sg, nb)
            new = _run(model, G, R, 1, 0, 0, q, r, rlen, om, open_, ext, sg, nb)
            flagged += int(old.violations > 0 and old.first_violation_kind == 4)
            clean += int(new.violations == 0 and (new.score, new.end_query, new.end_ref) == _oracle(orc, q, r, om, open_, ext, sg))
    assert flagged >= 3 and clean >= 9, (flagged, clean)
