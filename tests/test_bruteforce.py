"""The oracle against an exhaustive enumeration that contains no DP (`-m "not gpu"`).

The reference's own known answers pin the oracle only for gaps 0/0 on <= 12-mers; nothing reference-held
exists for non-zero gap penalties, the nine semi-global free-end variants (/root/reference/src/aligner/mod.rs:270-331) or local
end cells.  tests/brute/brute_align.c enumerates EVERY alignment of two short sequences as a string of columns and scores each
string from the definition of the affine gap model (src/aligner/mod.rs:139-153).  Checked here, for every pair of sequences of
<= 4 letters over a 3-letter alphabet (14 400 pairs, exhaustive) and for samples up to 6 x 6, under gap models 0/0, open ==
extend, open >> extend and a mismatch-heavy matrix:
  * the oracle's score is the optimum over all alignments -- nw, sw, and sg with each of the 16 free-end combinations (the
    reference's grammar reaches 9 of them);
  * the alignment the oracle reports (end cell + CIGAR from its traceback) is ONE optimal alignment: re-scored column by column
    by the enumerator's scorer it gives that score, consumes exactly what the end cell says, labels =/X correctly and, for nw
    and sw, has the reported matches / length.
What it cannot decide is WHICH optimal alignment upstream parasail reports on ties (end-cell and traceback priorities stay
[UNPINNED], oracle/pmx_oracle.c header)."""
import ctypes as C
import itertools
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "brute", "brute_align.c")
SO = os.path.join(HERE, "brute", "_build", "libbrute.so")
NEG = -1000000


@pytest.fixture(scope="module")
def brute():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-fopenmp", "-Wall", "-Wextra", "-shared", "-o", SO, SRC])
    lib = C.CDLL(SO)
    lib.brute_optimum_batch.restype = C.c_int
    lib.brute_score_ops.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _all_seqs(maxlen, letters=b"ACG"):
    return [bytes(t) for L in range(1, maxlen + 1) for t in itertools.product(letters, repeat=L)]


def _optima(brute, orc, qs, rs, open_, ext, m):
    qb, qo = orc.pack(qs)
    rb, ro = orc.pack(rs)
    out = np.zeros((len(qs), 18), dtype=np.int32)
    rc = brute.brute_optimum_batch(C.c_long(len(qs)), _p(qb), _p(qo), _p(rb), _p(ro), int(open_), int(ext),
                                   _p(m.scores), m.size, _p(m.mapper), _p(out))
    assert rc == 0
    return out, (qb, qo, rb, ro)


# letters of the compiled convention (include/pmx_conventions.h): which CIGAR letter consumes the query
def _to_brute_letters(orc, text_ops):
    w = orc.align(orc.NW, b"AA", b"A", 1, 1, orc.Matrix.create("ACG", 1, -1), trace=True)     # one query character must be deleted
    ops, _, _ = orc.walk(w)
    consumes_query = [c for c in ops if c in "ID"][0]
    return text_ops if consumes_query == "I" else text_ops.translate(str.maketrans("ID", "DI"))


def _check(brute, orc, qs, rs, open_, ext, m, cigar_every=1):
    best, (qb, qo, rb, ro) = _optima(brute, orc, qs, rs, open_, ext, m)
    n = len(qs)
    modes = [(orc.NW, 0, 0)] + [(orc.SG, f, 1 + f) for f in range(16)] + [(orc.SW, 0, 17)]
    for mode, flags, col in modes:
        got = orc.align_batch(mode, qb, qo, rb, ro, open_, ext, m, sg_flags=flags)
        bad = np.nonzero(got[:, 0] != best[:, col])[0]
        assert len(bad) == 0, ("score is not the optimum", mode, flags, open_, ext, qs[bad[0]], rs[bad[0]], got[bad[0]], best[bad[0], col])
        # the reported alignment is one optimal alignment
        idx = np.arange(0, n, cigar_every, dtype=np.int64)
        texts, rec = orc.cigar_sample(mode, idx, qb, qo, rb, ro, open_, ext, m, sg_flags=flags)
        for t, k in enumerate(idx):
            q, r = qs[k], rs[k]
            ops = "".join(c * int(cnt) for cnt, c in __import__("re").findall(r"(\d+)([=XID])", texts[t]))
            assert len(ops) == sum(int(x) for x in __import__("re").findall(r"\d+", texts[t]))
            ops = _to_brute_letters(orc, ops)
            score, eq, er, bq, br = (int(x) for x in rec[t])
            assert (score, eq, er) == tuple(int(x) for x in got[k]), (mode, flags, k)
            uq, ur = C.c_int(), C.c_int()
            qa = np.frombuffer(q, dtype=np.uint8); ra = np.frombuffer(r, dtype=np.uint8)
            if mode == orc.SW:
                if score == 0:
                    continue
                s = brute.brute_score_ops(ops.encode(), _p(qa), len(q), _p(ra), len(r), bq, br, int(open_), int(ext),
                                          _p(m.scores), m.size, _p(m.mapper), 0, C.byref(uq), C.byref(ur))
                assert s == score and (bq + uq.value - 1, br + ur.value - 1) == (eq, er), (mode, q, r, texts[t], s, score)
                assert ops[0] in "=X" and ops[-1] in "=X"
            else:
                assert (bq, br) == (0, 0)
                s = brute.brute_score_ops(ops.encode(), _p(qa), len(q), _p(ra), len(r), 0, 0, int(open_), int(ext),
                                          _p(m.scores), m.size, _p(m.mapper), flags if mode == orc.SG else 0,
                                          C.byref(uq), C.byref(ur))
                assert s == score and (uq.value, ur.value) == (len(q), len(r)), (mode, flags, q, r, texts[t], s, score)
                # the end cell is where the free tail (if any) starts
                tail_i = len(ops) - len(ops.rstrip("I")); tail_d = len(ops) - len(ops.rstrip("D"))
                if mode == orc.NW or not (flags & (orc.S1_END | orc.S2_END)):
                    assert (eq, er) == (len(q) - 1, len(r) - 1)
                else:
                    assert eq == len(q) - 1 or er == len(r) - 1
                    if eq < len(q) - 1:
                        assert (flags & orc.S1_END) and tail_i >= len(q) - 1 - eq
                    if er < len(r) - 1:
                        assert (flags & orc.S2_END) and tail_d >= len(r) - 1 - er


GAP_MODELS = [(0, 0), (1, 1), (2, 2), (3, 1), (7, 1), (5, 0)]


@pytest.mark.parametrize("gaps", GAP_MODELS)
def test_oracle_is_optimal_on_every_pair_up_to_4_by_4(brute, orc, gaps):
    seqs = _all_seqs(4)
    assert len(seqs) == 120
    qs = [a for a in seqs for _ in seqs]
    rs = [b for _ in seqs for b in seqs]
    m = orc.Matrix.create("ACG", 2, -1)
    _check(brute, orc, qs, rs, gaps[0], gaps[1], m, cigar_every=7)


@pytest.mark.parametrize("gaps,scores", [((0, 0), (1, -1)), ((2, 2), (3, -2)), ((6, 1), (2, -3)), ((4, 2), (1, -4))])
def test_oracle_is_optimal_on_samples_up_to_6_by_6(brute, orc, gaps, scores):
    rng = np.random.default_rng(7100 + gaps[0])
    seqs = _all_seqs(6)
    assert len(seqs) == 1092
    pick = rng.integers(0, len(seqs), size=(700, 2))
    qs = [seqs[a] for a, _ in pick]
    rs = [seqs[b] for _, b in pick]
    # half of the pairs related (the interesting ties sit near the diagonal)
    for k in range(0, len(qs), 2):
        q = bytearray(qs[k])
        if len(q) > 1 and rng.random() < 0.5:
            del q[rng.integers(0, len(q))]
        if rng.random() < 0.5:
            q[rng.integers(0, len(q))] = b"ACG"[rng.integers(0, 3)]
        rs[k] = bytes(q)
    m = orc.Matrix.create("ACG", scores[0], scores[1])
    _check(brute, orc, qs, rs, gaps[0], gaps[1], m, cigar_every=1)


def test_statistics_follow_the_reported_alignment(brute, orc):
    """nw / sw: matches = number of '=' columns and length = number of columns of the CIGAR the same call reports."""
    import re
    seqs = _all_seqs(4)
    m = orc.Matrix.create("ACG", 2, -1)
    rng = np.random.default_rng(7200)
    for _ in range(1500):
        q, r = seqs[rng.integers(0, len(seqs))], seqs[rng.integers(0, len(seqs))]
        for mode in (orc.NW, orc.SW):
            for o, e in ((3, 1), (1, 1), (0, 0)):
                w = orc.align(mode, q, r, o, e, m, stats=True, trace=True)
                text = orc.cigar(w)
                runs = [(int(c), op) for c, op in re.findall(r"(\d+)([=XID])", text)]
                assert w.matches == sum(c for c, op in runs if op == "=") and w.length == sum(c for c, _ in runs), (mode, q, r, text)
                assert w.similar == w.matches                     # match > 0 > mismatch: positive-scoring columns are the '=' ones
