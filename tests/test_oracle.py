"""CPU tests of the oracle itself: every known-answer assertion of the reference's
tests/test_parasail.rs (tests/golden/reference_kats.json), internal consistency properties, and
the striped CPU port against the scalar DP."""
import numpy as np
import pytest

from util import random_seqs, mutate, score_from_cigar, DNA, AA

MODES = {"nw": 0, "sg": 1, "sw": 2}


def _matrix(orc, spec):
    return orc.Matrix.create(spec["alphabet"], spec["match"], spec["mismatch"])


def test_reference_kats(orc, kats):
    checked = 0
    for k in kats:
        if k["mode"] not in MODES:
            continue
        m = _matrix(orc, k["matrix"])
        bits = 0 if k["width"] == "sat" else int(k["width"])
        r = orc.align(MODES[k["mode"]], k["query"].encode(), k["ref"].encode(), k["open"], k["extend"], m,
                      bits=bits, stats=k["stats"], table=k["table"], rowcol=k["rowcol"], trace=k["trace"])
        e = k["expect"]
        for f in ("score", "end_query", "end_ref", "matches", "length"):
            if f in e:
                assert getattr(r, f) == e[f], (k["name"], f)
        for t in ("score", "matches", "similar", "length"):
            if t + "_table_rows" in e:
                assert getattr(r, t + "_table").shape == (e[t + "_table_rows"], e[t + "_table_cols"]), k["name"]
            if t + "_table_last" in e:
                assert getattr(r, t + "_table")[-1, -1] == e[t + "_table_last"], k["name"]
            for w in ("row", "col"):
                if "%s_%s" % (t, w) in e:
                    assert list(getattr(r, "%s_%s" % (t, w))) == e["%s_%s" % (t, w)], k["name"]
        if "trace_table_len" in e:
            assert r.trace_table.size == e["trace_table_len"]
            assert ((r.trace_table & 7) <= 4).all()
        assert not r.saturated, k["name"]
        checked += 1
    assert checked >= 28


def test_banded_and_ssw_kats_are_plain_dp_consistent(orc, kats):
    # tests/test_parasail.rs:726-756: ACGT/ACGT scores 4 under any sane band; ssw = sw with begin 0/0
    m = orc.Matrix.default()
    r = orc.align(orc.SW, b"ACGT", b"ACGT", 0, 0, m, trace=True)
    ops, bq, br = orc.walk(r)
    assert (r.score, r.end_query, r.end_ref, bq, br) == (4, 3, 3, 0, 0)


def test_blosum62_fixture(orc):
    m = orc.Matrix.from_file("tests/golden/blosum62.txt")
    assert m.size == 24 and (m.scores == m.scores.T).all()
    assert m.scores[m.mapper[ord("W")], m.mapper[ord("W")]] == 11
    assert m.scores[m.mapper[ord("a")], m.mapper[ord("R")]] == -1
    assert m.mapper[ord("J")] == 23        # outside the alphabet -> wildcard column


def test_default_matrix_quirk(orc):
    # src/matrix/mod.rs:248: "ACGTA" -- the duplicated A maps to its last position and still matches
    m = orc.Matrix.default()
    assert m.size == 6 and m.mapper[ord("A")] == 4
    assert orc.align(orc.NW, b"AAAA", b"AAAA", 0, 0, m).score == 4


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_table_score_consistency(orc, mode):
    rng = np.random.default_rng(100 + mode)
    m = orc.Matrix.create("ACGT", 2, -3)
    for _ in range(40):
        q = random_seqs(rng, 1, 1, 40)[0]
        r = mutate(rng, q) if rng.random() < 0.5 else random_seqs(rng, 1, 1, 40)[0]
        res = orc.align(mode, q, r, 5, 2, m, stats=True, table=True, rowcol=True)
        t = res.score_table
        assert t[res.end_query, res.end_ref] == res.score
        assert (res.score_row == t[-1, :]).all() and (res.score_col == t[:, -1]).all()
        assert res.matches_table[res.end_query, res.end_ref] == res.matches
        assert res.length_table[res.end_query, res.end_ref] == res.length
        if mode == 0:
            assert (res.end_query, res.end_ref) == (len(q) - 1, len(r) - 1)
        if mode == 2:
            assert res.score == t.max() and t.min() >= 0
            jj = np.argmax((t == t.max()).any(axis=0))
            ii = np.argmax(t[:, jj] == t.max())
            assert (res.end_query, res.end_ref) == (ii, jj)
        if mode == 1:
            assert res.score == max(t[-1, :].max(), t[:, -1].max())


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (11, 1), (3, 3)])
def test_cigar_reproduces_score(orc, mode, gaps):
    """The walked alignment must re-score to the DP score (validates trace bits + walk)."""
    rng = np.random.default_rng(200 + mode * 10 + gaps[0])
    m = orc.Matrix.create("ACGT", 2, -3)
    for _ in range(40):
        q = random_seqs(rng, 1, 1, 50)[0]
        r = mutate(rng, q, 0.15, 0.08)
        res = orc.align(mode, q, r, gaps[0], gaps[1], m, trace=True, stats=True)
        ops, bq, br = orc.walk(res)
        text = orc.cigar(res)
        if mode == 1:
            # free end gaps are part of the sg CIGAR but cost nothing: strip them before re-scoring
            continue
        s, i, j = score_from_cigar(text, q, r, bq, br, m.scores, m.mapper, gaps[0], gaps[1])
        assert s == res.score, (text, q, r)
        assert (i - 1, j - 1) == (res.end_query, res.end_ref)
        if mode == 0:
            assert (bq, br) == (0, 0)
        n_eq = sum(n for n, op in __import__("util").cigar_ops(text) if op == "=")
        assert n_eq == res.matches
        assert len(ops) == res.length


def test_sg_variants(orc):
    m = orc.Matrix.create("ACGT", 2, -3)
    q, r = b"ACGTACGT", b"TTTTACGTACGTTTTT"
    full = orc.align(orc.SG, q, r, 5, 2, m, sg_flags=orc.SG_ALL)
    assert full.score == 16 and full.end_query == 7 and full.end_ref == 11
    dx = orc.align(orc.SG, q, r, 5, 2, m, sg_flags=orc.S2_BEG | orc.S2_END)   # sg_dx: reference ends free
    assert dx.score == 16
    none = orc.align(orc.SG, q, r, 5, 2, m, sg_flags=0)
    nw = orc.align(orc.NW, q, r, 5, 2, m)
    assert none.score == nw.score
    qx = orc.align(orc.SG, q, r, 5, 2, m, sg_flags=orc.S1_BEG | orc.S1_END)   # query ends free only
    assert qx.score < 16


def test_saturation_rule(orc):
    m = orc.Matrix.create("ACGT", 40, -40)
    q = b"ACGT" * 250                         # perfect 1 kbp match = 40 000 > int16
    assert orc.align(orc.SW, q, q, 5, 2, m, bits=16).saturated == 1
    assert orc.align(orc.SW, q, q, 5, 2, m, bits=32).saturated == 0
    assert orc.align(orc.SW, q, q, 5, 2, m, bits=0).score == 40000
    assert orc.align(orc.SW, b"ACGT" * 40, b"ACGT" * 40, 5, 2, orc.Matrix.create("ACGT", 2, -3), bits=8).saturated == 1


@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (1, 1), (11, 1)])
def test_striped_cpu_port_matches_scalar(orc, gaps):
    rng = np.random.default_rng(300 + gaps[0])
    m = orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 300, 1, 200)
    rs = [mutate(rng, q, 0.1, 0.05) if rng.random() < 0.6 else random_seqs(rng, 1, 1, 200)[0] for q in qs]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, gaps[0], gaps[1], m)
    for lanes in (16, 32) if orc.cpu_striped_lanes() == 32 else (16,):      # both vector widths where the CPU has AVX-512BW
        got, used = orc.cpu_sw_striped16_batch(qb, qo, rb, ro, gaps[0], gaps[1], m, threads=2, lanes=lanes)
        assert used >= 1
        assert (got == want).all(), lanes


def test_striped_cpu_port_blosum62(orc):
    rng = np.random.default_rng(301)
    m = orc.Matrix.from_file("tests/golden/blosum62.txt")
    qs = random_seqs(rng, 100, 10, 300, AA)
    rs = [mutate(rng, q, 0.3, 0.05, AA) for q in qs]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, 11, 1, m)
    got, _ = orc.cpu_sw_striped16_batch(qb, qo, rb, ro, 11, 1, m, threads=2)
    assert (got == want).all()
    # profile arm: one shared query, its striped profile built once per thread
    q = qs[7]
    qb2, qo2 = orc.pack([q] * len(rs))
    want = orc.align_batch(orc.SW, qb2, qo2, rb, ro, 11, 1, m)
    for lanes in (16, 32) if orc.cpu_striped_lanes() == 32 else (16,):
        got, _ = orc.cpu_sw_striped16_batch(None, None, rb, ro, 11, 1, m, threads=2, shared_query=q, lanes=lanes)
        assert (got == want).all(), lanes


def test_cfg1_fixture_is_reproduced(orc):
    """BASELINE config 1 (one protein pair, BLOSUM62 11/1): the committed oracle outputs (tests/golden/make_cfg1.py)
    pin the scalar oracle against regressions; the GPU suite replays the same pair through the mirror."""
    import json
    fx = json.load(open("tests/golden/cfg1_protein_pair.json"))
    m = orc.Matrix.from_file("tests/golden/blosum62.txt")
    q, r = fx["query"].encode(), fx["ref"].encode()
    for name, mode in (("sw", orc.SW), ("nw", orc.NW), ("sg", orc.SG)):
        w = orc.align(mode, q, r, fx["open"], fx["extend"], m, stats=True, trace=True)
        c = fx["cases"][name]
        assert (w.score, w.end_query, w.end_ref, w.matches, w.similar, w.length) == \
            (c["score"], c["end_query"], c["end_ref"], c["matches"], c["similar"], c["length"])
        assert orc.cigar(w) == c["cigar"]


def test_vectorised_cpu_ports_of_the_stats_and_trace_modes(orc):
    """oracle/pmx_cpu_inter16.c (the CPU timing baselines of BASELINE configs 3 and 4: 16 pairs per AVX2 vector) against the scalar
    oracle, bit for bit: score, ends, matches / similar / length of `nw_stats_*_profile_16` with one shared protein query, ragged
    references; score, ends, begin positions and CIGAR text of `sg_trace` / `nw_trace` on DNA with ragged pairs, tie-heavy gap
    models and wildcards."""
    from util import AA
    rng = np.random.default_rng(4242)
    b62 = orc.Matrix.from_file("tests/golden/blosum62.txt")
    for qlen, n, lo, hi, o, e in ((300, 45, 380, 520, 11, 1), (37, 33, 1, 90, 3, 3), (120, 17, 100, 140, 5, 0)):
        q = random_seqs(rng, 1, qlen, qlen, AA)[0]
        rs = [mutate(rng, q, 0.3, 0.05, AA) + random_seqs(rng, 1, 0, 60, AA)[0] if k % 2 else random_seqs(rng, 1, lo, hi, AA)[0] for k in range(n)]
        rb, ro = orc.pack(rs)
        got, used = orc.cpu_nw_stats_inter16(q, rb, ro, o, e, b62, threads=2)
        want = orc.align_stats_sample(orc.NW, np.arange(n), None, None, rb, ro, o, e, b62, shared_query=q)
        assert used >= 1 and (got == want[:, :6]).all(), (qlen, np.nonzero((got != want[:, :6]).any(axis=1))[0][:5], got[:3], want[:3])
    dna = orc.Matrix.create("ACGT", 2, -3)
    for mode in (orc.SG, orc.NW):
        for n, lo, hi, o, e in ((50, 1, 40, 5, 2), (35, 200, 260, 5, 2), (40, 10, 120, 2, 2), (33, 30, 90, 4, 0)):
            qs = random_seqs(rng, n, lo, hi)
            rs = [mutate(rng, q, 0.1, 0.05) if k % 3 else random_seqs(rng, 1, lo, hi)[0] for k, q in enumerate(qs)]
            qs[1] = qs[1][:1] + b"N" + qs[1][2:] if len(qs[1]) > 2 else qs[1]
            rs[2] = rs[2][:1] + b"N" + rs[2][2:] if len(rs[2]) > 2 else rs[2]
            qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
            texts, rec, used = orc.cpu_trace_cigar_inter16(mode, qb, qo, rb, ro, o, e, dna, threads=2)
            wtext, wrec = orc.cigar_sample(mode, np.arange(n), qb, qo, rb, ro, o, e, dna)
            assert (rec == wrec).all(), (mode, n, np.nonzero((rec != wrec).any(axis=1))[0][:5], rec[:3], wrec[:3])
            assert [t.decode() for t in texts] == wtext, (mode, n)
    with pytest.raises(RuntimeError):                       # not a match / mismatch matrix: refused, not mis-scored
        orc.cpu_trace_cigar_inter16(orc.SG, qb, qo, rb, ro, 11, 1, b62)
