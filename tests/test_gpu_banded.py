"""Banded alignment (`-m gpu`): Aligner::banded_nw (/root/reference/src/aligner/mod.rs:454-489, KAT tests/test_parasail.rs:726-736)
against the banded oracle on NARROW bands, the batch extension (any mode, per-pair band centre: BASELINE config 5's "banded SW"),
and inputs beyond every LDS-resident limit (long references in the general kernel, long pairs in the band-only kernel)."""
import os

import numpy as np
import pytest

import workloads as wl
from util import random_seqs, mutate, DNA, AA

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [0, 1, 2, 8, 32, 63, 64, 400])
def test_banded_nw_narrow_bands_match_the_banded_oracle(pkg, orc, k):
    """one pair per call through Aligner::banded_nw; k <= 63 runs the band-only kernel, wider bands the masked general kernel"""
    rng = np.random.default_rng(8100 + k)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    al = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).bandwidth(k).build()
    qs = random_seqs(rng, 24, 1, 220)
    rs = [mutate(rng, q, 0.1, 0.05) if i % 3 else random_seqs(rng, 1, 1, 260)[0] for i, q in enumerate(qs)]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_banded_batch(orc.NW, qb, qo, rb, ro, 5, 2, om, k)
    full = orc.align_batch(orc.NW, qb, qo, rb, ro, 5, 2, om)
    for t, (q, r) in enumerate(zip(qs, rs)):
        res = al.banded_nw(q, r)
        assert res.is_banded() and res.is_global()
        assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == tuple(want[t]), (k, t, len(q), len(r))
    assert (want[:, 0] <= full[:, 0]).all()
    if k == 400:
        assert (want[:, 0] == full[:, 0]).all()


@pytest.mark.parametrize("form", ["strip", "staged", "percell"])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 3, 15, 16, 31, 32, 63, 100])
def test_banded_batch_every_mode_with_band_centres(pkg, orc, mode, k, form, monkeypatch):
    """the band-strip kernel (band coordinates, C offsets per lane), both forms of the anti-diagonal band-only kernel (sequences
    staged in LDS + lean interior loop; per-cell form) and, for k > 63, the masked general kernel"""
    staged = form != "percell"
    if form == "percell":
        monkeypatch.setenv("PMX_BANDED_NO_STAGING", "1")
    if form == "staged":
        monkeypatch.setenv("PMX_BANDED_NO_STRIP", "1")
    rng = np.random.default_rng(8200 + 10 * k + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    n = 300
    qs = random_seqs(rng, n, 1, 180)
    rs = []
    diag = np.zeros(n, dtype=np.int32)
    for t, q in enumerate(qs):
        body = mutate(rng, q, 0.08, 0.04)
        pre = random_seqs(rng, 1, 0, 60)[0] if t % 2 else b""
        rs.append(pre + body + (random_seqs(rng, 1, 0, 40)[0] if t % 5 == 0 else b""))
        diag[t] = len(pre) + int(rng.integers(-4, 5))            # the band follows the planted copy, a little off-centre
    b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2)
    [b.global_, b.semi_global, b.local][mode]()
    al = b.build()
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for dg in (None, diag):
        got = al.align_batch_banded(qs, rs, k, dg)
        kernel = pkg.lib.pmx_last_kernel().decode()
        assert kernel == ("pmx_general_kernel/banded" if k > 63
                          else ("pmx_bstrip_kernel/local" if mode == 2 else "pmx_bstrip_kernel/double skew") if form == "strip"
                          else ("pmx_banded_packed_kernel/matrix rows" if mode == 2 else "pmx_banded_staged_kernel") if staged
                          else "pmx_banded_kernel"), kernel
        want = orc.align_banded_batch(mode, qb, qo, rb, ro, 5, 2, om, k, dg)
        bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
        assert len(bad) == 0, (mode, k, dg is None, bad[:5], got[bad[:3]], want[bad[:3]], [(len(qs[x]), len(rs[x])) for x in bad[:3]])


def test_banded_batch_sg_variants_and_protein(pkg, orc):
    rng = np.random.default_rng(8300)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    qs = random_seqs(rng, 200, 5, 150, AA)
    rs = [mutate(rng, q, 0.3, 0.05, AA) + random_seqs(rng, 1, 0, 30, AA)[0] for q in qs]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for sg, qg, dg in ((orc.S1_BEG | orc.S2_END, ["prefix"], ["suffix"]), (orc.S1_END, ["suffix"], []), (orc.S2_BEG | orc.S2_END, [], ["prefix", "suffix"])):
        al = pkg.Aligner.new().matrix(pm).gap_open(11).gap_extend(1).semi_global().allow_query_gaps(qg).allow_ref_gaps(dg).build()
        got = al.align_batch_banded(qs, rs, 20)
        want = orc.align_banded_batch(orc.SG, qb, qo, rb, ro, 11, 1, om, 20, sg_flags=sg)
        assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() and (got["end_ref"] == want[:, 2]).all(), sg


def test_cfg5_banded_sw_second_pass(pkg, orc):
    """BASELINE config 5's "banded SW": a first full pass (`sw_striped_profile_sat`) gives every pair's end cell; the banded pass
    around that cell's diagonal reproduces the full score wherever the optimal path stays inside the band (every planted copy:
    1 % indels drift a few columns), never exceeds it elsewhere, and agrees with the banded oracle on a sample."""
    n = 20000
    q, rbuf, roff, planted = wl.make_cfg5(n, rank=2)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    prof = pkg.Profile.new(q, False, pm)
    al = pkg.Aligner.new().local().profile(prof).matrix(pm).gap_open(5).gap_extend(2).build()
    full = al.align_batch_packed(None, None, rbuf, roff)
    diag = (full["end_ref"] - full["end_query"]).astype(np.int32)
    rs = [rbuf[roff[k]:roff[k + 1]].tobytes() for k in range(n)]
    band = 48
    got = al.align_batch_banded([], rs, band, diag)
    assert pkg.lib.pmx_last_kernel().decode() == "pmx_bstrip_kernel/local"
    assert (got["score"] <= full["score"]).all()
    assert (got["score"][planted] == full["score"][planted]).all()
    assert (got["end_query"][planted] == full["end_query"][planted]).all() and (got["end_ref"][planted] == full["end_ref"][planted]).all()
    idx = np.unique(np.concatenate([planted[:60], np.arange(0, n, 257)]))
    sub_r = [rs[k] for k in idx]
    rb2, ro2 = orc.pack(sub_r)
    want = orc.align_banded_batch(orc.SW, None, None, rb2, ro2, 5, 2, om, band, diag[idx], shared_query=q)
    assert (got["score"][idx] == want[:, 0]).all() and (got["end_query"][idx] == want[:, 1]).all() and (got["end_ref"][idx] == want[:, 2]).all()


def test_long_pairs_have_no_length_limit(pkg, orc):
    """ nothing on the path may abort on long inputs: the band-only kernel takes 300 kbp x 300 kbp pairs (the reference
    offers banded_nw "for aligning large sequences"), and a plain striped call whose reference does not fit the LDS runs in the
    general kernel from an HBM copy of the reference."""
    rng = np.random.default_rng(8400)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    L = 300_000
    q = DNA[rng.integers(0, 4, size=L)]
    r = q.copy()
    pos = np.sort(rng.choice(L, size=300, replace=False))
    r[pos] = DNA[(np.searchsorted(DNA, r[pos]) + 1) % 4]                     # 300 substitutions, no indels: the path is the main diagonal
    res = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).bandwidth(32).build().banded_nw(q.tobytes(), r.tobytes())
    assert res.get_score() == 2 * (L - 300) - 3 * 300 and res.get_end_query() == L - 1 and res.get_end_ref() == L - 1
    # a medium case with indels against the banded oracle (20 kbp: the oracle sweeps every cell)
    q2 = DNA[rng.integers(0, 4, size=20000)].tobytes()
    r2 = mutate(rng, q2, 0.05, 0.004)
    qb, qo = orc.pack([q2]); rb, ro = orc.pack([r2])
    for k in (16, 63):
        res = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).bandwidth(k).build().banded_nw(q2, r2)
        w = orc.align_banded_batch(orc.NW, qb, qo, rb, ro, 5, 2, om, k)[0]
        assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == tuple(w), k
    # one-off local alignment, query beyond 2 048 rows, reference beyond the 160 KB of LDS
    q3 = DNA[rng.integers(0, 4, size=2100)].tobytes()
    r3 = bytearray(DNA[rng.integers(0, 4, size=170_000)].tobytes())
    r3[90_000:90_000 + 2100] = mutate(rng, q3, 0.03, 0.01)[:2100].ljust(2100, b"A")
    r3 = bytes(r3)
    al = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).build()
    res = al.align(q3, r3)
    w = orc.align(orc.SW, q3, r3, 5, 2, om)
    assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (w.score, w.end_query, w.end_ref)
    st = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).use_stats().build().align(q3[:300], r3)
    w = orc.align(orc.NW, q3[:300], r3, 5, 2, om, stats=True)
    assert (st.get_score(), st.get_matches(), st.get_length()) == (w.score, w.matches, w.length)


def test_ssw_begin_positions_and_packed_cigar(pkg, orc):
    """SSW emulation (/root/reference/src/aligner/mod.rs:491-529, src/alignment/mod.rs:506-551; KAT tests/test_parasail.rs:738-756 is
    ACGT vs ACGT only): score, end AND begin coordinates and the packed `len << 4 | op` CIGAR on 520 random / mutated pairs with
    mismatches and gaps, DNA and protein, against the oracle's local alignment + walk."""
    import re
    rng = np.random.default_rng(8600)
    ops = "MIDNSHP=X"
    cases = []
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    for t in range(400):
        q = random_seqs(rng, 1, 8, 180)[0]
        core = mutate(rng, q[rng.integers(0, len(q) // 2):], 0.12, 0.06)
        r = random_seqs(rng, 1, 0, 40)[0] + core + random_seqs(rng, 1, 0, 40)[0] if t % 4 else random_seqs(rng, 1, 5, 200)[0]
        cases.append((q, r, pm, om, 5, 2))
    pb, ob = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    for t in range(120):
        q = random_seqs(rng, 1, 10, 150, AA)[0]
        r = random_seqs(rng, 1, 0, 30, AA)[0] + mutate(rng, q, 0.3, 0.05, AA) + random_seqs(rng, 1, 0, 30, AA)[0]
        cases.append((q, r, pb, ob, 11, 1))
    nontrivial = 0
    for q, r, m, o, go, ge in cases:
        res = pkg.Aligner.new().matrix(m).gap_open(go).gap_extend(ge).build().ssw(q, r)
        w = orc.align(orc.SW, q, r, go, ge, o, trace=True)
        ops_w, bq, br = orc.walk(w)
        text = orc.cigar(w)
        assert res.score() == min(max(w.score, 0), 65535)
        if w.score == 0:
            continue
        assert (res.query_end(), res.ref_end()) == (w.end_query, w.end_ref)
        assert (res.query_start(), res.ref_start()) == (bq, br), (q, r)
        packed = [(int(n) << 4) | ops.index(c) for n, c in re.findall(r"(\d+)([=XID])", text)]
        got = [res.cigar()[k] for k in range(res.cigar_len())]
        assert got == packed, (q, r, got, packed)
        nontrivial += len(packed) > 2
    assert nontrivial > 300


def _seeds(default):
    spec = os.environ.get("PMX_FUZZ_SEEDS")
    if not spec:
        return default
    a, c = spec.split(":")
    return list(range(int(a), int(c)))


@pytest.mark.parametrize("seed", _seeds([301, 302]))
def test_fuzz_banded_batches(pkg, orc, seed):
    """Random banded batches against the banded oracle: every mode and free-end set, random gap models and bands, band centres on
    and off the planted copy, lengths from 1 symbol to beyond the interior loop's blocks, DNA and protein (both forms of the
    band-only kernel take these: staged + lean interior loop, or per cell when the switch forces it)."""
    rng = np.random.default_rng(seed)
    mats = [(pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3), DNA),
            (pkg.Matrix.create(b"ACGT", 5, -4), orc.Matrix.create("ACGT", 5, -4), DNA),
            (pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt"), AA)]
    for it in range(10):
        pm, om, alpha = mats[int(rng.integers(0, len(mats)))]
        mode = int(rng.integers(0, 3))
        open_ = int(rng.choice([0, 1, 5, 11, 20])); ext = int(rng.choice([0, 1, 2, 5]))
        k = int(rng.choice([0, 1, 2, 5, 15, 16, 30, 31, 32, 48, 63]))
        lo, hi = [(1, 12), (5, 60), (40, 300), (200, 900)][int(rng.integers(0, 4))]
        n = int(rng.choice([1, 3, 64, 257]))
        qs = random_seqs(rng, n, lo, hi, alpha)
        rs, diag = [], np.zeros(n, dtype=np.int32)
        for t, q in enumerate(qs):
            body = mutate(rng, q, 0.1, 0.05, alpha) if rng.random() < 0.8 else random_seqs(rng, 1, lo, hi, alpha)[0]
            pre = random_seqs(rng, 1, 0, 50, alpha)[0] if rng.random() < 0.5 else b""
            post = random_seqs(rng, 1, 0, 50, alpha)[0] if rng.random() < 0.3 else b""
            rs.append((pre + body + post) or body or b"A")
            diag[t] = len(pre) + int(rng.integers(-6, 7))
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
            want = orc.align_banded_batch(mode, qb, qo, rb, ro, open_, ext, om, k, dg_, sg_flags=sg)
            bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
            assert len(bad) == 0, (seed, it, mode, sg, open_, ext, k, lo, hi, n, dg_ is None, pkg.lib.pmx_last_kernel().decode(),
                                   bad[:5], got[bad[:3]], want[bad[:3]], [(len(qs[x]), len(rs[x])) for x in bad[:3]])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_wide_bands_sweep_only_reachable_columns(pkg, orc, mode):
    """k > 63 runs in the general kernel, which sweeps per 64-row band of the query only the columns that band can reach: band
    centres far right of the main diagonal (the first query rows start at column d - k > 0: the boundary ROW is their diagonal
    source), far left (the band leaves the matrix before the query ends), bands that miss the corner, queries of several 64-row
    bands and references several times the query; every pair against the banded oracle."""
    rng = np.random.default_rng(8700 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2)
    [b.global_, b.semi_global, b.local][mode]()
    al = b.build()
    for k, qlo, qhi, shift in ((64, 100, 400, 250), (80, 60, 700, 600), (127, 300, 301, 40), (200, 65, 130, 900)):
        n = 120
        qs = random_seqs(rng, n, qlo, qhi)
        rs, diag = [], np.zeros(n, dtype=np.int32)
        for t, q in enumerate(qs):
            pre = random_seqs(rng, 1, 0, shift)[0] if t % 3 else b""
            post = random_seqs(rng, 1, 0, shift)[0] if t % 4 == 0 else b""
            rs.append(pre + mutate(rng, q, 0.08, 0.04) + post)
            diag[t] = len(pre) + int(rng.integers(-30, 31)) if t % 7 else -int(rng.integers(0, len(q)))     # (every seventh: left of the main diagonal)
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        got = al.align_batch_banded(qs, rs, k, diag)
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_general_kernel/banded"
        want = orc.align_banded_batch(mode, qb, qo, rb, ro, 5, 2, om, k, diag)
        bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
        assert len(bad) == 0, (mode, k, bad[:5], got[bad[:3]], want[bad[:3]], [(len(qs[x]), len(rs[x]), int(diag[x])) for x in bad[:3]])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_long_single_pairs_share_the_waves_of_a_workgroup(pkg, orc, mode):
    """Queries beyond the packed kernels' 2048 rows run in the general kernel; with few pairs in flight the 64-row bands of ONE
    pair are spread over the waves of a workgroup (a pipeline of bands, a barrier per step).  Score, ends and statistics of long
    similar and dissimilar pairs -- ragged last band, a query of exactly 16 bands, one just beyond -- against the oracle, and the
    same pairs inside a batch large enough to take the one-wave-per-pair form."""
    rng = np.random.default_rng(8800 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2)
    [b.global_, b.semi_global, b.local][mode]()
    al, als = b.build(), b.use_stats().build()
    qs = [random_seqs(rng, 1, L, L)[0] for L in (2049, 2500, 1024 + 2048, 1025 + 2048, 4000)]
    rs = [mutate(rng, q, 0.1, 0.03) if k % 2 == 0 else random_seqs(rng, 1, 700, 3000)[0] for k, q in enumerate(qs)]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_stats_sample(mode, np.arange(len(qs)), qb, qo, rb, ro, 5, 2, om)
    import os
    os.environ["PMX_NO_LONG_KERNEL"] = "1"                         # (score-only calls of this size now take pmx_long32_kernel: tests/test_gpu_long.py)
    try:
        got = al.align_batch(qs, rs)
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_general_kernel"
        one = al.align(qs[0], rs[0])
    finally:
        del os.environ["PMX_NO_LONG_KERNEL"]
    assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() and (got["end_ref"] == want[:, 2]).all()
    rec, st = als.align_batch(qs, rs)
    assert (rec["score"] == want[:, 0]).all() and (st["matches"] == want[:, 3]).all() and (st["similar"] == want[:, 4]).all() \
        and (st["length"] == want[:, 5]).all()
    assert (one.get_score(), one.get_end_query(), one.get_end_ref()) == tuple(want[0, :3])
    many_q, many_r = qs[:2] * 40, rs[:2] * 40                      # 80 pairs: one wave per pair
    big = al.align_batch(many_q, many_r)
    assert (big["score"][:2] == want[:2, 0]).all() and (big["score"][::2] == want[0, 0]).all() and (big["end_ref"][1::2] == want[1, 2]).all()


@pytest.mark.parametrize("k", [0, 1, 2, 7, 15, 16, 31, 32, 48, 63])
def test_banded_local_packed_kernel_edges_and_low_complexity(pkg, orc, k, monkeypatch):
    """The packed int16 form of banded LOCAL alignment (two pairs per lane group, pad margins instead of edge checks) against the
    banded oracle AND the 32-bit staged form: one-symbol sequences, bands that miss the matrix or clip a corner, pairs without a
    single match (score 0: the end cell is the band's first cell in column-major order), an odd pair count, neighbours of very
    different lengths in one lane group, and low-complexity sequences whose out-of-band diagonals are long runs of matches."""
    rng = np.random.default_rng(8700 + k)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs, rs, dg = [], [], []
    for t in range(401):
        kind = t % 8
        if kind == 0:
            q = random_seqs(rng, 1, 1, 200)[0]; r = mutate(rng, q, 0.1, 0.05); d = int(rng.integers(-3, 4))
        elif kind == 1:
            q = random_seqs(rng, 1, 1, 3)[0]; r = random_seqs(rng, 1, 1, 300)[0]; d = int(rng.integers(-5, 300))
        elif kind == 2:
            q = b"A" * int(rng.integers(1, 300)); r = b"A" * int(rng.integers(1, 300)); d = int(rng.integers(-40, 40))
        elif kind == 3:
            q = b"ACGT" * int(rng.integers(1, 60)); r = (b"ACGT" * 80)[int(rng.integers(0, 4)):][:int(rng.integers(1, 300))]; d = int(rng.integers(-20, 20))
        elif kind == 4:
            q = b"A" * int(rng.integers(1, 50)); r = b"C" * int(rng.integers(1, 50)); d = int(rng.integers(-60, 60))        # no match at all
        elif kind == 5:
            q = random_seqs(rng, 1, 100, 250)[0]; r = random_seqs(rng, 1, 5, 40)[0] + mutate(rng, q, 0.05, 0.02); d = len(r) - len(q) + int(rng.integers(-2, 3))
        elif kind == 6:
            q = random_seqs(rng, 1, 20, 60)[0]; r = random_seqs(rng, 1, 20, 60)[0]; d = int(rng.integers(-200, 200))       # bands that miss the matrix
        else:
            q = random_seqs(rng, 1, 180, 250)[0]; r = mutate(rng, q, 0.2, 0.1); d = 0
        qs.append(q); rs.append(r); dg.append(d)
    dg = np.array(dg, dtype=np.int32)
    al = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).build()
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_banded_batch(orc.SW, qb, qo, rb, ro, 5, 2, om, k, dg)
    got = al.align_batch_banded(qs, rs, k, dg)
    assert pkg.lib.pmx_last_kernel().decode() == "pmx_bstrip_kernel/local"
    bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
    assert len(bad) == 0, (k, bad[:6], got[bad[:4]], want[bad[:4]], [(len(qs[x]), len(rs[x]), int(dg[x])) for x in bad[:4]])
    monkeypatch.setenv("PMX_BANDED_NO_STRIP", "1")
    ref = al.align_batch_banded(qs, rs, k, dg)
    assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_banded_packed_kernel")
    assert (ref == got).all()
    monkeypatch.setenv("PMX_BANDED_NO_PACKED", "1")
    ref = al.align_batch_banded(qs, rs, k, dg)
    assert pkg.lib.pmx_last_kernel().decode() == "pmx_banded_staged_kernel"
    assert (ref == got).all()


@pytest.mark.parametrize("seed", _seeds([0]))
@pytest.mark.parametrize("k", [0, 3, 15, 16, 31, 48, 63])
def test_banded_local_shared_query_rows(pkg, orc, k, seed, monkeypatch):
    """One shared query (a reused profile) over a small alphabet: the packed kernel's third form starts both pairs of a lane group on
    the same query row (one matrix row per lane for both).  Band centres far below the main diagonal (the band enters the matrix deep
    in the query: its lane-group partner starts early on cells outside the matrix), far above it, bands that miss the matrix,
    references much shorter than the query, one-symbol references; an odd pair count below the sort threshold and 4 500 pairs
    above it (processing order by entry row) -- against the banded oracle and against the per-pair-row forms."""
    rng = np.random.default_rng(9100 + k + 1000 * seed)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = random_seqs(rng, 1, 380, 381)[0]
    for n in (77, 4500):
        rs, dg = [], []
        for t in range(n):
            kind = t % 7
            if kind == 0:       # a copy of a piece of the query somewhere in a longer reference
                a = int(rng.integers(0, 300)); body = mutate(rng, q[a:a + int(rng.integers(20, 380 - a + 1))], 0.08, 0.04)
                pre = random_seqs(rng, 1, 0, 200)[0]; r = pre + body + random_seqs(rng, 1, 0, 100)[0]; d = len(pre) - a + int(rng.integers(-3, 4))
            elif kind == 1:     # short reference, band entering deep in the query
                r = random_seqs(rng, 1, 1, 60)[0]; d = -int(rng.integers(0, 380))
            elif kind == 2:     # far above the diagonal
                r = random_seqs(rng, 1, 300, 900)[0]; d = int(rng.integers(0, len(r)))
            elif kind == 3:     # may miss the matrix altogether
                r = random_seqs(rng, 1, 1, 120)[0]; d = int(rng.integers(-600, 300))
            elif kind == 4:
                r = q[int(rng.integers(0, 200)):]; d = -(len(q) - len(r))
            elif kind == 5:
                r = b"A" * int(rng.integers(1, 400)); d = int(rng.integers(-380, 400))
            else:
                r = random_seqs(rng, 1, 1, 3)[0]; d = int(rng.integers(-380, 3))
            rs.append(r or b"C"); dg.append(d)
        dg = np.array(dg, dtype=np.int32)
        prof = pkg.Profile.new(q, False, pm)
        al = pkg.Aligner.new().local().profile(prof).matrix(pm).gap_open(5).gap_extend(2).build()
        got = al.align_batch_banded([], rs, k, dg)
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_bstrip_kernel/local"
        rb, ro = orc.pack(rs)
        idx = np.arange(n) if n < 1000 else np.unique(np.concatenate([np.arange(0, n, 9), np.arange(140)]))
        rb2, ro2 = orc.pack([rs[t] for t in idx])
        want = orc.align_banded_batch(orc.SW, None, None, rb2, ro2, 5, 2, om, k, dg[idx], shared_query=q)
        bad = np.nonzero((got["score"][idx] != want[:, 0]) | (got["end_query"][idx] != want[:, 1]) | (got["end_ref"][idx] != want[:, 2]))[0]
        assert len(bad) == 0, (k, n, idx[bad[:6]], got[idx[bad[:4]]], want[bad[:4]], [(len(rs[idx[x]]), int(dg[idx[x]])) for x in bad[:4]])
        with monkeypatch.context() as mp:
            mp.setenv("PMX_BANDED_NO_STRIP", "1")
            ref = al.align_batch_banded([], rs, k, dg)
            assert pkg.lib.pmx_last_kernel().decode() == "pmx_banded_packed_kernel/shared query rows"
            assert (ref == got).all()
            mp.setenv("PMX_BANDED_NO_SHARED_ROWS", "1")
            ref = al.align_batch_banded([], rs, k, dg)
            assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_banded_packed_kernel")
            assert (ref == got).all()


@pytest.mark.parametrize("letters", [b"ACGTN", b"ACGTNR", b"ACGTNRY"])
def test_banded_local_row_lookup_forms_on_wider_alphabets(pkg, orc, letters):
    """The packed kernel's matrix-row forms read a query symbol's scores as 8 bytes and pick one with `v_perm_b32`: alphabets of 5 and 6
    letters (+ wildcard = 6 and 7 symbols, selectors into the second source register, the pad symbol last) take them, 7 letters
    (8 symbols) the byte-lookup form -- per-pair queries and one shared query, against the banded oracle."""
    rng = np.random.default_rng(9300 + len(letters))
    alpha = np.frombuffer(letters, dtype=np.uint8)
    pm, om = pkg.Matrix.create(letters, 3, -2), orc.Matrix.create(letters.decode(), 3, -2)
    q = random_seqs(rng, 1, 300, 301, alpha)[0]
    rs, dg = [], []
    for t in range(4200):
        a = int(rng.integers(0, 200))
        pre = random_seqs(rng, 1, 0, 120, alpha)[0]
        rs.append(pre + mutate(rng, q[a:a + int(rng.integers(30, 100))], 0.1, 0.04, alpha) + random_seqs(rng, 1, 0, 60, alpha)[0])
        dg.append(len(pre) - a + int(rng.integers(-4, 5)))
    dg = np.array(dg, dtype=np.int32)
    idx = np.arange(0, len(rs), 11)
    rb2, ro2 = orc.pack([rs[t] for t in idx])
    expect = {5: "/shared query rows", 6: "/shared query rows", 7: ""}[len(letters)]
    for band in (12, 40):
        al = pkg.Aligner.new().local().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(4).gap_extend(1).build()
        got = al.align_batch_banded([], rs, band, dg)
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_banded_packed_kernel" + expect, pkg.lib.pmx_last_kernel()
        want = orc.align_banded_batch(orc.SW, None, None, rb2, ro2, 4, 1, om, band, dg[idx], shared_query=q)
        assert (got["score"][idx] == want[:, 0]).all() and (got["end_query"][idx] == want[:, 1]).all() and (got["end_ref"][idx] == want[:, 2]).all()
        # per-pair queries: the per-pair row form
        qs = [q[int(rng.integers(0, 50)):] for _ in idx]
        al2 = pkg.Aligner.new().local().matrix(pm).gap_open(4).gap_extend(1).build()
        got2 = al2.align_batch_banded(qs, [rs[t] for t in idx], band, dg[idx])
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_banded_packed_kernel" + ("/matrix rows" if len(letters) < 7 else ""), pkg.lib.pmx_last_kernel()
        qb, qo = orc.pack(qs)
        want2 = orc.align_banded_batch(orc.SW, qb, qo, rb2, ro2, 4, 1, om, band, dg[idx])
        assert (got2["score"] == want2[:, 0]).all() and (got2["end_query"] == want2[:, 1]).all() and (got2["end_ref"] == want2[:, 2]).all()
