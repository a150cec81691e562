"""GPU parity tests (run with `-m gpu` on an MI355X).  Everything goes through the C ABI of
libparasail_amd.so (ctypes mirror of the parasail-rs interface) and is compared bit-for-bit with
the CPU oracle on the same seeded inputs; the reference's own known-answer tests are replayed
through the mirrored `Aligner` API so they read like tests/test_parasail.rs."""
import threading

import os

import numpy as np
import pytest

from util import random_seqs, mutate, score_from_cigar, cigar_ops, DNA, AA

pytestmark = pytest.mark.gpu

MODES = {"nw": 0, "sg": 1, "sw": 2}


def _builder(pkg, k):
    b = pkg.Aligner.new()
    {"nw": b.global_, "sg": b.semi_global, "sw": b.local}[k["mode"]]()
    ms = k["matrix"]
    matrix = pkg.Matrix.create(ms["alphabet"].encode(), ms["match"], ms["mismatch"])
    b.matrix(matrix).gap_open(k["open"]).gap_extend(k["extend"]).striped()
    if k["width"] != "sat":
        b.solution_width(int(k["width"]))
    if k["trace"]:
        b.use_trace()
    if k["stats"]:
        b.use_stats()
    if k["table"]:
        b.use_table()
    if k["rowcol"]:
        b.use_last_rowcol()
    if k["profile"]:
        b.profile(pkg.Profile.new(k["query"].encode(), k["profile_stats"], matrix))
    return b


def test_reference_kats_through_aligner(pkg, kats):
    """Every assertion of /root/reference/tests/test_parasail.rs, via the mirrored API."""
    n = 0
    for k in kats:
        q, r = k["query"].encode(), k["ref"].encode()
        e = k["expect"]
        if k["mode"] == "nw_banded":
            res = pkg.Aligner.new().bandwidth(2).build().banded_nw(q, r)
            assert res.get_score() == e["score"] and res.is_banded()
            n += 1
            continue
        if k["mode"] == "ssw":
            res = pkg.Aligner.new().build().ssw(q, r)
            assert (res.score(), res.query_end(), res.ref_end(), res.query_start(), res.ref_start()) == \
                   (e["score"], e["query_end"], e["ref_end"], e["query_start"], e["ref_start"])
            assert res.cigar_len() == 1 and res.cigar()[0] == (4 << 4) | 7
            n += 1
            continue
        al = _builder(pkg, k).build()
        res = al.align(None if k["profile"] else q, r)
        if "score" in e: assert res.get_score() == e["score"], k["name"]
        if "end_query" in e: assert res.get_end_query() == e["end_query"], k["name"]
        if "end_ref" in e: assert res.get_end_ref() == e["end_ref"], k["name"]
        if "matches" in e: assert res.get_matches() == e["matches"], k["name"]
        if "length" in e: assert res.get_length() == e["length"], k["name"]
        for t in ("score", "matches", "similar", "length"):
            if t + "_table_rows" in e:
                tab = getattr(res, "get_%s_table" % t)()
                assert (tab.rows(), tab.cols()) == (e[t + "_table_rows"], e[t + "_table_cols"]), k["name"]
                assert tab.get(0, 0) is not None
            if t + "_table_last" in e:
                assert getattr(res, "get_%s_table" % t)().last() == e[t + "_table_last"], k["name"]
            for w in ("row", "col"):
                key = "%s_%s" % (t, w)
                if key in e:
                    assert list(getattr(res, "get_" + key)()) == e[key], k["name"]
        if "trace_table_len" in e:
            tt = res.get_trace_table()
            assert (tt.rows(), tt.cols(), len(tt.as_slice())) == (4, 4, 16)
            for i in range(4):
                for j in range(4):
                    assert tt.get(i, j) in (0, 1, 2, 4)
            assert res.get_cigar(q, r) == "4="
            tb = res.get_traceback_strings(q, r)
            assert (tb.query, tb.comparison, tb.reference) == ("ACGT", "||||", "ACGT")
            res.print_traceback(q, r)
        for f, v in k["flags"].items():
            assert getattr(res, f)() == v, (k["name"], f)
        n += 1
    assert n == len(kats)


def test_multithread_shared_profile(pkg):
    """tests/test_parasail.rs:689-723"""
    m = pkg.Matrix.default()
    al = pkg.Aligner.new().profile(pkg.Profile.new(b"ACGT", True, m)).use_stats().striped().build()
    out = []

    def work(ref):
        out.append(al.clone().align(None, ref).get_score())
    th = [threading.Thread(target=work, args=(b"ACGT",)) for _ in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert out == [4, 4]


def test_scan_and_diag_names_are_aliases(pkg, orc):
    """`_scan` / `_diag` strategy slots (src/aligner/mod.rs:198-208) resolve to the same kernels; only the
    strategy predicate differs (src/alignment/mod.rs:446-460)."""
    rng = np.random.default_rng(909)
    q = random_seqs(rng, 1, 80, 80)[0]
    r = mutate(rng, q, 0.1, 0.05)
    om = orc.Matrix.create("ACGT", 2, -3)
    for mode in (0, 1, 2):
        want = orc.align(mode, q, r, 5, 2, om, stats=True)
        for strat in ("striped", "scan", "diag"):
            b = pkg.Aligner.new().matrix(pkg.Matrix.create(b"ACGT", 2, -3)).gap_open(5).gap_extend(2).use_stats()
            [b.global_, b.semi_global, b.local][mode]()
            getattr(b, strat)()
            res = b.build().align(q, r)
            assert (res.get_score(), res.get_end_query(), res.get_end_ref(), res.get_matches(), res.get_length()) == \
                   (want.score, want.end_query, want.end_ref, want.matches, want.length)
            assert (res.is_striped(), res.is_scan(), res.is_diag()) == (strat == "striped", strat == "scan", strat == "diag")
            assert not res.is_blocked() and not res.is_banded()


def test_cfg1_protein_pair_fixture(pkg):
    """BASELINE config 1: one protein pair, BLOSUM62, gaps 11/1, through Aligner::align() semantics (one-off, profile
    arm, stats, traceback) and through the batch entries; expected values from tests/golden/cfg1_protein_pair.json"""
    import json
    fx = json.load(open("tests/golden/cfg1_protein_pair.json"))
    q, r = fx["query"].encode(), fx["ref"].encode()
    pm = pkg.Matrix.from_name("blosum62")
    for name, sel in (("sw", "local"), ("nw", "global_"), ("sg", "semi_global")):
        c = fx["cases"][name]
        def mk():
            b = pkg.Aligner.new().matrix(pm).gap_open(fx["open"]).gap_extend(fx["extend"])
            getattr(b, sel)()
            return b
        res = mk().build().align(q, r)                                   # e.g. sw_striped_sat
        assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (c["score"], c["end_query"], c["end_ref"])
        st = mk().use_stats().build().align(q, r)
        assert (st.get_score(), st.get_matches(), st.get_similar(), st.get_length()) == \
            (c["score"], c["matches"], c["similar"], c["length"])
        tr = mk().use_trace().build()
        assert tr.align(q, r).get_cigar(q, r) == c["cigar"]
        prof = mk().profile(pkg.Profile.new(q, False, pm)).build().align(None, r)
        assert prof.get_score() == c["score"]
        rec = mk().solution_width(16).build().align_batch([q, q], [r, r])
        assert (rec["score"] == c["score"]).all() and (rec["end_query"] == c["end_query"]).all() and (rec["end_ref"] == c["end_ref"]).all()
        rec, cig = tr.align_batch_cigar([q], [r])
        assert cig[0] == c["cigar"] and rec["score"][0] == c["score"]


def test_cross_check_against_system_parasail_if_present(pkg):
    """if the box has the reference's real library, compare with it; the build image has none."""
    from oracle import parasail_probe
    lib = parasail_probe.load()
    if lib is None:
        pytest.skip("parasail oracle unavailable; parity is against the in-repo scalar oracle (pinned by the reference's KATs)")
    rng = np.random.default_rng(99)
    qs = random_seqs(rng, 300, 20, 150)
    rs = [mutate(rng, q, 0.1, 0.03) for q in qs]
    want = parasail_probe.align_batch(lib, b"sw_striped_16", qs, rs, 5, 2, b"ACGT", 2, -3)
    got = pkg.Aligner.new().local().matrix(pkg.Matrix.create(b"ACGT", 2, -3)).gap_open(5).gap_extend(2).solution_width(16).build().align_batch(qs, rs)
    assert [tuple(int(x) for x in (g["score"], g["end_query"], g["end_ref"])) for g in got] == want


def test_accessor_guards(pkg):
    res = pkg.Aligner.new().build().align(b"ACGT", b"ACGT")
    for f, exc in (("get_matches", pkg.NoStats), ("get_length", pkg.NoStats), ("get_score_table", pkg.NoTable),
                   ("get_matches_table", pkg.NoStatsTable), ("get_score_row", pkg.NoRowCol),
                   ("get_matches_col", pkg.NoRowCol), ("get_trace_table", pkg.NoTrace)):
        with pytest.raises(exc):
            getattr(res, f)()
    with pytest.raises(pkg.NoTrace):
        res.get_cigar(b"ACGT", b"ACGT")
    assert res.get_similar() == 0          # no guard in the reference (src/alignment/mod.rs:87-89)
    rc = pkg.Aligner.new().use_last_rowcol().build().align(b"ACGT", b"ACG")
    assert list(rc.get_score_row()) == [1, 2, 3] and rc.is_rowcol() and not rc.is_stats_rowcol()
    with pytest.raises(pkg.NoRowCol):
        rc.get_matches_row()


def test_cpp_mirror(pkg):
    """The header-only C++ mirror (parasail-rs_amd/cpp/parasail_rs.hpp) replays the reference KATs.

    History (hygiene): gpurun_out/pytest_gpu4.log of round 1 shows this binary exiting with SIGSEGV and EMPTY stdout and
    stderr at 22:33, one minute before the mirror's first commit (a2ea2af, 22:34).  That run used the uncommitted first draft of
    test_mirror.cpp / parasail_rs.hpp (the snapshot gpurun ships is the working tree), which no longer exists; the "[stage N]" markers
    on stderr were added to the committed version to localise it and it has not reproduced since (7 full-suite runs in round 2, plus
    the driver's round-end run).  Empty stderr means it died before stage 1, i.e. in the draft's first Matrix / Aligner construction,
    not in the HIP runtime's teardown (the library keeps no static object with a destructor that calls into HIP: its scratch,
    streams and events are thread-local PODs that are never destroyed at exit)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_mirror")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "cpp mirror ok" in p.stdout, p.stdout + p.stderr


# ------------------------------------------------------------- the hot kernel (sw, int16) ----
def _fast_case(pkg, orc, qs, rs, open_, ext, pm, om, width=16):
    b = pkg.Aligner.new().local().matrix(pm).gap_open(open_).gap_extend(ext)
    if width:
        b.solution_width(width)
    got = b.build().align_batch(qs, rs)
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, open_, ext, om)
    bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) |
                     (got["end_ref"] != want[:, 2]))[0]
    assert len(bad) == 0, (bad[:5], got[bad[:5]], want[bad[:5]], [(qs[i], rs[i]) for i in bad[:2]])
    assert (got["flags"] == 0).all()


@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (1, 1), (11, 1), (3, 0)])
def test_sw16_uniform_150(pkg, orc, gaps):
    rng = np.random.default_rng(1000 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 2001, 150, 150)
    rs = [mutate(rng, q)[:150].ljust(150, b"A") if i % 2 else random_seqs(rng, 1, 150, 150)[0]
          for i, q in enumerate(qs)]
    _fast_case(pkg, orc, qs, rs, gaps[0], gaps[1], pm, om)


def test_sw16_ragged_lengths(pkg, orc):
    rng = np.random.default_rng(1100)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 1500, 1, 160)
    rs = [mutate(rng, q, 0.1, 0.05) if rng.random() < 0.5 else random_seqs(rng, 1, 1, 300)[0] for q in qs]
    _fast_case(pkg, orc, qs, rs, 5, 2, pm, om)
    _fast_case(pkg, orc, qs[:1], rs[:1], 5, 2, pm, om)          # n = 1
    _fast_case(pkg, orc, [b"A"], [b"C"], 5, 2, pm, om)          # all-zero table -> (0, 0, 0)
    _fast_case(pkg, orc, [b"A"] * 3, [b"A"] * 3, 0, 0, pm, om)


def test_sw16_default_matrix_and_lowercase(pkg, orc):
    rng = np.random.default_rng(1200)
    qs = random_seqs(rng, 400, 5, 150)
    rs = [mutate(rng, q).lower() if i % 3 == 0 else mutate(rng, q) for i, q in enumerate(qs)]
    qs = [q + b"N" if i % 5 == 0 else q for i, q in enumerate(qs)]      # wildcard symbol
    _fast_case(pkg, orc, qs, rs, 0, 0, pkg.Matrix.default(), orc.Matrix.default(), width=0)   # sw_striped_sat
    _fast_case(pkg, orc, qs, rs, 2, 1, pkg.Matrix.default(), orc.Matrix.default())


@pytest.mark.parametrize("maxlen", [160, 256, 512, 1000, 2000])
def test_sw16_blosum62_all_instantiations(pkg, orc, maxlen):
    """one batch per (G,R) instantiation of the fast kernel"""
    rng = np.random.default_rng(1300 + maxlen)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    n = 120 if maxlen <= 512 else 40
    qs = random_seqs(rng, n, max(1, maxlen // 3), maxlen, AA)
    qs[0] = random_seqs(rng, 1, maxlen, maxlen, AA)[0]
    rs = [mutate(rng, q, 0.3, 0.05, AA) if rng.random() < 0.7 else random_seqs(rng, 1, 10, maxlen, AA)[0] for q in qs]
    _fast_case(pkg, orc, qs, rs, 11, 1, pm, om)


def test_sw16_file_matrix_asymmetric(pkg, orc):
    """scores[q][r] orientation: make the matrix asymmetric with set_value"""
    rng = np.random.default_rng(1400)
    pm = pkg.Matrix.create(b"ACGT", 2, -3)
    pm.set_value(0, 1, 4)       # query A vs reference C
    om = orc.Matrix.create("ACGT", 2, -3)
    om.scores[0, 1] = 4
    qs = random_seqs(rng, 300, 20, 150)
    rs = random_seqs(rng, 300, 20, 150)
    _fast_case(pkg, orc, qs, rs, 5, 2, pm, om)


@pytest.mark.parametrize("env", [{}, {"PMX_SW16_NO_PERMTABLE": "1"}, {"PMX_SW16_NO_SKEW": "1"}, {"PMX_SW16_NO_U8": "1"},
                                 {"PMX_SW16_NO_U8": "1", "PMX_SW16_NO_SKEW": "1"},
                                 {"PMX_SW16_VARIANT": "1"}, {"PMX_SW16_VARIANT": "0"}])
def test_sw16_every_arithmetic_variant(pkg, orc, monkeypatch, env):
    """the six arithmetic variants of the hot kernel (skewed / byte profile / VOP2 / max3 / saturating) agree with the oracle"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(1460)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 600, 1, 160)
    rs = [mutate(rng, q, 0.1, 0.05) if rng.random() < 0.6 else random_seqs(rng, 1, 1, 400)[0] for q in qs]
    for gaps in ((5, 2), (2, 2), (7, 0), (0, 0)):
        _fast_case(pkg, orc, qs, rs, gaps[0], gaps[1], pm, om)
    for qmax in (100, 104, 125, 128, 150, 152):                      # the 8-lane shapes <8,13> <8,16> <8,19>
        qs = random_seqs(rng, 2100, qmax - 30, qmax)                   # > 2048 pairs: the 8-lane shapes are for full chips
        qs[0] = random_seqs(rng, 1, qmax, qmax)[0]
        rs = [mutate(rng, q, 0.1, 0.05) if rng.random() < 0.7 else random_seqs(rng, 1, 1, 200)[0] for q in qs]
        rs[0] = qs[0]
        _fast_case(pkg, orc, qs, rs, 5, 2, pm, om)
    qs = random_seqs(rng, 60, 100, 700)
    rs = [mutate(rng, q, 0.05, 0.05) for q in qs]                    # long, high-scoring
    _fast_case(pkg, orc, qs, rs, 5, 2, pm, om)
    bm, bo = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    qs = random_seqs(rng, 100, 30, 256, AA)
    rs = [mutate(rng, q, 0.3, 0.05, AA) for q in qs]
    _fast_case(pkg, orc, qs, rs, 11, 1, bm, bo)


@pytest.mark.parametrize("qmax", [36, 50, 56, 75, 80, 100, 128, 150, 160])
def test_sw16_permtable_variant_and_wildcard_retry(pkg, orc, qmax):
    """batches of >= 4096 short DNA pairs take the perm-table kernel; pairs with a wildcard in the query are
    handed back on the device and redone with the LDS profile (no flag may leak out)"""
    rng = np.random.default_rng(1470 + qmax)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    n = 6000
    qs = random_seqs(rng, n, qmax // 2, qmax)
    qs[0] = random_seqs(rng, 1, qmax, qmax)[0]
    rs = [mutate(rng, q, 0.08, 0.04) if rng.random() < 0.7 else random_seqs(rng, 1, 1, 220)[0] for q in qs]
    for i in range(0, n, 37):                                         # wildcards in queries -> retry path
        q = bytearray(qs[i]); q[int(rng.integers(len(q)))] = ord("N"); qs[i] = bytes(q)
    for i in range(5, n, 41):                                         # wildcards in references: handled in place
        r = bytearray(rs[i]); r[int(rng.integers(len(r)))] = ord("N"); rs[i] = bytes(r)
    qs[1] = b"N" * 10; rs[1] = b"N" * 10
    qs[2] = qs[2].lower()
    _fast_case(pkg, orc, qs, rs, 5, 2, pm, om)
    _fast_case(pkg, orc, qs, rs, 3, 3, pm, om, width=0)
    same = [random_seqs(rng, 1, qmax, qmax)[0]] * 4100                # uniform lengths: no sort, all blocks full but the last
    _fast_case(pkg, orc, same, same, 5, 2, pm, om)


@pytest.mark.parametrize("qmax,n,rmax", [(250, 6000, 300), (500, 4200, 400), (1000, 4100, 300), (2000, 4100, 150)])
def test_sw16_permtable_larger_shapes(pkg, orc, qmax, n, rmax):
    """the perm-table variant in the 16 / 32 / 64-lane shapes"""
    rng = np.random.default_rng(1480 + qmax)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, n, max(1, qmax // 2), qmax)
    qs[0] = random_seqs(rng, 1, qmax, qmax)[0]
    rs = [mutate(rng, q, 0.08, 0.04)[:rmax] if rng.random() < 0.5 else random_seqs(rng, 1, 1, rmax)[0] for q in qs]
    for i in range(0, n, 53):
        q = bytearray(qs[i]); q[int(rng.integers(len(q)))] = ord("N"); qs[i] = bytes(q)
    _fast_case(pkg, orc, qs, rs, 5, 2, pm, om)


def test_sw16_permtable_shared_query(pkg, orc):
    """profile arm: one shared query; the perm-table variant is used only when the query holds no wildcard"""
    rng = np.random.default_rng(1490)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    rs = random_seqs(rng, 4500, 20, 400)
    for q in (random_seqs(rng, 1, 300, 300)[0], random_seqs(rng, 1, 149, 149)[0] + b"N", random_seqs(rng, 1, 1000, 1000)[0]):
        for k in range(0, len(rs), 9):
            rs[k] = mutate(rng, q, 0.05, 0.02)[:400]
        al = pkg.Aligner.new().local().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(5).gap_extend(2).solution_width(16).build()
        got = al.align_batch([], rs)
        qb, qo = orc.pack([q] * len(rs)); rb, ro = orc.pack(rs)
        want = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
        assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() \
            and (got["end_ref"] == want[:, 2]).all() and (got["flags"] == 0).all()


def test_sw16_width8_batches(pkg, orc):
    """`sw_striped_8` batches run in the fast kernel: local H >= 0, so the 8-bit saturation rule is "score > 127" """
    rng = np.random.default_rng(1495)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 5000, 10, 150)
    rs = [mutate(rng, q, 0.1, 0.03) if i % 2 else random_seqs(rng, 1, 10, 150)[0] for i, q in enumerate(qs)]
    al = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(8).build()
    assert al.fn_name == "sw_striped_8"
    got = al.align_batch(qs, rs)
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
    sat = want[:, 0] > 127
    assert sat.sum() > 100 and (~sat).sum() > 100
    assert ((got["flags"] & pkg.FLAG_SATURATED) != 0).tolist() == sat.tolist()
    ok = ~sat
    assert (got["score"][ok] == want[ok, 0]).all() and (got["end_query"][ok] == want[ok, 1]).all() and (got["end_ref"][ok] == want[ok, 2]).all()
    for k in (0, 1, 2, 3):
        w = orc.align(orc.SW, qs[k], rs[k], 5, 2, om, bits=8)
        assert bool(got["flags"][k] & pkg.FLAG_SATURATED) == bool(w.saturated)
        one = al.align(qs[k], rs[k])
        assert one.is_saturated() == bool(w.saturated)


@pytest.mark.parametrize("qlen", [100, 250, 320, 500, 1000, 2000])
def test_sw16q_shared_query_protein_database_search(pkg, orc, qlen):
    """the profile arm of local alignment (one protein query, BLOSUM62 11/1, many references): the workgroup-shared
    profile kernel in each of its shapes, ragged reference lengths (length-sorted order), planted homologs"""
    rng = np.random.default_rng(1600 + qlen)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, qlen, qlen, AA)[0]
    n = 400 if qlen <= 500 else 120
    rs = random_seqs(rng, n, 30, 900, AA)
    for k in range(0, n, 5):
        pos = int(rng.integers(0, max(1, len(rs[k]) - 10)))
        rs[k] = (rs[k][:pos] + mutate(rng, q, 0.3, 0.04, AA)[: 600] + rs[k][pos:])[:1200]
    rs[1] = rs[1].lower()
    qb, qo = orc.pack([q] * n); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, 11, 1, om)
    for width in (16, 0):
        b = pkg.Aligner.new().local().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(11).gap_extend(1)
        if width:
            b.solution_width(width)
        got = b.build().align_batch([], rs)
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_sw16q_kernel")
        assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() \
            and (got["end_ref"] == want[:, 2]).all() and (got["flags"] == 0).all()


@pytest.mark.parametrize("alpha", ["dna", "protein"])
def test_sw16_long_references_fetch_variant(pkg, orc, alpha):
    """per-pair queries against long references: reference symbols come from HBM instead of LDS (VAR 8)"""
    rng = np.random.default_rng(1700)
    if alpha == "dna":
        pm, om, al, go, ge = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3), DNA, 5, 2
    else:
        pm, om, al, go, ge = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt"), AA, 11, 1
    for qlo, qhi, n in ((30, 160, 60), (200, 320, 40), (600, 1000, 20)):      # (more than 16 pairs: fewer would take pmx_long32_kernel)
        qs = random_seqs(rng, n, qlo, qhi, al)
        rs = [random_seqs(rng, 1, 1000, 4000, al)[0] + mutate(rng, q, 0.15, 0.04, al) + random_seqs(rng, 1, 100, 2500, al)[0] for q in qs]
        rs[0] = rs[0][:1030]
        _fast_case(pkg, orc, qs, rs, go, ge, pm, om)
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_sw16_kernel")


@pytest.mark.parametrize("qhi,rhi", [(160, 400), (256, 300), (320, 1500), (512, 200), (1024, 120)])
def test_sw16m_matrix_lookup_per_pair_protein(pkg, orc, qhi, rhi):
    """per-pair protein batches (> 2048 pairs): no LDS profile, scores are read from the transposed matrix in LDS"""
    rng = np.random.default_rng(1800 + qhi)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    n = 2100
    qs = random_seqs(rng, n, max(1, qhi // 3), qhi, AA)
    qs[0] = random_seqs(rng, 1, qhi, qhi, AA)[0]
    rs = [mutate(rng, q, 0.3, 0.05, AA)[:rhi] if rng.random() < 0.5 else random_seqs(rng, 1, 5, rhi, AA)[0] for q in qs]
    rs[3] = rs[3].lower(); qs[4] = qs[4][: len(qs[4]) // 2] + b"X*" + qs[4][len(qs[4]) // 2 + 2:]
    _fast_case(pkg, orc, qs, rs, 11, 1, pm, om)
    assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_sw16m_kernel")
    _fast_case(pkg, orc, qs, rs, 11, 1, pm, om, width=0)


def test_sw16_saturating_int16_variant(pkg, orc):
    """scores too large for the max3 lanes (matrix max > 2048) take the saturating-int16 variant"""
    rng = np.random.default_rng(1450)
    pm, om = pkg.Matrix.create(b"ACGT", 3000, -3000), orc.Matrix.create("ACGT", 3000, -3000)
    qs = random_seqs(rng, 200, 1, 40)
    rs = [mutate(rng, q, 0.2, 0.1) for q in qs]
    got = pkg.Aligner.new().local().matrix(pm).gap_open(4000).gap_extend(100).solution_width(16).build().align_batch(qs, rs)
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, 4000, 100, om)
    ok = want[:, 0] <= 32767
    assert ok.sum() > 50 and (~ok).sum() > 5
    assert (got["score"][ok] == want[ok, 0]).all() and (got["end_query"][ok] == want[ok, 1]).all() \
        and (got["end_ref"][ok] == want[ok, 2]).all() and (got["flags"][ok] == 0).all()
    assert (got["flags"][~ok] & pkg.FLAG_SATURATED).all()
    sat = pkg.Aligner.new().local().matrix(pm).gap_open(4000).gap_extend(100).build().align_batch(qs, rs)
    assert (sat["score"] == want[:, 0]).all() and (sat["flags"] == 0).all()


def test_sw16_saturation_flag_and_promotion(pkg, orc):
    pm, om = pkg.Matrix.create(b"ACGT", 40, -40), orc.Matrix.create("ACGT", 40, -40)
    q = b"ACGT" * 250
    fixed = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).build()
    got = fixed.align_batch([q, b"ACGT"], [q, b"ACGT"])
    assert got["flags"][0] & pkg.FLAG_SATURATED and not got["flags"][1]
    assert got["score"][1] == 160
    one = fixed.align(q, q)
    assert one.is_saturated()
    assert orc.align(orc.SW, q, q, 5, 2, om, bits=16).saturated == 1
    sat = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).build()     # sw_striped_sat
    one = sat.align(q, q)
    assert not one.is_saturated() and one.get_score() == 40000
    # batch promotion: overflowing pairs are re-run in 32 bits, the others keep their int16 result
    rng = np.random.default_rng(1500)
    qs = [q, b"ACGT", q[:900], random_seqs(rng, 1, 300, 300)[0], q]
    rs = [q, b"ACGT", q[:900], random_seqs(rng, 1, 300, 300)[0], q[4:]]
    for al in (sat, pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(32).build()):
        got = al.align_batch(qs, rs)
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        want = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
        assert (got["flags"] == 0).all()
        assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() \
            and (got["end_ref"] == want[:, 2]).all()
        assert got["score"][0] == 40000 and got["score"][2] == 36000


# ------------------------------------------------------ fast global / semi-global kernel ----
def _nwsg_case(pkg, orc, mode, sg, qs, rs, open_, ext, pm, om, expect_kernel="pmx_nwsg16_kernel"):
    import ctypes as C
    b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext).solution_width(16)
    [b.global_, b.semi_global][mode]()
    if mode == 1 and sg is not None:
        qg = [n for f, n in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
        dg = [n for f, n in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
        b.allow_query_gaps(qg).allow_ref_gaps(dg)
    al = b.build()
    cfg = al._config()
    if expect_kernel:
        assert pkg.lib.pmx_kernel_for(C.byref(cfg), max(map(len, qs)), max(map(len, rs))).decode() == expect_kernel
    got = al.align_batch(qs, rs)
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(mode, qb, qo, rb, ro, open_, ext, om, sg_flags=sg if sg is not None else orc.SG_ALL, bits=16)
    bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
    assert len(bad) == 0, (mode, sg, open_, ext, bad[:5], got[bad[:5]], want[bad[:5]], [(qs[i], rs[i]) for i in bad[:2]])
    assert (got["flags"] == 0).all()


@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (1, 1), (11, 1), (3, 0)])
@pytest.mark.parametrize("mode", [0, 1])
def test_nwsg16_uniform_and_ragged(pkg, orc, mode, gaps):
    rng = np.random.default_rng(3000 + 10 * mode + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 600, 150, 150) + random_seqs(rng, 601, 1, 159)
    rs = [mutate(rng, q, 0.1, 0.03) if i % 3 else random_seqs(rng, 1, 1, 250)[0] for i, q in enumerate(qs)]
    _nwsg_case(pkg, orc, mode, None, qs, rs, gaps[0], gaps[1], pm, om)


def test_nwsg16_sg_variants(pkg, orc):
    rng = np.random.default_rng(3100)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 400, 5, 150)
    rs = []
    for q in qs:
        core = mutate(rng, q, 0.1, 0.05)
        rs.append(random_seqs(rng, 1, 1, 30)[0] + core + random_seqs(rng, 1, 1, 30)[0] if rng.random() < 0.5 else (core[4:-4] or core))
    for sg in (orc.S1_BEG, orc.S1_END, orc.S2_BEG, orc.S2_END, orc.S1_BEG | orc.S1_END, orc.S2_BEG | orc.S2_END,
               orc.S1_BEG | orc.S2_END, orc.S1_END | orc.S2_BEG, orc.S1_BEG | orc.S2_BEG, orc.S1_END | orc.S2_END,
               orc.S1_BEG | orc.S1_END | orc.S2_BEG, orc.SG_ALL):
        _nwsg_case(pkg, orc, 1, sg, qs, rs, 5, 2, pm, om)


@pytest.mark.parametrize("maxlen", [159, 255, 511, 1000, 2000])
def test_nwsg16_blosum62_all_instantiations(pkg, orc, maxlen):
    rng = np.random.default_rng(3200 + maxlen)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    n = 100 if maxlen <= 511 else 30
    qs = random_seqs(rng, n, max(1, maxlen // 3), maxlen, AA)
    qs[0] = random_seqs(rng, 1, maxlen, maxlen, AA)[0]
    rs = [mutate(rng, q, 0.3, 0.05, AA) if rng.random() < 0.7 else random_seqs(rng, 1, 10, maxlen, AA)[0] for q in qs]
    for mode in (0, 1):
        _nwsg_case(pkg, orc, mode, None, qs, rs, 11, 1, pm, om)


def test_nwsg16_first_generation_kernel(pkg, orc, monkeypatch):
    """the first-generation kernel (int16 profile, -inf scores) stays in use when score + open does not fit a byte"""
    rng = np.random.default_rng(3250)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 300, 1, 159) + random_seqs(rng, 40, 200, 500)
    rs = [mutate(rng, q, 0.1, 0.04) if i % 3 else random_seqs(rng, 1, 1, 300)[0] for i, q in enumerate(qs)]
    _nwsg_case(pkg, orc, 0, None, qs, rs, 2, 2, pm, om)            # mismatch + open < 0
    monkeypatch.setenv("PMX_NWSG16_GEN1", "1")
    for sg in (None, orc.S1_BEG | orc.S2_END, orc.S2_BEG, orc.S1_END):
        _nwsg_case(pkg, orc, 1, sg, qs, rs, 5, 2, pm, om)
    _nwsg_case(pkg, orc, 0, None, qs, rs, 5, 2, pm, om)


@pytest.mark.parametrize("qmax", [36, 50, 56, 75, 80, 100, 103, 127, 151, 159])
def test_nwsg16_eight_lane_shapes(pkg, orc, qmax):
    """the 8-lane shapes <8,13> <8,16> <8,19> <8,20> of the second-generation kernel (batches above 2048 pairs)"""
    rng = np.random.default_rng(3270 + qmax)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 2100, max(1, qmax - 40), qmax)
    qs[0] = random_seqs(rng, 1, qmax, qmax)[0]
    rs = [mutate(rng, q, 0.1, 0.04) if rng.random() < 0.7 else random_seqs(rng, 1, 1, 220)[0] for q in qs]
    for mode, sg in ((0, None), (1, None), (1, orc.S1_END | orc.S2_BEG)):
        _nwsg_case(pkg, orc, mode, sg, qs, rs, 5, 2, pm, om)
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16v_kernel<8,")


def test_nwsg16_long_references_and_skew_growth(pkg, orc):
    """second-generation kernel: the column skew grows with the reference length; long references, ext up to open"""
    rng = np.random.default_rng(3260)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 40, 50, 300)
    rs = [random_seqs(rng, 1, 2000, 3000)[0][:1500] + mutate(rng, q, 0.05, 0.02) + random_seqs(rng, 1, 200, 1200)[0] for q in qs]
    for mode, sg in ((0, None), (1, None), (1, orc.S2_BEG | orc.S2_END), (1, orc.S1_BEG | orc.S1_END)):
        _nwsg_case(pkg, orc, mode, sg, qs, rs, 5, 2, pm, om, expect_kernel=None)
        _nwsg_case(pkg, orc, mode, sg, qs, rs, 3, 3, pm, om, expect_kernel=None)


@pytest.mark.parametrize("qlen", [90, 159, 300, 511, 1000, 2000])
def test_nwsg16q_shared_query_profile_arm(pkg, orc, qlen):
    """global / semi-global with one reused protein profile: the workgroup-shared-profile kernel in each shape"""
    rng = np.random.default_rng(3400 + qlen)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, qlen, qlen, AA)[0]
    n = 300 if qlen <= 511 else 60
    rs = [mutate(rng, q, 0.3, 0.05, AA) if rng.random() < 0.5 else random_seqs(rng, 1, 20, qlen + 300, AA)[0] for _ in range(n)]
    qb, qo = orc.pack([q] * n); rb, ro = orc.pack(rs)
    for mode, sg in ((0, None), (1, None), (1, orc.S1_BEG | orc.S2_END), (1, orc.S2_BEG | orc.S2_END), (1, orc.S1_END)):
        b = pkg.Aligner.new().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(11).gap_extend(1).solution_width(16)
        [b.global_, b.semi_global][mode]()
        if mode == 1 and sg is not None:
            qg = [t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
            dg = [t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
            b.allow_query_gaps(qg).allow_ref_gaps(dg)
        got = b.build().align_batch([], rs)
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16q_kernel")
        want = orc.align_batch(mode, qb, qo, rb, ro, 11, 1, om, sg_flags=sg if sg is not None else orc.SG_ALL, bits=16)
        bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
        assert len(bad) == 0, (mode, sg, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.parametrize("qlen", [160, 256, 320, 512, 1024, 2048])
def test_nwsg16_query_fills_every_row(pkg, orc, qlen):
    """second generation, scores only: queries of exactly G * R rows (no virtual row on top), per-pair and shared"""
    rng = np.random.default_rng(3500 + qlen)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    n = 2100 if qlen == 160 else 40
    qs = random_seqs(rng, n, qlen, qlen, AA)
    rs = [mutate(rng, q, 0.3, 0.05, AA) if i % 2 else random_seqs(rng, 1, 20, qlen + 100, AA)[0] for i, q in enumerate(qs)]
    for mode, sg in ((0, None), (1, None), (1, orc.S1_BEG | orc.S2_BEG)):
        _nwsg_case(pkg, orc, mode, sg, qs, rs, 11, 1, pm, om)
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16m_kernel" if n > 2048 else "pmx_nwsg16v_kernel")
    q = qs[0]
    qb, qo = orc.pack([q] * n); rb, ro = orc.pack(rs)
    got = pkg.Aligner.new().global_().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(11).gap_extend(1).solution_width(16).build().align_batch([], rs)
    assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16q_kernel")
    want = orc.align_batch(0, qb, qo, rb, ro, 11, 1, om, bits=16)
    assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() and (got["end_ref"] == want[:, 2]).all()


@pytest.mark.parametrize("qhi,rhi", [(159, 300), (256, 256), (320, 1400), (500, 200), (1024, 100)])
def test_nwsg16m_matrix_lookup_per_pair_protein(pkg, orc, qhi, rhi):
    """global / semi-global, per-pair protein batches (> 2048 pairs): scores read from the transposed matrix in LDS"""
    rng = np.random.default_rng(3600 + qhi)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    n = 2100
    qs = random_seqs(rng, n, max(1, qhi // 3), qhi, AA)
    qs[0] = random_seqs(rng, 1, qhi, qhi, AA)[0]
    rs = [mutate(rng, q, 0.3, 0.05, AA)[:rhi] if rng.random() < 0.5 else random_seqs(rng, 1, 5, rhi, AA)[0] for q in qs]
    for mode, sg in ((0, None), (1, None), (1, orc.S1_BEG | orc.S2_END), (1, orc.S1_END | orc.S2_BEG), (1, orc.S2_BEG | orc.S2_END)):
        _nwsg_case(pkg, orc, mode, sg, qs, rs, 11, 1, pm, om)
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16m_kernel")


def test_nwsg16_falls_back_outside_the_exact_window(pkg, orc):
    """long sequences with large penalties leave the biased 16-bit window: the general kernel takes over"""
    rng = np.random.default_rng(3300)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 20, 100, 150)
    rs = random_seqs(rng, 20, 100, 150)
    _nwsg_case(pkg, orc, 0, None, qs, rs, 60, 50, pm, om, expect_kernel=None)      # lo bound ~ -15000
    _nwsg_case(pkg, orc, 0, None, qs, rs, 2, 5, pm, om, expect_kernel="pmx_general_kernel")   # open < extend at a fixed width: the general kernel tracks the range (sat / 32 / 64: pmx_long32_kernel)


# --------------------------------------------------------------------- general kernel ----
def _check_general(pkg, orc, mode, sg, q, r, open_, ext, pm, om, width=0):
    want = orc.align(mode, q, r, open_, ext, om, sg_flags=sg if sg is not None else orc.SG_ALL, bits=width,
                     stats=True, table=True, rowcol=True, trace=True)

    def mk():
        b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
        [b.global_, b.semi_global, b.local][mode]()
        if width:
            b.solution_width(width)
        if mode == 1 and sg is not None:
            qg = [n for f, n in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
            dg = [n for f, n in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
            b.allow_query_gaps(qg).allow_ref_gaps(dg)
        return b
    plain = mk().build().align(q, r)
    assert (plain.get_score(), plain.get_end_query(), plain.get_end_ref()) == \
           (want.score, want.end_query, want.end_ref), (mode, sg, q, r)
    assert plain.is_saturated() == bool(want.saturated)
    st = mk().use_stats().use_table().build().align(q, r)
    assert (st.get_score(), st.get_matches(), st.get_similar(), st.get_length()) == \
           (want.score, want.matches, want.similar, want.length), (mode, sg, q, r)
    for t in ("score", "matches", "similar", "length"):
        got = np.asarray(getattr(st, "get_%s_table" % t)().as_slice()).reshape(len(q), len(r))
        assert (got == getattr(want, t + "_table")).all(), (t, mode, q, r)
    rc = mk().use_stats().use_last_rowcol().build().align(q, r)
    for t in ("score", "matches", "similar", "length"):
        assert (np.asarray(getattr(rc, "get_%s_row" % t)()) == getattr(want, t + "_row")).all(), (t, mode)
        assert (np.asarray(getattr(rc, "get_%s_col" % t)()) == getattr(want, t + "_col")).all(), (t, mode)
    tr = mk().use_trace().build().align(q, r)
    got = np.asarray(tr.get_trace_table().as_slice()).reshape(len(q), len(r))
    assert (got == want.trace_table).all(), (mode, q, r)
    assert tr.get_cigar(q, r) == orc.cigar(want)
    tb = tr.get_traceback_strings(q, r)
    assert (tb.query, tb.comparison, tb.reference) == orc.traceback_strings(want)
    return want, tr


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_general_all_outputs_small(pkg, orc, mode):
    rng = np.random.default_rng(2000 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    for it in range(12):
        q = random_seqs(rng, 1, 1, 90)[0]
        r = mutate(rng, q, 0.15, 0.08) if it % 2 else random_seqs(rng, 1, 1, 90)[0]
        gaps = [(5, 2), (0, 0), (1, 1), (11, 1)][it % 4]
        want, tr = _check_general(pkg, orc, mode, None, q, r, gaps[0], gaps[1], pm, om)
        if mode != 1:
            bq, br = tr.get_cigar_begin(q, r)
            s, i, j = score_from_cigar(tr.get_cigar(q, r), q, r, bq, br, om.scores, om.mapper, *gaps)
            assert s == want.score


def test_general_multi_band_protein(pkg, orc):
    """queries longer than one 64-row band, BLOSUM62 11/1 (BASELINE config 1 style)"""
    rng = np.random.default_rng(2100)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    for ql in (64, 65, 128, 129, 200):
        q = random_seqs(rng, 1, ql, ql, AA)[0]
        r = mutate(rng, q, 0.3, 0.05, AA)
        for mode in (0, 1, 2):
            _check_general(pkg, orc, mode, None, q, r, 11, 1, pm, om)


def test_general_sg_variants(pkg, orc):
    rng = np.random.default_rng(2200)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    for sg in (orc.S1_BEG, orc.S1_END, orc.S2_BEG, orc.S2_END, orc.S1_BEG | orc.S1_END, orc.S2_BEG | orc.S2_END,
               orc.S1_BEG | orc.S2_END, orc.S1_END | orc.S2_BEG, orc.S1_BEG | orc.S2_BEG, orc.S1_END | orc.S2_END):
        for _ in range(3):
            q = random_seqs(rng, 1, 5, 80)[0]
            core = mutate(rng, q, 0.1, 0.05)
            r = random_seqs(rng, 1, 0, 20)[0] + core + random_seqs(rng, 1, 0, 20)[0] if rng.random() < 0.5 else core[3:-3] or core
            _check_general(pkg, orc, 1, sg, q, r, 5, 2, pm, om)


def test_general_fixed_width_saturation(pkg, orc):
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = b"ACGT" * 40
    for mode in (0, 1, 2):
        w8 = orc.align(mode, q, q, 5, 2, om, bits=8)
        assert w8.saturated == 1
        b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(8)
        [b.global_, b.semi_global, b.local][mode]()
        assert b.build().align(q, q).is_saturated()
    # negative overflow in nw: long unrelated sequences at 8 bit
    rng = np.random.default_rng(2300)
    q, r = random_seqs(rng, 1, 80, 80)[0], random_seqs(rng, 1, 5, 5)[0]
    want = orc.align(0, q, r, 5, 2, om, bits=8)
    got = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(8).build().align(q, r)
    assert got.is_saturated() == bool(want.saturated) and want.saturated == 1


@pytest.mark.parametrize("gaps", [(5, 2), (1, 1), (11, 1), (0, 0), (3, 0)])
@pytest.mark.parametrize("n", [300, 4200])
def test_nwsg_width8_range_tracking_in_the_packed_kernel(pkg, orc, gaps, n):
    """`nw_striped_8`, `sg*_striped_8` (src/aligner/mod.rs:125-130; saturation: src/alignment/mod.rs:436-440): the packed int16
    kernel computes the table and tracks the range of H; the flag equals the oracle's rule (some H, boundaries included, outside
    [-128, 127]) on every pair, and unsaturated pairs carry the exact score and ends.  Lengths straddle the int8 range."""
    rng = np.random.default_rng(2350 + gaps[0] * 7 + n)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, n, 1, 150)
    rs = [mutate(rng, q, 0.12, 0.05) if k % 3 else random_seqs(rng, 1, 1, 160)[0] for k, q in enumerate(qs)]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    idx = np.arange(n)
    seen = set()
    for mode, sg, qg, dg in ((0, 15, None, None), (1, 15, None, None), (1, 1 | 8, ["prefix"], ["suffix"]), (1, 2, ["suffix"], []), (1, 4 | 8, [], ["prefix", "suffix"])):
        b = pkg.Aligner.new().matrix(pm).gap_open(gaps[0]).gap_extend(gaps[1]).solution_width(8)
        [b.global_, b.semi_global][mode]()
        if qg is not None:
            b.allow_query_gaps(qg).allow_ref_gaps(dg)
        got = b.build().align_batch(qs, rs)
        # (mismatch + open >= 0: the window of the byte-profile arithmetic; outside it the general kernel tracks the range)
        assert ("pmx_nwsg16v_kernel" in pkg.lib.pmx_last_kernel().decode()) == (gaps[0] >= 3), pkg.lib.pmx_last_kernel()
        want = orc.align_stats_sample(mode, idx, qb, qo, rb, ro, gaps[0], gaps[1], om, sg_flags=sg, bits=8)
        sat = want[:, 6] == 1
        bad = np.nonzero((got["flags"] & 1) != want[:, 6])[0]
        assert len(bad) == 0, (mode, sg, gaps, bad[:5], [(len(qs[k]), len(rs[k])) for k in bad[:5]], want[bad[:5]])
        ok = ~sat
        assert (got["score"][ok] == want[ok, 0]).all() and (got["end_query"][ok] == want[ok, 1]).all() and (got["end_ref"][ok] == want[ok, 2]).all()
        seen |= {bool(x) for x in sat}
    assert seen == {True, False}


def test_nwsg_width8_protein_profile_arm_and_long_references(pkg, orc):
    rng = np.random.default_rng(2360)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, 12, 12, AA)[0]
    rs = random_seqs(rng, 3000, 1, 160, AA) + [mutate(rng, q, 0.2, 0.05, AA) for _ in range(200)]
    rb, ro = orc.pack(rs)
    idx = np.arange(len(rs))
    for mode in (0, 1):
        b = pkg.Aligner.new().matrix(pm).gap_open(11).gap_extend(1).solution_width(8).profile(pkg.Profile.new(q, False, pm))
        [b.global_, b.semi_global][mode]()
        got = b.build().align_batch([], rs)
        assert "pmx_nwsg16v_kernel" in pkg.lib.pmx_last_kernel().decode()
        want = orc.align_stats_sample(mode, idx, None, None, rb, ro, 11, 1, om, bits=8, shared_query=q)
        assert ((got["flags"] & 1) == want[:, 6]).all()
        ok = want[:, 6] == 0
        assert ok.any() and ((~ok).any() or mode == 1)
        assert (got["score"][ok] == want[ok, 0]).all() and (got["end_ref"][ok] == want[ok, 2]).all()
    # references of >= 1024 symbols (the fetch variant): every pair saturates at 8 bits, one way or the other
    dm, dom = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 2100, 20, 60); rl = random_seqs(rng, 2100, 1024, 1300)
    got = pkg.Aligner.new().semi_global().matrix(dm).gap_open(5).gap_extend(2).solution_width(8).build().align_batch(qs, rl)
    qb, qo = orc.pack(qs); rb2, ro2 = orc.pack(rl)
    want = orc.align_stats_sample(1, np.arange(0, 2100, 7), qb, qo, rb2, ro2, 5, 2, dom, bits=8)
    assert ((got["flags"][::7] & 1) == want[:, 6]).all()
    ok = want[:, 6] == 0
    assert (got["score"][::7][ok] == want[ok, 0]).all()


def test_general_batch_modes_and_stats(pkg, orc):
    rng = np.random.default_rng(2400)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 200, 1, 200)
    rs = [mutate(rng, q, 0.1, 0.05) for q in qs]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for mode in (0, 1, 2):
        b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).use_stats()
        [b.global_, b.semi_global, b.local][mode]()
        rec, st = b.build().align_batch(qs, rs)
        want = orc.align_batch(mode, qb, qo, rb, ro, 5, 2, om)
        assert (rec["score"] == want[:, 0]).all() and (rec["end_query"] == want[:, 1]).all() \
            and (rec["end_ref"] == want[:, 2]).all()
        for k in range(0, 200, 17):
            w = orc.align(mode, qs[k], rs[k], 5, 2, om, stats=True)
            assert (st["matches"][k], st["similar"][k], st["length"][k]) == (w.matches, w.similar, w.length)


def test_profile_batch_nw_stats_blosum62(pkg, orc):
    """BASELINE config 3 shape, reduced: one reused protein query, global + stats"""
    rng = np.random.default_rng(2500)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, 300, 300, AA)[0]
    rs = [mutate(rng, q, 0.4, 0.05, AA) + random_seqs(rng, 1, 0, 200, AA)[0] for _ in range(60)]
    prof = pkg.Profile.new(q, True, pm)
    al = pkg.Aligner.new().profile(prof).matrix(pm).gap_open(11).gap_extend(1).solution_width(16).build()
    assert al.fn_name == "nw_stats_striped_profile_16"
    rec, st = al.align_batch([], rs)
    for k, r in enumerate(rs):
        w = orc.align(0, q, r, 11, 1, om, stats=True, bits=16)
        assert (rec["score"][k], rec["end_query"][k], rec["end_ref"][k]) == (w.score, w.end_query, w.end_ref)
        assert (st["matches"][k], st["similar"][k], st["length"][k]) == (w.matches, w.similar, w.length)
        assert bool(rec["flags"][k] & 1) == bool(w.saturated)
    one = al.align(None, rs[0])
    assert one.get_score() == rec["score"][0] and one.get_matches() == st["matches"][0]


def test_profile_batch_sw_sat_mixed_lengths(pkg, orc):
    """BASELINE config 5 shape, reduced: one 1 kbp query (reused profile) against references of mixed
    length, `sw_striped_profile_sat`; a second matrix forces the int16 -> 32-bit promotion."""
    rng = np.random.default_rng(2550)
    q = random_seqs(rng, 1, 1000, 1000)[0]
    lens = np.exp(rng.uniform(np.log(500), np.log(5000), size=60)).astype(int)
    rs = [DNA[rng.integers(0, 4, size=int(l))].tobytes() for l in lens]
    for k in range(0, 60, 7):                       # plant noisy copies of the query
        pos = int(rng.integers(0, max(1, len(rs[k]) - 10)))
        rs[k] = rs[k][:pos] + mutate(rng, q, 0.05, 0.01) + rs[k][pos:]
    for match, mism in ((2, -3), (40, -40)):
        pm, om = pkg.Matrix.create(b"ACGT", match, mism), orc.Matrix.create("ACGT", match, mism)
        al = pkg.Aligner.new().local().profile(pkg.Profile.new(q, False, pm)).matrix(pm).gap_open(5).gap_extend(2).build()
        assert al.fn_name == "sw_striped_profile_sat"
        got = al.align_batch([], rs)
        qb, qo = orc.pack([q] * len(rs)); rb, ro = orc.pack(rs)
        want = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
        assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() \
            and (got["end_ref"] == want[:, 2]).all() and (got["flags"] == 0).all()
        if match == 40:
            assert got["score"].max() > 32767


def test_batch_cigar_semi_global(pkg, orc):
    """BASELINE config 4 shape, reduced: sg + traceback + CIGAR, walk done on the device"""
    rng = np.random.default_rng(2600)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 150, 200, 250)
    rs = [mutate(rng, q, 0.1, 0.02) for q in qs]
    for mode in (1, 0, 2):
        b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).use_trace()
        [b.global_, b.semi_global, b.local][mode]()
        rec, cig = b.build().align_batch_cigar(qs, rs)
        for k in range(len(qs)):
            w = orc.align(mode, qs[k], rs[k], 5, 2, om, trace=True)
            assert (rec["score"][k], rec["end_query"][k], rec["end_ref"][k]) == (w.score, w.end_query, w.end_ref)
            assert cig[k] == orc.cigar(w), (mode, k)


def test_cigar_letter_convention_is_selectable_at_run_time(pkg, orc, monkeypatch):
    """The letters of the two gap states are unpinned (include/pmx_conventions.h; the reference's test only prints its CIGAR,
    tests/test_parasail.rs:606-616).  PMX_CIGAR_SWAP_ID=1 exchanges I and D -- in get_cigar's packed ops and decoded text, in
    ssw's packed CIGAR and in batch CIGAR text -- without a rebuild; traceback strings and everything else are unchanged."""
    import re
    swap = str.maketrans("ID", "DI")
    bam = "MIDNSHP=X"
    rng = np.random.default_rng(2602)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 2200, 60, 140)
    rs = [mutate(rng, q, 0.08, 0.08) for q in qs]
    for env, tr in ((None, lambda t: t), ("1", lambda t: t.translate(swap)), ("0", lambda t: t)):
        if env is None:
            monkeypatch.delenv("PMX_CIGAR_SWAP_ID", raising=False)
        else:
            monkeypatch.setenv("PMX_CIGAR_SWAP_ID", env)
        seen = set()
        for mode in (1, 0, 2):
            b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).use_trace()
            [b.global_, b.semi_global, b.local][mode]()
            al = b.build()
            for n in (40, len(qs)):                                  # the host pipeline of small batches and the device render
                rec, cig = al.align_batch_cigar(qs[:n], rs[:n])
                for k in range(0, n, 1 if n == 40 else 37):
                    w = orc.align(mode, qs[k], rs[k], 5, 2, om, trace=True)
                    assert cig[k] == tr(orc.cigar(w)), (env, mode, n, k)
                    seen |= set(re.findall(r"[IDX=]", cig[k]))
            for k in range(6):                                       # one pair: get_cigar (ops + text) and the traceback strings
                res = al.align(qs[k], rs[k])
                w = orc.align(mode, qs[k], rs[k], 5, 2, om, trace=True)
                assert res.get_cigar(qs[k], rs[k]) == tr(orc.cigar(w))
                tb = res.get_traceback_strings(qs[k], rs[k])
                assert (tb.query, tb.comparison, tb.reference) == orc.traceback_strings(w)
        assert {"I", "D"} <= seen
        for k in range(20):                                          # ssw: packed ops
            res = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).build().ssw(qs[k], rs[k])
            w = orc.align(orc.SW, qs[k], rs[k], 5, 2, om, trace=True)
            want = [(int(n) << 4) | bam.index(c) for n, c in re.findall(r"(\d+)([=XID])", tr(orc.cigar(w)))]
            assert [res.cigar()[x] for x in range(res.cigar_len())] == want


def test_batch_cigar_text_outlives_every_derived_view(pkg):
    """The packed CIGAR text is a view of the callee's block; the block must stay alive for DERIVED arrays too (numpy
    collapses base chains: np.asarray / .view(np.ndarray) / slices drop a subclass attribute), also across a second call
    that would otherwise be handed the recycled block."""
    import gc
    rng = np.random.default_rng(2601)
    pm = pkg.Matrix.create(b"ACGT", 2, -3)
    a = pkg.Aligner.new().semi_global().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).use_trace().build()
    qs = random_seqs(rng, 300, 100, 150)
    rs = [mutate(rng, q, 0.1, 0.02) for q in qs]
    qb, qo = pkg.pack(qs)
    rb, ro = pkg.pack(rs)
    _, text, coff = a.align_batch_cigar_packed(qb, qo, rb, ro)
    want = text.tobytes()
    kept = [np.asarray(text), text.view(np.ndarray), text[3:], np.ascontiguousarray(text)]
    del text
    gc.collect()
    qs2 = random_seqs(rng, 300, 100, 150)                      # another batch of the same size: same pool bucket
    rs2 = [mutate(rng, q, 0.3, 0.05) for q in qs2]
    qb2, qo2 = pkg.pack(qs2)
    rb2, ro2 = pkg.pack(rs2)
    _, text2, _ = a.align_batch_cigar_packed(qb2, qo2, rb2, ro2)
    assert text2.tobytes() != want
    assert kept[0].tobytes() == want and kept[1].tobytes() == want and kept[3].tobytes() == want
    assert kept[2].tobytes() == want[3:]
    assert text2.ctypes.data != kept[0].ctypes.data


def _stats_case(pkg, orc, mode, sg, qs, rs, open_, ext, pm, om, shared_query=None):
    b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext).solution_width(16)
    [b.global_, b.semi_global, b.local][mode]()
    if mode == 1 and sg is not None:
        qg = [n for f, n in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
        dg = [n for f, n in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
        b.allow_query_gaps(qg).allow_ref_gaps(dg)
    if shared_query is not None:
        b.profile(pkg.Profile.new(shared_query, True, pm))
        rec, st = b.build().align_batch([], rs)
        qs = [shared_query] * len(rs)
    else:
        rec, st = b.use_stats().build().align_batch(qs, rs)
    for k in range(len(rs)):
        w = orc.align(mode, qs[k], rs[k], open_, ext, om, sg_flags=sg if sg is not None else orc.SG_ALL, stats=True)
        got = (rec["score"][k], rec["end_query"][k], rec["end_ref"][k], st["matches"][k], st["similar"][k], st["length"][k])
        want = (w.score, w.end_query, w.end_ref, w.matches, w.similar, w.length)
        assert got == want, (mode, sg, k, got, want, qs[k], rs[k])


@pytest.mark.parametrize("gaps", [(5, 2), (1, 1), (11, 1), (3, 3)])
@pytest.mark.parametrize("by_trace", [True, False])
def test_stats16_gap_models(pkg, orc, gaps, monkeypatch, by_trace):
    # small alphabets: statistics by traceback for >= 2048 pairs (forced here), the statistics kernel otherwise
    monkeypatch.setenv("PMX_STATS_BY_TRACE" if by_trace else "PMX_NO_STATS_BY_TRACE", "1")
    rng = np.random.default_rng(5000 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 300, 1, 160)          # ragged and >= 256 pairs: also exercises the length-sorted order
    rs = [mutate(rng, q, 0.12, 0.06) if i % 4 else random_seqs(rng, 1, 1, 300)[0] for i, q in enumerate(qs)]
    for mode in (0, 1):
        _stats_case(pkg, orc, mode, None, qs, rs, gaps[0], gaps[1], pm, om)


@pytest.mark.parametrize("gaps", [(5, 2), (1, 1), (11, 1)])
@pytest.mark.parametrize("by_trace", [True, False])
def test_stats16_local(pkg, orc, gaps, monkeypatch, by_trace):
    # small alphabets: statistics by traceback for >= 2048 pairs (forced here), the statistics kernel otherwise
    monkeypatch.setenv("PMX_STATS_BY_TRACE" if by_trace else "PMX_NO_STATS_BY_TRACE", "1")
    rng = np.random.default_rng(5050 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 300, 1, 160)
    rs = [random_seqs(rng, 1, 0, 40)[0] + mutate(rng, q, 0.12, 0.06) + random_seqs(rng, 1, 0, 40)[0] if i % 4
          else random_seqs(rng, 1, 1, 300)[0] for i, q in enumerate(qs)]
    qs += [b"A", b"ACGT"]; rs += [b"C", b"TTTT"]
    _stats_case(pkg, orc, 2, None, qs, rs, gaps[0], gaps[1], pm, om)
    b62, ob62 = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, 300, 300, AA)[0]
    rs2 = [random_seqs(rng, 1, 0, 100, AA)[0] + mutate(rng, q, 0.4, 0.05, AA)[50:250] + random_seqs(rng, 1, 0, 300, AA)[0] for _ in range(40)]
    _stats_case(pkg, orc, 2, None, None, rs2, 11, 1, b62, ob62, shared_query=q)


@pytest.mark.parametrize("by_trace", [True, False])
def test_stats16_sg_variants_and_sizes(pkg, orc, monkeypatch, by_trace):
    # small alphabets: statistics by traceback for >= 2048 pairs (forced here), the statistics kernel otherwise
    monkeypatch.setenv("PMX_STATS_BY_TRACE" if by_trace else "PMX_NO_STATS_BY_TRACE", "1")
    rng = np.random.default_rng(5100)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 60, 20, 150)
    rs = [random_seqs(rng, 1, 0, 25)[0] + mutate(rng, q, 0.1, 0.05) + random_seqs(rng, 1, 0, 25)[0] for q in qs]
    for sg in (orc.S1_BEG, orc.S1_END, orc.S2_BEG, orc.S2_END, orc.S1_BEG | orc.S1_END, orc.S2_BEG | orc.S2_END,
               orc.S1_END | orc.S2_BEG, orc.S1_BEG | orc.S2_END):
        _stats_case(pkg, orc, 1, sg, qs, rs, 5, 2, pm, om)
    for maxlen in (160, 256, 320, 512, 1024, 1100):      # one per instantiation, last one -> general kernel
        q2 = random_seqs(rng, 10, maxlen // 2, maxlen)
        q2[0] = random_seqs(rng, 1, maxlen, maxlen)[0]
        r2 = [mutate(rng, q, 0.1, 0.03) for q in q2]
        _stats_case(pkg, orc, 0, None, q2, r2, 5, 2, pm, om)
        _stats_case(pkg, orc, 1, None, q2, r2, 5, 2, pm, om)


@pytest.mark.parametrize("gaps", [(5, 2), (3, 3), (11, 1)])
def test_stats16p_packed_kernel_everywhere(pkg, orc, monkeypatch, gaps):
    """the packed statistics kernel forced for every global / semi-global case it can express (it is normally used
    for the profile arm without free ends only): all free-end sets, ragged lengths, ties (open == ext)"""
    monkeypatch.setenv("PMX_STATS16P_ALWAYS", "1")
    monkeypatch.setenv("PMX_NO_STATS_BY_TRACE", "1")
    rng = np.random.default_rng(5300 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 200, 1, 300)
    rs = [mutate(rng, q, 0.12, 0.06) if i % 4 else random_seqs(rng, 1, 1, 400)[0] for i, q in enumerate(qs)]
    _stats_case(pkg, orc, 0, None, qs, rs, gaps[0], gaps[1], pm, om)
    assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_stats16p_kernel")
    for sg in (orc.S1_BEG, orc.S2_END, orc.S1_END | orc.S2_BEG, orc.S1_BEG | orc.S2_BEG, orc.SG_ALL):
        _stats_case(pkg, orc, 1, sg, qs[:80], rs[:80], gaps[0], gaps[1], pm, om)
    q2 = random_seqs(rng, 12, 400, 640)
    r2 = [mutate(rng, q, 0.1, 0.03) for q in q2]
    _stats_case(pkg, orc, 0, None, q2, r2, gaps[0], gaps[1], pm, om)


def test_stats_by_traceback_protein_short_references(pkg, orc):
    """large alphabets, per-pair queries, short references, a full batch: the matrix-lookup statistics kernel without
    free ends, statistics counted along the packed traceback otherwise"""
    rng = np.random.default_rng(5400)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    qs = random_seqs(rng, 2100, 40, 200, AA)
    rs = [mutate(rng, q, 0.3, 0.06, AA) if i % 3 else random_seqs(rng, 1, 10, 260, AA)[0] for i, q in enumerate(qs)]
    for mode, sg, route in ((2, None, "trace"), (0, None, "lookup"), (1, None, "trace"), (1, orc.S1_BEG | orc.S2_END, "trace"),
                            (1, orc.S1_BEG | orc.S2_BEG, "lookup")):
        _stats_case(pkg, orc, mode, sg, qs, rs, 11, 1, pm, om)
        k = pkg.lib.pmx_last_kernel().decode()
        # no free end: the packed statistics kernel with matrix lookup; otherwise counts along the packed traceback
        assert k.endswith("pmx_walkp_kernel/stats") if route == "trace" else k.startswith("pmx_stats16p_kernel") and k.endswith("matrix lookup")


def test_stats16p_matrix_lookup_long_references(pkg, orc):
    """config 3's one-off form: per-pair protein queries against long references, global + statistics"""
    rng = np.random.default_rng(5500)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    for qlo, qhi, n in ((40, 160, 24), (250, 320, 16), (500, 640, 8)):
        qs = random_seqs(rng, n, qlo, qhi, AA)
        rs = [random_seqs(rng, 1, 300, 1500, AA)[0] + mutate(rng, q, 0.3, 0.05, AA) + random_seqs(rng, 1, 300, 1500, AA)[0] for q in qs]
        for mode, sg in ((0, None), (1, orc.S2_BEG), (1, orc.S1_BEG | orc.S2_BEG)):
            _stats_case(pkg, orc, mode, sg, qs, rs, 11, 1, pm, om)
            assert pkg.lib.pmx_last_kernel().decode().endswith("matrix lookup")


def test_stats16_shared_query_blosum62(pkg, orc):
    """config 3 shape through the 4-wave shared-profile variant"""
    rng = np.random.default_rng(5200)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, 300, 300, AA)[0]
    rs = [mutate(rng, q, 0.4, 0.05, AA) + random_seqs(rng, 1, 0, 300, AA)[0] for _ in range(41)]
    rs += random_seqs(rng, 9, 800, 1500, AA)
    for mode in (0, 1):
        _stats_case(pkg, orc, mode, None, None, rs, 11, 1, pm, om, shared_query=q)


def _cigar_case(pkg, orc, mode, sg, qs, rs, open_, ext, pm, om):
    b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext).solution_width(16).use_trace()
    [b.global_, b.semi_global, b.local][mode]()
    if mode == 1 and sg is not None:
        qg = [n for f, n in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
        dg = [n for f, n in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
        b.allow_query_gaps(qg).allow_ref_gaps(dg)
    rec, cig = b.build().align_batch_cigar(qs, rs)
    for k in range(len(qs)):
        w = orc.align(mode, qs[k], rs[k], open_, ext, om, sg_flags=sg if sg is not None else orc.SG_ALL, trace=True)
        assert (rec["score"][k], rec["end_query"][k], rec["end_ref"][k]) == (w.score, w.end_query, w.end_ref), (mode, sg, k)
        assert cig[k] == orc.cigar(w), (mode, sg, k, qs[k], rs[k], cig[k], orc.cigar(w))


@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (1, 1), (11, 1), (3, 0)])
def test_trace16_cigar_gap_models(pkg, orc, gaps):
    """fast traceback kernel (4-bit trace + device walk) against the oracle's byte trace + walk"""
    rng = np.random.default_rng(4000 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 120, 1, 250)
    rs = [mutate(rng, q, 0.12, 0.06) if i % 4 else random_seqs(rng, 1, 1, 300)[0] for i, q in enumerate(qs)]
    for mode in (0, 1):
        _cigar_case(pkg, orc, mode, None, qs, rs, gaps[0], gaps[1], pm, om)


def test_trace16_cigar_sg_variants_and_sizes(pkg, orc):
    rng = np.random.default_rng(4100)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 60, 20, 250)
    rs = [random_seqs(rng, 1, 0, 25)[0] + mutate(rng, q, 0.1, 0.05) + random_seqs(rng, 1, 0, 25)[0] for q in qs]
    for sg in (orc.S1_BEG, orc.S1_END, orc.S2_BEG, orc.S2_END, orc.S1_BEG | orc.S1_END, orc.S2_BEG | orc.S2_END,
               orc.S1_END | orc.S2_BEG, orc.S1_BEG | orc.S2_END):
        _cigar_case(pkg, orc, 1, sg, qs, rs, 5, 2, pm, om)
    for maxlen in (255, 400, 511, 700, 1023, 1100):      # <32,8>, <64,8>, <64,16>, general kernel
        q2 = random_seqs(rng, 12, maxlen // 2, maxlen)
        q2[0] = random_seqs(rng, 1, maxlen, maxlen)[0]
        r2 = [mutate(rng, q, 0.1, 0.03) for q in q2]
        _cigar_case(pkg, orc, 1, None, q2, r2, 5, 2, pm, om)
        _cigar_case(pkg, orc, 0, None, q2, r2, 5, 2, pm, om)


@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (1, 1), (11, 1), (3, 0)])
def test_trace16_cigar_local(pkg, orc, gaps):
    """sw_trace through the fast traceback kernel: zero floor in the DP, the walk stops where the score is used up"""
    rng = np.random.default_rng(4300 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 150, 1, 250)
    rs = [random_seqs(rng, 1, 0, 40)[0] + mutate(rng, q, 0.12, 0.06) + random_seqs(rng, 1, 0, 40)[0] if i % 4
          else random_seqs(rng, 1, 1, 300)[0] for i, q in enumerate(qs)]
    qs += [b"A", b"ACGT"]; rs += [b"C", b"TTTT"]                  # all-zero tables
    _cigar_case(pkg, orc, 2, None, qs, rs, gaps[0], gaps[1], pm, om)
    q2 = random_seqs(rng, 10, 300, 1000)
    r2 = [mutate(rng, q, 0.1, 0.03) for q in q2]
    _cigar_case(pkg, orc, 2, None, q2, r2, gaps[0], gaps[1], pm, om)
    if gaps == (5, 2):
        assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_sw16_kernel/packed trace")


def test_cigar_short_queries_long_references(pkg, orc):
    """traceback shapes are picked by LDS footprint too: short queries against long references move to wider lane groups"""
    rng = np.random.default_rng(4340)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 24, 40, 120)
    rs = [random_seqs(rng, 1, 3000, 9000)[0] + mutate(rng, q, 0.08, 0.03) + random_seqs(rng, 1, 1000, 3000)[0] for q in qs]
    for mode, sg, ext in ((2, None, 2), (1, None, 1), (1, orc.S2_BEG | orc.S2_END, 1)):   # (ext 2 would leave the 16-bit window in sg)
        _cigar_case(pkg, orc, mode, sg, qs, rs, 5, ext, pm, om)
        assert "packed trace" in pkg.lib.pmx_last_kernel().decode()


def test_trace16_first_generation_kernels(pkg, orc, monkeypatch):
    """the unpacked traceback kernel stays in use outside the second generation's window (forced here)"""
    monkeypatch.setenv("PMX_TRACE16_GEN1", "1")
    rng = np.random.default_rng(4350)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 120, 1, 250)
    rs = [mutate(rng, q, 0.12, 0.06) if i % 3 else random_seqs(rng, 1, 1, 300)[0] for i, q in enumerate(qs)]
    for mode, sg in ((2, None), (1, None), (0, None), (1, orc.S1_BEG | orc.S2_END)):
        _cigar_case(pkg, orc, mode, sg, qs, rs, 5, 2, pm, om)
    assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_trace16_kernel")


def test_batch_cigar_chunking(pkg, orc, monkeypatch):
    """the batch CIGAR entry processes chunks that bound the trace scratch: force several chunks"""
    rng = np.random.default_rng(4150)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 70, 50, 200)
    rs = [mutate(rng, q, 0.1, 0.04) for q in qs]
    monkeypatch.setenv("PMX_CIGAR_CHUNK_BYTES", "300000")        # ~5-10 pairs per chunk
    _cigar_case(pkg, orc, 1, None, qs, rs, 5, 2, pm, om)
    monkeypatch.setenv("PMX_CIGAR_CHUNK_BYTES", "1")             # one pair per chunk
    _cigar_case(pkg, orc, 0, None, qs[:9], rs[:9], 5, 2, pm, om)


def test_trace16_cigar_blosum62(pkg, orc):
    rng = np.random.default_rng(4200)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    qs = random_seqs(rng, 60, 30, 250, AA)
    rs = [mutate(rng, q, 0.3, 0.06, AA) for q in qs]
    for mode in (0, 1):
        _cigar_case(pkg, orc, mode, None, qs, rs, 11, 1, pm, om)


def test_pssm_single_pair(pkg, orc):
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    q, r = b"MKVLAAGIVGL", b"MKVIAGGLVGL"
    ps = pm.to_pssm(q)
    got = pkg.Aligner.new().matrix(ps).gap_open(11).gap_extend(1).build().align(q, r)
    assert got.get_score() == orc.align(0, q, r, 11, 1, om).score


def test_banded_matches_full_when_band_is_wide(pkg, orc):
    rng = np.random.default_rng(2700)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    for _ in range(10):
        q = random_seqs(rng, 1, 20, 150)[0]
        r = mutate(rng, q, 0.1, 0.03)
        full = orc.align(0, q, r, 5, 2, om).score
        wide = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).bandwidth(400).build().banded_nw(q, r)
        assert wide.get_score() == full and wide.is_banded() and wide.is_global()
        narrow = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).bandwidth(20).build().banded_nw(q, r)
        assert narrow.get_score() <= full


# ------------------------------------------------------------------------------- fuzz ----
def _seeds(default):
    """PMX_FUZZ_SEEDS=a:b widens a fuzz test's seed list for soak runs (the committed default keeps the suite short)."""
    spec = os.environ.get("PMX_FUZZ_SEEDS")
    if not spec:
        return default
    a, b = spec.split(":")
    return list(range(int(a), int(b)))


@pytest.mark.parametrize("seed", _seeds([101, 102, 103, 5018]))
def test_fuzz_fast_kernel_windows(pkg, orc, seed):
    """Randomised batches aimed at the edges of the fast kernels' exact windows: custom 4-letter matrices whose
    score + open touches 0 and 255, gap models with ext = 0 / ext = open, long references (large column skew),
    wildcards and lower case, batch sizes on both sides of the perm-table threshold; score and end positions
    of all three modes against the oracle."""
    rng = np.random.default_rng(seed)
    for it in range(14):
        match = int(rng.choice([1, 2, 5, 40, 200]))
        mism = -int(rng.choice([1, 3, 4, 30]))
        open_ = int(rng.choice([-mism, -mism + 1, 255 - match, 5, 11, 60]))
        open_ = max(0, min(open_, 255 - match))
        ext = int(rng.choice([0, 1, 2, open_])) if open_ else 0
        ext = min(ext, open_)
        pm, om = pkg.Matrix.create(b"ACGT", match, mism), orc.Matrix.create("ACGT", match, mism)
        shape = int(rng.integers(0, 4))
        if shape == 0:   n, qlo, qhi, rlo, rhi = 4500, 20, 150, 20, 200          # perm-table threshold crossed
        elif shape == 1: n, qlo, qhi, rlo, rhi = 40, 50, 200, 2500, 7000         # long references: skew growth
        elif shape == 2: n, qlo, qhi, rlo, rhi = 300, 200, 1000, 100, 500
        else:            n, qlo, qhi, rlo, rhi = 2200, 90, 128, 60, 150
        qs = random_seqs(rng, n, qlo, qhi)
        rs = []
        for q in qs:
            r = random_seqs(rng, 1, rlo, rhi)[0]
            if rng.random() < 0.6:
                pos = int(rng.integers(0, max(1, len(r) - len(q))))
                r = (r[:pos] + mutate(rng, q, 0.1, 0.04) + r[pos + len(q):])[:rhi]
            rs.append(r)
        for i in range(0, n, 29):
            q = bytearray(qs[i]); q[int(rng.integers(len(q)))] = ord("N"); qs[i] = bytes(q)
        for i in range(3, n, 31):
            r = bytearray(rs[i]); r[int(rng.integers(len(r)))] = ord("n"); rs[i] = bytes(r)
        qs[0] = qs[0].lower()
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        for mode in (2, 1, 0):
            if mode != 2 and shape == 1 and rng.random() < 0.5:
                continue
            sg = int(rng.integers(1, 16)) if mode == 1 else None
            b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
            [b.global_, b.semi_global, b.local][mode]()
            if mode == 1:
                qg = [t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
                dg = [t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
                b.allow_query_gaps(qg).allow_ref_gaps(dg)
            got = b.build().align_batch(qs, rs)                     # `sat`: promotion hides overflow
            want = orc.align_batch(mode, qb, qo, rb, ro, open_, ext, om, sg_flags=sg if sg is not None else orc.SG_ALL)
            bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
            assert len(bad) == 0, (seed, it, match, mism, open_, ext, shape, mode, sg, bad[:5], got[bad[:5]], want[bad[:5]])
            assert (got["flags"] == 0).all()


def test_fuzz_every_dispatch_path(pkg, orc):
    """Randomised differential test over the whole batch dispatcher: random mode, free-end set, gap
    model, matrix, width, length distribution and output kind (records / stats / CIGAR), every
    result compared with the oracle.  Seeds are fixed; failures print the configuration."""
    rng = np.random.default_rng(77)
    mats = [("dna23", pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3), DNA),
            ("dna11", pkg.Matrix.default(), orc.Matrix.default(), DNA),
            ("dna54", pkg.Matrix.create(b"ACGTN", 5, -4), orc.Matrix.create("ACGTN", 5, -4), DNA),
            ("b62", pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt"), AA)]
    for it in range(60):
        name, pm, om, alpha = mats[int(rng.integers(0, len(mats)))]
        mode = int(rng.integers(0, 3))
        sg = int(rng.integers(1, 16)) if mode == 1 else None
        open_ = int(rng.choice([0, 1, 2, 5, 11, 20]))
        ext = int(rng.choice([0, 1, 2, 5]))
        width = int(rng.choice([0, 16, 32]))
        kind = ["rec", "stats", "cigar"][int(rng.integers(0, 3))]
        n = int(rng.choice([1, 2, 7, 33, 300]))
        lo, hi = [(1, 30), (20, 160), (100, 400), (1, 700)][int(rng.integers(0, 4))]
        qs = random_seqs(rng, n, lo, hi, alpha)
        rs = [mutate(rng, q, 0.15, 0.05, alpha) if rng.random() < 0.7 else random_seqs(rng, 1, lo, hi, alpha)[0] for q in qs]
        b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
        [b.global_, b.semi_global, b.local][mode]()
        if width:
            b.solution_width(width)
        if mode == 1:
            qg = [t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
            dg = [t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
            b.allow_query_gaps(qg).allow_ref_gaps(dg)
            if not qg and not dg:
                sg = orc.SG_ALL
        ctx = (it, name, mode, sg, open_, ext, width, kind, n, lo, hi)
        if kind == "cigar":
            rec, cig = b.use_trace().build().align_batch_cigar(qs, rs)
        elif kind == "stats":
            rec, st = b.use_stats().build().align_batch(qs, rs)
        else:
            rec = b.build().align_batch(qs, rs)
        for k in range(n):
            w = orc.align(mode, qs[k], rs[k], open_, ext, om, sg_flags=sg if sg is not None else orc.SG_ALL,
                          stats=kind == "stats", trace=kind == "cigar")
            assert (rec["score"][k], rec["end_query"][k], rec["end_ref"][k]) == (w.score, w.end_query, w.end_ref), (ctx, k, qs[k], rs[k])
            assert rec["flags"][k] == 0, ctx
            if kind == "stats":
                assert (st["matches"][k], st["similar"][k], st["length"][k]) == (w.matches, w.similar, w.length), (ctx, k)
            if kind == "cigar":
                assert cig[k] == orc.cigar(w), (ctx, k, qs[k], rs[k])


# ------------------------------------------------------------------ full-size properties ----
def test_headline_config_properties(pkg, orc):
    """BASELINE config 2 at full size (1M x 150 x 150): sampled oracle parity plus
    size-independent properties (self alignment, order invariance, checksum stability)."""
    import bench
    n = 1_000_000
    qbuf, qoff, rbuf, roff = bench.make_cfg2_inputs(n)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    al = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).build()
    got = al.align_batch_packed(qbuf, qoff, rbuf, roff)
    assert (got["flags"] == 0).all() and got["score"].min() >= 0 and got["score"].max() <= 300
    assert (got["end_query"] < 150).all() and (got["end_ref"] < 150).all()
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(n, size=3000, replace=False))
    sq = [qbuf[qoff[i]:qoff[i + 1]].tobytes() for i in idx]
    sr = [rbuf[roff[i]:roff[i + 1]].tobytes() for i in idx]
    b1, o1 = orc.pack(sq); b2, o2 = orc.pack(sr)
    want = orc.align_batch(orc.SW, b1, o1, b2, o2, 5, 2, om)
    assert (got["score"][idx] == want[:, 0]).all()
    assert (got["end_query"][idx] == want[:, 1]).all() and (got["end_ref"][idx] == want[:, 2]).all()
    # self alignment: score 2*L, ends L-1
    selfa = al.align_batch_packed(qbuf[:150 * 4096], qoff[:4097], qbuf[:150 * 4096], qoff[:4097])
    assert (selfa["score"] == 300).all() and (selfa["end_query"] == 149).all() and (selfa["end_ref"] == 149).all()
    # order invariance: reversing the batch reverses the records
    m = 8191
    rev_q = qbuf[:150 * m].reshape(m, 150)[::-1].copy().reshape(-1)
    rev_r = rbuf[:150 * m].reshape(m, 150)[::-1].copy().reshape(-1)
    rev = al.align_batch_packed(rev_q, qoff[:m + 1], rev_r, roff[:m + 1])
    assert (rev[::-1] == got[:m]).all()
    # deterministic
    again = al.align_batch_packed(qbuf, qoff, rbuf, roff)
    assert (again == got).all()


def test_host_entry_pipelined_slices_with_mixed_lengths(pkg, orc):
    """pmx_align_batch sends large batches up in slices that overlap the kernels; every pair of a 300k batch with
    mixed (not sorted-worthy) lengths against the striped CPU port, and against the one-shot device path"""
    import torch
    rng = np.random.default_rng(6100)
    n = 300_000
    ql = rng.integers(125, 151, size=n); rl = rng.integers(125, 151, size=n)
    qoff = np.zeros(n + 1, dtype=np.int64); np.cumsum(ql, out=qoff[1:])
    roff = np.zeros(n + 1, dtype=np.int64); np.cumsum(rl, out=roff[1:])
    qbuf = DNA[rng.integers(0, 4, size=int(qoff[-1]))]
    rbuf = DNA[rng.integers(0, 4, size=int(roff[-1]))]
    for k in range(0, n, 1000):                       # some wildcards: the perm-table kernel's retry path inside slices
        qbuf[qoff[k] + 3] = ord("N")
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    al = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).build()
    got = al.align_batch_packed(qbuf, qoff, rbuf, roff)
    want, _ = orc.cpu_sw_striped16_batch(qbuf, qoff, rbuf, roff, 5, 2, om)
    assert (got["score"] == want[:, 0]).all() and (got["end_query"] == want[:, 1]).all() and (got["end_ref"] == want[:, 2]).all()
    assert (got["flags"] == 0).all()
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(x).to(dev) for x in (qbuf, qoff, rbuf, roff)]
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    pkg.align_batch_device(al._config(), n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 150, 150,
                           out.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert (o[:, 0] == got["score"]).all() and (o[:, 1] == got["end_query"]).all() and (o[:, 2] == got["end_ref"]).all()


@pytest.mark.parametrize("qlen,rlen,n", [(150, 150, 6000), (250, 250, 4200), (480, 700, 4100), (1000, 2000, 4100)])
def test_sw16_end_position_under_ties(pkg, orc, qlen, rlen, n):
    """End cells when the best score occurs several times (the first in column-major order must win): references that
    carry the same query segment twice (equal scores in two columns), queries that carry a segment twice (equal scores
    in two rows of one column, in different lanes of the pair's group), both at once, and exact duplicates of whole
    reads -- the cases the kernel's shared score bound and its tie rule decide (pmx_sw16.hip, share_bound).  Every pair
    against the oracle, batch sizes above the perm-table kernel's threshold, one shape per lane-group width."""
    rng = np.random.default_rng(31000 + qlen)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs, rs = [], []
    for k in range(n):
        seg = random_seqs(rng, 1, 12, 40)[0]
        kind = k % 5
        q = bytearray(random_seqs(rng, 1, qlen, qlen)[0]); r = bytearray(random_seqs(rng, 1, rlen, rlen)[0])
        def put(buf, pos, s):
            buf[pos:pos + len(s)] = s
        if kind in (0, 2):                       # the segment twice in the reference
            a = int(rng.integers(0, rlen // 2 - 40)); b = int(rng.integers(rlen // 2, rlen - 40))
            put(r, a, seg); put(r, b, seg)
            put(q, int(rng.integers(0, qlen - 40)), seg)
        if kind in (1, 2):                       # the segment twice in the query, far apart (different lanes of the group)
            a = int(rng.integers(0, qlen // 2 - 40)); b = int(rng.integers(qlen // 2, qlen - 40))
            put(q, a, seg); put(q, b, seg)
            if kind == 1:
                put(r, int(rng.integers(0, rlen - 40)), seg)
        if kind == 3:                            # identical reads (the diagonal), and a shifted copy
            m = min(qlen, rlen)
            r[:m] = q[:m]
        if kind == 4 and k % 10 == 4:            # periodic sequences: ties everywhere
            unit = random_seqs(rng, 1, 3, 7)[0]
            q = bytearray((unit * (qlen // len(unit) + 1))[:qlen]); r = bytearray((unit * (rlen // len(unit) + 1))[:rlen])
        qs.append(bytes(q)); rs.append(bytes(r))
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    al = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2).solution_width(16).build()
    got = al.align_batch_packed(qb, qo, rb, ro)
    assert "permtable" in pkg.lib.pmx_last_kernel().decode()
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
    bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
    assert len(bad) == 0, (len(bad), bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.parametrize("seed", _seeds([201, 202]))
def test_fuzz_local_end_cells_any_scoring(pkg, orc, seed):
    """The local kernels' end-cell bookkeeping (strip saves under a wave-uniform branch, the shared score bound, the tie rule)
    under random scoring -- match / mismatch / gap models including ext = 0 and ext = open -- on batches built to tie: repeated
    segments in query and reference, duplicates, periodic reads; uniform lengths above the perm-table threshold and ragged
    lengths (sorted order, LDS profiles); every pair against the oracle."""
    rng = np.random.default_rng(seed)
    for it in range(6):
        match = int(rng.choice([1, 2, 3, 5, 10]))
        mism = -int(rng.choice([1, 2, 3, 4, 9]))
        open_ = int(rng.choice([1, 2, 5, 7, 12]))
        ext = int(rng.choice([0, 1, 2, open_]))
        ext = min(ext, open_)
        pm, om = pkg.Matrix.create(b"ACGT", match, mism), orc.Matrix.create("ACGT", match, mism)
        uniform = it % 2 == 0
        L = int(rng.choice([64, 100, 150, 200, 260]))
        n = 4300 if uniform else 2500
        qs, rs = [], []
        for k in range(n):
            ql = L if uniform else int(rng.integers(30, L + 1)); rl = L if uniform else int(rng.integers(30, L + 40))
            q = bytearray(random_seqs(rng, 1, ql, ql)[0]); r = bytearray(random_seqs(rng, 1, rl, rl)[0])
            seg = random_seqs(rng, 1, 8, 24)[0][: min(ql, rl) // 3]
            kind = int(rng.integers(0, 6))
            def put(buf, s_):
                pos = int(rng.integers(0, len(buf) - len(s_) + 1)); buf[pos:pos + len(s_)] = s_
            if kind == 0: put(q, seg); put(r, seg); put(r, seg)
            elif kind == 1: put(q, seg); put(q, seg); put(r, seg)
            elif kind == 2: put(q, seg); put(q, seg); put(r, seg); put(r, seg)
            elif kind == 3: m_ = min(ql, rl); r[:m_] = q[:m_]
            elif kind == 4:
                unit = random_seqs(rng, 1, 2, 6)[0]
                q = bytearray((unit * (ql // len(unit) + 1))[:ql]); r = bytearray((unit * (rl // len(unit) + 1))[:rl])
            qs.append(bytes(q)); rs.append(bytes(r))
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        al = pkg.Aligner.new().local().matrix(pm).gap_open(open_).gap_extend(ext).solution_width(16).build()
        got = al.align_batch_packed(qb, qo, rb, ro)
        want = orc.align_batch(orc.SW, qb, qo, rb, ro, open_, ext, om)
        bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
        assert len(bad) == 0, (seed, it, match, mism, open_, ext, uniform, L, pkg.lib.pmx_last_kernel().decode(), bad[:5], got[bad[:5]], want[bad[:5]])


# --------------------------------------------------- perm-table (top-aligned) form of the global / semi-global kernels ----
@pytest.mark.parametrize("mode,sg", [(0, None), (1, None), (1, 1 | 8), (1, 2 | 4), (1, 2), (1, 8)])
@pytest.mark.parametrize("L", [75, 100, 125, 150, 250])
def test_nwsg16_permtable_form_uniform_reads(pkg, orc, mode, sg, L):
    """Equal-length DNA reads (the common case of a read set): global / semi-global scores without an LDS profile -- the v_perm
    looks the score up, rows top-aligned -- every free-end variant against the oracle; blocks that hold a query with a letter
    beyond the first four, or queries of different lengths, fall to the LDS-profile form inside the same call."""
    rng = np.random.default_rng(4400 + L + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    n = 4300
    qs = random_seqs(rng, n, L, L)
    rs = [mutate(rng, q, 0.1, 0.03) if k % 3 else random_seqs(rng, 1, L - 20, L + 30)[0] for k, q in enumerate(qs)]
    qs[7] = qs[7][:L // 2] + b"N" + qs[7][L // 2 + 1:]            # a wildcard in one query: its block takes the LDS-profile form
    qs[100] = qs[100][:L - 9]                                    # a shorter query: likewise
    qs[3000] = qs[3000].lower()
    b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(16)
    [b.global_, b.semi_global][mode]()
    if sg is not None:
        b.allow_query_gaps([n_ for f, n_ in ((1, "prefix"), (2, "suffix")) if sg & f]).allow_ref_gaps([n_ for f, n_ in ((4, "prefix"), (8, "suffix")) if sg & f])
    got = b.build().align_batch(qs, rs)
    kernel = pkg.lib.pmx_last_kernel().decode()
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(mode, qb, qo, rb, ro, 5, 2, om, sg_flags=sg if sg is not None else orc.SG_ALL)
    bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
    assert len(bad) == 0, (mode, sg, L, kernel, bad[:8], got[bad[:4]], want[bad[:4]])
    assert (got["flags"] == 0).all()
    if L <= 152:
        assert "permtable" in kernel, kernel


def test_nwsg16_permtable_form_with_traceback_and_statistics(pkg, orc):
    """the traceback sweeps' perm-table form: CIGAR text and statistics of equal-length reads (BASELINE config 4's shape), with
    blocks that fall to the LDS-profile form mixed in; the walk reads every block's row alignment from the sweep's flags"""
    rng = np.random.default_rng(4500)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    n = 3000
    for L in (250, 120):
        qs = random_seqs(rng, n, L, L)
        rs = [mutate(rng, q, 0.1, 0.03)[:L + 10] for q in qs]
        qs[5] = qs[5][:40] + b"N" + qs[5][41:]
        qs[1000] = qs[1000][:L - 31]
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        idx = np.unique(np.concatenate([np.arange(0, n, 13), [5, 6, 7, 1000, 1001, 2999]]))
        for mode in (1, 0):
            b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(16)
            [b.global_, b.semi_global][mode]()
            rec, text, coff = b.use_trace().build().align_batch_cigar_packed(qb, qo, rb, ro)
            want_text, want = orc.cigar_sample(mode, idx, qb, qo, rb, ro, 5, 2, om)
            raw = text.tobytes()
            for t, k in enumerate(idx):
                assert (rec["score"][k], rec["end_query"][k], rec["end_ref"][k]) == tuple(want[t, :3]), (L, mode, k)
                assert raw[coff[k]:coff[k + 1]].decode() == want_text[t], (L, mode, k)
            b2 = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(16)
            [b2.global_, b2.semi_global][mode]()
            r2, st = b2.use_stats().build().align_batch_packed(qb, qo, rb, ro)
            ws = orc.align_stats_sample(mode, idx, qb, qo, rb, ro, 5, 2, om)
            got = np.stack([r2["score"][idx], r2["end_query"][idx], r2["end_ref"][idx], st["matches"][idx], st["similar"][idx], st["length"][idx]], axis=1)
            assert (got == ws[:, :6]).all(), (L, mode)


@pytest.mark.parametrize("seed", _seeds([401, 402]))
def test_fuzz_permtable_form_blocks_and_fallback(pkg, orc, seed):
    """random equal-length DNA batches with blocks the perm-table form has to leave (ragged lengths, wildcards, lower case), random
    scoring inside and outside its window, every mode / free-end set / width incl. 8: both forms inside one call against the oracle"""
    rng = np.random.default_rng(seed)
    for it in range(10):
        L = int(rng.choice([20, 49, 50, 64, 75, 100, 127, 128, 150, 152, 160, 200, 250, 256]))
        n = int(rng.integers(2100, 4400))
        match, mismatch = [(2, -3), (1, -1), (3, -2), (5, -4), (1, -3)][int(rng.integers(0, 5))]
        open_, ext = [(5, 2), (3, 1), (11, 1), (4, 4), (6, 0)][int(rng.integers(0, 5))]
        pm, om = pkg.Matrix.create(b"ACGT", match, mismatch), orc.Matrix.create("ACGT", match, mismatch)
        qs = random_seqs(rng, n, L, L)
        rs = [mutate(rng, q, 0.1, 0.04) if k % 2 else random_seqs(rng, 1, max(1, L - 30), L + 30)[0] for k, q in enumerate(qs)]
        for k in rng.integers(0, n, size=12):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                p = int(rng.integers(0, L)); qs[k] = qs[k][:p] + b"N" + qs[k][p + 1:]
            elif kind == 1:
                qs[k] = qs[k][:int(rng.integers(1, L + 1))]
            else:
                qs[k] = qs[k].lower(); rs[k] = rs[k].lower()
        mode = int(rng.integers(0, 2))
        sg = int(rng.integers(1, 16)) if mode == 1 else 15
        width = [16, None, 32, 8][int(rng.integers(0, 4))]
        b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
        if width:
            b.solution_width(width)
        [b.global_, b.semi_global][mode]()
        if mode == 1:
            b.allow_query_gaps([n_ for f, n_ in ((1, "prefix"), (2, "suffix")) if sg & f]).allow_ref_gaps([n_ for f, n_ in ((4, "prefix"), (8, "suffix")) if sg & f])
        got = b.build().align_batch(qs, rs)
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        want = orc.align_stats_sample(mode, np.arange(n), qb, qo, rb, ro, open_, ext, om, sg_flags=sg, bits=width or 0)
        ok = want[:, 6] == 0
        assert ((got["flags"] & 1) == want[:, 6]).all(), (seed, it, mode, sg, width, L)
        bad = np.nonzero(ok & ((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2])))[0]
        assert len(bad) == 0, (seed, it, mode, sg, width, L, open_, ext, match, mismatch, pkg.lib.pmx_last_kernel(), bad[:5], got[bad[:3]], want[bad[:3]])


@pytest.mark.parametrize("gaps", [(5, 2), (3, 3), (20, 1), (11, 11)])
def test_nwsg16_row_offset_very_negative_scores_and_every_free_end(pkg, orc, gaps):
    """The row-offset form (every row stored + row * extend: no F - extend per row) takes the decline along the gaps out of what is
    STORED, so the bias no longer covers it -- but the free-end captures compare values with skew and offset taken off again, and
    the boundary values of long queries / references are very negative: long queries against short references and the other way
    round, every mode and all 15 free-end sets, per-pair DNA (perm-table and LDS-profile forms), per-pair protein (matrix lookup:
    2 100 pairs) and one shared protein query; gap models up to extend = open."""
    open_, ext = gaps
    rng = np.random.default_rng(3900 + open_ * 7 + ext)
    dpm, dom = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    ppm, pom = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    # DNA: equal-length long queries vs short references (perm-table blocks), and ragged both ways (LDS-profile form)
    qs = random_seqs(rng, 64, 900, 900) + random_seqs(rng, 40, 5, 40) + random_seqs(rng, 40, 400, 1000)
    rs = random_seqs(rng, 64, 3, 60) + random_seqs(rng, 40, 800, 2500) + random_seqs(rng, 40, 1, 30)
    for mode, sg in [(0, None)] + [(1, f) for f in range(1, 16)]:           # (no free end at all is `nw`: the builder cannot say sg with none)
        _nwsg_case(pkg, orc, mode, sg, qs, rs, open_, ext, dpm, dom, expect_kernel=None)
    if open_ + 11 + ext <= 255 and open_ >= 4:
        pq = random_seqs(rng, 2100, 200, 250, AA)
        pr = [random_seqs(rng, 1, 2, 30, AA)[0] if k % 2 else random_seqs(rng, 1, 200, 400, AA)[0] for k in range(2100)]
        for mode, sg in ((0, None), (1, None), (1, orc.S1_END), (1, orc.S2_BEG | orc.S2_END), (1, orc.S1_BEG | orc.S1_END)):
            _nwsg_case(pkg, orc, mode, sg, pq, pr, open_, ext, ppm, pom, expect_kernel=None)
            assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16m_kernel")
        q = random_seqs(rng, 1, 700, 700, AA)[0]
        rs2 = random_seqs(rng, 60, 2, 40, AA) + random_seqs(rng, 60, 600, 1500, AA)
        qb, qo = orc.pack([q] * len(rs2)); rb, ro = orc.pack(rs2)
        for mode, sg in [(0, None)] + [(1, f) for f in (1, 2, 4, 8, 3, 12, 15)]:
            b = pkg.Aligner.new().profile(pkg.Profile.new(q, False, ppm)).matrix(ppm).gap_open(open_).gap_extend(ext).solution_width(16)
            [b.global_, b.semi_global][mode]()
            if mode == 1:
                b.allow_query_gaps([t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f])
                b.allow_ref_gaps([t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f])
            got = b.build().align_batch([], rs2)
            if ext <= 3:                                    # (extend = 11 over 700 + 1 500 symbols leaves the int16 window: general kernel)
                assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_nwsg16q_kernel")
            want = orc.align_batch(mode, qb, qo, rb, ro, open_, ext, pom, sg_flags=sg if sg is not None else orc.SG_ALL, bits=16)
            bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]))[0]
            assert len(bad) == 0, (mode, sg, open_, ext, bad[:5], got[bad[:5]], want[bad[:5]])


def test_nwsg_width8_boundary_saturated_blocks_on_the_permtable_form(pkg, orc):
    """Width 8, equal-length reads (the perm-table form's blocks): blocks whose pairs all saturate by a penalised boundary alone
    (-(open + (len - 1) extend) < -128) run untracked on the perm-table form, every other block -- short reads, a free boundary on the
    side that is long, a read with a wildcard -- is marked there and tracked by the LDS-profile form.  The flag equals the oracle's
    on every pair; unsaturated pairs carry the exact result."""
    rng = np.random.default_rng(2370)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs, rs = [], []
    for blk in range(300):                          # blocks of 16 pairs (the <8,R> shapes) with one read length each
        ql = [150, 150, 40, 61, 62, 63, 64, 100][blk % 8]
        for t in range(16):
            q = random_seqs(rng, 1, ql, ql)[0]
            if blk % 37 == 5 and t == 3:
                q = q[:7] + b"N" + q[8:]
            qs.append(q)
            rs.append(mutate(rng, q, 0.1, 0.03) if t % 2 else random_seqs(rng, 1, 1, 170)[0])
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    idx = np.arange(len(qs))
    seen = set()
    for mode, sg, qg, dg in ((0, 15, None, None), (1, 2 | 8, ["suffix"], ["suffix"]), (1, 1 | 8, ["prefix"], ["suffix"]), (1, 4 | 2, ["suffix"], ["prefix"])):
        b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).solution_width(8)
        [b.global_, b.semi_global][mode]()
        if qg is not None:
            b.allow_query_gaps(qg).allow_ref_gaps(dg)
        got = b.build().align_batch(qs, rs)
        assert "permtable" in pkg.lib.pmx_last_kernel().decode(), pkg.lib.pmx_last_kernel()
        want = orc.align_stats_sample(mode, idx, qb, qo, rb, ro, 5, 2, om, sg_flags=sg, bits=8)
        bad = np.nonzero((got["flags"] & 1) != want[:, 6])[0]
        assert len(bad) == 0, (mode, sg, bad[:5], [(len(qs[k]), len(rs[k])) for k in bad[:5]], want[bad[:5]])
        ok = want[:, 6] == 0
        assert (got["score"][ok] == want[ok, 0]).all() and (got["end_query"][ok] == want[ok, 1]).all() and (got["end_ref"][ok] == want[ok, 2]).all()
        seen |= {bool(x) for x in ok}
    assert seen == {True, False}


@pytest.mark.parametrize("seed", _seeds([701, 702]))
def test_fuzz_traceback_and_statistics_global_and_semi_global(pkg, orc, seed, monkeypatch):
    """Random CIGAR and statistics batches over the traceback sweeps of the global / semi-global kernels (row offset, perm-table and
    LDS-profile forms, matrix lookup, shared profile): random gap models inside the one-instruction-merge window, every free-end set,
    lengths from 1 to 500 with long-against-short pairs both ways, DNA with wildcards, protein; CIGAR text, score, end cell and
    matches / similar / length against the oracle's byte trace on every pair."""
    rng = np.random.default_rng(seed)
    monkeypatch.setenv("PMX_STATS_BY_TRACE", "1")
    dpm, dom = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    ppm, pom = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    for it in range(6):
        protein = rng.random() < 0.35
        pm, om, alpha = (ppm, pom, AA) if protein else (dpm, dom, DNA)
        open_ = int(rng.choice([4, 5, 11, 20] if protein else [0, 1, 3, 5, 11, 40])); ext = int(rng.choice([0, 1, 2, open_])); ext = min(ext, open_)
        lo, hi = [(1, 30), (20, 160), (100, 260), (200, 500)][int(rng.integers(0, 4))]
        n = int(rng.choice([3, 40, 130]))
        qs = random_seqs(rng, n, lo, hi, alpha)
        rs = []
        for q in qs:
            u = rng.random()
            if u < 0.6: r = random_seqs(rng, 1, 0, 20, alpha)[0] + mutate(rng, q, 0.12, 0.06, alpha) + random_seqs(rng, 1, 0, 20, alpha)[0]
            elif u < 0.8: r = random_seqs(rng, 1, 1, 12, alpha)[0]                      # long query against a few symbols
            else: r = random_seqs(rng, 1, hi, hi + 300, alpha)[0]
            rs.append(r or b"A")
        if not protein:
            for i in range(0, n, 17):
                q = bytearray(qs[i]); q[int(rng.integers(len(q)))] = ord("N"); qs[i] = bytes(q)
        for mode, sg in ((0, None), (1, int(rng.integers(1, 16))), (1, None)):
            _cigar_case(pkg, orc, mode, sg, qs, rs, open_, ext, pm, om)
            _stats_case(pkg, orc, mode, sg, qs, rs, open_, ext, pm, om)
        if protein:
            _stats_case(pkg, orc, 0, None, None, rs, open_, ext, pm, om, shared_query=qs[0])
            _stats_case(pkg, orc, 1, int(rng.integers(1, 16)), None, rs, open_, ext, pm, om, shared_query=qs[0])


def test_deferred_results_behind_the_unchanged_abi(pkg, orc, monkeypatch):
    """PMX_DEFER_ALIGN (round-3 review, missing #4): the reference's calling pattern is one align() per pair
    (/root/reference/src/aligner/mod.rs:397-452, tests/test_parasail.rs:702-717).  With the switch on, a score / statistics call
    queues the pair and returns a pending result; the first accessor runs the queue as ONE batch.  Same values as the immediate
    path for every mode, mixed configurations (a change of configuration runs the queue), results freed before they are read,
    results read from another thread, the profile arm, and long pairs (which are not deferred)."""
    import threading
    rng = np.random.default_rng(7800)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 300, 20, 200)
    rs = [mutate(rng, q, 0.1, 0.04) for q in qs]
    builders = {}
    for mode in (0, 1, 2):
        b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2)
        [b.global_, b.semi_global, b.local][mode]()
        builders[mode] = b.build()
    bs = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).local().use_stats().build()
    want = {mode: [(r.get_score(), r.get_end_query(), r.get_end_ref()) for r in (builders[mode].align(q, s) for q, s in zip(qs, rs))] for mode in builders}
    want_st = [(r.get_score(), r.get_matches(), r.get_similar(), r.get_length()) for r in (bs.align(q, s) for q, s in zip(qs[:60], rs[:60]))]
    monkeypatch.setenv("PMX_DEFER_ALIGN", "1")
    for mode, al in builders.items():
        res = [al.align(q, s) for q, s in zip(qs, rs)]            # 300 calls, nothing has run yet
        assert res[0].is_local() == (mode == 2) and res[0].is_striped()          # predicates answer at once
        del res[5], res[17]                                       # freed while pending: withdrawn
        keep = [k for k in range(300) if k not in (5, 18)]
        got = [(r.get_score(), r.get_end_query(), r.get_end_ref()) for r in res]
        assert "pmx_" in pkg.lib.pmx_last_kernel().decode()
        assert got == [want[mode][k] for k in keep], mode
    # interleaved configurations: each change runs the queue so far
    mixed = []
    for k in range(60):
        mixed.append((k % 3, builders[k % 3].align(qs[k], rs[k])))
        if k % 7 == 0:
            mixed.append(("st", bs.align(qs[k], rs[k])))
    for (tag, r), k in zip(mixed, [x for k in range(60) for x in ([k, k] if k % 7 == 0 else [k])]):
        if tag == "st":
            assert (r.get_score(), r.get_matches(), r.get_similar(), r.get_length()) == want_st[k], k
        else:
            assert (r.get_score(), r.get_end_query(), r.get_end_ref()) == want[tag][k], (tag, k)
    # pending results read by another thread
    res = [builders[2].align(q, s) for q, s in zip(qs[:40], rs[:40])]
    box = []
    t = threading.Thread(target=lambda: box.append([r.get_score() for r in res]))
    t.start(); t.join()
    assert box[0] == [w[0] for w in want[2][:40]]
    # the profile arm, and an explicit flush
    prof = pkg.Profile.new(qs[0], False, pm)
    alp = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).local().profile(prof).build()
    res = [alp.align(None, s) for s in rs[:30]]
    pkg.lib.pmx_flush_deferred()
    qb, qo = orc.pack([qs[0]] * 30); rb, ro = orc.pack(rs[:30])
    w = orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)
    assert [r.get_score() for r in res] == list(w[:, 0])
    # a long pair is not deferred (it fills the chip on its own)
    lq = random_seqs(rng, 1, 3000, 3000)[0]; lr = mutate(rng, lq, 0.1, 0.03)
    one = builders[2].align(lq, lr)
    assert "long32" in pkg.lib.pmx_last_kernel().decode()
    qb, qo = orc.pack([lq]); rb, ro = orc.pack([lr])
    assert one.get_score() == orc.align_batch(orc.SW, qb, qo, rb, ro, 5, 2, om)[0, 0]


@pytest.mark.parametrize("mode,sg", [(0, 15), (1, 15), (1, 5), (1, 10), (1, 8), (1, 1)])
def test_global_kernels_at_the_edge_of_their_int16_window(pkg, orc, mode, sg):
    """(round-3 review, weak #3 / advice: the admissibility windows of pmx_nwsg16.hip are argued in comments and sampled by fuzz.)
    For several scoring schemes the LONGEST reference the packed global / semi-global kernels still accept is searched for by
    launching (pmx_last_kernel() tells which kernel ran), for a short query (where the launcher may pick a larger shape than the
    window proof's estimate) and for a query as long as the shapes allow; at that length and just below it the inputs that stretch
    the value range -- all matches, no match at all, the query at the far end behind a long gap, poly-A ties -- are compared with
    the oracle.  No promotion pass exists behind these kernels: a window that is one step too wide shows here as a wrong score."""
    rng = np.random.default_rng(7900 + mode * 16 + sg)
    n = 2100                                                        # above the thresholds of the perm-table / block-flag forms
    for match, mis, open_, ext in ((2, -3, 5, 2), (1, -1, 1, 1), (5, -4, 10, 1), (3, -2, 4, 4), (9, -9, 20, 3)):
        pm, om = pkg.Matrix.create(b"ACGT", match, mis), orc.Matrix.create("ACGT", match, mis)
        b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
        [b.global_, b.semi_global][mode]()
        if mode == 1:
            b.allow_query_gaps([t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f])
            b.allow_ref_gaps([t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f])
        al = b.build()
        for qlen in (50, 1000):
            q0 = random_seqs(rng, 1, qlen, qlen)[0]

            def batch(rlen):
                far = random_seqs(rng, 1, rlen, rlen)[0]
                rs = [(q0 * (rlen // qlen + 1))[:rlen],                       # matches all the way
                      bytes(b"ACGT"[(b"ACGT".index(bytes([c])) + 1) % 4] for c in (q0 * (rlen // qlen + 1))[:rlen]),     # no match on the diagonal
                      far[:rlen - qlen] + q0 if rlen > qlen else far,          # the query's copy behind a long gap
                      q0 + far[:rlen - qlen] if rlen > qlen else far,          # ... and in front of one
                      b"A" * rlen, far]
                qs = [q0, q0, q0, q0, b"A" * qlen, q0]
                k = len(rs)
                return qs * (n // k + 1), rs * (n // k + 1), k

            def fast(rlen):
                qs, rs, k = batch(rlen)
                al.align_batch(qs[:n], rs[:n])
                return "pmx_nwsg16" in pkg.lib.pmx_last_kernel().decode()
            lo, hi = 64, 30500
            if not fast(lo):
                continue                                                     # this scheme never takes the packed kernels
            while hi - lo > 1:                                               # the longest reference they still take
                mid = (lo + hi) // 2
                lo, hi = (mid, hi) if fast(mid) else (lo, mid)
            for rlen in {lo, lo - 1, max(64, lo - 37), max(64, lo // 2)}:
                qs, rs, k = batch(rlen)
                got = al.align_batch(qs[:n], rs[:n])
                assert "pmx_nwsg16" in pkg.lib.pmx_last_kernel().decode()
                qb, qo = orc.pack(qs[:k]); rb, ro = orc.pack(rs[:k])
                want = orc.align_batch(mode, qb, qo, rb, ro, open_, ext, om, sg_flags=sg)
                for t in range(k):
                    rows = slice(t, n, k)
                    g = np.stack([got["score"][rows], got["end_query"][rows], got["end_ref"][rows]], axis=1)
                    assert (g == want[t]).all() and (got["flags"][rows] == 0).all(), \
                        (mode, sg, match, mis, open_, ext, qlen, rlen, t, g[0], want[t], pkg.lib.pmx_last_kernel().decode())


@pytest.mark.parametrize("match", [16, 40, 100, 200])
def test_local_kernels_around_their_rerun_limit(pkg, orc, match):
    """(round-3 review, weak #3: the local kernels' exact ranges end at internal limits -- biased max3 lanes, offset int16 -- behind
    which a pair is re-run in 32 bits.)  All-match pairs whose scores climb through 24 000 ... 36 000 in steps of about one match
    straddle every such limit and the int16 boundary itself: `sat` must return the oracle's score and ends for all of them, the
    fixed width 16 must flag exactly the pairs above 32 767 -- as a small call (one wave per pair) and inside a batch large enough
    for the kernels that hand pairs back through the retry list."""
    rng = np.random.default_rng(1700 + match)
    pm, om = pkg.Matrix.create(b"ACGT", match, -match), orc.Matrix.create("ACGT", match, -match)
    lens = sorted({min(2040, max(1, s // match)) for s in range(24000, 36001, max(match, 150))})
    qs = [random_seqs(rng, 1, L, L)[0] for L in lens]
    rs = [q if k % 3 else b"G" + q + b"T" for k, q in enumerate(qs)]            # (some embedded: end positions off the corner)
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    want = orc.align_batch(orc.SW, qb, qo, rb, ro, 7, 3, om)
    assert want[:, 0].max() > 32767 or match * 2040 <= 32767
    filler_q = random_seqs(rng, 4200, 30, 60); filler_r = [mutate(rng, q, 0.1, 0.05) for q in filler_q]
    sat = pkg.Aligner.new().local().matrix(pm).gap_open(7).gap_extend(3).build()
    w16 = pkg.Aligner.new().local().matrix(pm).gap_open(7).gap_extend(3).solution_width(16).build()
    for Q, Rr in ((qs, rs), (qs + filler_q, rs + filler_r)):
        got = sat.align_batch(Q, Rr)
        k = len(qs)
        assert (got["score"][:k] == want[:, 0]).all() and (got["end_query"][:k] == want[:, 1]).all() and (got["end_ref"][:k] == want[:, 2]).all(), \
            (match, np.nonzero(got["score"][:k] != want[:, 0])[0][:5], pkg.lib.pmx_last_kernel())
        assert (got["flags"][:k] == 0).all()
        g16 = w16.align_batch(Q, Rr)
        over = want[:, 0] > 32767
        assert ((g16["flags"][:k] & pkg.FLAG_SATURATED) != 0).tolist() == over.tolist(), (match, lens, g16["flags"][:k], want[:, 0])
        ok = ~over
        assert (g16["score"][:k][ok] == want[ok, 0]).all() and (g16["end_query"][:k][ok] == want[ok, 1]).all() and (g16["end_ref"][:k][ok] == want[ok, 2]).all()
