"""One long pair across the chip (`-m gpu`): pmx_long32_kernel (parasail-rs_amd/csrc/pmx_long.hip).

`Aligner::align()` has no length limit (/root/reference/src/aligner/mod.rs:397-430; :454-456 "for aligning large sequences").
A call (or a handful of pairs) of >= 512 query rows and >= 3 M cells spreads the query's 256-row bands over the CUs, each band one
wave, chained through 8-byte granules in HBM.  Everything here is compared with the scalar oracle: every mode and free-end
variant, lengths around the 64-column chunks and the 256-row bands, tie-heavy inputs for the end-position rules, ragged
batches, the profile arm, fixed-width saturation, and 20 kbp x 20 kbp."""
import os

import numpy as np
import pytest

from util import random_seqs, mutate, DNA, AA

pytestmark = pytest.mark.gpu

LONG = "pmx_long32_kernel"


def _rec(a):
    return np.stack([a["score"], a["end_query"], a["end_ref"]], axis=1)


def _builder(pkg, pm, o, e, mode, width=None):
    b = pkg.Aligner.new().matrix(pm).gap_open(o).gap_extend(e)
    [b.global_, b.semi_global, b.local][mode]()
    if width is not None:
        b.solution_width(width)
    return b


FORMS = {"dispatch": {}, "two columns": {"PMX_LONG_TWO_COLUMNS": "1"}, "two columns, 128-row bands": {"PMX_LONG_TWO_COLUMNS": "1", "PMX_LONG_ROWS_PER_LANE": "2"},
         "two columns, 64-step chunks": {"PMX_LONG_TWO_COLUMNS": "1", "PMX_LONG_CHUNK_COLS": "64"}}


@pytest.fixture(params=list(FORMS))
def form(request, monkeypatch):
    """the kernel form: what the dispatcher's time model picks (one column per step for batches and square pairs), or forced"""
    for k, v in FORMS[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (1, 1), (7, 0)])
def test_ragged_batches_around_chunk_and_band_edges(pkg, orc, mode, gaps, form):
    rng = np.random.default_rng(9900 + mode * 10 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qlens = [3000, 512, 513, 255, 256, 257, 1, 700, 1024, 1025, 2049, 100, 1279, 1280, 1281, 40]
    rlens = [3100, 63, 64, 65, 127, 128, 129, 191, 192, 193, 1, 2, 6000, 640, 1000, 5000]
    qs = [random_seqs(rng, 1, L, L)[0] for L in qlens]
    rs = []
    for k, (q, L) in enumerate(zip(qs, rlens)):
        base = mutate(rng, q, 0.08, 0.04) if k % 2 == 0 else random_seqs(rng, 1, L, L)[0]
        r = (base + random_seqs(rng, 1, L, L)[0])[:L]
        rs.append(r)
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for sg in ((15,) if mode != 1 else (15, 1, 2, 4, 8, 1 | 8, 2 | 4, 3, 12, 5, 10)):
        b = _builder(pkg, pm, gaps[0], gaps[1], mode)
        if mode == 1:
            b.allow_query_gaps([n for f, n in ((1, "prefix"), (2, "suffix")) if sg & f]).allow_ref_gaps([n for f, n in ((4, "prefix"), (8, "suffix")) if sg & f])
        got = _rec(b.build().align_batch(qs, rs))
        assert LONG in pkg.lib.pmx_last_kernel().decode(), pkg.lib.pmx_last_kernel()
        want = orc.align_batch(mode, qb, qo, rb, ro, gaps[0], gaps[1], om, sg_flags=sg)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (mode, sg, gaps, bad, [(qlens[k], rlens[k]) for k in bad[:4]], got[bad[:4]], want[bad[:4]])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_end_positions_under_ties(pkg, orc, mode, form):
    """periodic sequences: the maximum occurs in many cells, several of them in different bands and lanes"""
    rng = np.random.default_rng(9950 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 1, -1), orc.Matrix.create("ACGT", 1, -1)
    unit = random_seqs(rng, 1, 37, 37)[0]
    qs = [(unit * 100)[:L] for L in (2000, 1500, 3000, 777)]
    rs = [(unit * 100)[:L] if k % 2 == 0 else (unit[5:] + unit * 100)[:L] for k, L in enumerate((1800, 2100, 2999, 3500))]
    qs += [b"A" * 1200, b"ACGT" * 400]; rs += [b"A" * 2600, b"ACGT" * 700]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for o, e in ((0, 0), (1, 1), (3, 1)):
        got = _rec(_builder(pkg, pm, o, e, mode).build().align_batch(qs, rs))
        assert LONG in pkg.lib.pmx_last_kernel().decode()
        want = orc.align_batch(mode, qb, qo, rb, ro, o, e, om)
        assert (got == want).all(), (mode, o, e, got, want)


def test_protein_and_profile_arm(pkg, orc):
    rng = np.random.default_rng(9960)
    pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
    q = random_seqs(rng, 1, 1500, 1500, AA)[0]
    rs = [mutate(rng, q, 0.3, 0.05, AA), random_seqs(rng, 1, 2500, 2500, AA)[0], (random_seqs(rng, 1, 900, 900, AA)[0] + mutate(rng, q[300:900], 0.2, 0.04, AA))]
    rb, ro = orc.pack(rs)
    qb = np.tile(np.frombuffer(q, dtype=np.uint8), len(rs)); qo = np.arange(len(rs) + 1, dtype=np.int64) * len(q)
    for mode in (0, 1, 2):
        want = orc.align_batch(mode, qb, qo, rb, ro, 11, 1, om)
        got = _rec(_builder(pkg, pm, 11, 1, mode).build().align_batch([q] * len(rs), rs))
        assert LONG in pkg.lib.pmx_last_kernel().decode()
        assert (got == want).all(), (mode, got, want)
        prof = pkg.Profile.new(q, False, pm)
        gotp = _rec(_builder(pkg, pm, 11, 1, mode).profile(prof).build().align_batch([], rs))      # one shared query (profile arm)
        assert LONG in pkg.lib.pmx_last_kernel().decode()
        assert (gotp == want).all(), (mode, gotp, want)
        one = _builder(pkg, pm, 11, 1, mode).profile(prof).build().align(None, rs[1])                # Aligner::align(None, reference)
        assert (one.get_score(), one.get_end_query(), one.get_end_ref()) == tuple(want[1])


def test_fixed_widths_and_saturation(pkg, orc):
    rng = np.random.default_rng(9970)
    pm, om = pkg.Matrix.create(b"ACGT", 40, -40), orc.Matrix.create("ACGT", 40, -40)
    q = random_seqs(rng, 1, 2000, 2000)[0]; r = mutate(rng, q, 0.02, 0.01)
    for mode in (0, 1, 2):
        for width in (None, 8, 16, 32, 64):
            res = _builder(pkg, pm, 50, 5, mode, width).build().align(q, r)
            w = orc.align(mode, q, r, 50, 5, om, bits=width or 0)
            if width in (None, 32, 64) or not w.saturated:
                assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (w.score, w.end_query, w.end_ref), (mode, width)
            assert bool(res.is_saturated()) == bool(w.saturated), (mode, width)
            if width in (None, 32, 64):
                assert w.score > 32767 and LONG in pkg.lib.pmx_last_kernel().decode()
    # global, width 16, boundaries inside the range (short gaps): the general kernel tracks the range; results agree anyway
    pm2, om2 = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = random_seqs(rng, 1, 2200, 2200)[0]; r = mutate(rng, q, 0.05, 0.02)
    res = _builder(pkg, pm2, 5, 2, 0, 16).build().align(q, r)
    w = orc.align(0, q, r, 5, 2, om2, bits=16)
    assert (res.get_score(), bool(res.is_saturated())) == (w.score, bool(w.saturated))


@pytest.mark.parametrize("mode", [0, 2])
def test_20_kbp_pair(pkg, orc, mode):
    rng = np.random.default_rng(9980 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = random_seqs(rng, 1, 20000, 20000)[0]; r = mutate(rng, q, 0.08, 0.03)
    res = _builder(pkg, pm, 5, 2, mode).build().align(q, r)
    assert LONG in pkg.lib.pmx_last_kernel().decode()
    w = orc.align(mode, q, r, 5, 2, om)
    assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (w.score, w.end_query, w.end_ref)
    assert abs(w.score) > 20000


def test_short_query_against_a_very_long_reference_and_the_switch(pkg, orc, monkeypatch):
    rng = np.random.default_rng(9990)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = random_seqs(rng, 1, 600, 600)[0]
    r = random_seqs(rng, 1, 150000, 150000)[0][:70000] + mutate(rng, q, 0.05, 0.02) + random_seqs(rng, 1, 80000, 80000)[0]
    for mode in (1, 2):
        res = _builder(pkg, pm, 5, 2, mode).build().align(q, r)
        assert LONG in pkg.lib.pmx_last_kernel().decode()
        w = orc.align(mode, q, r, 5, 2, om)
        assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (w.score, w.end_query, w.end_ref)
        monkeypatch.setenv("PMX_NO_LONG_KERNEL", "1")
        res2 = _builder(pkg, pm, 5, 2, mode).build().align(q, r)
        assert LONG not in pkg.lib.pmx_last_kernel().decode()
        monkeypatch.delenv("PMX_NO_LONG_KERNEL")
        assert (res2.get_score(), res2.get_end_query(), res2.get_end_ref()) == (w.score, w.end_query, w.end_ref)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_batches_outside_every_packed_window_take_the_band_kernel(pkg, orc, mode, monkeypatch):
    """Score-only batches no packed int16 kernel serves -- a gap model with open < extend, an alphabet of 40 letters, queries beyond
    2 048 rows in numbers -- run in pmx_long32_kernel (one wave per 256 query rows, any number of pairs, chunks of bounded
    scratch) instead of the general kernel; every pair against the oracle."""
    rng = np.random.default_rng(9995 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 3000, 1, 300)
    rs = [mutate(rng, q, 0.1, 0.04) if k % 2 else random_seqs(rng, 1, 1, 350)[0] for k, q in enumerate(qs)]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    got = _rec(_builder(pkg, pm, 2, 5, mode, 32).build().align_batch(qs, rs))                       # open < extend
    assert (LONG in pkg.lib.pmx_last_kernel().decode()) == (mode != 2), pkg.lib.pmx_last_kernel()   # (the local packed kernel's max3 variant needs no open >= extend)
    assert (got == orc.align_batch(mode, qb, qo, rb, ro, 2, 5, om)).all()
    # 40 letters
    letters = bytes(range(65, 91)) + b"0123456789@+=?"                    # (the mapper folds case: 40 distinct symbols without lower case)
    rows = np.random.default_rng(5).integers(-4, 6, size=(41, 41))
    rows = np.minimum(rows, rows.T)
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        names = [chr(c) for c in letters] + ["*"]
        fh.write("   " + "  ".join(names) + "\n")
        for i, nm in enumerate(names):
            fh.write(nm + " " + " ".join("%3d" % v for v in rows[i]) + "\n")
        path = fh.name
    try:
        pm40 = pkg.Matrix.from_file(path)
        al40 = np.frombuffer(letters, dtype=np.uint8)
        q40 = random_seqs(rng, 500, 5, 200, al40); r40 = random_seqs(rng, 500, 5, 260, al40)
        got = _rec(_builder(pkg, pm40, 6, 1, mode).build().align_batch(q40, r40))
        assert LONG in pkg.lib.pmx_last_kernel().decode(), pkg.lib.pmx_last_kernel()
        om40 = orc.Matrix(rows.astype(np.int32), np.array([letters.find(bytes([c])) if bytes([c]) in letters else 40 for c in range(256)], dtype=np.int32))
        qb4, qo4 = orc.pack(q40); rb4, ro4 = orc.pack(r40)
        assert (got == orc.align_batch(mode, qb4, qo4, rb4, ro4, 6, 1, om40)).all()
    finally:
        os.remove(path)
    # 40 pairs of 2 100 - 2 600 rows, in chunks of a few pairs each
    lq = random_seqs(rng, 40, 2100, 2600); lr = [mutate(rng, q, 0.1, 0.03)[:2400] for q in lq]
    qb, qo = orc.pack(lq); rb, ro = orc.pack(lr)
    want = orc.align_batch(mode, qb, qo, rb, ro, 5, 2, om)
    for chunk in (None, "2.5e6"):
        if chunk:
            monkeypatch.setenv("PMX_LONG_CHUNK_BYTES", chunk)
        got = _rec(_builder(pkg, pm, 5, 2, mode).build().align_batch(lq, lr))
        assert LONG in pkg.lib.pmx_last_kernel().decode()
        assert (got == want).all(), chunk


def _seeds(default):
    """PMX_FUZZ_SEEDS=a:b (or a comma list) widens the seed list for soak runs"""
    spec = os.environ.get("PMX_FUZZ_SEEDS")
    if not spec:
        return default
    if ":" in spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return [int(x) for x in spec.split(",")]


@pytest.mark.parametrize("seed", _seeds([301, 302, 303]))
def test_fuzz_band_kernel_shapes_modes_and_scoring(pkg, orc, seed, form):
    """random modes, free-end sets, gap models (open < extend and extend = 0 included), matrices over 2-9 letters, lengths around the
    256-row bands and the 64-column chunks, 1-16 pairs per call: pmx_long32_kernel against the oracle"""
    rng = np.random.default_rng(seed)
    for it in range(14):
        na = int(rng.integers(2, 10))
        letters = bytes(b"ACGTRYKMS"[:na])
        match, mismatch = int(rng.integers(1, 9)), -int(rng.integers(0, 9))
        pm, om = pkg.Matrix.create(letters, match, mismatch), orc.Matrix.create(letters.decode(), match, mismatch)
        al = np.frombuffer(letters, dtype=np.uint8)
        open_, ext = int(rng.integers(0, 12)), int(rng.integers(0, 7))
        n = int(rng.integers(1, 17))
        big_q = int(rng.choice([512, 513, 767, 768, 1024, 1025, 1500, 2049, 2600]))
        big_r = max(int(3.2e6 / big_q) + int(rng.integers(1, 400)), 64)
        qs = [random_seqs(rng, 1, big_q, big_q, al)[0]] + random_seqs(rng, n - 1, 1, big_q, al)
        rs = []
        for k, q in enumerate(qs):
            L = big_r if k == 0 else int(rng.choice([1, 63, 64, 65, 127, 128, 129, 200, 1000, big_r]))
            base = mutate(rng, q, 0.1, 0.05, al) if rng.random() < 0.6 else random_seqs(rng, 1, L, L, al)[0]
            rs.append((base + random_seqs(rng, 1, L, L, al)[0])[:L])
        mode = int(rng.integers(0, 3))
        sg = int(rng.integers(1, 16)) if mode == 1 else 15
        b = _builder(pkg, pm, open_, ext, mode)
        if mode == 1:
            b.allow_query_gaps([n_ for f, n_ in ((1, "prefix"), (2, "suffix")) if sg & f]).allow_ref_gaps([n_ for f, n_ in ((4, "prefix"), (8, "suffix")) if sg & f])
            if sg in (3, 12, 15) or (sg & 3) == 0 or (sg & 12) == 0:
                pass
        got = _rec(b.build().align_batch(qs, rs))
        # (a free-end set the builder cannot name -- no gaps allowed on one sequence -- resolves to that sequence's default)
        eff = sg if mode == 1 else 15
        if mode == 1:
            qpart, dpart = sg & 3, sg & 12
            eff = (qpart | dpart) if (qpart or dpart) else 15
            if qpart == 3 and dpart == 12:
                eff = 15
        assert LONG in pkg.lib.pmx_last_kernel().decode(), (pkg.lib.pmx_last_kernel(), mode, open_, ext)
        qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
        want = orc.align_batch(mode, qb, qo, rb, ro, open_, ext, om, sg_flags=eff)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (seed, it, mode, eff, open_, ext, match, mismatch, na, bad[:4], got[bad[:4]], want[bad[:4]],
                               [(len(qs[k]), len(rs[k])) for k in bad[:4]])


def test_bounded_wait_gives_up_and_the_call_is_redone(pkg, orc, monkeypatch):
    """(round-3 review) a band's wait for the band above is bounded: with the limit forced to 0 every wait gives up at once, the
    launch's abort word makes every other band leave, and the call is redone on the per-pair kernels -- same records, the give-up
    reported through pmx_last_error()"""
    rng = np.random.default_rng(9990)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = [random_seqs(rng, 1, L, L)[0] for L in (3000, 2500, 2100)]
    rs = [mutate(rng, q, 0.1, 0.05) for q in qs]
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    for mode in (0, 1, 2):
        al = _builder(pkg, pm, 5, 2, mode).build()
        want = orc.align_batch(mode, qb, qo, rb, ro, 5, 2, om)
        got = _rec(al.align_batch(qs, rs))
        assert LONG in pkg.lib.pmx_last_kernel().decode()
        assert (got == want).all()
        with monkeypatch.context() as mp:
            mp.setenv("PMX_LONG_SPIN_LIMIT", "0")
            got0 = al.align_batch(qs, rs)
            assert LONG not in pkg.lib.pmx_last_kernel().decode(), pkg.lib.pmx_last_kernel()
            assert b"bounded wait" in pkg.lib.pmx_last_error()
            assert (_rec(got0) == want).all() and (got0["flags"] == 0).all()
        one = al.align(qs[0], rs[0])                                       # and the normal road is back
        assert LONG in pkg.lib.pmx_last_kernel().decode() and one.get_score() == want[0, 0]


@pytest.mark.timeout(300)
def test_two_host_threads_run_long_grids_side_by_side(pkg, orc):
    """(round-3 review) scratch and streams are per host thread, so two threads can have two long-pair grids on the chip at once,
    each with spinning consumers: 2 x 600 pairs of 5 kbp oversubscribe the chip several times.  Both drain (the lowest unfinished
    band of either grid is always resident) and every record equals the oracle's."""
    import threading
    rng = np.random.default_rng(9991)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    batches = []
    for t in range(2):
        qs = random_seqs(rng, 600, 4800, 5200)
        rs = [mutate(rng, q, 0.08, 0.03) if k % 3 else random_seqs(rng, 1, 4000, 5200)[0] for k, q in enumerate(qs)]
        batches.append((qs, rs))
    al = _builder(pkg, pm, 5, 2, 2).build()
    out, names, errs = [None, None], [None, None], []

    def work(t):
        try:
            for _ in range(2):
                out[t] = al.align_batch(*batches[t])
                names[t] = pkg.lib.pmx_last_kernel().decode()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(2):
        assert LONG in names[t], names
        qs, rs = batches[t]
        idx = np.arange(0, 600, 7)
        qb, qo = orc.pack([qs[k] for k in idx]); rb, ro = orc.pack([rs[k] for k in idx])
        want = orc.align_batch(2, qb, qo, rb, ro, 5, 2, om)
        assert (_rec(out[t])[idx] == want).all() and (out[t]["flags"] == 0).all()
