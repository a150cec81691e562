"""Band-strip kernel (`-m gpu`): parasail-rs_amd/csrc/pmx_bstrip.hip against the banded oracle.

Reference counterpart: Aligner::banded_nw -> parasail_nw_banded (/root/reference/src/aligner/mod.rs:454-489; KAT
tests/test_parasail.rs:726-736) and the batch extension with a per-pair band centre.  tests/test_gpu_banded.py runs its band tests
through this kernel as well (it is the default for alphabets of <= 4 letters); here: every free-end set, tie-heavy gap models,
every lane shape, the pairs handed back for a wildcard, the window's fallback, the processing order of large batches."""
import numpy as np
import pytest

from util import random_seqs, mutate

pytestmark = pytest.mark.gpu


def _batch(rng, n, lo, hi, far=0.15, wild=False):
    qs = random_seqs(rng, n, lo, hi)
    rs, diag = [], np.zeros(n, dtype=np.int32)
    for t, q in enumerate(qs):
        body = mutate(rng, q, 0.1, 0.05) if rng.random() < 0.8 else random_seqs(rng, 1, lo, hi)[0]
        pre = random_seqs(rng, 1, 0, 50)[0] if rng.random() < 0.5 else b""
        post = random_seqs(rng, 1, 0, 50)[0] if rng.random() < 0.3 else b""
        rs.append((pre + body + post) or b"A")
        diag[t] = len(pre) + int(rng.integers(-6, 7)) if rng.random() > far else int(rng.integers(-hi - 5, hi + 5))
    if wild:
        for t in range(0, n, 7):
            r = bytearray(rs[t]); r[int(rng.integers(0, len(r)))] = ord("N"); rs[t] = bytes(r)
        for t in range(3, n, 11):
            q = bytearray(qs[t]); q[int(rng.integers(0, len(q)))] = ord("N"); qs[t] = bytes(q)
    return qs, rs, diag


def _aligner(pkg, orc, pm, mode, sg, open_, ext):
    b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext)
    [b.global_, b.semi_global, b.local][mode]()
    if mode == 1:
        qg = [t for f, t in ((orc.S1_BEG, "prefix"), (orc.S1_END, "suffix")) if sg & f]
        dg = [t for f, t in ((orc.S2_BEG, "prefix"), (orc.S2_END, "suffix")) if sg & f]
        b.allow_query_gaps(qg).allow_ref_gaps(dg)
    return b.build()


def _check(pkg, orc, al, om, mode, sg, open_, ext, qs, rs, k, diag, expect_kernel=None, tag=()):
    qb, qo = orc.pack(qs); rb, ro = orc.pack(rs)
    got = al.align_batch_banded(qs, rs, k, diag)
    kernel = pkg.lib.pmx_last_kernel().decode()
    if expect_kernel is not None:
        assert kernel.startswith(expect_kernel), (kernel, tag)
    want = orc.align_banded_batch(mode, qb, qo, rb, ro, open_, ext, om, k, diag, sg_flags=sg)
    bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]) | (got["flags"] != 0))[0]
    assert len(bad) == 0, (tag, kernel, mode, sg, open_, ext, k, bad[:5], got[bad[:3]], want[bad[:3]],
                           [(len(qs[x]), len(rs[x]), int(diag[x]) if diag is not None else 0) for x in bad[:3]])
    return got


@pytest.mark.parametrize("sg", list(range(1, 16)))
def test_every_free_end_set(pkg, orc, sg):
    rng = np.random.default_rng(9400 + sg)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    for open_, ext, k in ((5, 2, 15), (3, 3, 7), (4, 0, 31), (11, 1, 3)):
        al = _aligner(pkg, orc, pm, 1, sg, open_, ext)
        qs, rs, diag = _batch(rng, 200, 1, 140)
        for dg in (None, diag):
            _check(pkg, orc, al, om, 1, sg, open_, ext, qs, rs, k, dg, "pmx_bstrip_kernel/", (open_, ext, k))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("skew", ["double", "one"])
def test_every_lane_shape_and_both_skews(pkg, orc, mode, skew, monkeypatch):
    """bands that fill their shape exactly, leave one offset or many in front of the band (guarded cells, whole guarded lanes)"""
    if skew == "one":
        monkeypatch.setenv("PMX_BSTRIP_ONE_SKEW", "1")
    rng = np.random.default_rng(9500 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 3, -2), orc.Matrix.create("ACGT", 3, -2)
    al = _aligner(pkg, orc, pm, mode, orc.SG_ALL, 4, 1)
    name = "pmx_bstrip_kernel/local" if mode == 2 else "pmx_bstrip_kernel/%s skew" % skew
    for k in (0, 2, 3, 4, 5, 7, 8, 11, 12, 15, 17, 23, 24, 31, 33, 40, 47, 48, 50, 51, 52, 63):
        qs, rs, diag = _batch(rng, 150, 1, 200)
        _check(pkg, orc, al, om, mode, orc.SG_ALL, 4, 1, qs, rs, k, diag, name, (k,))
    for shape, k in (("4x8", 15), ("8x8", 31), ("8x8", 20), ("4x16", 9), ("8x16", 40), ("2x16", 3)):
        monkeypatch.setenv("PMX_BSTRIP_SHAPE", shape)
        qs, rs, diag = _batch(rng, 150, 1, 200)
        _check(pkg, orc, al, om, mode, orc.SG_ALL, 4, 1, qs, rs, k, diag, name, (shape, k))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_wildcards_are_handed_back_and_large_batches_are_sorted(pkg, orc, mode):
    """a reference letter beyond the first four has no selector: those pairs come back through the retry list and run in
    pmx_banded_kernel; a query wildcard is a table row of its own.  5 000 pairs: the processing order by band length is built"""
    rng = np.random.default_rng(9600 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    al = _aligner(pkg, orc, pm, mode, orc.SG_ALL, 5, 2)
    for n, lo, hi, k in ((300, 1, 150, 15), (5000, 20, 260, 31), (4100, 1, 40, 48)):
        qs, rs, diag = _batch(rng, n, lo, hi, wild=True)
        _check(pkg, orc, al, om, mode, orc.SG_ALL, 5, 2, qs, rs, k, diag, "pmx_bstrip_kernel/", (n, k))


def test_window_fallbacks(pkg, orc):
    """outside the int16 window (long sequences under a large extension penalty, scores below -open, open < extend) the
    anti-diagonal kernels run; inside it the strip kernel does -- same records either way"""
    rng = np.random.default_rng(9700)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs, rs, diag = _batch(rng, 40, 2500, 3000, far=0.0)
    al = _aligner(pkg, orc, pm, 0, 0, 12, 6)
    _check(pkg, orc, al, om, 0, 0, 12, 6, qs, rs, 31, diag, "pmx_banded_", "long + large ext")
    al = _aligner(pkg, orc, pm, 0, 0, 5, 1)
    _check(pkg, orc, al, om, 0, 0, 5, 1, qs, rs, 31, diag, "pmx_bstrip_kernel/", "long, small ext")
    qs, rs, diag = _batch(rng, 100, 10, 200)
    al = _aligner(pkg, orc, pm, 2, 0, 1, 2)
    _check(pkg, orc, al, om, 2, 0, 1, 2, qs, rs, 15, diag, "pmx_banded_", "open < extend")
    al = _aligner(pkg, orc, pm, 1, orc.SG_ALL, 2, 0)
    _check(pkg, orc, al, om, 1, orc.SG_ALL, 2, 0, qs, rs, 15, diag, "pmx_banded_", "mismatch below -open")
    pm2, om2 = pkg.Matrix.create(b"ACGT", 1, -1), orc.Matrix.create("ACGT", 1, -1)
    al = _aligner(pkg, orc, pm2, 1, orc.SG_ALL, 1, 1)
    _check(pkg, orc, al, om2, 1, orc.SG_ALL, 1, 1, qs, rs, 15, diag, "pmx_bstrip_kernel/", "1/-1, 1/1")


def test_banded_nw_single_pair_entry(pkg, orc):
    """Aligner::banded_nw (one pair per call) reaches the strip kernel too"""
    rng = np.random.default_rng(9800)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    for k in (2, 16, 40):
        al = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).bandwidth(k).build()
        for _ in range(6):
            q = random_seqs(rng, 1, 1, 300)[0]; r = mutate(rng, q, 0.1, 0.05)
            qb, qo = orc.pack([q]); rb, ro = orc.pack([r])
            want = orc.align_banded_batch(orc.NW, qb, qo, rb, ro, 5, 2, om, k)[0]
            res = al.banded_nw(q, r)
            assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_bstrip_kernel/")
            assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == tuple(want)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_one_shared_query_every_mode(pkg, orc, mode):
    """the profile arm (one shared query, staged once per wave): band centres that enter the matrix deep in the query or miss it,
    references much shorter and much longer than the query, bands 5 / 31 / 48"""
    rng = np.random.default_rng(9850 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = random_seqs(rng, 1, 420, 420)[0]
    rs, dg = [], []
    for t in range(900):
        kind = t % 6
        if kind == 0:
            a = int(rng.integers(0, 300)); body = mutate(rng, q[a:a + int(rng.integers(20, 420 - a + 1))], 0.08, 0.04)
            pre = random_seqs(rng, 1, 0, 200)[0]; r = pre + body + random_seqs(rng, 1, 0, 100)[0]; d = len(pre) - a + int(rng.integers(-3, 4))
        elif kind == 1:
            r = random_seqs(rng, 1, 1, 60)[0]; d = -int(rng.integers(0, 420))
        elif kind == 2:
            r = random_seqs(rng, 1, 300, 900)[0]; d = int(rng.integers(0, len(r)))
        elif kind == 3:
            r = random_seqs(rng, 1, 1, 120)[0]; d = int(rng.integers(-600, 300))
        elif kind == 4:
            r = mutate(rng, q, 0.1, 0.05); d = int(rng.integers(-4, 5))
        else:
            r = q[int(rng.integers(0, 200)):]; d = -(len(q) - len(r))
        rs.append(r or b"C"); dg.append(d)
    dg = np.array(dg, dtype=np.int32)
    b = pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).profile(pkg.Profile.new(q, False, pm))
    [b.global_, b.semi_global, b.local][mode]()
    al = b.build()
    rb, ro = orc.pack(rs)
    for k in (5, 31, 48):
        for d_ in (None, dg):
            got = al.align_batch_banded([], rs, k, d_)
            assert pkg.lib.pmx_last_kernel().decode().startswith("pmx_bstrip_kernel/")
            want = orc.align_banded_batch(mode, None, None, rb, ro, 5, 2, om, k, d_, shared_query=q)
            bad = np.nonzero((got["score"] != want[:, 0]) | (got["end_query"] != want[:, 1]) | (got["end_ref"] != want[:, 2]) | (got["flags"] != 0))[0]
            assert len(bad) == 0, (mode, k, d_ is None, bad[:5], got[bad[:3]], want[bad[:3]], [(len(rs[x]), int(dg[x])) for x in bad[:3]])
