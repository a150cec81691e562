"""Score tables and last rows / columns through the row-by-row kernel (`-m gpu`): cell by cell against the oracle.
Reference accessors: /root/reference/src/alignment/mod.rs:123-288, layout src/alignment/table.rs:4-9."""
import numpy as np
import pytest

import workloads as wl
from util import random_seqs, mutate, DNA, AA

pytestmark = pytest.mark.gpu


def _batch_tables(pkg, cfg, qs, rs, want_table=True):
    import torch
    dev = torch.device("cuda", 0)
    qb, qo = pkg.pack(qs); rb, ro = pkg.pack(rs)
    n = len(qs)
    ql = (qo[1:] - qo[:-1]); rl = (ro[1:] - ro[:-1])
    toff = np.zeros(n + 1, dtype=np.int64); np.cumsum(ql * rl, out=toff[1:])
    d = [torch.from_numpy(x).to(dev) for x in (qb, qo, rb, ro, toff)]
    table = torch.full((int(toff[-1]),), -7, dtype=torch.int32, device=dev) if want_table else None
    row = torch.full((int(ro[-1]),), -7, dtype=torch.int32, device=dev)
    col = torch.full((int(qo[-1]),), -7, dtype=torch.int32, device=dev)
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    rc = pkg.lib.pmx_align_batch_table_device(__import__("ctypes").byref(cfg), n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                              int(ql.max()), int(rl.max()), d[4].data_ptr(), table.data_ptr() if want_table else None,
                                              row.data_ptr(), col.data_ptr(), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, pkg.lib.pmx_last_error().decode()
    torch.cuda.synchronize()
    return (table.cpu().numpy() if want_table else None), row.cpu().numpy(), col.cpu().numpy(), out.cpu().numpy(), toff, qo, ro


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("gaps", [(5, 2), (0, 0), (3, 3), (11, 1)])
def test_table_kernel_cell_by_cell(pkg, orc, mode, gaps):
    rng = np.random.default_rng(9100 + mode * 10 + gaps[0])
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 90, 1, 300)
    rs = [mutate(rng, q, 0.1, 0.05) if i % 3 else random_seqs(rng, 1, 1, 1000)[0] for i, q in enumerate(qs)]
    rs[0] = random_seqs(rng, 1, 1024, 1024)[0]; rs[1] = b"A"; qs[2] = b"C"                  # the widest shape, single-symbol edges
    for sg in ((15,) if mode != 1 else (15, 0, 1 | 8, 2 | 4, 2)):
        cfg = pkg.pmx_config_t(mode, sg, gaps[0], gaps[1], 32, 0, pm.inner)
        table, row, col, out, toff, qo, ro = _batch_tables(pkg, cfg, qs, rs)
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_table_kernel"
        for k in range(len(qs)):
            w = orc.align(mode, qs[k], rs[k], gaps[0], gaps[1], om, sg_flags=sg, table=True, rowcol=True)
            t = table[toff[k]:toff[k + 1]].reshape(len(qs[k]), len(rs[k]))
            bad = np.argwhere(t != w.score_table)
            assert len(bad) == 0, (mode, sg, gaps, k, len(qs[k]), len(rs[k]), bad[:3], t[tuple(bad[0])], w.score_table[tuple(bad[0])])
            assert (row[ro[k]:ro[k + 1]] == w.score_row).all() and (col[qo[k]:qo[k + 1]] == w.score_col).all()
            assert tuple(out[k, :3]) == (w.score, w.end_query, w.end_ref), (mode, sg, k)


def test_table_kernel_protein_1000_by_1000_and_rowcol_only(pkg, orc):
    """>= 1 000 pairs in total across the two table tests, up to 1 000 x 1 000, BLOSUM62 11/1; and the rowcol-only form"""
    rng = np.random.default_rng(9200)
    pm = pkg.Matrix.from_name("blosum62")
    om = orc.Matrix.from_file("tests/golden/blosum62.txt")
    qs = random_seqs(rng, 40, 600, 1000, AA) + random_seqs(rng, 900, 20, 120, AA)
    rs = [mutate(rng, q, 0.3, 0.05, AA)[:1000] for q in qs]
    for mode in (0, 2):
        cfg = pkg.pmx_config_t(mode, 15, 11, 1, 32, 0, pm.inner)
        table, row, col, out, toff, qo, ro = _batch_tables(pkg, cfg, qs, rs)
        _, row2, col2, out2, _, _, _ = _batch_tables(pkg, cfg, qs, rs, want_table=False)
        assert (row == row2).all() and (col == col2).all() and (out == out2).all()
        for k in range(len(qs)):
            w = orc.align(mode, qs[k], rs[k], 11, 1, om, table=True, rowcol=True)
            assert (table[toff[k]:toff[k + 1]].reshape(len(qs[k]), len(rs[k])) == w.score_table).all(), (mode, k)
            assert (row[ro[k]:ro[k + 1]] == w.score_row).all() and (col[qo[k]:qo[k + 1]] == w.score_col).all()
            assert tuple(out[k, :3]) == (w.score, w.end_query, w.end_ref)


@pytest.mark.parametrize("chunk_bytes", [None, "1", "71000", "142000"])
def test_table_batches_beyond_the_table_kernel_in_chunks(pkg, orc, monkeypatch, chunk_bytes):
    """References beyond the row-by-row kernel's 1 024 columns: the general kernel, in chunks of bounded scratch.  Chunks of
    one pair (n = k * chunk + 1 leaves one; a forced tiny chunk makes every chunk one pair) keep the packed row / column
    addressing of the batch (round-2 advisor finding: they used to be rejected)."""
    rng = np.random.default_rng(9250)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    qs = random_seqs(rng, 5, 30, 90)
    rs = random_seqs(rng, 5, 1030, 1100)
    if chunk_bytes is not None:
        monkeypatch.setenv("PMX_GENERAL_CHUNK_BYTES", chunk_bytes)      # 1 -> one pair per chunk; 71000 -> two (5 = 2 * 2 + 1); 142000 -> four
    for mode in (0, 1, 2):
        cfg = pkg.pmx_config_t(mode, 15, 5, 2, 32, 0, pm.inner)
        table, row, col, out, toff, qo, ro = _batch_tables(pkg, cfg, qs, rs)
        assert pkg.lib.pmx_last_kernel().decode() == "pmx_general_kernel/tables"
        for k in range(len(qs)):
            w = orc.align(mode, qs[k], rs[k], 5, 2, om, table=True, rowcol=True)
            assert (table[toff[k]:toff[k + 1]].reshape(len(qs[k]), len(rs[k])) == w.score_table).all(), (mode, k)
            assert (row[ro[k]:ro[k + 1]] == w.score_row).all() and (col[qo[k]:qo[k + 1]] == w.score_col).all(), (mode, k)
            assert tuple(out[k, :3]) == (w.score, w.end_query, w.end_ref)


def test_single_pair_table_accessors_use_the_table_kernel(pkg, orc):
    """Aligner::use_table / use_last_rowcol (src/aligner/mod.rs:225-246) -> get_score_table / row / col on one pair"""
    rng = np.random.default_rng(9300)
    pm, om = pkg.Matrix.create(b"ACGT", 3, -2), orc.Matrix.create("ACGT", 3, -2)
    q = random_seqs(rng, 1, 700, 700)[0]; r = mutate(rng, q, 0.1, 0.03)
    for sel, mode in (("global_", 0), ("semi_global", 1), ("local", 2)):
        b = pkg.Aligner.new().matrix(pm).gap_open(4).gap_extend(1).use_table(); getattr(b, sel)()
        res = b.build().align(q, r)
        w = orc.align(mode, q, r, 4, 1, om, table=True, rowcol=True)
        t = res.get_score_table()
        assert (t.rows(), t.cols()) == (len(q), len(r)) and (np.array(t.as_slice()).reshape(len(q), len(r)) == w.score_table).all()
        assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (w.score, w.end_query, w.end_ref)
        b = pkg.Aligner.new().matrix(pm).gap_open(4).gap_extend(1).use_last_rowcol(); getattr(b, sel)()
        res = b.build().align(q, r)
        assert (np.array(res.get_score_row()) == w.score_row).all() and (np.array(res.get_score_col()) == w.score_col).all()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_single_pair_trace_table_from_the_table_kernel(pkg, orc, mode):
    """Aligner::use_trace on one pair (src/aligner/mod.rs:251-267 -> get_trace_table / get_cigar / get_traceback_strings): the
    row-by-row kernel writes the reference's byte per cell; every byte, the CIGAR and the three strings against the oracle for
    lengths that exercise all three column widths per lane (references <= 256, <= 512, <= 1024), ragged row ends, one-symbol
    sequences, gap models with open == extend (every tie in the E / F bits) and with extend = 0, DNA and protein."""
    rng = np.random.default_rng(9400 + mode)
    cases = [(b"ACGT", 2, -3, 5, 2, DNA), (b"ACGT", 1, -1, 1, 1, DNA), (b"ACGT", 3, -2, 4, 0, DNA), (None, 0, 0, 11, 1, AA), (None, 0, 0, 3, 3, AA)]
    shapes = [(1, 1), (1, 40), (33, 1), (5, 300), (150, 150), (70, 257), (64, 256), (300, 513), (200, 1024), (129, 1023)]
    for alpha, ma, mi, open_, ext, letters in cases:
        if alpha is None:
            pm, om = pkg.Matrix.from_name("blosum62"), orc.Matrix.from_file("tests/golden/blosum62.txt")
        else:
            pm, om = pkg.Matrix.create(alpha, ma, mi), orc.Matrix.create(alpha.decode(), ma, mi)
        for ql, rl in shapes:
            q = random_seqs(rng, 1, ql, ql, letters)[0]
            r = (mutate(rng, q, 0.15, 0.05, letters) + random_seqs(rng, 1, rl, rl, letters)[0])[:rl] if ql > 20 and rng.random() < 0.7 \
                else random_seqs(rng, 1, rl, rl, letters)[0]
            b = pkg.Aligner.new().matrix(pm).gap_open(open_).gap_extend(ext).use_trace()
            [b.global_, b.semi_global, b.local][mode]()
            res = b.build().align(q, r)
            w = orc.align(mode, q, r, open_, ext, om, trace=True)
            ctx = (mode, open_, ext, ql, rl)
            assert (res.get_score(), res.get_end_query(), res.get_end_ref()) == (w.score, w.end_query, w.end_ref), ctx
            got = np.asarray(res.get_trace_table().as_slice()).reshape(len(q), len(r))
            bad = np.argwhere(got != w.trace_table)
            assert len(bad) == 0, (ctx, bad[:5], [(int(got[i, j]), int(w.trace_table[i, j])) for i, j in bad[:5]])
            assert res.get_cigar(q, r) == orc.cigar(w), ctx
            tb = res.get_traceback_strings(q, r)
            assert (tb.query, tb.comparison, tb.reference) == orc.traceback_strings(w), ctx


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_trace_and_statistics_tables_of_long_pairs(pkg, orc, mode):
    """Outputs of one long pair from the general kernel in its pipelined multi-wave form (references beyond the table kernel's
    1 024 columns, queries of several 64-row bands): trace table bytes + CIGAR, and the four statistics tables with last rows /
    columns, against the oracle."""
    rng = np.random.default_rng(9500 + mode)
    pm, om = pkg.Matrix.create(b"ACGT", 2, -3), orc.Matrix.create("ACGT", 2, -3)
    q = random_seqs(rng, 1, 333, 333)[0]
    r = mutate(rng, q, 0.1, 0.04) + random_seqs(rng, 1, 900, 900)[0]
    mk = lambda: [pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).global_,
                  pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).semi_global,
                  pkg.Aligner.new().matrix(pm).gap_open(5).gap_extend(2).local][mode]()
    w = orc.align(mode, q, r, 5, 2, om, stats=True, table=True, rowcol=True, trace=True)
    tr = mk().use_trace().build().align(q, r)
    got = np.asarray(tr.get_trace_table().as_slice()).reshape(len(q), len(r))
    assert (got == w.trace_table).all() and tr.get_cigar(q, r) == orc.cigar(w)
    st = mk().use_stats().use_table().build().align(q, r)
    for t in ("score", "matches", "similar", "length"):
        assert (np.asarray(getattr(st, "get_%s_table" % t)().as_slice()).reshape(len(q), len(r)) == getattr(w, t + "_table")).all(), t
    rc = mk().use_stats().use_last_rowcol().build().align(q, r)
    for t in ("score", "matches", "similar", "length"):
        assert (np.asarray(getattr(rc, "get_%s_row" % t)()) == getattr(w, t + "_row")).all(), t
        assert (np.asarray(getattr(rc, "get_%s_col" % t)()) == getattr(w, t + "_col")).all(), t
    assert (st.get_score(), st.get_matches(), st.get_similar(), st.get_length()) == (w.score, w.matches, w.similar, w.length)
