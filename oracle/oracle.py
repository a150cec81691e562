"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.  See oracle/pmx_oracle.c for what
is restated and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpmx_oracle.so")

NW, SG, SW = 0, 1, 2
S1_BEG, S1_END, S2_BEG, S2_END = 1, 2, 4, 8
SG_ALL = S1_BEG | S1_END | S2_BEG | S2_END


class _Result(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("score", "end_query", "end_ref", "matches", "similar", "length", "saturated")]


class _Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("score_table", "matches_table", "similar_table", "length_table",
                 "score_row", "matches_row", "similar_row", "length_row",
                 "score_col", "matches_col", "similar_col", "length_col",
                 "trace_table")]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("pmx_oracle.c", "pmx_oracle_batch.c", "pmx_striped_cpu.c", "pmx_cpu_inter16.c", "pmx_striped_body.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_align.restype = C.c_int
        _lib.orc_walk.restype = C.c_int
        _lib.orc_align_batch.restype = C.c_int
        _lib.pmx_cpu_sw_striped16_batch.restype = C.c_int
        _lib.pmx_cpu_sw_striped16_batch2.restype = C.c_int
        _lib.pmx_cpu_striped_lanes.restype = C.c_int
        _lib.orc_align_stats_sample.restype = C.c_int
        _lib.orc_cigar_sample.restype = C.c_int
        _lib.orc_rescore_cigars.restype = C.c_long
        _lib.orc_align_banded_batch.restype = C.c_int
        _lib.pmx_cpu_nw_stats_inter16.restype = C.c_int
        _lib.pmx_cpu_trace_cigar_inter16.restype = C.c_int
    return _lib


class Matrix:
    """size x size int32 scores + 256-entry mapper (byte -> row/col index)."""

    def __init__(self, scores, mapper, alphabet=""):
        self.scores = np.ascontiguousarray(scores, dtype=np.int32)
        self.size = int(self.scores.shape[0])
        self.mapper = np.ascontiguousarray(mapper, dtype=np.int32)
        self.alphabet = alphabet

    @classmethod
    def create(cls, alphabet, match, mismatch):
        if isinstance(alphabet, bytes):
            alphabet = alphabet.decode()
        n = len(alphabet) + 1
        m = np.zeros((n, n), dtype=np.int32)
        mp = np.zeros(256, dtype=np.int32)
        lib().orc_matrix_create(alphabet.encode(), int(match), int(mismatch),
                                m.ctypes.data_as(C.c_void_p), mp.ctypes.data_as(C.c_void_p))
        return cls(m, mp, alphabet)

    @classmethod
    def default(cls):
        return cls.create("ACGTA", 1, -1)     # src/matrix/mod.rs:246-250

    @classmethod
    def from_file(cls, path):
        """Square-matrix file format of tests/square.txt (comment lines '#', header row of
        symbols, one row per symbol with a leading repeat of the symbol)."""
        rows, alphabet = [], None
        with open(path) as fh:
            for line in fh:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                tok = line.split()
                if alphabet is None:
                    alphabet = tok
                    continue
                rows.append([int(x) for x in tok[1:]])
        n = len(alphabet)
        m = np.array(rows, dtype=np.int32).reshape(n, n)
        mp = np.full(256, n - 1, dtype=np.int32)
        for i, ch in enumerate(alphabet[:-1]):
            mp[ord(ch.upper())] = i
            mp[ord(ch.lower())] = i
        mp[ord(alphabet[-1])] = n - 1
        return cls(m, mp, "".join(alphabet))


class Result:
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def align(mode, query, ref, open_, ext, matrix, sg_flags=SG_ALL, bits=0,
          stats=False, table=False, rowcol=False, trace=False):
    """One pair through the scalar oracle.  Returns an object with score, end_query,
    end_ref, saturated and (if requested) stats / tables / rows / cols / trace."""
    q = np.frombuffer(bytes(query), dtype=np.uint8)
    r = np.frombuffer(bytes(ref), dtype=np.uint8)
    ql, rl = len(q), len(r)
    res, out = _Result(), _Outputs()
    keep = {}

    def mk(name, shape, dtype=np.int32):
        a = np.zeros(shape, dtype=dtype)
        keep[name] = a
        setattr(out, name, a.ctypes.data)

    if table:
        mk("score_table", (ql, rl))
        if stats:
            for n in ("matches_table", "similar_table", "length_table"):
                mk(n, (ql, rl))
    if rowcol:
        mk("score_row", (rl,)); mk("score_col", (ql,))
        if stats:
            for n in ("matches", "similar", "length"):
                mk(n + "_row", (rl,)); mk(n + "_col", (ql,))
    if trace:
        mk("trace_table", (ql, rl), np.int8)
    rc = lib().orc_align(mode, sg_flags, _ptr(q), ql, _ptr(r), rl, int(open_), int(ext),
                         _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper),
                         int(bits), int(stats), C.byref(res), C.byref(out))
    if rc:
        raise RuntimeError("orc_align failed rc=%d" % rc)
    o = Result()
    for n, _ in _Result._fields_:
        setattr(o, n, getattr(res, n))
    for n, a in keep.items():
        setattr(o, n, a)
    o.mode, o.query, o.ref, o.matrix = mode, bytes(query), bytes(ref), matrix
    return o


def walk(res):
    """Traceback walk -> (ops string, beg_query, beg_ref)."""
    q = np.frombuffer(res.query, dtype=np.uint8)
    r = np.frombuffer(res.ref, dtype=np.uint8)
    buf = C.create_string_buffer(len(q) + len(r) + 2)
    bq, br = C.c_int(), C.c_int()
    n = lib().orc_walk(res.mode, _ptr(res.trace_table), _ptr(q), len(q), _ptr(r), len(r),
                       _ptr(res.matrix.mapper), res.end_query, res.end_ref, buf,
                       C.byref(bq), C.byref(br))
    if n < 0:
        raise RuntimeError("orc_walk failed")
    return buf.value.decode(), bq.value, br.value


def cigar(res):
    ops, _, _ = walk(res)
    out = C.create_string_buffer(12 * (len(ops) + 1))
    lib().orc_cigar_text(ops.encode(), len(ops), out, len(out))
    return out.value.decode()


def traceback_strings(res, match="|", pos=" ", neg=" "):
    ops, bq, br = walk(res)
    q = np.frombuffer(res.query, dtype=np.uint8)
    r = np.frombuffer(res.ref, dtype=np.uint8)
    n = len(ops)
    qs, cs, rs = (C.create_string_buffer(n + 1) for _ in range(3))
    lib().orc_traceback_strings(ops.encode(), n, _ptr(q), _ptr(r), bq, br,
                                _ptr(res.matrix.scores), res.matrix.size, _ptr(res.matrix.mapper),
                                C.c_char(match.encode()), C.c_char(pos.encode()), C.c_char(neg.encode()),
                                qs, cs, rs)
    return qs.value.decode(), cs.value.decode(), rs.value.decode()


def pack(seqs):
    """list of bytes -> (uint8 buffer, int64 offsets[n+1])."""
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    buf = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8).copy()
    return buf, off


def align_batch(mode, qbuf, qoff, rbuf, roff, open_, ext, matrix, sg_flags=SG_ALL, bits=0):
    n = len(qoff) - 1
    out = np.zeros((n, 3), dtype=np.int32)
    qoff = np.ascontiguousarray(qoff, dtype=np.int64)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    rc = lib().orc_align_batch(mode, sg_flags, C.c_long(n), _ptr(qbuf), _ptr(qoff), _ptr(rbuf), _ptr(roff),
                               int(open_), int(ext), _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper),
                               int(bits), _ptr(out))
    if rc:
        raise RuntimeError("orc_align_batch: some pair failed")
    return out


def cpu_sw_striped16_batch(qbuf, qoff, rbuf, roff, open_, ext, matrix, threads=0, shared_query=None, lanes=0):
    """CPU timing baseline (Farrar striped int16; AVX-512BW or AVX2 picked at run time, + OpenMP).
    shared_query: bytes of one query whose profile is built once per thread (profile arm); qbuf/qoff are ignored then.
    lanes: 16 / 32 force a vector width (tests), 0 = what the CPU supports.  Returns (out[n,3], threads_used)."""
    n = len(roff) - 1
    out = np.zeros((n, 3), dtype=np.int32)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    lib().pmx_cpu_striped_force_lanes(int(lanes))
    if shared_query is not None:
        qb = np.frombuffer(bytes(shared_query), dtype=np.uint8)
        used = lib().pmx_cpu_sw_striped16_batch2(C.c_long(n), _ptr(qb), None, len(qb), _ptr(rbuf), _ptr(roff),
                                                 int(open_), int(ext), _ptr(matrix.scores), matrix.size,
                                                 _ptr(matrix.mapper), _ptr(out), int(threads))
    else:
        qoff = np.ascontiguousarray(qoff, dtype=np.int64)
        used = lib().pmx_cpu_sw_striped16_batch2(C.c_long(n), _ptr(qbuf), _ptr(qoff), 0, _ptr(rbuf), _ptr(roff),
                                                 int(open_), int(ext), _ptr(matrix.scores), matrix.size,
                                                 _ptr(matrix.mapper), _ptr(out), int(threads))
    lib().pmx_cpu_striped_force_lanes(0)
    return out, used


def cpu_striped_lanes():
    return int(lib().pmx_cpu_striped_lanes())


def align_stats_sample(mode, index, qbuf, qoff, rbuf, roff, open_, ext, matrix, sg_flags=SG_ALL, bits=0, shared_query=None):
    """Pairs `index` of a packed batch with statistics -> int32 [m, 7]: score, end_query, end_ref, matches, similar,
    length, saturated.  shared_query: bytes of the one query every pair uses (profile arm)."""
    index = np.ascontiguousarray(index, dtype=np.int64)
    out = np.zeros((len(index), 7), dtype=np.int32)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    if shared_query is not None:
        qb = np.frombuffer(bytes(shared_query), dtype=np.uint8)
        rc = lib().orc_align_stats_sample(mode, sg_flags, C.c_long(len(index)), _ptr(index), _ptr(qb), None, len(qb),
                                          _ptr(rbuf), _ptr(roff), int(open_), int(ext), _ptr(matrix.scores), matrix.size,
                                          _ptr(matrix.mapper), int(bits), _ptr(out))
    else:
        qoff = np.ascontiguousarray(qoff, dtype=np.int64)
        rc = lib().orc_align_stats_sample(mode, sg_flags, C.c_long(len(index)), _ptr(index), _ptr(qbuf), _ptr(qoff), 0,
                                          _ptr(rbuf), _ptr(roff), int(open_), int(ext), _ptr(matrix.scores), matrix.size,
                                          _ptr(matrix.mapper), int(bits), _ptr(out))
    if rc:
        raise RuntimeError("orc_align_stats_sample: some pair failed")
    return out


def cigar_sample(mode, index, qbuf, qoff, rbuf, roff, open_, ext, matrix, sg_flags=SG_ALL):
    """Pairs `index` of a packed batch: (list of CIGAR text, int32 [m, 5]: score, end_query, end_ref, beg_query, beg_ref)."""
    index = np.ascontiguousarray(index, dtype=np.int64)
    qoff = np.ascontiguousarray(qoff, dtype=np.int64)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    m = len(index)
    stride = int(12 * ((qoff[1:] - qoff[:-1])[index] + (roff[1:] - roff[:-1])[index]).max() + 16) if m else 16
    text = np.zeros((m, stride), dtype=np.uint8)
    rec = np.zeros((m, 5), dtype=np.int32)
    rc = lib().orc_cigar_sample(mode, sg_flags, C.c_long(m), _ptr(index), _ptr(qbuf), _ptr(qoff), _ptr(rbuf), _ptr(roff),
                                int(open_), int(ext), _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper),
                                _ptr(text), stride, _ptr(rec))
    if rc:
        raise RuntimeError("orc_cigar_sample: some pair failed")
    return [bytes(row).split(b"\0", 1)[0].decode() for row in text], rec


def rescore_cigars(text, toff, qbuf, qoff, rbuf, roff, open_, ext, matrix, beg=None, free_mask=0):
    """Independent re-scoring of CIGAR text (uint8 buffer + int64 offsets) -> (int32 [n, 4]: score, query consumed,
    reference consumed, mislabelled =/X columns; number of malformed texts)."""
    n = len(toff) - 1
    out = np.zeros((n, 4), dtype=np.int32)
    toff = np.ascontiguousarray(toff, dtype=np.int64)
    qoff = np.ascontiguousarray(qoff, dtype=np.int64)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    if beg is not None:
        beg = np.ascontiguousarray(beg, dtype=np.int32)
    bad = lib().orc_rescore_cigars(C.c_long(n), _ptr(text), _ptr(toff), _ptr(qbuf), _ptr(qoff), _ptr(rbuf), _ptr(roff),
                                   _ptr(beg), int(open_), int(ext), int(free_mask), _ptr(matrix.scores), matrix.size,
                                   _ptr(matrix.mapper), _ptr(out))
    return out, int(bad)


def align_banded_batch(mode, qbuf, qoff, rbuf, roff, open_, ext, matrix, band, diag=None, sg_flags=SG_ALL, shared_query=None):
    """Banded alignment of a packed batch (cells with |(j - i) - diag| > band are excluded; diag: int32 per pair or None
    for the main diagonal) -> int32 [n, 3]: score, end_query, end_ref."""
    n = len(roff) - 1
    out = np.zeros((n, 3), dtype=np.int32)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    if diag is not None:
        diag = np.ascontiguousarray(diag, dtype=np.int32)
    if shared_query is not None:
        qb = np.frombuffer(bytes(shared_query), dtype=np.uint8)
        rc = lib().orc_align_banded_batch(mode, sg_flags, C.c_long(n), _ptr(qb), None, len(qb), _ptr(rbuf), _ptr(roff),
                                          int(open_), int(ext), _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper),
                                          int(band), _ptr(diag), _ptr(out))
    else:
        qoff = np.ascontiguousarray(qoff, dtype=np.int64)
        rc = lib().orc_align_banded_batch(mode, sg_flags, C.c_long(n), _ptr(qbuf), _ptr(qoff), 0, _ptr(rbuf), _ptr(roff),
                                          int(open_), int(ext), _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper),
                                          int(band), _ptr(diag), _ptr(out))
    if rc:
        raise RuntimeError("orc_align_banded_batch: some pair failed")
    return out


def cpu_nw_stats_inter16(query, rbuf, roff, open_, ext, matrix, threads=0):
    """Vectorised CPU restatement of `nw_stats_*_profile_16` (one shared query; 16 references per AVX2 vector: pmx_cpu_inter16.c)
    -> (int32 [n, 6]: score, end_query, end_ref, matches, similar, length; threads used)."""
    n = len(roff) - 1
    out = np.zeros((n, 6), dtype=np.int32)
    qb = np.frombuffer(bytes(query), dtype=np.uint8)
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    used = lib().pmx_cpu_nw_stats_inter16(C.c_long(n), _ptr(qb), len(qb), _ptr(rbuf), _ptr(roff), int(open_), int(ext),
                                          _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper), _ptr(out), int(threads))
    if used < 0:
        raise RuntimeError("pmx_cpu_nw_stats_inter16 failed (%d)" % used)
    return out, used


def cpu_trace_cigar_inter16(mode, qbuf, qoff, rbuf, roff, open_, ext, matrix, threads=0, decode=True):
    """Vectorised CPU restatement of `{nw,sg}_trace_*_16` + get_cigar (16 pairs per AVX2 vector, byte trace table, scalar walk;
    semi-global = all four ends free; match / mismatch matrices only) -> (list of CIGAR texts, int32 [n, 5]: score, end_query,
    end_ref, beg_query, beg_ref; threads used)."""
    n = len(roff) - 1
    qoff = np.ascontiguousarray(qoff, dtype=np.int64); roff = np.ascontiguousarray(roff, dtype=np.int64)
    stride = int(max(16, 2 * (int(np.max(qoff[1:] - qoff[:-1])) + int(np.max(roff[1:] - roff[:-1]))) + 16))
    text = np.zeros((n, stride), dtype=np.uint8)
    rec = np.zeros((n, 5), dtype=np.int32)
    used = lib().pmx_cpu_trace_cigar_inter16(1 if mode == SG else 0, C.c_long(n), _ptr(qbuf), _ptr(qoff), _ptr(rbuf), _ptr(roff),
                                             int(open_), int(ext), _ptr(matrix.scores), matrix.size, _ptr(matrix.mapper),
                                             _ptr(text), stride, _ptr(rec), int(threads))
    if used < 0:
        raise RuntimeError("pmx_cpu_trace_cigar_inter16 failed (%d)" % used)
    blob = text.tobytes()                                    # (NUL-terminated slots: one split per pair, outside any timed region)
    texts = [blob[k * stride:(k + 1) * stride].split(b"\0", 1)[0] for k in range(n)] if decode else text
    return texts, rec, used
