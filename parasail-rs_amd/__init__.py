"""parasail_rs_amd -- host-side mirror of the parasail-rs interface over libparasail_amd.so.

Rust is not available in the build image, so the reference's L2/L3 layer
(`Aligner` / `AlignerBuilder` / `Matrix` / `Profile` / `Alignment`,
/root/reference/src/{aligner,matrix,profile,alignment}/mod.rs) is mirrored here
with the same names, argument meaning, defaults, quirks and error behaviour, on
top of the C ABI in include/parasail_amd.h (ctypes; nothing here computes).
`global()` is spelled `global_()` because `global` is a Python keyword; Rust
panics become `PanicError`; `Err(...)` values become the exception classes below.

The DP always runs in the HIP kernels.  There is no CPU fallback: if the shared
library is missing, importing this package raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libparasail_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "parasail_amd.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libparasail_amd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C parasail-rs_amd`. There is no CPU fallback." % LIB_PATH)
lib = C.CDLL(LIB_PATH)


# ------------------------------------------------------------------ C types ----
class parasail_matrix_t(C.Structure):
    _fields_ = [("name", C.c_char_p), ("matrix", C.POINTER(C.c_int)), ("mapper", C.POINTER(C.c_int)),
                ("size", C.c_int), ("max", C.c_int), ("min", C.c_int),
                ("user_matrix", C.POINTER(C.c_int)), ("type_", C.c_int), ("length", C.c_int),
                ("alphabet", C.c_char_p), ("query", C.c_char_p)]


class parasail_traceback_t(C.Structure):
    _fields_ = [("query", C.c_void_p), ("comp", C.c_void_p), ("ref_", C.c_void_p)]


class parasail_cigar_t(C.Structure):
    _fields_ = [("seq", C.POINTER(C.c_uint32)), ("len", C.c_int), ("beg_query", C.c_int), ("beg_ref", C.c_int)]


class parasail_result_ssw_t(C.Structure):
    _fields_ = [("score1", C.c_uint16), ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32),
                ("read_begin1", C.c_int32), ("read_end1", C.c_int32),
                ("cigar", C.POINTER(C.c_uint32)), ("cigarLen", C.c_int32)]


class pmx_config_t(C.Structure):
    _fields_ = [("mode", C.c_int), ("sg_flags", C.c_int), ("open", C.c_int), ("extend", C.c_int),
                ("width", C.c_int), ("want", C.c_int), ("matrix", C.POINTER(parasail_matrix_t))]


RECORD_DTYPE = np.dtype([("score", "<i4"), ("end_query", "<i4"), ("end_ref", "<i4"), ("flags", "<i4")])
STATS_DTYPE = np.dtype([("matches", "<i4"), ("similar", "<i4"), ("length", "<i4")])

MODE_NW, MODE_SG, MODE_SW = 0, 1, 2
SG_QB, SG_QE, SG_DB, SG_DE, SG_ALL = 1, 2, 4, 8, 15
WANT_STATS, WANT_CIGAR, WANT_SORTED = 1, 2, 4
FLAG_SATURATED = 1

_MP = C.POINTER(parasail_matrix_t)
_FN = C.CFUNCTYPE(C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, _MP)
_PFN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int)


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("parasail_lookup_function", C.c_void_p, C.c_char_p)
_sig("parasail_lookup_pfunction", C.c_void_p, C.c_char_p)
_sig("parasail_nw_banded", C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _MP)
_sig("parasail_ssw", C.POINTER(parasail_result_ssw_t), C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, _MP)
_sig("parasail_ssw_init", C.c_void_p, C.c_char_p, C.c_int, _MP, C.c_int8)
_sig("parasail_result_ssw_free", None, C.POINTER(parasail_result_ssw_t))
for _n in ("score", "end_query", "end_ref", "matches", "similar", "length"):
    _sig("parasail_result_get_" + _n, C.c_int, C.c_void_p)
for _k in ("score", "matches", "similar", "length"):
    for _w in ("table", "row", "col"):
        _sig("parasail_result_get_%s_%s" % (_k, _w), C.POINTER(C.c_int), C.c_void_p)
_sig("parasail_result_get_trace_table", C.POINTER(C.c_int), C.c_void_p)
for _n in ("nw", "sg", "sw", "saturated", "banded", "scan", "striped", "diag", "blocked", "stats",
           "stats_table", "table", "rowcol", "stats_rowcol", "trace"):
    _sig("parasail_result_is_" + _n, C.c_int, C.c_void_p)
_sig("parasail_result_free", None, C.c_void_p)
_sig("parasail_result_get_traceback", C.POINTER(parasail_traceback_t), C.c_void_p, C.c_char_p, C.c_int,
     C.c_char_p, C.c_int, _MP, C.c_char, C.c_char, C.c_char)
_sig("parasail_traceback_free", None, C.POINTER(parasail_traceback_t))
_sig("parasail_traceback_generic", None, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, _MP,
     C.c_void_p, C.c_char, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int)
_sig("parasail_result_get_cigar", C.POINTER(parasail_cigar_t), C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, _MP)
_sig("parasail_cigar_decode", C.c_void_p, C.POINTER(parasail_cigar_t))
_sig("parasail_cigar_free", None, C.POINTER(parasail_cigar_t))
_sig("parasail_matrix_create", _MP, C.c_char_p, C.c_int, C.c_int)
_sig("parasail_matrix_lookup", _MP, C.c_char_p)
_sig("parasail_matrix_from_file", _MP, C.c_char_p)
_sig("parasail_matrix_pssm_create", _MP, C.c_char_p, C.POINTER(C.c_int), C.c_int)
_sig("parasail_matrix_convert_square_to_pssm", _MP, _MP, C.c_char_p, C.c_int)
_sig("parasail_matrix_copy", _MP, _MP)
_sig("parasail_matrix_set_value", None, _MP, C.c_int, C.c_int, C.c_int)
_sig("parasail_matrix_free", None, _MP)
_sig("parasail_profile_free", None, C.c_void_p)
_sig("pmx_free", None, C.c_void_p)
_sig("pmx_last_error", C.c_char_p)
_sig("pmx_version", C.c_char_p)
_sig("pmx_device_count", C.c_int)
_sig("pmx_set_device", C.c_int, C.c_int)
_sig("pmx_kernel_for", C.c_char_p, C.POINTER(pmx_config_t), C.c_int32, C.c_int32)
_sig("pmx_last_kernel", C.c_char_p)
_sig("pmx_switches", C.c_char_p)
_sig("pmx_align_batch", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_void_p, C.c_void_p)
_sig("pmx_align_batch_device", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("pmx_align_profile_batch", C.c_int, C.POINTER(pmx_config_t), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_void_p)
_sig("pmx_align_profile_batch_device", C.c_int, C.POINTER(pmx_config_t), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("pmx_align_batch_cigar", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p)
_sig("pmx_align_batch_cigar_device", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
_sig("pmx_align_batch_banded", C.c_int, C.POINTER(pmx_config_t), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)
_sig("pmx_align_batch_banded_device", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("pmx_align_profile_batch_banded_device", C.c_int, C.POINTER(pmx_config_t), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("pmx_align_batch_multi", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
_sig("pmx_align_profile_batch_multi", C.c_int, C.POINTER(pmx_config_t), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
_sig("pmx_shard_bounds_by_cells", C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
_sig("pmx_align_batch_table_device", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("pmx_host_register", C.c_int, C.c_void_p, C.c_size_t)
_sig("pmx_host_unregister", C.c_int, C.c_void_p)
_sig("pmx_align_batch_2bit", C.c_int, C.POINTER(pmx_config_t), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_void_p, C.c_void_p, C.c_void_p)
_libc_free = C.CDLL(None).free
_libc_free.argtypes = [C.c_void_p]


# ------------------------------------------------------------------ errors ----
class Error(Exception):
    """src/error.rs:8-17"""


class PanicError(RuntimeError):
    """A Rust `panic!` / `assert!` in the reference."""


class InteriorNulByte(Error): pass          # src/aligner/error.rs:7
class NoBandwidth(Error): pass              # src/aligner/error.rs:8
class NoStats(Error): pass                  # src/alignment/error.rs:6-10
class NoTable(Error): pass
class NoStatsTable(Error): pass
class NoRowCol(Error): pass
class NoTrace(Error): pass
class FailedLookup(Error): pass             # src/matrix/mod.rs:65-67
class NullMatrix(Error): pass
class NotSquare(Error): pass
class NotBuiltIn(Error): pass
class InvalidIndex(Error): pass
class FileNotFound(Error): pass
class NullProfile(Error): pass              # src/profile/mod.rs:101-103
class QueryIsEmpty(Error): pass             # src/profile/mod.rs:299-301
class BatchError(Error): pass               # additive batch API


def _cstring(b):
    b = bytes(b)
    if b"\0" in b:
        raise InteriorNulByte("nul byte found in provided data")     # CString::new, src/aligner/mod.rs:399,:409
    return b


# ------------------------------------------------------------------ Matrix ----
class Matrix:
    """src/matrix/mod.rs:25-312"""

    def __init__(self, inner, builtin):
        self.inner = inner
        self.builtin = builtin

    @classmethod
    def create(cls, alphabet, match_score, mismatch_score):
        if not (match_score >= 0 and mismatch_score <= 0):
            raise PanicError("Match score should be a positive integer and mismatch score should be a negative integer.")
        if len(alphabet) == 0:
            raise PanicError("Alphabet should not be empty.")
        return cls(lib.parasail_matrix_create(_cstring(alphabet), match_score, mismatch_score), False)

    @classmethod
    def from_name(cls, matrix_name):
        """`Matrix::from(name)`, src/matrix/mod.rs:57-73"""
        if not matrix_name:
            raise PanicError("Matrix name should not be empty.")
        m = lib.parasail_matrix_lookup(_cstring(matrix_name.encode()))
        if not m:
            raise FailedLookup(matrix_name)
        return cls(m, True)

    @classmethod
    def from_file(cls, file):
        if not os.path.exists(file):
            raise FileNotFound(file)
        m = lib.parasail_matrix_from_file(_cstring(file.encode()))
        if not m:
            raise NullMatrix()
        return cls(m, False)

    @classmethod
    def create_pssm(cls, alphabet, values, rows):
        vals = (C.c_int * max(len(values), rows * len(alphabet)))(*values)
        m = lib.parasail_matrix_pssm_create(_cstring(alphabet.encode() if isinstance(alphabet, str) else alphabet),
                                            vals, rows)
        if not m:
            raise NullMatrix()
        return cls(m, False)

    def to_pssm(self, pssm_query):
        if len(pssm_query) == 0:
            raise PanicError("PSSM query sequence should not be empty.")
        if self.inner.contents.type_ != 0:
            raise NotSquare()
        m = lib.parasail_matrix_convert_square_to_pssm(self.inner, _cstring(pssm_query), len(pssm_query))
        if not m:
            raise NullMatrix()
        return Matrix(m, False)

    def set_value(self, row, col, value):
        if self.builtin:
            raise NotBuiltIn()
        size = self.inner.contents.size - 2
        if size < 0:
            raise NullMatrix()
        if row < 0 or row > size or col < 0 or col > size:
            raise InvalidIndex(row, col)
        lib.parasail_matrix_set_value(self.inner, row, col, value)

    @classmethod
    def default(cls):
        return cls.create(b"ACGTA", 1, -1)           # src/matrix/mod.rs:246-250

    def clone(self):
        return Matrix(lib.parasail_matrix_copy(self.inner), False)

    @property
    def size(self):
        return self.inner.contents.size

    @property
    def length(self):
        return self.inner.contents.length

    def to_numpy(self):
        c = self.inner.contents
        return np.ctypeslib.as_array(c.matrix, shape=(c.length, c.size)).copy()

    def mapper(self):
        return np.ctypeslib.as_array(self.inner.contents.mapper, shape=(256,)).copy()

    def __str__(self):                               # src/matrix/mod.rs:253-268
        return "".join(" ".join(str(v) for v in row) + " \n" for row in self.to_numpy())

    def __del__(self):
        try:
            if not self.builtin and self.inner:
                lib.parasail_matrix_free(self.inner)
        except Exception:
            pass


# ------------------------------------------------------------------ Profile ---
class SolutionWidth:
    Sat, Bit8, Bit16, Bit32, Bit64 = "sat", "8", "16", "32", "64"      # src/prelude.rs:9-15


class InstructionSet:
    Best, SSE2, SSE41, AVX2, AltiVec, Neon = "", "_sse_128", "_sse_128", "_avx_256", "_altivec_128", "_neon_128"


class Profile:
    """src/profile/mod.rs:281-395"""

    def __init__(self, inner, use_stats, query_len, matrix=None):
        self.inner = inner
        self.use_stats = use_stats
        self.query_len = query_len
        self._matrix = matrix          # keep the matrix alive (the C profile borrows it)

    @classmethod
    def new(cls, query_bytes, with_stats, matrix):
        if len(query_bytes) == 0:
            raise QueryIsEmpty()
        q = _cstring(query_bytes)
        name = "parasail_profile_create_stats_sat" if with_stats else "parasail_profile_create_sat"
        return cls._create(name, q, with_stats, matrix)

    @classmethod
    def _create(cls, name, q, with_stats, matrix):
        f = getattr(lib, name)
        f.restype = C.c_void_p
        f.argtypes = [C.c_char_p, C.c_int, _MP]
        p = f(q, len(q), matrix.inner)
        if not p:
            raise NullProfile()
        return cls(p, bool(with_stats), len(q), matrix)

    @classmethod
    def builder(cls, query, matrix):
        return ProfileBuilder(query, matrix)

    @classmethod
    def new_ssw(cls, query_bytes, matrix, score_size):
        if len(query_bytes) == 0:
            raise PanicError("Query sequence has length 0.")
        q = _cstring(query_bytes)
        p = lib.parasail_ssw_init(q, len(q), matrix.inner, score_size)
        if not p:
            raise NullProfile()
        return cls(p, True, len(q), matrix)

    @classmethod
    def default(cls):
        return cls(None, False, 0)                   # null profile = "no profile", src/profile/mod.rs:365-373

    def is_null(self):
        return not self.inner

    def __del__(self):
        try:
            if self.inner:
                lib.parasail_profile_free(self.inner)
        except Exception:
            pass


class ProfileBuilder:
    """src/profile/mod.rs:42-278"""

    def __init__(self, query, matrix):
        self.query, self.matrix = query, matrix
        self._stats, self._width, self._isa = False, SolutionWidth.Sat, InstructionSet.Best

    def use_stats(self):
        self._stats = True
        return self

    def solution_width(self, w):
        self._width = w
        return self

    def instruction_set(self, isa):
        self._isa = isa
        return self

    def build(self):
        name = "parasail_profile_create%s%s_%s" % ("_stats" if self._stats else "", self._isa, self._width)
        return Profile._create(name, _cstring(self.query), self._stats, self.matrix)


# ------------------------------------------------------------------ tables ----
class TraceFlags:
    """src/alignment/table.rs:127-142"""
    ZERO_MASK, E_MASK, F_MASK = 120, 103, 31
    ZERO, INS, DEL, DIAG, DIAG_E, INS_E, DIAG_F, DEL_F = 0, 1, 2, 4, 8, 16, 32, 64


class Table:
    """src/alignment/table.rs:33-108 (row-major [query_len][ref_len] int32 view)"""

    def __init__(self, data, rows, cols, owner):
        self.inner, self._rows, self._cols, self._owner = data, rows, cols, owner

    def get(self, row, col):
        if row < self._rows and col < self._cols:
            return int(self.inner[row * self._cols + col])
        return None

    def rows(self):
        return self._rows

    def cols(self):
        return self._cols

    def as_slice(self):
        return self.inner

    def last(self):
        return int(self.inner[len(self.inner) - 1])


class TracebackTable(Table):
    """src/alignment/table.rs:197-300 (1 byte per cell)"""

    def get(self, row, col):
        v = self.get_detailed(row, col)
        return None if v is None else v & (TraceFlags.DIAG | TraceFlags.INS | TraceFlags.DEL)

    def get_detailed(self, row, col):
        if row < self._rows and col < self._cols:
            return int(self.inner[row * self._cols + col]) & 127
        return None


class Traceback:
    def __init__(self, query, comparison, reference):
        self.query, self.comparison, self.reference = query, comparison, reference


# ------------------------------------------------------------------ Alignment --
class Alignment:
    """src/alignment/mod.rs:54-504"""

    def __init__(self, inner, matrix, query_len, ref_len):
        self.inner, self.matrix, self.query_len, self.ref_len = inner, matrix, query_len, ref_len

    def get_score(self): return lib.parasail_result_get_score(self.inner)
    def get_end_query(self): return lib.parasail_result_get_end_query(self.inner)
    def get_end_ref(self): return lib.parasail_result_get_end_ref(self.inner)

    def get_matches(self):
        if self.is_stats():
            return lib.parasail_result_get_matches(self.inner)
        raise NoStats("get_matches()")

    def get_similar(self):                       # no is_stats guard in the reference (:87-89)
        return lib.parasail_result_get_similar(self.inner)

    def get_length(self):
        if self.is_stats():
            return lib.parasail_result_get_length(self.inner)
        raise NoStats("get_length()")

    def _table(self, kind):
        p = getattr(lib, "parasail_result_get_%s_table" % kind)(self.inner)
        n = self.query_len * self.ref_len
        return Table(np.ctypeslib.as_array(p, shape=(n,)), self.query_len, self.ref_len, self)

    def get_score_table(self):
        if self.is_table() or self.is_stats_table():
            return self._table("score")
        raise NoTable("get_score_table()")

    def get_matches_table(self):
        if self.is_stats_table():
            return self._table("matches")
        raise NoStatsTable("get_matches_table()")

    def get_similar_table(self):
        if self.is_stats_table():
            return self._table("similar")
        raise NoStatsTable("get_similar_table()")

    def get_length_table(self):
        if self.is_stats_table():
            return self._table("length")
        raise NoStatsTable("get_length_table()")

    def _rowcol(self, kind, which, fname):
        plain_ok = kind == "score" and self.is_rowcol()
        if not (plain_ok or self.is_stats_rowcol()):
            raise NoRowCol(fname)
        p = getattr(lib, "parasail_result_get_%s_%s" % (kind, which))(self.inner)
        n = self.ref_len if which == "row" else self.query_len
        return np.ctypeslib.as_array(p, shape=(n,))

    def get_score_row(self): return self._rowcol("score", "row", "get_score_row()")
    def get_matches_row(self): return self._rowcol("matches", "row", "get_matches_row()")
    def get_similar_row(self): return self._rowcol("similar", "row", "get_similar_row()")
    def get_length_row(self): return self._rowcol("length", "row", "get_length_row()")
    def get_score_col(self): return self._rowcol("score", "col", "get_score_col()")
    def get_matches_col(self): return self._rowcol("matches", "col", "get_matches_col()")
    def get_similar_col(self): return self._rowcol("similar", "col", "get_similar_col()")
    def get_length_col(self): return self._rowcol("length", "col", "get_length_col()")

    def get_trace_table(self):
        if not self.is_trace():
            raise NoTrace("get_trace_table()")
        p = C.cast(lib.parasail_result_get_trace_table(self.inner), C.POINTER(C.c_int8))
        n = self.query_len * self.ref_len
        return TracebackTable(np.ctypeslib.as_array(p, shape=(n,)), self.query_len, self.ref_len, self)

    def print_traceback(self, query, reference):
        if self.is_trace():
            lib.parasail_traceback_generic(_cstring(query), len(query), _cstring(reference), len(reference),
                                           b"Query:", b"Target:", self.matrix.inner, self.inner,
                                           b"|", b" ", b" ", 80, 7, 1)
        else:
            print("Alignment string is not available without traceback enabled. "
                  "Consider using the `use_trace` method on AlignerBuilder.")

    def get_traceback_strings(self, query, reference):
        if not self.is_trace():
            raise NoTrace("get_traceback_strings()")
        tb = lib.parasail_result_get_traceback(self.inner, _cstring(query), len(query), _cstring(reference),
                                               len(reference), self.matrix.inner, b"|", b" ", b" ")
        if not tb:
            raise NoTrace("get_traceback_strings()")
        out = Traceback(*(C.string_at(getattr(tb.contents, f)).decode() for f in ("query", "comp", "ref_")))
        lib.parasail_traceback_free(tb)
        return out

    def get_cigar(self, query, reference):
        if not self.is_trace():
            raise NoTrace("get_cigar()")
        c = lib.parasail_result_get_cigar(self.inner, _cstring(query), len(query), _cstring(reference),
                                          len(reference), self.matrix.inner)
        if not c:
            raise NoTrace("get_cigar()")
        s = lib.parasail_cigar_decode(c)
        text = C.string_at(s).decode()
        _libc_free(s)                 # plain malloc block (Rust adopts it with CString::from_raw, :410)
        lib.parasail_cigar_free(c)
        return text

    def get_cigar_begin(self, query, reference):
        """(beg_query, beg_ref) of the walked alignment -- fields of parasail_cigar_t."""
        c = lib.parasail_result_get_cigar(self.inner, _cstring(query), len(query), _cstring(reference),
                                          len(reference), self.matrix.inner)
        out = (c.contents.beg_query, c.contents.beg_ref)
        lib.parasail_cigar_free(c)
        return out

    def is_global(self): return lib.parasail_result_is_nw(self.inner) != 0
    def is_semi_global(self): return lib.parasail_result_is_sg(self.inner) != 0
    def is_local(self): return lib.parasail_result_is_sw(self.inner) != 0
    def is_saturated(self): return lib.parasail_result_is_saturated(self.inner) != 0
    def is_banded(self): return lib.parasail_result_is_banded(self.inner) != 0
    def is_scan(self): return lib.parasail_result_is_scan(self.inner) != 0
    def is_striped(self): return lib.parasail_result_is_striped(self.inner) != 0
    def is_diag(self): return lib.parasail_result_is_diag(self.inner) != 0
    def is_blocked(self): return lib.parasail_result_is_blocked(self.inner) != 0
    def is_stats(self): return lib.parasail_result_is_stats(self.inner) != 0
    def is_stats_table(self): return lib.parasail_result_is_stats_table(self.inner) != 0
    def is_table(self): return lib.parasail_result_is_table(self.inner) != 0
    def is_rowcol(self): return lib.parasail_result_is_rowcol(self.inner) != 0
    def is_stats_rowcol(self): return lib.parasail_result_is_stats_rowcol(self.inner) != 0
    def is_trace(self): return lib.parasail_result_is_trace(self.inner) != 0

    def __del__(self):
        try:
            lib.parasail_result_free(self.inner)
        except Exception:
            pass


class SSWResult:
    """src/alignment/mod.rs:506-551"""

    def __init__(self, inner):
        self.inner = inner

    def score(self): return self.inner.contents.score1
    def ref_start(self): return self.inner.contents.ref_begin1
    def ref_end(self): return self.inner.contents.ref_end1
    def query_start(self): return self.inner.contents.read_begin1
    def query_end(self): return self.inner.contents.read_end1
    def cigar(self): return self.inner.contents.cigar
    def cigar_len(self): return self.inner.contents.cigarLen

    def __del__(self):
        try:
            lib.parasail_result_ssw_free(self.inner)
        except Exception:
            pass


# ------------------------------------------------------------------ Aligner ---
class AlignerBuilder:
    """src/aligner/mod.rs:67-370"""

    def __init__(self):
        self._mode = "nw"
        self._solution_width = "sat"
        self._matrix = Matrix.default()
        self._gap_open = 0                       # defaults are 0/0 (:92-93) although the docs say 5 and 2
        self._gap_extend = 0
        self._profile = Profile.default()
        self._allow_query_gaps = []
        self._allow_ref_gaps = []
        self._vec_strategy = "_striped"
        self._use_stats = ""
        self._use_table = ""
        self._use_trace = ""
        self._bandwidth = None

    def global_(self): self._mode = "nw"; return self
    def semi_global(self): self._mode = "sg"; return self
    def local(self): self._mode = "sw"; return self
    def solution_width(self, w): self._solution_width = str(int(w)); return self
    def matrix(self, m): self._matrix = m; return self
    def gap_open(self, v): self._gap_open = v; return self
    def gap_extend(self, v): self._gap_extend = v; return self
    def profile(self, p): self._profile = p; return self
    def allow_query_gaps(self, g): self._allow_query_gaps = list(g); return self
    def allow_ref_gaps(self, g): self._allow_ref_gaps = list(g); return self
    def striped(self): self._vec_strategy = "_striped"; return self
    def scan(self): self._vec_strategy = "_scan"; return self
    def diag(self): self._vec_strategy = "_diag"; return self

    def use_stats(self):                         # :213-223
        self._use_stats = "_stats"
        self._use_trace = ""
        return self

    def use_table(self):                         # :228-237
        self._use_table = "_table"
        self._use_trace = ""
        return self

    def use_last_rowcol(self):                   # :243-246 (does not clear trace)
        self._use_table = "_rowcol"
        return self

    def use_trace(self):                         # :251-267
        self._use_trace = "_trace"
        self._use_table = ""
        self._use_stats = ""
        return self

    def bandwidth(self, k): self._bandwidth = k; return self

    @staticmethod
    def _allowed_gaps(prefix, gaps):             # :270-286
        if gaps:
            if "prefix" in gaps and "suffix" in gaps:
                return "_%sx" % prefix
            if "prefix" in gaps:
                return "_%sb" % prefix
            if "suffix" in gaps:
                return "_%se" % prefix
        return ""

    def get_parasail_fn_name(self):              # :289-331
        sg = ""
        if self._mode == "sg":
            sg = self._allowed_gaps("q", self._allow_query_gaps) + self._allowed_gaps("d", self._allow_ref_gaps)
            if sg == "_qx_dx":
                sg = ""
        if self._profile.is_null():
            profile, stats = "", self._use_stats
        else:
            if self._vec_strategy not in ("_striped", "_scan"):
                raise PanicError("Vectorization strategy must be striped or scan for alignment with a profile.")
            profile = "_profile"
            stats = "_stats" if self._profile.use_stats else ""
        return "%s%s%s%s%s%s%s_%s" % (self._mode, sg, self._use_trace, stats, self._use_table,
                                      self._vec_strategy, profile, self._solution_width)

    def build(self):                             # :339-369
        name = self.get_parasail_fn_name()
        if self._profile.is_null():
            f = lib.parasail_lookup_function(name.encode())
            fn = _FN(f) if f else None
        else:
            f = lib.parasail_lookup_pfunction(name.encode())
            fn = _PFN(f) if f else None
        if fn is None:
            raise PanicError("Parasail function: %s, not found." % name)
        return Aligner(fn, name, self._matrix, self._gap_open, self._gap_extend, self._profile,
                       self._vec_strategy, self._bandwidth)


class Aligner:
    """src/aligner/mod.rs:372-535"""

    def __init__(self, fn, fn_name, matrix, gap_open, gap_extend, profile, vec_strategy, bandwidth):
        self._fn, self.fn_name = fn, fn_name
        self.matrix, self.gap_open, self.gap_extend = matrix, gap_open, gap_extend
        self._profile, self.vec_strategy, self._bandwidth = profile, vec_strategy, bandwidth

    @staticmethod
    def new():
        return AlignerBuilder()

    def clone(self):
        return Aligner(self._fn, self.fn_name, self.matrix, self.gap_open, self.gap_extend, self._profile,
                       self.vec_strategy, self._bandwidth)

    def align(self, query, reference):           # :397-452
        ref_len = len(reference)
        reference = _cstring(reference)
        if self._profile.is_null():
            if query is None:
                raise PanicError("Query sequence is required for alignment without a profile.")
            query_len = len(query)
            q = _cstring(query)
            res = self._fn(q, query_len, reference, ref_len, self.gap_open, self.gap_extend, self.matrix.inner)
            return Alignment(res, self.matrix, query_len, ref_len)
        res = self._fn(self._profile.inner, reference, ref_len, self.gap_open, self.gap_extend)
        return Alignment(res, self.matrix, self._profile.query_len, ref_len)

    def banded_nw(self, query, reference):       # :457-489
        ref_len, query_len = len(reference), len(query)
        reference, q = _cstring(reference), _cstring(query)
        if self._bandwidth is None:
            raise NoBandwidth()
        res = lib.parasail_nw_banded(q, query_len, reference, ref_len, self.gap_open, self.gap_extend,
                                     self._bandwidth, self.matrix.inner)
        return Alignment(res, self.matrix, query_len, ref_len)

    def ssw(self, query, reference):             # :492-529
        ref_len = len(reference)
        reference = _cstring(reference)
        if query is None:
            raise PanicError("Query sequence is required for SSW alignment for now.")
        q = _cstring(query)
        return SSWResult(lib.parasail_ssw(q, len(q), reference, ref_len, self.gap_open, self.gap_extend,
                                          self.matrix.inner))

    # ---- additive batch interface (no reference counterpart) -----------------------------
    def _config(self, want=0):
        name = self.fn_name
        mode = {"nw": MODE_NW, "sg": MODE_SG, "sw": MODE_SW}[name[:2]]
        flags = 0
        if mode == MODE_SG:
            head = name.split("_striped")[0].split("_scan")[0].split("_diag")[0]
            q = [t for t in ("_qb", "_qe", "_qx") if t in head]
            d = [t for t in ("_db", "_de", "_dx") if t in head]
            if not q and not d:
                flags = SG_ALL
            else:
                for t in q + d:
                    flags |= {"_qb": SG_QB, "_qe": SG_QE, "_qx": SG_QB | SG_QE,
                              "_db": SG_DB, "_de": SG_DE, "_dx": SG_DB | SG_DE}[t]
        width = name.rsplit("_", 1)[1]
        if "_stats" in name:
            want |= WANT_STATS
        cfg = pmx_config_t(mode, flags, self.gap_open, self.gap_extend, 0 if width == "sat" else int(width),
                           want, self.matrix.inner)
        return cfg

    def align_batch(self, queries, references):
        """Many independent pairs in one call.  Returns a structured array with fields
        score, end_query, end_ref, flags (and, for a stats aligner, a second array with
        matches, similar, length)."""
        qbuf, qoff = pack(queries)
        rbuf, roff = pack(references)
        return self.align_batch_packed(qbuf, qoff, rbuf, roff)

    def align_batch_packed(self, qbuf, qoff, rbuf, roff, out=None):
        """`out`: a RECORD_DTYPE array of n records to fill (a caller that aligns batch after batch reuses one, already paged in)."""
        n = len(roff) - 1
        cfg = self._config()
        out = _record_buffer(out, n)
        stats = np.zeros(n, dtype=STATS_DTYPE) if cfg.want & WANT_STATS else None
        if self._profile.is_null():
            if len(qoff) - 1 != n:
                raise BatchError("queries and references differ in count")
            rc = lib.pmx_align_batch(C.byref(cfg), n, qbuf.ctypes.data, qoff.ctypes.data, rbuf.ctypes.data,
                                     roff.ctypes.data, out.ctypes.data, stats.ctypes.data if stats is not None else None)
        else:
            rc = lib.pmx_align_profile_batch(C.byref(cfg), self._profile.inner, n, rbuf.ctypes.data,
                                             roff.ctypes.data, out.ctypes.data,
                                             stats.ctypes.data if stats is not None else None)
        if rc:
            raise BatchError(lib.pmx_last_error().decode())
        return (out, stats) if stats is not None else out

    def align_batch_2bit(self, q2, qoff, r2, roff, out=None):
        """2-bit packed input (see pack_2bit): offsets count bases.  `out` as in align_batch_packed."""
        n = len(roff) - 1
        cfg = self._config()
        out = _record_buffer(out, n)
        stats = np.zeros(n, dtype=STATS_DTYPE) if cfg.want & WANT_STATS else None
        rc = lib.pmx_align_batch_2bit(C.byref(cfg), n, q2.ctypes.data, qoff.ctypes.data, r2.ctypes.data, roff.ctypes.data,
                                      out.ctypes.data, stats.ctypes.data if stats is not None else None)
        if rc:
            raise BatchError(lib.pmx_last_error().decode())
        return (out, stats) if stats is not None else out

    def align_batch_multi(self, qbuf, qoff, rbuf, roff, devices):
        """One batch across several GPUs of the node (cell-balanced contiguous blocks, records in input order)."""
        n = len(roff) - 1
        cfg = self._config()
        out = np.zeros(n, dtype=RECORD_DTYPE)
        stats = np.zeros(n, dtype=STATS_DTYPE) if cfg.want & WANT_STATS else None
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        if self._profile.is_null():
            rc = lib.pmx_align_batch_multi(C.byref(cfg), n, qbuf.ctypes.data, qoff.ctypes.data, rbuf.ctypes.data, roff.ctypes.data,
                                           dev.ctypes.data, len(dev), out.ctypes.data, stats.ctypes.data if stats is not None else None)
        else:
            rc = lib.pmx_align_profile_batch_multi(C.byref(cfg), self._profile.inner, n, rbuf.ctypes.data, roff.ctypes.data,
                                                   dev.ctypes.data, len(dev), out.ctypes.data,
                                                   stats.ctypes.data if stats is not None else None)
        if rc:
            raise BatchError(lib.pmx_last_error().decode())
        return (out, stats) if stats is not None else out

    def align_batch_banded(self, queries, references, band, diag=None):
        """Banded batch (extension): cells with |(j - i) - diag[k]| > band are excluded; score and end positions."""
        rbuf, roff = pack(references)
        n = len(roff) - 1
        cfg = self._config()
        cfg.want = 0
        out = np.zeros(n, dtype=RECORD_DTYPE)
        d = None if diag is None else np.ascontiguousarray(diag, dtype=np.int32)
        if self._profile.is_null():
            qbuf, qoff = pack(queries)
            rc = lib.pmx_align_batch_banded(C.byref(cfg), None, n, qbuf.ctypes.data, qoff.ctypes.data, rbuf.ctypes.data,
                                            roff.ctypes.data, int(band), d.ctypes.data if d is not None else None, out.ctypes.data)
        else:
            rc = lib.pmx_align_batch_banded(C.byref(cfg), self._profile.inner, n, None, None, rbuf.ctypes.data,
                                            roff.ctypes.data, int(band), d.ctypes.data if d is not None else None, out.ctypes.data)
        if rc:
            raise BatchError(lib.pmx_last_error().decode())
        return out

    def align_batch_cigar(self, queries, references):
        qbuf, qoff = pack(queries)
        rbuf, roff = pack(references)
        out, text, coff = self.align_batch_cigar_packed(qbuf, qoff, rbuf, roff)
        raw = text.tobytes()
        return out, [raw[coff[k]:coff[k + 1]].decode() for k in range(len(roff) - 1)]

    def align_batch_cigar_packed(self, qbuf, qoff, rbuf, roff, out=None, coff=None):
        """Packed in, packed out: (records, CIGAR text as one uint8 array, int64 offsets[n+1]).  `out` / `coff`: arrays to fill
        (a caller that aligns batch after batch reuses them; the text block is recycled by the library once its array is gone)."""
        n = len(roff) - 1
        cfg = self._config()
        cfg.want &= ~WANT_STATS
        out = _record_buffer(out, n)
        if coff is None:
            coff = np.zeros(n + 1, dtype=np.int64)
        elif coff.dtype != np.int64 or coff.shape != (n + 1,) or not coff.flags.c_contiguous:
            raise BatchError("coff must be a contiguous int64 array of %d offsets" % (n + 1))
        cbuf = C.c_void_p()
        rc = lib.pmx_align_batch_cigar(C.byref(cfg), n, qbuf.ctypes.data, qoff.ctypes.data, rbuf.ctypes.data,
                                       roff.ctypes.data, out.ctypes.data, C.byref(cbuf), coff.ctypes.data)
        if rc:
            raise BatchError(lib.pmx_last_error().decode())
        # a view of the callee's malloc block (no copy); released with pmx_free when the array goes away
        nbytes = int(coff[n])
        # The owner hangs on the ctypes array, the ULTIMATE base of every numpy view (numpy collapses the base chain of
        # derived arrays down to it: np.asarray(text), text.view(np.ndarray) and slices all keep the block alive).
        raw = (C.c_ubyte * max(nbytes, 1)).from_address(cbuf.value)
        raw._pmx_owner = _OwnedBuffer(cbuf)
        text = np.frombuffer(raw, dtype=np.uint8, count=nbytes)
        return out, text, coff


class _OwnedBuffer:
    """Keeps a callee-allocated block alive for numpy views of it; pmx_free on collection."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                lib.pmx_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pack(seqs):
    """list of bytes -> (uint8 buffer, int64 offsets[n+1]) in the layout of include/parasail_amd.h."""
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    buf = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8).copy()
    if len(buf) == 0:
        buf = np.zeros(1, dtype=np.uint8)
    return buf, off


def align_batch_device(cfg, n, d_qbuf, d_qoff, d_rbuf, d_roff, max_qlen, max_rlen, d_out, d_stats=None, stream=0):
    """Device-pointer entry (ints are raw device addresses, `stream` a hipStream_t value)."""
    rc = lib.pmx_align_batch_device(C.byref(cfg), n, d_qbuf, d_qoff, d_rbuf, d_roff, max_qlen, max_rlen,
                                    d_out, d_stats, stream)
    if rc:
        raise BatchError(lib.pmx_last_error().decode())


def align_batch_cigar_device(cfg, n, d_qbuf, d_qoff, d_rbuf, d_roff, max_qlen, max_rlen, d_out, d_text, capacity, d_text_off, stream=0):
    """Device-pointer CIGAR entry: records + CIGAR text + n+1 text offsets, all in device memory."""
    rc = lib.pmx_align_batch_cigar_device(C.byref(cfg), n, d_qbuf, d_qoff, d_rbuf, d_roff, max_qlen, max_rlen,
                                          d_out, d_text, capacity, d_text_off, stream)
    if rc:
        raise BatchError(lib.pmx_last_error().decode())


def align_profile_batch_device(cfg, profile, n, d_rbuf, d_roff, max_rlen, d_out, d_stats=None, stream=0):
    """One reused query profile against device-resident references."""
    rc = lib.pmx_align_profile_batch_device(C.byref(cfg), profile.inner, n, d_rbuf, d_roff, max_rlen, d_out, d_stats, stream)
    if rc:
        raise BatchError(lib.pmx_last_error().decode())


def pack_2bit(buf, alphabet=b"ACGT"):
    """ASCII letters -> 2 bits per base (base b in byte b / 4 at bits 2 (b % 4)), the input form of pmx_align_batch_2bit."""
    lut = np.zeros(256, dtype=np.uint8)
    for i, ch in enumerate(alphabet[:4]):
        lut[ch] = i; lut[ord(chr(ch).lower())] = i
    codes = lut[buf]
    pad = (-len(codes)) % 4
    if pad:
        codes = np.concatenate([codes, np.zeros(pad, dtype=np.uint8)])
    c = codes.reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def _record_buffer(out, n):
    if out is None:
        return np.zeros(n, dtype=RECORD_DTYPE)
    if out.dtype != RECORD_DTYPE or out.shape != (n,) or not out.flags.c_contiguous:
        raise BatchError("out must be a contiguous RECORD_DTYPE array of %d records" % n)
    return out


def switches():
    """[(name, kind, what)] for every environment switch the library reads (csrc/pmx_switches.h)."""
    return [tuple(line.split("\t")) for line in lib.pmx_switches().decode().splitlines()]


def host_register(*arrays):
    """Page-lock numpy arrays handed to the host-buffer batch entries (full PCIe rate); undo with host_unregister."""
    for a in arrays:
        if lib.pmx_host_register(a.ctypes.data, a.nbytes):
            raise BatchError(lib.pmx_last_error().decode())


def host_unregister(*arrays):
    for a in arrays:
        lib.pmx_host_unregister(a.ctypes.data)


def shard_bounds_by_cells(qoff, roff, parts):
    """The C planner behind pmx_align_batch_multi (qoff None: one shared query)."""
    roff = np.ascontiguousarray(roff, dtype=np.int64)
    n = len(roff) - 1
    b = np.zeros(parts + 1, dtype=np.int64)
    q = None if qoff is None else np.ascontiguousarray(qoff, dtype=np.int64)
    if lib.pmx_shard_bounds_by_cells(n, q.ctypes.data if q is not None else None, roff.ctypes.data, parts, b.ctypes.data):
        raise BatchError("planner failed")
    return [int(x) for x in b]


def version():
    return lib.pmx_version().decode()
