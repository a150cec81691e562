"""Opportunistic binding of a SYSTEM libparasail (the reference's real arithmetic, jeffdaily/parasail via
libparasail-sys).  Test infrastructure only, like everything under oracle/.  The library
is not in the build image and cannot be installed there; if a box ever has one, the GPU tests cross-check against
it and bench.py times it as the "reference" CPU baseline.  `load()` returns None when it is absent -- callers then
say "parasail oracle unavailable" and fall back to the in-repo restatement."""
import ctypes as C
import ctypes.util
import os


def load():
    cand = [os.environ.get("PMX_SYSTEM_PARASAIL"), ctypes.util.find_library("parasail"), "libparasail.so", "libparasail.so.3", "libparasail.so.8"]
    for name in cand:
        if not name:
            continue
        try:
            lib = C.CDLL(name)
        except OSError:
            continue
        if hasattr(lib, "pmx_version"):          # our own drop-in under the reference's name: not an oracle
            continue
        try:
            lib.parasail_matrix_create.restype = C.c_void_p
            lib.parasail_matrix_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
            lib.parasail_matrix_free.argtypes = [C.c_void_p]
            lib.parasail_lookup_function.restype = C.c_void_p
            lib.parasail_lookup_function.argtypes = [C.c_char_p]
            for f in ("score", "end_query", "end_ref"):
                g = getattr(lib, "parasail_result_get_" + f)
                g.restype = C.c_int
                g.argtypes = [C.c_void_p]
            lib.parasail_result_free.argtypes = [C.c_void_p]
        except AttributeError:
            continue
        return lib
    return None


def align_batch(lib, fn_name, qs, rs, open_, ext, alphabet, match, mismatch):
    """[(score, end_query, end_ref)] through the system library's `fn_name` (e.g. b"sw_striped_16")."""
    proto = C.CFUNCTYPE(C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p)
    ptr = lib.parasail_lookup_function(fn_name)
    if not ptr:
        raise RuntimeError("system parasail has no function %r" % fn_name)
    fn = proto(ptr)
    m = lib.parasail_matrix_create(alphabet, match, mismatch)
    out = []
    for q, r in zip(qs, rs):
        res = fn(q, len(q), r, len(r), open_, ext, m)
        out.append((lib.parasail_result_get_score(res), lib.parasail_result_get_end_query(res), lib.parasail_result_get_end_ref(res)))
        lib.parasail_result_free(res)
    lib.parasail_matrix_free(m)
    return out
