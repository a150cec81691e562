"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol that
include/parasail_amd.h declares, resolves exactly the dispatch names of the reference's name
grammar (src/aligner/mod.rs:289-331) and implements the host-only Matrix/Profile/builder logic.
No kernel is launched here."""
import ctypes as C
import itertools
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "parasail_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b((?:parasail|pmx)_[a-z0-9_]+)\s*\(", text))
    names -= {"parasail_function_t", "parasail_pfunction_t", "parasail_pcreator_t"}
    # macro-declared profile creators
    for isa in ("", "_sse_128", "_avx_256", "_neon_128", "_altivec_128"):
        for st in ("", "_stats"):
            for w in ("sat", "8", "16", "32", "64"):
                names.add("parasail_profile_create%s%s_%s" % (st, isa, w))
    return {n for n in names if "##" not in n and not n.endswith("_")}


def test_library_exports_every_declared_symbol(pkg):
    syms = _declared_symbols()
    assert len(syms) > 110
    missing = [s for s in sorted(syms) if not hasattr(pkg.lib, s)]
    assert not missing, missing


def test_symbols_the_reference_binds(pkg):
    # spot list taken from the `use libparasail_sys::{...}` blocks of the reference
    for s in ["parasail_lookup_function", "parasail_lookup_pfunction", "parasail_nw_banded", "parasail_ssw",
              "parasail_cigar_decode", "parasail_cigar_free", "parasail_result_free", "parasail_result_get_cigar",
              "parasail_result_get_traceback", "parasail_traceback_generic", "parasail_result_ssw_free",
              "parasail_matrix_convert_square_to_pssm", "parasail_matrix_copy", "parasail_matrix_create",
              "parasail_matrix_free", "parasail_matrix_from_file", "parasail_matrix_lookup",
              "parasail_matrix_pssm_create", "parasail_matrix_set_value", "parasail_profile_free",
              "parasail_ssw_init", "parasail_profile_create_stats_avx_256_sat", "parasail_profile_create_neon_128_64"]:
        assert hasattr(pkg.lib, s), s


def test_name_grammar_resolves(pkg):
    modes = ["nw", "sw", "sg", "sg_qb", "sg_qe", "sg_qx", "sg_db", "sg_de", "sg_dx", "sg_qb_de", "sg_qe_db",
             "sg_qb_db", "sg_qe_de", "sg_qx_db", "sg_qb_dx"]
    outs = ["", "_trace", "_stats", "_table", "_rowcol", "_stats_table", "_stats_rowcol"]
    n = 0
    for m, o, v, w in itertools.product(modes, outs, ["_striped", "_scan", "_diag"], ["sat", "8", "16", "32", "64"]):
        name = "%s%s%s_%s" % (m, o, v, w)
        assert pkg.lib.parasail_lookup_function(name.encode()), name
        assert not pkg.lib.parasail_lookup_pfunction(name.encode()), name
        if v != "_diag":
            pname = "%s%s%s_profile_%s" % (m, o, v, w)
            assert pkg.lib.parasail_lookup_pfunction(pname.encode()), pname
            assert not pkg.lib.parasail_lookup_function(pname.encode()), pname
        n += 1
    assert n == 15 * 7 * 3 * 5
    assert pkg.lib.parasail_lookup_function(b"parasail_sw_striped_16")
    f1 = pkg.lib.parasail_lookup_function(b"sw_striped_16")
    f2 = pkg.lib.parasail_lookup_function(b"sw_striped_sat")
    assert f1 != f2


@pytest.mark.parametrize("bad", ["nw_trace_rowcol_striped_sat", "nw_striped_7", "nw_striped_", "xx_striped_sat",
                                 "nw_blocked_sat", "nw_trace_stats_striped_sat", "nw_diag_profile_sat", "sw_striped"])
def test_unknown_names_do_not_resolve(pkg, bad):
    assert not pkg.lib.parasail_lookup_function(bad.encode())
    assert not pkg.lib.parasail_lookup_pfunction(bad.encode())


def test_builder_name_grammar_and_exclusions(pkg):
    A = pkg.Aligner
    assert A.new().get_parasail_fn_name() == "nw_striped_sat"                       # defaults :86-104
    assert A.new().local().solution_width(16).get_parasail_fn_name() == "sw_striped_16"
    assert A.new().semi_global().allow_query_gaps(["prefix", "suffix"]).allow_ref_gaps(["prefix", "suffix"]) \
        .get_parasail_fn_name() == "sg_striped_sat"                                  # _qx_dx collapses :296-298
    assert A.new().semi_global().allow_query_gaps(["suffix"]).allow_ref_gaps(["prefix"]) \
        .get_parasail_fn_name() == "sg_qe_db_striped_sat"
    assert A.new().use_trace().use_stats().get_parasail_fn_name() == "nw_stats_striped_sat"   # stats clears trace
    assert A.new().use_stats().use_table().use_trace().get_parasail_fn_name() == "nw_trace_striped_sat"
    assert A.new().use_table().use_last_rowcol().scan().get_parasail_fn_name() == "nw_rowcol_scan_sat"
    with pytest.raises(pkg.PanicError):
        A.new().use_trace().use_last_rowcol().build()                                # :243-246 + :353-358
    with pytest.raises(pkg.PanicError):
        A.new().solution_width(7).build()
    # aligner_construction, tests/test_parasail.rs:48-62: null profile + use_stats builds
    b = A.new().matrix(pkg.Matrix.default()).gap_open(10).gap_extend(1).profile(pkg.Profile.default()) \
        .allow_query_gaps(["prefix", "suffix"]).striped().use_stats()
    assert b.get_parasail_fn_name() == "nw_stats_striped_sat"
    al = b.build()
    assert (al.gap_open, al.gap_extend, al.vec_strategy) == (10, 1, "_striped")
    # profile decides the stats slot (:301-317); profile + diag panics (:307-310)
    m = pkg.Matrix.default()
    p = pkg.Profile.new(b"ACGT", True, m)
    assert A.new().profile(p).get_parasail_fn_name() == "nw_stats_striped_profile_sat"
    p2 = pkg.Profile.new(b"ACGT", False, m)
    assert A.new().profile(p2).use_stats().get_parasail_fn_name() == "nw_striped_profile_sat"
    with pytest.raises(pkg.PanicError):
        A.new().profile(p2).diag().build()


def test_matrix_construction(pkg):
    """tests/test_parasail.rs:5-34 (matrix_construction)"""
    pkg.Matrix.default()
    m = pkg.Matrix.create(b"ACGT", 3, -2)
    m.set_value(2, 2, 100)
    assert m.to_numpy()[2, 2] == 100
    b = pkg.Matrix.from_name("blosum62")
    b.to_pssm(b"ACGT")
    sq = pkg.Matrix.from_file(os.path.join(ROOT, "tests/golden/square.txt"))
    assert (sq.size, sq.length, sq.inner.contents.type_) == (17, 17, 0)
    ps = pkg.Matrix.from_file(os.path.join(ROOT, "tests/golden/pssm.txt"))
    assert (ps.length, ps.inner.contents.type_) == (10, 1)
    assert ps.to_numpy()[2, 9] == 12 and ps.inner.contents.query == b"YSCDGCLKPI"
    pkg.Matrix.create_pssm("abcdef", [1, 2, 3, 4, 5, 6, 7, 8], 2)


def test_matrix_semantics(pkg, orc):
    d = pkg.Matrix.default()
    c = d.inner.contents
    assert (c.size, c.length, c.type_, c.max, c.min) == (6, 6, 0, 1, -1)
    om = orc.Matrix.default()
    assert (d.to_numpy() == om.scores).all() and (d.mapper() == om.mapper).all()
    with pytest.raises(pkg.NotBuiltIn):
        pkg.Matrix.from_name("blosum62").set_value(0, 0, 1)
    m = pkg.Matrix.create(b"ACGT", 3, -2)
    for bad in [(-1, 0), (0, 4), (4, 0)]:
        with pytest.raises(pkg.InvalidIndex):                 # valid range 0..=size-2, src/matrix/mod.rs:228-236
            m.set_value(bad[0], bad[1], 1)
    m.set_value(3, 3, 9)
    cl = m.clone()
    m.set_value(0, 0, -7)
    assert cl.to_numpy()[0, 0] == 3 and cl.to_numpy()[3, 3] == 9
    with pytest.raises(pkg.FailedLookup):
        pkg.Matrix.from_name("nosuchmatrix")
    with pytest.raises(pkg.FileNotFound):
        pkg.Matrix.from_file("/nonexistent/matrix.txt")
    with pytest.raises(pkg.PanicError):
        pkg.Matrix.create(b"ACGT", -1, -2)
    with pytest.raises(pkg.InteriorNulByte):
        pkg.Matrix.create(b"AC\0T", 1, -1)
    with pytest.raises(pkg.NotSquare):
        pkg.Matrix.from_name("blosum62").to_pssm(b"ACGT").to_pssm(b"AC")


def test_blosum62_matches_fixture(pkg, orc):
    b = pkg.Matrix.from_name("blosum62")
    f = orc.Matrix.from_file(os.path.join(ROOT, "tests/golden/blosum62.txt"))
    assert (b.to_numpy() == f.scores).all()
    assert (b.mapper() == f.mapper).all()
    ff = pkg.Matrix.from_file(os.path.join(ROOT, "tests/golden/blosum62.txt"))
    assert (ff.to_numpy() == f.scores).all() and (ff.mapper() == f.mapper).all()


def test_nuc44_matches_reference_fixture(pkg, orc):
    """built-in "nuc44" = the reference's tests/square.txt without its extra U row/column"""
    n = pkg.Matrix.from_name("nuc44")
    sq = orc.Matrix.from_file(os.path.join(ROOT, "tests/golden/square.txt"))
    keep = [i for i, c in enumerate(sq.alphabet) if c != "U"]
    assert (n.to_numpy() == sq.scores[np.ix_(keep, keep)]).all()
    assert n.size == 16 and n.mapper()[ord("a")] == 0 and n.mapper()[ord("N")] == 14 and n.mapper()[ord("U")] == 15
    with pytest.raises(pkg.NotBuiltIn):
        n.set_value(0, 0, 1)


def test_profile_construction(pkg):
    """tests/test_parasail.rs:36-45, :758-765 and src/profile/mod.rs:298-358"""
    q = b"ATGGCACTATAA"
    m = pkg.Matrix.default()
    p = pkg.Profile.new(q, False, m)
    ps = pkg.Profile.new(q, True, m)
    assert (p.use_stats, ps.use_stats, p.query_len) == (False, True, 12)
    with pytest.raises(pkg.QueryIsEmpty):
        pkg.Profile.new(b"", False, m)
    with pytest.raises(pkg.PanicError):
        pkg.Profile.new_ssw(b"", m, 2)
    assert pkg.Profile.new_ssw(b"ACGT", m, 2).use_stats is True
    assert pkg.Profile.default().is_null()
    for isa in (pkg.InstructionSet.Best, pkg.InstructionSet.SSE2, pkg.InstructionSet.SSE41, pkg.InstructionSet.AVX2,
                pkg.InstructionSet.AltiVec, pkg.InstructionSet.Neon):
        for w in (pkg.SolutionWidth.Sat, pkg.SolutionWidth.Bit8, pkg.SolutionWidth.Bit16, pkg.SolutionWidth.Bit32,
                  pkg.SolutionWidth.Bit64):
            pkg.Profile.builder(q, m).instruction_set(isa).solution_width(w).build()
            pkg.Profile.builder(q, m).use_stats().instruction_set(isa).solution_width(w).build()


def test_error_paths_without_gpu(pkg):
    al = pkg.Aligner.new().build()
    with pytest.raises(pkg.InteriorNulByte):
        al.align(b"AC\0GT", b"ACGT")                     # src/aligner/mod.rs:399,:409
    with pytest.raises(pkg.InteriorNulByte):
        al.align(b"ACGT", b"A\0")
    with pytest.raises(pkg.PanicError):
        al.align(None, b"ACGT")                          # :403-406
    with pytest.raises(pkg.NoBandwidth):
        al.banded_nw(b"ACGT", b"ACGT")                   # :464-468
    with pytest.raises(pkg.PanicError):
        al.ssw(None, b"ACGT")                            # :512
    cfg = pkg.pmx_config_t(5, 0, 1, 1, 0, 0, al.matrix.inner)
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 10, 10) == b"invalid"
    cfg = pkg.pmx_config_t(pkg.MODE_SW, 0, 5, 2, 16, 0, al.matrix.inner)
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 150, 150) == b"pmx_sw16_kernel"
    cfg = pkg.pmx_config_t(pkg.MODE_NW, 0, 5, 2, 16, 0, al.matrix.inner)
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 150, 150) == b"pmx_nwsg16_kernel"
    cfg = pkg.pmx_config_t(pkg.MODE_SG, pkg.SG_ALL, 11, 1, 16, pkg.WANT_STATS, al.matrix.inner)
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 300, 5000) == b"pmx_stats16_kernel"
    cfg = pkg.pmx_config_t(pkg.MODE_NW, 0, 5, 2, 8, 0, al.matrix.inner)           # width 8: the int16 kernel tracks the range
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 150, 150) == b"pmx_nwsg16_kernel"
    cfg = pkg.pmx_config_t(pkg.MODE_NW, 0, 2, 5, 32, 0, al.matrix.inner)           # open < extend -> the 32-bit band kernel
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 150, 150) == b"pmx_long32_kernel"
    cfg = pkg.pmx_config_t(pkg.MODE_NW, 0, 2, 5, 16, pkg.WANT_STATS, al.matrix.inner)   # ... with statistics: general kernel
    assert pkg.lib.pmx_kernel_for(C.byref(cfg), 150, 150) == b"pmx_general_kernel"


def test_matrix_lookup_resolves_documented_names_from_a_directory(pkg, tmp_path, monkeypatch):
    """src/matrix/mod.rs:46-50 documents blosum{30..100} and pam{10..500}; embedded here are blosum62 and nuc44, every other name
    resolves from $PMX_MATRIX_DIR (NCBI-format files through the loader) and then behaves like a built-in.  The file used is
    tests/golden/blosum62.txt under other names: the mechanism is what is tested, not table values."""
    import shutil
    src = os.path.join(ROOT, "tests", "golden", "blosum62.txt")
    shutil.copy(src, tmp_path / "blosum50")
    shutil.copy(src, tmp_path / "pam250.txt")
    with pytest.raises(pkg.FailedLookup):
        pkg.Matrix.from_name("blosum50")                      # no directory configured
    monkeypatch.setenv("PMX_MATRIX_DIR", str(tmp_path))
    b62 = pkg.Matrix.from_name("blosum62")
    for name in ("blosum50", "BLOSUM50", "pam250"):
        m = pkg.Matrix.from_name(name)
        assert m.size == 24 and (m.to_numpy() == b62.to_numpy()).all()
        with pytest.raises(pkg.NotBuiltIn):                   # built-in semantics: set_value refused (src/matrix/mod.rs:222-239)
            m.set_value(0, 0, 5)
    a, b = pkg.Matrix.from_name("blosum50"), pkg.Matrix.from_name("blosum50")
    assert C.addressof(a.inner.contents) == C.addressof(b.inner.contents)       # cached: one table per name
    for bad in ("blosum45", "../blosum50", "/etc/passwd", ""):
        with pytest.raises((pkg.FailedLookup, pkg.PanicError)):
            pkg.Matrix.from_name(bad)


def test_matrix_lookup_default_directory_and_aliases(pkg, tmp_path, monkeypatch):
    """Without $PMX_MATRIX_DIR the library looks in <libdir>/../matrices (shipped with the package; nothing is written there by
    this test); `dnafull` is the NUC.4.4 table under its EMBOSS name; a miss is NOT remembered -- a file dropped into the
    directory after a failed lookup is found by the same process (matrices/README.md promises names resolve with no rebuild)."""
    import shutil
    monkeypatch.delenv("PMX_MATRIX_DIR", raising=False)
    assert (pkg.Matrix.from_name("dnafull").to_numpy() == pkg.Matrix.from_name("nuc44").to_numpy()).all()
    assert os.path.isdir(os.path.join(ROOT, "parasail-rs_amd", "matrices"))
    with pytest.raises(pkg.FailedLookup):
        pkg.Matrix.from_name("pmxtestonly77")
    assert b"unknown matrix name" in pkg.lib.pmx_last_error() and b"parasail-rs_amd" in pkg.lib.pmx_last_error()
    monkeypatch.setenv("PMX_MATRIX_DIR", str(tmp_path))
    with pytest.raises(pkg.FailedLookup):
        pkg.Matrix.from_name("pmxtestonly77")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "blosum62.txt"), tmp_path / "pmxtestonly77.txt")
    m = pkg.Matrix.from_name("pmxtestonly77")                 # the miss above was not cached
    assert (m.to_numpy() == pkg.Matrix.from_name("blosum62").to_numpy()).all()


def _write_ncbi(path, alphabet, table):
    with open(path, "w") as fh:
        fh.write("# generated by tests/test_abi.py\n   " + "  ".join(alphabet) + "\n")
        for a, row in zip(alphabet, table):
            fh.write(a + " " + " ".join("%2d" % v for v in row) + "\n")


def test_all_66_documented_matrix_names_through_the_directory(pkg, tmp_path, monkeypatch):
    """src/matrix/mod.rs:46-50 documents 16 BLOSUM and 50 PAM names.  The library lists them (`pmx_documented_matrix_names`); each
    one resolves from $PMX_MATRIX_DIR once its NCBI file is there (generated here: structurally valid tables, a different one per
    name -- the loader and the name table are what is tested, not table values), says "documented name ... no file" while it is
    not, and a damaged file under a documented name is refused with the reason (load-time self-check: symmetric, positive
    diagonal, `*` = the minimum, B / Z / X inside the range of the residues they stand for)."""
    import numpy as np
    buf = C.create_string_buffer(4096)
    need = pkg.lib.pmx_documented_matrix_names(buf, 4096)
    names = buf.value.decode().split()
    assert need <= 4096 and len(names) == 66 and len(set(names)) == 66
    assert names[:3] == ["blosum30", "blosum35", "blosum40"] and "blosum62" in names and names[-1] == "pam500" and "pam10" in names
    monkeypatch.setenv("PMX_MATRIX_DIR", str(tmp_path))
    b62 = pkg.Matrix.from_name("blosum62").to_numpy().astype(np.int64)
    alphabet = list("ARNDCQEGHILKMFPSTWYVBZX*")
    for k, name in enumerate(names):
        if name == "blosum62":
            continue
        with pytest.raises(pkg.FailedLookup):
            pkg.Matrix.from_name(name)
        msg = pkg.lib.pmx_last_error()
        assert b"documented name" in msg and name.encode() in msg, msg
        t = b62.copy()
        t[:23, :23] += (k % 3)                                # another table per name, same structure (symmetric shift of the letters' block)
        t[23, :23] = t[:23, 23] = t[:23, :23].min() - 1       # `*`: the minimum; */* stays 1
        _write_ncbi(tmp_path / (name + (".txt" if k % 2 else "")), alphabet, t)
        m = pkg.Matrix.from_name(name.upper() if k % 5 == 0 else name)
        assert m.size == 24 and (m.to_numpy() == t).all(), name
        with pytest.raises(pkg.NotBuiltIn):
            m.set_value(0, 0, 5)
    for bad in ("blosum63", "pam15", "pam510", "pam0", "blosum", "pam", "blosum062"):
        with pytest.raises(pkg.FailedLookup):
            pkg.Matrix.from_name(bad)
        assert b"unknown matrix name" in pkg.lib.pmx_last_error(), (bad, pkg.lib.pmx_last_error())

    A = {c: i for i, c in enumerate(alphabet)}

    def asym(t): t[A["W"], A["Y"]] += 1
    def diag(t): t[A["S"], A["S"]] = 0
    def star(t): t[A["*"], A["K"]] = t[A["K"], A["*"]] = 0
    def bout(t): t[A["B"], A["N"]] = t[A["N"], A["B"]] = 9
    def xout(t): t[A["X"], A["W"]] = t[A["W"], A["X"]] = 12
    for name, edit, reason in (("blosum45", asym, b"not symmetric"), ("pam250", diag, b"diagonal of S"), ("pam120", star, b"* against K"),
                               ("blosum80", bout, b"B against N"), ("pam30", xout, b"X against W")):
        # force a cache miss: these names were loaded above under this directory, so use a second directory
        d2 = tmp_path / ("second_" + name)
        d2.mkdir()
        monkeypatch.setenv("PMX_MATRIX_DIR", str(d2))
        t = b62.copy(); edit(t)
        _write_ncbi(d2 / name, alphabet, t)
        with pytest.raises(pkg.FailedLookup):
            pkg.Matrix.from_name(name)
        assert reason in pkg.lib.pmx_last_error(), (name, pkg.lib.pmx_last_error())
        # an UNdocumented name is the user's own table: loaded as it is
        _write_ncbi(d2 / "mytable", alphabet, t)
        assert (pkg.Matrix.from_name("mytable").to_numpy() == t).all()
        monkeypatch.setenv("PMX_MATRIX_DIR", str(tmp_path))


def test_embedded_blosum62_against_the_published_description(pkg):
    """(round-3 review: csrc/pmx_matrices.h and tests/golden/blosum62.txt were typed by the same hand.)  An independent check written
    from the literature's description of BLOSUM62 (Henikoff & Henikoff 1992; the NCBI file's layout), not from the table: order
    ARNDCQEGHILKMFPSTWYVBZX*, symmetric, the diagonal 4 5 6 6 9 5 5 6 8 4 4 5 5 6 7 4 5 11 7 4, extremes -4 / 11 (W/W), the
    well-known pairs (I/V 3, F/Y 3, K/R 2, D/E 2, L/M 2, I/L 2, W/Y 2, H/Y 2, N/D 1, S/T 1, Q/E 2, C against W -2), B and Z between
    the residues they stand for, `*` = -4 against every letter and 1 against itself."""
    import numpy as np
    m = pkg.Matrix.from_name("blosum62")
    t = m.to_numpy()
    al = "ARNDCQEGHILKMFPSTWYVBZX*"
    assert m.size == 24 and m.inner.contents.alphabet.decode() == al
    A = {c: i for i, c in enumerate(al)}
    assert (t == t.T).all()
    assert [int(t[A[c], A[c]]) for c in al[:20]] == [4, 5, 6, 6, 9, 5, 5, 6, 8, 4, 4, 5, 5, 6, 7, 4, 5, 11, 7, 4]
    assert t.min() == -4 and t.max() == 11 and t[A["W"], A["W"]] == 11
    known = (("I", "V", 3), ("F", "Y", 3), ("K", "R", 2), ("D", "E", 2), ("L", "M", 2), ("I", "L", 2), ("W", "Y", 2), ("H", "Y", 2),
                    ("N", "D", 1), ("S", "T", 1), ("Q", "E", 2), ("C", "W", -2), ("G", "P", -2), ("A", "S", 1), ("A", "G", 0), ("L", "V", 1),
                    ("M", "V", 1), ("I", "M", 1), ("Q", "K", 1), ("Q", "R", 1), ("E", "K", 1), ("N", "S", 1), ("N", "H", 1), ("W", "F", 1))
    for a, b, v in known:
        assert t[A[a], A[b]] == v, (a, b, int(t[A[a], A[b]]))
    for amb, x, y in (("B", "N", "D"), ("Z", "Q", "E")):
        for c in al[:20]:
            lo, hi = sorted((int(t[A[x], A[c]]), int(t[A[y], A[c]])))
            assert lo <= t[A[amb], A[c]] <= hi, (amb, c)
    assert (t[A["*"], :23] == -4).all() and t[A["*"], A["*"]] == 1
    # every residue scores itself higher than any other residue, and the conservative substitutions listed above are ALL the
    # positive residue pairs of the matrix (21 of the 190)
    blk = t[:20, :20]
    assert all(blk[i, i] > np.delete(blk[i], i).max() for i in range(20))
    positive = {frozenset((al[i], al[j])) for i in range(20) for j in range(i + 1, 20) if blk[i, j] > 0}
    assert positive == {frozenset((a, b)) for a, b, v in known if v > 0} and len(positive) == 21


def test_environment_switches_are_tabled(pkg):
    """The library reads the environment only through pmx_env(), and only names listed in csrc/pmx_switches.h (which
    tests/test_gpu_switches.py sweeps on the GPU): no stray getenv, no unlisted or unused switch."""
    import glob
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parasail-rs_amd", "csrc")
    table = pkg.switches()
    names = {t[0] for t in table}
    assert len(names) == len(table) and all(t[1] in ("force", "value", "path", "diag", "convention") and t[2] for t in table)
    used = set()
    for f in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        src = open(f).read()
        assert not re.search(r"(?<![A-Za-z_])getenv\(", src.replace("return getenv(name);", "")), f
        used |= set(re.findall(r'pmx_env\("(PMX_[A-Z0-9_]+)"\)', src))
    assert used == names, (used - names, names - used)
