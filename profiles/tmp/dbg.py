import sys, os
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_pkg()
pm = pkg.Matrix.create(b"ACGT", 40, -40)
q = b"ACGT" * 250
for w in (16, 0, 32):
    b = pkg.Aligner.new().local().matrix(pm).gap_open(5).gap_extend(2)
    if w: b.solution_width(w)
    al = b.build()
    print(w, al.align_batch([q, b"ACGT"], [q, b"ACGT"]))
    print(w, al.align_batch([q, b"ACGT", q[:900]], [q, b"ACGT", q[:900]]))
