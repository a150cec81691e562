// C++ replay of the reference's known-answer tests (/root/reference/tests/test_parasail.rs)
// through the header-only mirror parasail-rs_amd/cpp/parasail_rs.hpp.  Needs a GPU; built by
// __graft_entry__.build(), run by tests/test_gpu_parity.py::test_cpp_mirror.
#include <cassert>
#include <cstdio>
#include <thread>
#include "../../parasail-rs_amd/cpp/parasail_rs.hpp"

using namespace parasail_rs;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    const Bytes q = "ACGT", r = "ACGT";
    fprintf(stderr, "[stage 1]\n");
    {   // global/semi_global/local_alignment (:65-122)
        auto g = Aligner::builder().striped().build().align(&q, r);
        CHECK(g.get_score() == 4 && g.get_end_query() == 3 && g.get_end_ref() == 3 && g.is_global() && !g.is_local() && g.is_striped());
        auto s = Aligner::builder().semi_global().striped().build().align(&q, r);
        CHECK(s.get_score() == 4 && s.is_semi_global() && !s.is_global());
        auto l = Aligner::builder().local().striped().build().align(&q, r);
        CHECK(l.get_score() == 4 && l.get_end_query() == 3 && l.is_local());
    }
    fprintf(stderr, "[stage 2]\n");
    {   // *_with_stats (:125-173)
        auto a = Aligner::builder().local().use_stats().striped().build().align(&q, r);
        CHECK(a.get_matches() == 4 && a.get_length() == 4);
    }
    for (int w : {8, 16, 32, 64}) {   // global_{8,16,32,64}bit (:176-253)
        const Bytes a = "ACTGACTGACTG", b = "ACTGTCTGACTG";
        auto x = Aligner::builder().striped().solution_width(w).build().align(&a, b);
        CHECK(x.get_score() == 11 && x.get_end_query() == 11 && x.get_end_ref() == 11);
    }
    fprintf(stderr, "[stage 3]\n");
    {   // score_table (:256-325)
        auto x = Aligner::builder().use_table().striped().build().align(&q, r);
        CHECK(x.is_table() && !x.is_stats() && !x.is_stats_table());
        auto t = x.get_score_table();
        int v; CHECK(t.rows() == 4 && t.cols() == 4 && t.last() == 4 && t.get(0, 0, &v));
        auto m = Matrix::create("ACGT", 3, -2);
        auto p = Profile::new_(q, true, m);
        auto y = Aligner::builder().profile(std::move(p)).matrix(std::move(m)).use_stats().use_table().striped().build().align(nullptr, r);
        CHECK(y.is_stats() && y.is_stats_table() && y.is_table() && y.get_score_table().last() == 12);
    }
    fprintf(stderr, "[stage 4]\n");
    {   // rows / cols (:386-543)
        const Bytes r3 = "ACG";
        auto x = Aligner::builder().use_last_rowcol().use_stats().striped().build().align(&q, r3);
        CHECK(x.is_stats_rowcol() && x.is_stats() && !x.is_stats_table());
        CHECK((x.get_score_row() == std::vector<int>{1, 2, 3}) && (x.get_length_row() == std::vector<int>{4, 4, 4}));
        auto y = Aligner::builder().use_last_rowcol().use_stats().striped().build().align(&r3, q);
        CHECK((y.get_matches_col() == std::vector<int>{1, 2, 3}) && (y.get_length_col() == std::vector<int>{4, 4, 4}));
    }
    fprintf(stderr, "[stage 5]\n");
    {   // trace (:546-616)
        auto x = Aligner::builder().use_trace().striped().build().align(&q, r);
        auto t = x.get_trace_table();
        CHECK(x.is_trace() && t.rows() == 4 && t.cols() == 4 && t.len() == 16);
        CHECK(x.get_cigar(q, r) == "4=");
        auto tb = x.get_traceback_strings(q, r);
        CHECK(tb.query == "ACGT" && tb.comparison == "||||" && tb.reference == "ACGT");
    }
    fprintf(stderr, "[stage 6]\n");
    {   // multithread_global_alignment (:689-723): shared profile, cloned handle semantics
        auto m = Matrix::default_();
        auto p = Profile::new_(q, true, m);
        auto al = std::make_shared<Aligner>(Aligner::builder().profile(std::move(p)).use_stats().striped().build());
        int s[2] = {0, 0};
        std::thread t0([&] { s[0] = al->align(nullptr, r).get_score(); });
        std::thread t1([&] { s[1] = al->align(nullptr, r).get_score(); });
        t0.join(); t1.join();
        CHECK(s[0] == 4 && s[1] == 4);
    }
    fprintf(stderr, "[stage 7]\n");
    {   // banded / ssw (:726-756)
        CHECK(Aligner::builder().bandwidth(2).build().banded_nw(q, r).get_score() == 4);
        auto w = Aligner::builder().build().ssw(&q, r);
        CHECK(w.score() == 4 && w.query_end() == 3 && w.ref_end() == 3 && w.query_start() == 0 && w.ref_start() == 0);
        bool threw = false;
        try { Aligner::builder().build().banded_nw(q, r); } catch (const Error &e) { threw = e.kind == ErrorKind::NoBandwidth; }
        CHECK(threw);
    }
    fprintf(stderr, "[stage 8]\n");
    {   // error behaviour
        bool panic = false;
        try { Aligner::builder().use_trace().use_last_rowcol().build(); } catch (const Panic &) { panic = true; }
        CHECK(panic);
        bool nul = false;
        try { Bytes bad("AC\0GT", 5); Aligner::builder().build().align(&bad, r); } catch (const Error &e) { nul = e.kind == ErrorKind::InteriorNulByte; }
        CHECK(nul);
    }
    fprintf(stderr, "[stage 9]\n");
    {   // additive batch
        auto al = Aligner::builder().local().matrix(Matrix::create("ACGT", 2, -3)).gap_open(5).gap_extend(2).solution_width(16).build();
        auto out = al.align_batch({"ACGTACGTAC", "TTTT", "ACGT"}, {"ACGTACGTAC", "AAAA", "TACGTT"});
        CHECK(out[0].score == 20 && out[0].end_query == 9 && out[1].score == 0 && out[2].score == 8 && out[2].end_ref == 4);
        std::vector<std::string> cig;
        auto tr = Aligner::builder().global().matrix(Matrix::create("ACGT", 2, -3)).gap_open(5).gap_extend(2).use_trace().build();
        auto rec = tr.align_batch_cigar({"ACGTACGTAC", "ACGT"}, {"ACGTACGTAC", "ACGGT"}, cig);
        CHECK(rec[0].score == 20 && cig[0] == "10=" && cig[1] == "2=1D2=" && rec[1].score == 3);
        auto m = Matrix::create("ACGT", 2, -3);
        auto prof = Profile::new_("ACGTACGTAC", false, m);
        auto pal = Aligner::builder().local().profile(std::move(prof)).matrix(std::move(m)).gap_open(5).gap_extend(2).build();
        auto pr = pal.align_profile_batch({"ACGTACGTAC", "TTTT", "GTACG"});
        CHECK(pr[0].score == 20 && pr[1].score == 2 && pr[2].score == 10);
    }
    printf("cpp mirror ok\n"); fflush(stdout);
    return 0;
}
