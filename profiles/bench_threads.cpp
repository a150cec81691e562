// Throughput of the reference's own calling pattern through the C ABI: N host threads, each looping a one-pair alignment function
// over its slice of 150 x 150 DNA pairs (/root/reference/tests/test_parasail.rs:702-717: threads sharing one aligner, one align()
// per pair; src/aligner/mod.rs:397-452), immediate and with deferred results (PMX_DEFER_ALIGN=1: all calls first, then all scores),
// beside pmx_align_batch on the same pairs.   g++ -O2 -std=c++17 bench_threads.cpp -L../parasail-rs_amd/lib -lparasail_amd -lpthread
#include "../include/parasail_amd.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 1;
    const long per_thread = argc > 2 ? atol(argv[2]) : 20000;
    const int L = 150;
    const long n = per_thread * threads;
    std::mt19937 rng(12345);
    std::vector<std::string> qs((size_t)n), rs((size_t)n);
    for (long k = 0; k < n; ++k) {
        qs[k].resize(L); rs[k].resize(L);
        for (int i = 0; i < L; ++i) { qs[k][i] = "ACGT"[rng() & 3]; rs[k][i] = (rng() % 10) ? qs[k][i] : "ACGT"[rng() & 3]; }
    }
    parasail_matrix_t *m = parasail_matrix_create("ACGT", 2, -3);
    parasail_function_t *fn = parasail_lookup_function("sw_striped_16");
    if (!m || !fn) { fprintf(stderr, "setup failed\n"); return 1; }
    parasail_result_free(fn(qs[0].data(), L, rs[0].data(), L, 5, 2, m));          // first call: context, matrix upload
    std::vector<int> scores((size_t)n, 0), check((size_t)n, 0);
    auto run = [&](bool deferred) {
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                const long lo = t * per_thread, hi = lo + per_thread;
                if (!deferred) {
                    for (long k = lo; k < hi; ++k) {
                        parasail_result_t *r = fn(qs[k].data(), L, rs[k].data(), L, 5, 2, m);
                        scores[k] = parasail_result_get_score(r);
                        parasail_result_free(r);
                    }
                } else {
                    std::vector<parasail_result_t *> res((size_t)per_thread);
                    for (long k = lo; k < hi; ++k) res[k - lo] = fn(qs[k].data(), L, rs[k].data(), L, 5, 2, m);
                    for (long k = lo; k < hi; ++k) { scores[k] = parasail_result_get_score(res[k - lo]); parasail_result_free(res[k - lo]); }
                }
            });
        for (auto &x : th) x.join();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    unsetenv("PMX_DEFER_ALIGN");
    const long n_imm = std::min<long>(n, 4000L * threads);           // the immediate form is slow: a bounded share of the pairs
    double t_imm;
    {
        const long save = per_thread; (void)save;
        std::vector<std::thread> th;
        const long pt = n_imm / threads;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                for (long k = t * per_thread; k < t * per_thread + pt; ++k) {
                    parasail_result_t *r = fn(qs[k].data(), L, rs[k].data(), L, 5, 2, m);
                    check[k] = parasail_result_get_score(r);
                    parasail_result_free(r);
                }
            });
        for (auto &x : th) x.join();
        t_imm = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    setenv("PMX_DEFER_ALIGN", "1", 1);
    run(true);
    const double t_def = run(true);
    unsetenv("PMX_DEFER_ALIGN");
    // the batch entry on the same pairs (packing by the caller included, and without it)
    std::vector<pmx_record_t> rec((size_t)n);
    std::vector<uint8_t> qb, rb; std::vector<int64_t> qo(1, 0), ro(1, 0);
    pmx_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.mode = PMX_MODE_SW; cfg.open = 5; cfg.extend = 2; cfg.width = 16; cfg.matrix = m;
    auto pack = [&] {
        qb.clear(); rb.clear(); qo.assign(1, 0); ro.assign(1, 0);
        for (long k = 0; k < n; ++k) { qb.insert(qb.end(), qs[k].begin(), qs[k].end()); rb.insert(rb.end(), rs[k].begin(), rs[k].end());
                                       qo.push_back((int64_t)qb.size()); ro.push_back((int64_t)rb.size()); }
    };
    pack();
    pmx_align_batch(&cfg, n, qb.data(), qo.data(), rb.data(), ro.data(), rec.data(), nullptr);
    auto t0 = std::chrono::steady_clock::now();
    pack();
    const double t_pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    t0 = std::chrono::steady_clock::now();
    if (pmx_align_batch(&cfg, n, qb.data(), qo.data(), rb.data(), ro.data(), rec.data(), nullptr)) { fprintf(stderr, "%s\n", pmx_last_error()); return 1; }
    const double t_batch = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long bad = 0;
    for (long k = 0; k < n; ++k) bad += scores[k] != rec[k].score;
    for (int t = 0; t < threads; ++t) for (long k = t * per_thread; k < t * per_thread + n_imm / threads; ++k) bad += check[k] != rec[k].score;
    printf("threads %2d  pairs %8ld  immediate align(): %9.0f pairs/s (%.1f us per call)   deferred: %10.0f pairs/s (%.1f ms)   "
           "pmx_align_batch: %.1f ms (+ %.1f ms packing by the caller)   deferred / batch = %.1fx (%.1fx with packing)   mismatches %ld\n",
           threads, n, n_imm / t_imm, t_imm / (n_imm / threads) * 1e6, n / t_def, t_def * 1e3, t_batch * 1e3, t_pack * 1e3,
           t_def / t_batch, t_def / (t_batch + t_pack), bad);
    parasail_matrix_free(m);
    return bad ? 2 : 0;
}
