/*
 * pmx_oracle_batch.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Batch drivers around the scalar oracle of pmx_oracle.c (OpenMP over pairs) for the full-shape parity
 * tests: sampled pairs with statistics, sampled CIGAR text, and an independent re-scoring of CIGAR text
 * (the size-independent property "a CIGAR re-scored with the gap model reproduces the DP score", checked
 * on every pair of a full-size batch).  Same pinning as pmx_oracle.c (see its header).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int score, end_query, end_ref;
    int matches, similar, length;
    int saturated;
} orc_result_t;
typedef struct {
    int32_t *score_table, *matches_table, *similar_table, *length_table;
    int32_t *score_row, *matches_row, *similar_row, *length_row;
    int32_t *score_col, *matches_col, *similar_col, *length_col;
    int8_t *trace_table;
} orc_outputs_t;

int orc_align(int mode, int sg_flags, const uint8_t *q, int qlen, const uint8_t *r, int rlen, int open, int ext,
              const int32_t *matrix, int msize, const int32_t *mapper, int bits, int want_stats,
              orc_result_t *res, orc_outputs_t *out);
int orc_walk(int mode, const int8_t *trace, const uint8_t *q, int qlen, const uint8_t *r, int rlen,
             const int32_t *mapper, int end_query, int end_ref, char *ops_out, int *beg_query, int *beg_ref);
int orc_cigar_text(const char *ops, int n, char *out, int cap);

/* Pairs `index[0..m)` of a packed batch (qoff == NULL: one shared query of qshared bytes at qbuf), with
 * statistics.  out[7*k..]: score, end_query, end_ref, matches, similar, length, saturated. */
int orc_align_stats_sample(int mode, int sg_flags, long m, const int64_t *index,
                           const uint8_t *qbuf, const int64_t *qoff, int qshared,
                           const uint8_t *rbuf, const int64_t *roff,
                           int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                           int bits, int32_t *out)
{
    long k; int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(|:bad)
    for (k = 0; k < m; ++k) {
        const int64_t p = index ? index[k] : k;
        const uint8_t *q = qoff ? qbuf + qoff[p] : qbuf;
        const int qlen = qoff ? (int)(qoff[p + 1] - qoff[p]) : qshared;
        orc_result_t res;
        const int rc = orc_align(mode, sg_flags, q, qlen, rbuf + roff[p], (int)(roff[p + 1] - roff[p]),
                                 open, ext, matrix, msize, mapper, bits, 1, &res, NULL);
        int32_t *o = out + 7 * k;
        if (rc) { bad |= 1; memset(o, 0, 7 * sizeof(int32_t)); continue; }
        o[0] = res.score; o[1] = res.end_query; o[2] = res.end_ref;
        o[3] = res.matches; o[4] = res.similar; o[5] = res.length; o[6] = res.saturated;
    }
    return bad;
}

/* Pairs `index[0..m)`: DP with the byte trace table, walk, run-length text.  text: m slots of `stride`
 * bytes (NUL-terminated); rec[5*k..]: score, end_query, end_ref, beg_query, beg_ref. */
int orc_cigar_sample(int mode, int sg_flags, long m, const int64_t *index,
                     const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                     int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                     char *text, int stride, int32_t *rec)
{
    long k; int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(|:bad)
    for (k = 0; k < m; ++k) {
        const int64_t p = index ? index[k] : k;
        const uint8_t *q = qbuf + qoff[p], *r = rbuf + roff[p];
        const int qlen = (int)(qoff[p + 1] - qoff[p]), rlen = (int)(roff[p + 1] - roff[p]);
        orc_result_t res; orc_outputs_t out;
        int8_t *trace = malloc((size_t)qlen * rlen);
        char *ops = malloc((size_t)qlen + rlen + 2);
        int bq = 0, br = 0, n;
        memset(&out, 0, sizeof out);
        out.trace_table = trace;
        text[(size_t)k * stride] = 0;
        if (!trace || !ops || orc_align(mode, sg_flags, q, qlen, r, rlen, open, ext, matrix, msize, mapper, 0, 0, &res, &out)) {
            bad |= 1; free(trace); free(ops); continue;
        }
        n = orc_walk(mode, trace, q, qlen, r, rlen, mapper, res.end_query, res.end_ref, ops, &bq, &br);
        if (n < 0 || orc_cigar_text(ops, n, text + (size_t)k * stride, stride) < 0) bad |= 1;
        rec[5 * k] = res.score; rec[5 * k + 1] = res.end_query; rec[5 * k + 2] = res.end_ref;
        rec[5 * k + 3] = bq; rec[5 * k + 4] = br;
        free(trace); free(ops);
    }
    return bad;
}

/* Re-score CIGAR text with the gap model (a run of k gap columns costs open + (k-1)*ext): pair k's text is
 * text[toff[k]..toff[k+1]).  'I' consumes the query, 'D' the reference (the oracle's letters), '='/'X' both.
 * The walk starts at (beg[2k], beg[2k+1]) when beg != NULL, else at (0, 0).  free_mask (semi-global free
 * ends: 1 query begin, 2 query end, 4 reference begin, 8 reference end): a first / last run of gap columns on
 * a free side costs nothing (the oracle's walk emits the unaligned ends as gap runs).
 * out[4*k..]: score, query symbols consumed, reference symbols consumed, '=' columns whose symbols differ
 * after mapping + 'X' columns whose symbols agree (must be 0).  Returns the number of malformed texts. */
long orc_rescore_cigars(long n, const char *text, const int64_t *toff,
                        const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                        const int32_t *beg, int open, int ext, int free_mask,
                        const int32_t *matrix, int msize, const int32_t *mapper, int32_t *out)
{
    long k, bad = 0;
#pragma omp parallel for schedule(static) reduction(+:bad)
    for (k = 0; k < n; ++k) {
        const uint8_t *q = qbuf + qoff[k], *r = rbuf + roff[k];
        const int qlen = (int)(qoff[k + 1] - qoff[k]), rlen = (int)(roff[k + 1] - roff[k]);
        int i = beg ? beg[2 * k] : 0, j = beg ? beg[2 * k + 1] : 0;
        const int i0 = i, j0 = j;
        long s = 0; int wrong = 0, malformed = 0, first = 1;
        int64_t p = toff[k];
        while (p < toff[k + 1]) {
            long run = 0; int digits = 0;
            while (p < toff[k + 1] && text[p] >= '0' && text[p] <= '9') { run = run * 10 + (text[p] - '0'); ++p; ++digits; }
            if (!digits || p >= toff[k + 1] || run <= 0) { malformed = 1; break; }
            const char op = text[p++];
            if (op == '=' || op == 'X') {
                if (i + run > qlen || j + run > rlen) { malformed = 1; break; }
                for (long t = 0; t < run; ++t, ++i, ++j) {
                    const int a = mapper[q[i]], b = mapper[r[j]];
                    s += matrix[(size_t)msize * a + b];
                    if ((a == b) != (op == '=')) ++wrong;
                }
            } else if (op == 'I' || op == 'D') {
                const int last = (p == toff[k + 1]);
                const int beg_bit = op == 'I' ? 1 : 4, end_bit = op == 'I' ? 2 : 8;
                const int free_run = (first && (free_mask & beg_bit)) || (last && (free_mask & end_bit));
                if (op == 'I' ? (i + run > qlen) : (j + run > rlen)) { malformed = 1; break; }
                if (!free_run) s -= open + (run - 1) * (long)ext;
                if (op == 'I') i += (int)run; else j += (int)run;
            } else { malformed = 1; break; }
            first = 0;
        }
        out[4 * k] = (int32_t)s; out[4 * k + 1] = i - i0; out[4 * k + 2] = j - j0; out[4 * k + 3] = wrong;
        bad += malformed;
    }
    return bad;
}

int orc_align_ex(int mode, int sg_flags, const uint8_t *q, int qlen, const uint8_t *r, int rlen, int open, int ext,
                 const int32_t *matrix, int msize, const int32_t *mapper, int bits, int want_stats, int band, int diag,
                 orc_result_t *res, orc_outputs_t *out);

/* Banded batch (see orc_align_ex): diag == NULL means the main diagonal for every pair; qoff == NULL: one shared query.
 * out[3*k..]: score, end_query, end_ref. */
int orc_align_banded_batch(int mode, int sg_flags, long n,
                           const uint8_t *qbuf, const int64_t *qoff, int qshared,
                           const uint8_t *rbuf, const int64_t *roff,
                           int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                           int band, const int32_t *diag, int32_t *out)
{
    long k; int bad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(|:bad)
    for (k = 0; k < n; ++k) {
        const uint8_t *q = qoff ? qbuf + qoff[k] : qbuf;
        const int qlen = qoff ? (int)(qoff[k + 1] - qoff[k]) : qshared;
        orc_result_t res;
        const int rc = orc_align_ex(mode, sg_flags, q, qlen, rbuf + roff[k], (int)(roff[k + 1] - roff[k]), open, ext,
                                    matrix, msize, mapper, 0, 0, band, diag ? diag[k] : 0, &res, NULL);
        if (rc) { bad |= 1; out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = 0; continue; }
        out[3 * k] = res.score; out[3 * k + 1] = res.end_query; out[3 * k + 2] = res.end_ref;
    }
    return bad;
}
