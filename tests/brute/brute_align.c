/*
 * brute_align.c -- TEST INFRASTRUCTURE ONLY: an exhaustive checker that contains NO dynamic programming.
 *
 * The oracle (oracle/pmx_oracle.c) is pinned by the reference's known-answer tests only for gaps 0/0 on <= 12-mers
 * (SURVEY.md section 8c).  This file narrows the unpinned area from a direction that shares nothing with any DP code:
 * it ENUMERATES every alignment of two short sequences as a string of column operations
 *      M  one query and one reference character   (score = matrix entry)
 *      I  a query character against a gap         (vertical move,   the F table of a DP)
 *      D  a reference character against a gap     (horizontal move, the E table of a DP)
 * and scores each string directly from the definition of the affine gap model the reference documents
 * (/root/reference/src/aligner/mod.rs:139-153: a gap of k characters costs open + (k - 1) * extend):
 *   nw  every maximal run of I or D is charged.
 *   sg  as nw, except that the FIRST run is free when it is an I run and the query begin is free (S1_BEG) or a D run and the
 *       reference begin is free (S2_BEG), and that a SUFFIX of the LAST run is free when it is an I run and S1_END / a D run and
 *       S2_END -- as long as one character of each sequence is consumed before the free suffix starts (the alignment has to end
 *       on a cell of the table; name grammar and free-end flags: src/aligner/mod.rs:270-331).
 *   sw  the best contiguous piece of any alignment, every run charged, never below 0.
 * The optimum over all strings is the score any correct implementation must report; ties, end cells and CIGARs are not decided
 * here -- tests/test_bruteforce.py checks that what the oracle reports is ONE optimal alignment by re-scoring its CIGAR with
 * brute_score_ops() below.
 */
#include <stdint.h>
#include <string.h>

enum { S1_BEG = 1, S1_END = 2, S2_BEG = 4, S2_END = 8 };
#define NEG (-1000000)

typedef struct {
    const uint8_t *q, *r; int ql, rl;
    const int32_t *matrix; int msize; const int32_t *mapper;
    int open, ext;
    char ops[64]; int sub[64]; int n;
    int best_nw, best_sg[16], best_sw;
} ctx_t;

/* score of ops[a..b) as a global alignment of what it consumes, free-end flags f */
static int score_ops(const char *ops, const int *sub, int n, int open, int ext, int f)
{
    int k = 0, s = 0, cq = 0, cr = 0;
    while (k < n) {
        if (ops[k] == 'M') { s += sub[k]; ++cq; ++cr; ++k; continue; }
        {
            const char t = ops[k];
            int len = 1, freelen = 0;
            while (k + len < n && ops[k + len] == t) ++len;
            if (k == 0 && ((t == 'I' && (f & S1_BEG)) || (t == 'D' && (f & S2_BEG)))) freelen = len;      /* whole first run */
            else if (k + len == n && ((t == 'I' && (f & S1_END)) || (t == 'D' && (f & S2_END)))) {
                /* a suffix of the last run; before it starts one character of each sequence must have been consumed */
                freelen = len;
                if (t == 'I' && cr == 0) freelen = 0;                 /* (no reference character at all: impossible for rl >= 1) */
                if (t == 'D' && cq == 0) freelen = 0;
                if (t == 'I' && cq == 0 && freelen == len) freelen = len - 1;
                if (t == 'D' && cr == 0 && freelen == len) freelen = len - 1;
            }
            if (len - freelen > 0) s -= open + (len - freelen - 1) * ext;
            if (t == 'I') cq += len; else cr += len;
            k += len;
        }
    }
    return s;
}

static void leaf(ctx_t *c)
{
    int f, a, b, s;
    s = score_ops(c->ops, c->sub, c->n, c->open, c->ext, 0);
    if (s > c->best_nw) c->best_nw = s;
    for (f = 0; f < 16; ++f) {
        s = score_ops(c->ops, c->sub, c->n, c->open, c->ext, f);
        if (s > c->best_sg[f]) c->best_sg[f] = s;
    }
    for (a = 0; a < c->n; ++a) {
        if (c->ops[a] != 'M') continue;                               /* a best piece never starts or ends inside a gap */
        for (b = a + 1; b <= c->n; ++b) {
            if (c->ops[b - 1] != 'M') continue;
            s = score_ops(c->ops + a, c->sub + a, b - a, c->open, c->ext, 0);
            if (s > c->best_sw) c->best_sw = s;
        }
    }
}

static void rec(ctx_t *c, int i, int j)
{
    if (i == c->ql && j == c->rl) { leaf(c); return; }
    if (i < c->ql && j < c->rl) {
        c->ops[c->n] = 'M';
        c->sub[c->n] = c->matrix[(size_t)c->msize * c->mapper[c->q[i]] + c->mapper[c->r[j]]];
        ++c->n; rec(c, i + 1, j + 1); --c->n;
    }
    if (i < c->ql) { c->ops[c->n] = 'I'; c->sub[c->n] = 0; ++c->n; rec(c, i + 1, j); --c->n; }
    if (j < c->rl) { c->ops[c->n] = 'D'; c->sub[c->n] = 0; ++c->n; rec(c, i, j + 1); --c->n; }
}

/* out[18] = nw, sg[0..15] (index = free-end flags), sw.  Sequences of at most 16 + 16 characters. */
int brute_optimum(const uint8_t *q, int ql, const uint8_t *r, int rl, int open, int ext,
                  const int32_t *matrix, int msize, const int32_t *mapper, int32_t *out)
{
    ctx_t c; int f;
    if (ql < 1 || rl < 1 || ql + rl > 32) return -1;
    c.q = q; c.r = r; c.ql = ql; c.rl = rl; c.matrix = matrix; c.msize = msize; c.mapper = mapper; c.open = open; c.ext = ext;
    c.n = 0; c.best_nw = NEG; c.best_sw = 0;
    for (f = 0; f < 16; ++f) c.best_sg[f] = NEG;
    rec(&c, 0, 0);
    out[0] = c.best_nw;
    for (f = 0; f < 16; ++f) out[1 + f] = c.best_sg[f];
    out[17] = c.best_sw;
    return 0;
}

int brute_optimum_batch(long n, const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                        int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper, int32_t *out /* n * 18 */)
{
    long k; int bad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(|:bad)
    for (k = 0; k < n; ++k)
        bad |= brute_optimum(qbuf + qoff[k], (int)(qoff[k + 1] - qoff[k]), rbuf + roff[k], (int)(roff[k + 1] - roff[k]),
                             open, ext, matrix, msize, mapper, out + 18 * k) != 0;
    return bad;
}

/* Score of ONE alignment given as a forward ops string over {=, X, I, D} ('I' consumes the query, 'D' the reference -- the
 * caller translates letters), starting at (beg_query, beg_ref); f: free-end flags for a global string (nw: 0), local: 0.
 * Writes the characters consumed; returns NEG on a malformed string or a mislabelled =/X column. */
int brute_score_ops(const char *text, const uint8_t *q, int ql, const uint8_t *r, int rl, int beg_query, int beg_ref,
                    int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper, int f,
                    int *used_q, int *used_r)
{
    char ops[80]; int sub[80]; int n = 0, i = beg_query, j = beg_ref;
    const char *p;
    for (p = text; *p; ++p) {
        if (n >= 80) return NEG;
        if (*p == '=' || *p == 'X') {
            if (i >= ql || j >= rl) return NEG;
            if ((mapper[q[i]] == mapper[r[j]]) != (*p == '=')) return NEG;
            ops[n] = 'M'; sub[n] = matrix[(size_t)msize * mapper[q[i]] + mapper[r[j]]]; ++i; ++j;
        } else if (*p == 'I') { if (i >= ql) return NEG; ops[n] = 'I'; sub[n] = 0; ++i; }
        else if (*p == 'D') { if (j >= rl) return NEG; ops[n] = 'D'; sub[n] = 0; ++j; }
        else return NEG;
        ++n;
    }
    *used_q = i - beg_query; *used_r = j - beg_ref;
    return score_ops(ops, sub, n, open, ext, f);
}
