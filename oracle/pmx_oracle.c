/*
 * pmx_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Scalar CPU restatement of the pairwise-alignment arithmetic that
 * nsbuitrago/parasail-rs reaches through `Aligner::align()`
 * (/root/reference/src/aligner/mod.rs:397-452).  The arithmetic itself is not
 * in the reference tree: it lives in the un-vendored crate
 * libparasail-sys 0.2.1 (Cargo.toml:15, Cargo.lock:149-158) which wraps the C
 * library jeffdaily/parasail.  This file restates the published algorithm of
 * that library (affine-gap Gotoh DP; Daily 2016, cited at README.md:74) and
 * is anchored on the reference's own call sites and known-answer tests:
 *
 *   - gap model "open alone is charged when a gap is opened"
 *       -> gap of length k costs open + (k-1)*extend      src/aligner/mod.rs:139-153
 *   - dispatch grammar nw / sg[_q?][_d?] / sw             src/aligner/mod.rs:289-331
 *   - tables are [query_len][ref_len] int32 row-major      src/alignment/table.rs:4-9
 *   - trace table is 1 byte per cell, same layout          src/alignment/mod.rs:291-307
 *   - TraceFlags bit values                                src/alignment/table.rs:127-142
 *   - rows have ref_len entries, cols have query_len       src/alignment/mod.rs:195-288
 *   - end positions are 0-based inclusive                  tests/test_parasail.rs:73-75
 *   - matrix = alphabet + 1 wildcard row/col               src/matrix/mod.rs:222-239, tests/square.txt:7-8
 *
 * PARITY PINNING: pinned by every known-answer assertion in
 * /root/reference/tests/test_parasail.rs (transcribed to
 * tests/golden/reference_kats.json).  Everything those KATs do not exercise
 * (tie-breaking of end positions, stats/trace tie-breaks, CIGAR letters,
 * saturation reporting, wildcard scores of Matrix::create) follows the
 * upstream convention as published, is marked [UNPINNED] below, and is kept in
 * one place each so it can be flipped if a real libparasail becomes available.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <ctype.h>

#define ORC_NEG_INF (INT32_MIN / 2)

enum { ORC_NW = 0, ORC_SG = 1, ORC_SW = 2 };
/* semi-global free-end flags (name grammar src/aligner/mod.rs:270-299):
 *   q = query = s1, d = database/reference = s2; b = begin, e = end. */
enum { ORC_S1_BEG = 1, ORC_S1_END = 2, ORC_S2_BEG = 4, ORC_S2_END = 8 };

/* TraceFlags, src/alignment/table.rs:127-142 */
enum {
    ORC_ZERO = 0, ORC_INS = 1, ORC_DEL = 2, ORC_DIAG = 4,
    ORC_DIAG_E = 8, ORC_INS_E = 16, ORC_DIAG_F = 32, ORC_DEL_F = 64
};

/* [UNPINNED] CIGAR letters.  State INS (E table: horizontal move, consumes a
 * reference character, gap character in the query) and state DEL (F table:
 * vertical move, consumes a query character).  SAM convention with
 * query = s1, reference = s2: consuming only the reference is 'D',
 * consuming only the query is 'I'. */
#include "../include/pmx_conventions.h"
#define ORC_CIGAR_FOR_INS_STATE PMX_CIGAR_LETTER_FOR_INS_STATE
#define ORC_CIGAR_FOR_DEL_STATE PMX_CIGAR_LETTER_FOR_DEL_STATE

typedef struct {
    int score, end_query, end_ref;
    int matches, similar, length;
    int saturated;
} orc_result_t;

typedef struct {
    int32_t *score_table, *matches_table, *similar_table, *length_table; /* [qlen*rlen] or NULL */
    int32_t *score_row, *matches_row, *similar_row, *length_row;         /* [rlen] or NULL */
    int32_t *score_col, *matches_col, *similar_col, *length_col;         /* [qlen] or NULL */
    int8_t  *trace_table;                                                /* [qlen*rlen] or NULL */
} orc_outputs_t;

/* ---- Matrix::create (src/matrix/mod.rs:34-44) -------------------------------
 * size = strlen(alphabet)+1; last row/col is the wildcard for characters outside
 * the alphabet (bound `size-2` at src/matrix/mod.rs:228-236).  Diagonal = match,
 * off-diagonal = mismatch.  [UNPINNED] wildcard row/col scores 0; letters map
 * case-insensitively; a duplicated letter ("ACGTA", src/matrix/mod.rs:248) maps
 * to its LAST position. */
int orc_matrix_create(const char *alphabet, int match, int mismatch,
                      int32_t *matrix_out /* (n+1)*(n+1) */, int32_t *mapper_out /* 256 */)
{
    int n = (int)strlen(alphabet), size = n + 1, i, j, c = 0;
    for (i = 0; i < n; ++i) {
        for (j = 0; j < n; ++j) matrix_out[c++] = (i == j) ? match : mismatch;
        matrix_out[c++] = 0;
    }
    for (j = 0; j < size; ++j) matrix_out[c++] = 0;
    for (i = 0; i < 256; ++i) mapper_out[i] = n;
    for (i = 0; i < n; ++i) {
        mapper_out[toupper((unsigned char)alphabet[i])] = i;
        mapper_out[tolower((unsigned char)alphabet[i])] = i;
    }
    return size;
}

static int width_max(int bits) {
    switch (bits) { case 8: return INT8_MAX; case 16: return INT16_MAX; default: return INT32_MAX; }
}
static int width_min(int bits) {
    switch (bits) { case 8: return INT8_MIN; case 16: return INT16_MIN; default: return INT32_MIN; }
}

/* ---- the DP ------------------------------------------------------------------
 * H[i][j] = max(H[i-1][j-1] + S(q_i, r_j), E[i][j], F[i][j]  [, 0 for sw])
 * E[i][j] = max(E[i][j-1] - extend, H[i][j-1] - open)   (horizontal, along the reference)
 * F[i][j] = max(F[i-1][j] - extend, H[i-1][j] - open)   (vertical, along the query)
 * nw boundary H[i][-1] = -(open + i*extend), H[-1][j] likewise; sg: 0 on each free side.
 *
 * [UNPINNED] tie-breaks (upstream convention):
 *   H source priority  DIAG, then F (DEL), then E (INS); sw: ZERO when H <= 0.
 *   E/F open-vs-extend: "open" only when strictly greater.
 *   sw end            first maximum in column-major order (smallest end_ref, then smallest end_query).
 *   sg end            last row scanned by ascending j (strict >), then last column must be strictly
 *                     greater to win; inside the last column the smallest i wins.
 *   saturation        fixed width w: flagged iff some H (boundary included) leaves [MIN_w, MAX_w];
 *                     "sat" (bits == 0) never saturates below 32 bits.
 */
/* [UNPINNED] band (Aligner::banded_nw -> parasail_nw_banded(..., k, matrix), src/aligner/mod.rs:454-489; the one KAT,
 * tests/test_parasail.rs:726-736, has k = 2 on identical 4-mers): a cell (i, j) (0-based) belongs to the band iff
 * |(j - i) - diag| <= band; outside it H = E = F = -inf (set after the recurrences and the local floor, so a masked cell can
 * neither be entered nor left).  The boundary row / column keep their values.  diag = 0 is the reference's main-diagonal band;
 * a per-pair diag is this repo's extension for banded local alignment around a seed / a first-pass end diagonal
 * (BASELINE config 5 "banded SW"; no reference counterpart).  band < 0: no band. */
int orc_align_ex(int mode, int sg_flags,
                 const uint8_t *q, int qlen, const uint8_t *r, int rlen,
                 int open, int ext,
                 const int32_t *matrix, int msize, const int32_t *mapper,
                 int bits, int want_stats, int band, int diag,
                 orc_result_t *res, orc_outputs_t *out);

int orc_align(int mode, int sg_flags,
              const uint8_t *q, int qlen, const uint8_t *r, int rlen,
              int open, int ext,
              const int32_t *matrix, int msize, const int32_t *mapper,
              int bits, int want_stats,
              orc_result_t *res, orc_outputs_t *out)
{
    return orc_align_ex(mode, sg_flags, q, qlen, r, rlen, open, ext, matrix, msize, mapper, bits, want_stats, -1, 0, res, out);
}

int orc_align_ex(int mode, int sg_flags,
                 const uint8_t *q, int qlen, const uint8_t *r, int rlen,
                 int open, int ext,
                 const int32_t *matrix, int msize, const int32_t *mapper,
                 int bits, int want_stats, int band, int diag,
                 orc_result_t *res, orc_outputs_t *out)
{
    int i, j;
    int32_t *Hp, *HMp, *HSp, *HLp, *F, *FM, *FS, *FL;
    int64_t hmax = 0, hmin = 0;
    int score = ORC_NEG_INF, end_query = 0, end_ref = 0, matches = 0, similar = 0, length = 0;
    const int s1_beg = (mode == ORC_SG) && (sg_flags & ORC_S1_BEG);
    const int s1_end = (mode == ORC_SG) && (sg_flags & ORC_S1_END);
    const int s2_beg = (mode == ORC_SG) && (sg_flags & ORC_S2_BEG);
    const int s2_end = (mode == ORC_SG) && (sg_flags & ORC_S2_END);
    static orc_outputs_t none;
    if (!out) out = &none;
    if (qlen <= 0 || rlen <= 0) return -1;

    int32_t *LC;   /* last column: H, M, S, L per query row */
    Hp = malloc(sizeof(int32_t) * ((size_t)(rlen + 1) * 8 + (size_t)qlen * 4));
    if (!Hp) return -2;
    HMp = Hp + (rlen + 1); HSp = HMp + (rlen + 1); HLp = HSp + (rlen + 1);
    F = HLp + (rlen + 1); FM = F + (rlen + 1); FS = FM + (rlen + 1); FL = FS + (rlen + 1);
    LC = FL + (rlen + 1);

    /* row -1 */
    Hp[0] = 0; HMp[0] = HSp[0] = HLp[0] = 0;
    for (j = 1; j <= rlen; ++j) {
        if (mode == ORC_NW || (mode == ORC_SG && !s2_beg)) { Hp[j] = -(open + (j - 1) * ext); HLp[j] = j; }
        else { Hp[j] = 0; HLp[j] = 0; }
        HMp[j] = HSp[j] = 0;
        if (Hp[j] < hmin) hmin = Hp[j];
    }
    for (j = 0; j <= rlen; ++j) { F[j] = ORC_NEG_INF; FM[j] = FS[j] = FL[j] = 0; }
    if (mode == ORC_SW) score = ORC_NEG_INF;

    for (i = 1; i <= qlen; ++i) {
        const int qi = mapper[q[i - 1]];
        const int32_t *matrow = &matrix[(size_t)msize * qi];
        int NH = Hp[0], NM = HMp[0], NS = HSp[0], NL = HLp[0];
        int WH, WM = 0, WS = 0, WL;
        int E = ORC_NEG_INF, EM = 0, ES = 0, EL = 0;
        if (mode == ORC_NW || (mode == ORC_SG && !s1_beg)) { WH = -(open + (i - 1) * ext); WL = i; }
        else { WH = 0; WL = 0; }
        if (WH < hmin) hmin = WH;
        Hp[0] = WH; HMp[0] = 0; HSp[0] = 0; HLp[0] = WL;
        for (j = 1; j <= rlen; ++j) {
            const int rj = mapper[r[j - 1]];
            const int s = matrow[rj];
            const int NWH = NH, NWM = NM, NWS = NS, NWL = NL;
            int F_opn, F_ext, E_opn, E_ext, H_dag, H, HM, HS, HL;
            int8_t T = 0;
            NH = Hp[j]; NM = HMp[j]; NS = HSp[j]; NL = HLp[j];
            F_opn = NH - open; F_ext = F[j] - ext;
            if (F_opn > F_ext) { F[j] = F_opn; FM[j] = NM; FS[j] = NS; FL[j] = NL + 1; T |= ORC_DIAG_F; }
            else               { F[j] = F_ext; FL[j] = FL[j] + 1; T |= ORC_DEL_F; }
            E_opn = WH - open; E_ext = E - ext;
            if (E_opn > E_ext) { E = E_opn; EM = WM; ES = WS; EL = WL + 1; T |= ORC_DIAG_E; }
            else               { E = E_ext; EL = EL + 1; T |= ORC_INS_E; }
            if (F[j] < ORC_NEG_INF) F[j] = ORC_NEG_INF;
            if (E < ORC_NEG_INF) E = ORC_NEG_INF;
            H_dag = NWH + s;
            if (H_dag >= E && H_dag >= F[j]) {
                H = H_dag; HM = NWM + (qi == rj); HS = NWS + (s > 0); HL = NWL + 1; T |= ORC_DIAG;
            } else if (F[j] >= E) {
                H = F[j]; HM = FM[j]; HS = FS[j]; HL = FL[j]; T |= ORC_DEL;
            } else {
                H = E; HM = EM; HS = ES; HL = EL; T |= ORC_INS;
            }
            if (mode == ORC_SW && H <= 0) {
                H = 0; HM = HS = HL = 0;
                T &= ~(ORC_INS | ORC_DEL | ORC_DIAG);   /* ZERO */
            }
            if (band >= 0) {
                const int dj = (j - 1) - (i - 1) - diag;
                if (dj > band || dj < -band) { H = ORC_NEG_INF; E = ORC_NEG_INF; F[j] = ORC_NEG_INF; HM = HS = HL = 0; }
            }
            if (H > hmax) hmax = H;
            if (H < hmin && band < 0) hmin = H;
            Hp[j] = H; HMp[j] = HM; HSp[j] = HS; HLp[j] = HL;
            WH = H; WM = HM; WS = HS; WL = HL;
            if (out->score_table)   out->score_table[(size_t)(i - 1) * rlen + (j - 1)] = H;
            if (out->matches_table) out->matches_table[(size_t)(i - 1) * rlen + (j - 1)] = HM;
            if (out->similar_table) out->similar_table[(size_t)(i - 1) * rlen + (j - 1)] = HS;
            if (out->length_table)  out->length_table[(size_t)(i - 1) * rlen + (j - 1)] = HL;
            if (out->trace_table)   out->trace_table[(size_t)(i - 1) * rlen + (j - 1)] = T;
            if (mode == ORC_SW) {
                if (H > score || (H == score && (j - 1) < end_ref)) {
                    score = H; end_query = i - 1; end_ref = j - 1;
                    matches = HM; similar = HS; length = HL;
                }
            }
        }
        /* last column */
        LC[4 * (i - 1) + 0] = Hp[rlen]; LC[4 * (i - 1) + 1] = HMp[rlen];
        LC[4 * (i - 1) + 2] = HSp[rlen]; LC[4 * (i - 1) + 3] = HLp[rlen];
        if (out->score_col)   out->score_col[i - 1] = Hp[rlen];
        if (out->matches_col) out->matches_col[i - 1] = HMp[rlen];
        if (out->similar_col) out->similar_col[i - 1] = HSp[rlen];
        if (out->length_col)  out->length_col[i - 1] = HLp[rlen];
    }
    /* last row */
    for (j = 1; j <= rlen; ++j) {
        if (out->score_row)   out->score_row[j - 1] = Hp[j];
        if (out->matches_row) out->matches_row[j - 1] = HMp[j];
        if (out->similar_row) out->similar_row[j - 1] = HSp[j];
        if (out->length_row)  out->length_row[j - 1] = HLp[j];
    }

    if (mode == ORC_NW || (mode == ORC_SG && !s1_end && !s2_end)) {
        score = Hp[rlen]; end_query = qlen - 1; end_ref = rlen - 1;
        matches = HMp[rlen]; similar = HSp[rlen]; length = HLp[rlen];
    } else if (mode == ORC_SG) {
        score = ORC_NEG_INF;
        if (s2_end) {             /* reference end free: any cell of the last row */
            for (j = 1; j <= rlen; ++j) {
                if (Hp[j] > score) {
                    score = Hp[j]; end_query = qlen - 1; end_ref = j - 1;
                    matches = HMp[j]; similar = HSp[j]; length = HLp[j];
                }
            }
        }
        if (s1_end) {             /* query end free: any cell of the last column */
            for (i = 1; i <= qlen; ++i) {
                const int h = LC[4 * (i - 1)];
                if (h > score) {
                    score = h; end_query = i - 1; end_ref = rlen - 1;
                    matches = LC[4 * (i - 1) + 1]; similar = LC[4 * (i - 1) + 2]; length = LC[4 * (i - 1) + 3];
                }
            }
        }
    }

    res->score = score; res->end_query = end_query; res->end_ref = end_ref;
    res->matches = matches; res->similar = similar; res->length = length;
    res->saturated = 0;
    if (bits == 8 || bits == 16) {
        if (hmax > width_max(bits) || hmin < width_min(bits)) res->saturated = 1;
    }
    (void)want_stats;
    free(Hp);
    return 0;
}

/* ---- traceback walk -----------------------------------------------------------
 * Restates what Alignment::get_cigar / get_traceback_strings obtain from
 * parasail_result_get_cigar / parasail_result_get_traceback
 * (src/alignment/mod.rs:347-419).  Walk from (end_query, end_ref):
 *   state DIAG: H flag DIAG -> emit =/X, i--, j--; INS -> state INS; DEL -> state DEL; ZERO -> stop
 *   state INS : emit (gap in query, reference char), leave INS when the E flag is DIAG_E, j--
 *   state DEL : emit (query char, gap in reference), leave DEL when the F flag is DIAG_F, i--
 * nw/sg: once one sequence is exhausted the rest of the other is emitted as gaps.
 * sg: the unaligned tail beyond (end_query, end_ref) is emitted as end gaps.
 * Ops are returned in forward order as a string over {'=','X','I','D'}.
 * [UNPINNED] '=' means equal after mapping through the matrix alphabet.
 */
int orc_walk(int mode, const int8_t *trace, const uint8_t *q, int qlen, const uint8_t *r, int rlen,
             const int32_t *mapper, int end_query, int end_ref,
             char *ops_out /* cap >= qlen+rlen+1 */, int *beg_query, int *beg_ref)
{
    int i = end_query, j = end_ref, n = 0, k;
    int where = ORC_DIAG;
    char *rev = malloc((size_t)qlen + rlen + 2);
    if (!rev) return -1;
    if (mode == ORC_SG) {
        if (end_query + 1 == qlen) { for (k = rlen - 1; k > j; --k) rev[n++] = ORC_CIGAR_FOR_INS_STATE; }
        else if (end_ref + 1 == rlen) { for (k = qlen - 1; k > i; --k) rev[n++] = ORC_CIGAR_FOR_DEL_STATE; }
    }
    while (i >= 0 || j >= 0) {
        if (i < 0) {
            if (mode == ORC_SW) break;
            rev[n++] = ORC_CIGAR_FOR_INS_STATE; --j; continue;
        }
        if (j < 0) {
            if (mode == ORC_SW) break;
            rev[n++] = ORC_CIGAR_FOR_DEL_STATE; --i; continue;
        }
        {
            const int t = trace[(size_t)i * rlen + j];
            if (where == ORC_DIAG) {
                if (t & ORC_DIAG) {
                    rev[n++] = (mapper[q[i]] == mapper[r[j]]) ? '=' : 'X';
                    --i; --j;
                } else if (t & ORC_INS) where = ORC_INS;
                else if (t & ORC_DEL) where = ORC_DEL;
                else break; /* ZERO */
            } else if (where == ORC_INS) {
                rev[n++] = ORC_CIGAR_FOR_INS_STATE;
                if (t & ORC_DIAG_E) where = ORC_DIAG;
                --j;
            } else {
                rev[n++] = ORC_CIGAR_FOR_DEL_STATE;
                if (t & ORC_DIAG_F) where = ORC_DIAG;
                --i;
            }
        }
    }
    *beg_query = i + 1; *beg_ref = j + 1;
    for (k = 0; k < n; ++k) ops_out[k] = rev[n - 1 - k];
    ops_out[n] = 0;
    free(rev);
    return n;
}

/* Run-length text form ("4=1X2D"), the format parasail_cigar_decode returns
 * (src/alignment/mod.rs:410). */
int orc_cigar_text(const char *ops, int n, char *out, int cap)
{
    int k = 0, w = 0;
    while (k < n) {
        int run = 1, len;
        char buf[16];
        while (k + run < n && ops[k + run] == ops[k]) ++run;
        len = 0;
        { int v = run; char tmp[12]; int t = 0; while (v) { tmp[t++] = (char)('0' + v % 10); v /= 10; } while (t) buf[len++] = tmp[--t]; }
        buf[len++] = ops[k];
        if (w + len >= cap) return -1;
        memcpy(out + w, buf, (size_t)len); w += len;
        k += run;
    }
    out[w] = 0;
    return w;
}

/* Gapped strings (query / comparison / reference) from an ops string.
 * match '|' ; mismatch with positive score -> pos_char; otherwise neg_char
 * (src/alignment/mod.rs:353-365 passes '|', ' ', ' '). */
int orc_traceback_strings(const char *ops, int n, const uint8_t *q, const uint8_t *r,
                          int beg_query, int beg_ref,
                          const int32_t *matrix, int msize, const int32_t *mapper,
                          char match_c, char pos_c, char neg_c,
                          char *qs, char *cs, char *rs)
{
    int i = beg_query, j = beg_ref, k;
    for (k = 0; k < n; ++k) {
        const char o = ops[k];
        if (o == '=' || o == 'X') {
            const int s = matrix[(size_t)msize * mapper[q[i]] + mapper[r[j]]];
            qs[k] = (char)q[i]; rs[k] = (char)r[j];
            cs[k] = (o == '=') ? match_c : (s > 0 ? pos_c : neg_c);
            ++i; ++j;
        } else if (o == ORC_CIGAR_FOR_INS_STATE) {
            qs[k] = '-'; rs[k] = (char)r[j]; cs[k] = ' '; ++j;
        } else {
            qs[k] = (char)q[i]; rs[k] = '-'; cs[k] = ' '; ++i;
        }
    }
    qs[n] = cs[n] = rs[n] = 0;
    return n;
}

/* Score / end positions for a packed batch (same inputs as the GPU batch entry).
 * Sequences are concatenated; off[k]..off[k+1] delimits pair k.  Used as the
 * scalar checker and, with OpenMP, as a simple multi-core timing leg. */
int orc_align_batch(int mode, int sg_flags, long n,
                    const uint8_t *qbuf, const int64_t *qoff,
                    const uint8_t *rbuf, const int64_t *roff,
                    int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                    int bits, int32_t *out /* n*3: score,end_query,end_ref */)
{
    long k; int bad = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(|:bad)
    for (k = 0; k < n; ++k) {
        orc_result_t res;
        int rc = orc_align(mode, sg_flags, qbuf + qoff[k], (int)(qoff[k + 1] - qoff[k]),
                           rbuf + roff[k], (int)(roff[k + 1] - roff[k]),
                           open, ext, matrix, msize, mapper, bits, 0, &res, NULL);
        if (rc) { bad |= 1; out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = 0; continue; }
        out[3 * k] = res.score; out[3 * k + 1] = res.end_query; out[3 * k + 2] = res.end_ref;
    }
    return bad;
}
