/* nwsgv_model.c -- CPU model of the STORED arithmetic of the packed global / semi-global kernels (test infrastructure only).
 *
 * parasail-rs_amd/csrc/pmx_nwsg16.hip keeps every value of the DP as a 16-bit pattern that must stay inside [1024, 31743] (the
 * range on which v_pk_maximum3_f16 orders bit patterns like integers) after a bias nb, a column skew (+ ext per column) and, in the
 * row-offset form, + ext per row of the shape.  No promotion pass exists behind these kernels: the host predicate
 * (pmx_nwsgv_bias) must PROVE the window from lengths and scoring alone.  This file replays one pair lane for lane and step for step
 * the way pmx_nwsg16v_kernel does -- virtual rows above (or padding rows below) the query, virtual columns left and right of the
 * reference, the closed-form row above lane 0, the hand-offs, the captures with their own bias -- in plain ints, and records every
 * value that the kernel would feed to a max3 / a 16-bit compare / a byte of the profile: its smallest and largest, and how many
 * left their domain.  tests/test_window_models.py drives it at the corners of the predicate: the result must equal the oracle's
 * and nothing may leave its domain whenever the predicate admits the batch.
 *
 * Follows (does not copy from) the reference semantics of /root/reference/src/aligner/mod.rs:397-452 via the oracle's rules.
 */
#include <stdlib.h>
#include <string.h>

typedef struct {
    int score, end_query, end_ref;
    int lo, hi;                 /* extreme patterns fed to max3 */
    int violations;             /* values outside [1024, 31743], profile bytes outside [0, 255], capture values outside [0, 32767] */
    int first_violation_kind;   /* 1 max3 operand, 2 profile byte, 3 last-row capture, 4 last-column capture */
} nwsgv_model_out;

#define WLO 1024
#define WHI 31743

/* every operand of a max3 goes through here: the extremes are kept in locals of the caller (vlo, vhi) and compared with the window once
 * per lane and step */
#define MAX3C(dst, a, b, c) do { const int a_ = (a), b_ = (b), c_ = (c); \
        int lo3 = a_ < b_ ? a_ : b_; lo3 = lo3 < c_ ? lo3 : c_; int hi3 = a_ > b_ ? a_ : b_; hi3 = hi3 > c_ ? hi3 : c_; \
        vlo = vlo < lo3 ? vlo : lo3; vhi = vhi > hi3 ? vhi : hi3; (dst) = hi3; } while (0)

/* q, r: mapped symbols (0 .. msize-1).  top_aligned: the perm-table form (query in rows 0 .. qlen-1, padding rows below score
 * -open); otherwise the query sits in the LAST qlen rows of the G*R and P = G*R - qlen virtual rows lie above it.
 * legacy_capture: the last-column capture form before the round-3 fix (values + cb - row offset): kept so that the test can show
 * the model flags it.  Returns 0, or -1 on bad arguments. */
int nwsgv_model(int G, int R, int rowx, int top_aligned, int legacy_capture,
                const unsigned char *q, int qlen, const unsigned char *r, int rlen, int max_rlen,
                const int *mat, int msize, int open, int ext, int col_pen, int row_pen, int s1_end, int s2_end, int nb,
                nwsgv_model_out *out)
{
    const int QP = G * R;
    if (qlen < 1 || qlen > QP || rlen < 1 || rlen > max_rlen || G < 1 || R < 1) return -1;
    nwsgv_model_out *o = out;
    memset(o, 0, sizeof *o);
    o->lo = 1 << 30; o->hi = -(1 << 30);
    const int P = top_aligned ? 0 : QP - qlen;
    const int rx = rowx ? ext : 0;
    const int vrow_b = (row_pen ? 0 : open) + rx, vcol_b = (col_pen ? 0 : open) + rx;

    /* profile bytes [er][sym], sym == msize: the pad symbol (virtual / padding column) */
    unsigned char *prof = (unsigned char *)malloc((size_t)QP * (msize + 1));
    for (int er = 0; er < QP; ++er) {
        for (int sym = 0; sym <= msize; ++sym) {
            int v;
            if (top_aligned) {
                if (er < qlen) v = sym < msize ? mat[q[er] * msize + sym] + open + rx : vcol_b;
                else v = 0;                                     /* selector constant 0x0C: byte 0 */
            } else if (er >= P) v = sym < msize ? mat[q[er - P] * msize + sym] + open + rx : vcol_b;
            else v = sym < msize ? vrow_b : open + rx;
            if (v < 0 || v > 255) { if (!o->violations) o->first_violation_kind = 2; o->violations++; }
            prof[(size_t)er * (msize + 1) + sym] = (unsigned char)v;
        }
    }

    /* transposed for the sweep: [sym][er] */
    unsigned char *profT = (unsigned char *)malloc((size_t)QP * (msize + 1));
    for (int er = 0; er < QP; ++er) for (int sym = 0; sym <= msize; ++sym) profT[(size_t)sym * QP + er] = prof[(size_t)er * (msize + 1) + sym];
    const int fsub = rowx ? 0 : ext;
    int *H = (int *)malloc(sizeof(int) * QP), *E = (int *)malloc(sizeof(int) * QP), *Hn = (int *)malloc(sizeof(int) * R);
    int *Hout = (int *)malloc(sizeof(int) * G), *Fout = (int *)malloc(sizeof(int) * G), *diag0 = (int *)malloc(sizeof(int) * G);
    int *Hin = (int *)malloc(sizeof(int) * G), *Fin = (int *)malloc(sizeof(int) * G);
    int *skewX = (int *)malloc(sizeof(int) * G), *jj = (int *)malloc(sizeof(int) * G), *res = (int *)calloc(G, sizeof(int));
    int *bestrow = (int *)calloc(G, sizeof(int)), *bestrowj = (int *)calloc(G, sizeof(int));
    int *bestcol = (int *)calloc(G, sizeof(int)), *bestcoli = (int *)calloc(G, sizeof(int));

#define LEFT_H(erow) ({ int i_ = (erow) - P; if (top_aligned && i_ > qlen - 1) i_ = qlen - 1; (i_ >= 0 && col_pen) ? -(open + i_ * ext) : 0; })
#define BELOW_F(erow) ({ int i_ = (erow) - P; if (top_aligned && i_ > qlen - 1) i_ = qlen - 1; (i_ >= 0 && col_pen) ? -(open + i_ * ext) : -open; })
    const int roL = ((top_aligned ? qlen : QP) - 1) * rx;
    const int cb = rowx ? 4 * open + (QP + max_rlen + 2) * ext : 0;
    const int gs = top_aligned ? (qlen - 1) / R : G - 1, ks = top_aligned ? (qlen - 1) % R : R - 1;
    for (int g = 0; g < G; ++g) {
        const int base = nb + (G - g) * ext - open;
        for (int k = 0; k < R; ++k) {
            const int er = g * R + k;
            H[er] = base + LEFT_H(er) + er * rx;
            E[er] = H[er];
        }
        Hout[g] = H[g * R + R - 1];
        const int roF = ((g + 1) * R - 1) * rx, roD = (g * R - 1) * rx;
        Fout[g] = base + open + BELOW_F((g + 1) * R) + roF;
        diag0[g] = g == 0 ? base + roD : base + LEFT_H(g * R - 1) + roD;
        skewX[g] = (G - g + 1) * ext - open + roL - cb;
        jj[g] = -g;
    }
    int topX = row_pen ? nb + (G + 1) * ext - 2 * open - rx : nb + (G + 1) * ext - open - rx;
    const int topStep = row_pen ? 0 : ext;
    const int T = (max_rlen + G - 1 + 1) & ~1;

    for (int t = 0; t < T; ++t) {
        for (int g = 0; g < G; ++g) { Hin[g] = g ? Hout[g - 1] : topX; Fin[g] = g ? Fout[g - 1] : topX; }
        for (int g = 0; g < G; ++g) {
            const int col = t - g;
            const int sym = (col >= 0 && col < rlen) ? r[col] : msize;
            int F = Fin[g];
            int vlo = 1 << 30, vhi = -(1 << 30);
            const unsigned char *ps = profT + (size_t)sym * QP + g * R;
            int *Hg = H + g * R, *Eg = E + g * R;
            int dprev = diag0[g];
            for (int k = 0; k < R; ++k) {
                const int Tpk = dprev + ps[k];
                dprev = Hg[k];
                const int Fe = F - fsub;
                int Hh; MAX3C(Hh, Tpk, Eg[k], Fe);
                const int X = Hh - (open - ext);
                MAX3C(Eg[k], Eg[k], X, X);
                MAX3C(F, Fe, X, X);
                Hn[k] = X;
                Hg[k] = X;
            }
            if (vlo < o->lo) o->lo = vlo;
            if (vhi > o->hi) o->hi = vhi;
            if (vlo < WLO || vhi > WHI) { if (!o->violations) o->first_violation_kind = 1; o->violations++; }
            diag0[g] = Hin[g]; Hout[g] = Hn[R - 1]; Fout[g] = F;
            /* captures */
            const int j16 = jj[g] & 0xFFFF;
            const int mLast = j16 == ((rlen - 1) & 0xFFFF);
            const int Hlast = top_aligned ? Hn[ks] : Hout[g];
            if (mLast) res[g] = Hlast;
            if (s2_end) {
                const int inside = (unsigned)j16 < (unsigned)rlen;
                const int cand = Hlast - skewX[g];                          /* nb + cb + true H */
                if (inside && g == gs) {                                   /* (only the lane that holds the last row is read at the end) */
                    if (cand < 0 || cand > 32767) { if (!o->violations) o->first_violation_kind = 3; o->violations++; }
                }
                const int c16 = (int)(short)(cand & 0xFFFF);
                if (inside && bestrow[g] < c16) { bestrow[g] = c16; bestrowj[g] = j16; }
            }
            if (s1_end && mLast) {
                int cm = 0, krow = 0;
                for (int k = 0; k < R; ++k) {
                    const int er = g * R + k;
                    const int real = top_aligned ? er < qlen : er >= P;
                    int v = 0;
                    if (real) {
                        v = legacy_capture ? Hn[k] + cb - er * rx : Hn[k] + (QP - er) * rx;
                        if (v < WLO || v > WHI) { if (!o->violations) o->first_violation_kind = 4; o->violations++; }
                        v = (int)(short)(v & 0xFFFF);
                    }
                    Hn[k] = v;                                              /* (reused as vals[]) */
                    if (v > cm) cm = v;
                }
                for (int k = R - 1; k >= 0; --k) if (Hn[k] == cm) krow = g * R + k;
                if (bestcol[g] < cm) { bestcol[g] = cm; bestcoli[g] = krow; }
            }
            jj[g] += 1;
            skewX[g] += ext;
        }
        topX += topStep;
    }

    /* combine (lane 0 of the group) */
    const int unsk = nb + (rlen - 1 + G) * ext - open + ext;
    const int corner = res[gs] - unsk - roL;
    if (!s1_end && !s2_end) { o->score = corner; o->end_query = qlen - 1; o->end_ref = rlen - 1; }
    else {
        long long best = -(1LL << 40); int ei = 0, ej = 0;
        if (s2_end) { best = bestrow[gs] - nb - cb; ei = qlen - 1; ej = bestrowj[gs]; }
        if (s1_end) {
            unsigned key = 0;
            for (int g = 0; g < G; ++g) {
                const unsigned kk = ((unsigned)(bestcol[g] & 0xFFFF) << 16) | (0xFFFFu - (unsigned)(bestcoli[g] & 0xFFFF));
                if (kk > key) key = kk;
            }
            const int cv = (int)(key >> 16) - unsk - (legacy_capture ? cb : QP * rx);
            if (cv > best) { best = cv; ei = (int)(0xFFFFu - (key & 0xFFFFu)) - P; ej = rlen - 1; }
        }
        o->score = (int)best; o->end_query = ei; o->end_ref = ej;
    }
    free(prof); free(profT); free(H); free(E); free(Hn); free(Hout); free(Fout); free(diag0); free(Hin); free(Fin);
    free(skewX); free(jj); free(res); free(bestrow); free(bestrowj); free(bestcol); free(bestcoli);
    return 0;
}
