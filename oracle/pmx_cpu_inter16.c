/*
 * pmx_cpu_inter16.c -- TEST INFRASTRUCTURE ONLY (CPU timing baselines for the statistics and traceback modes + a second
 * checker).  NOT PART OF THE PRODUCT PATH.
 *
 * BASELINE configs 3 (`nw_stats_striped_profile_16`: one reused query profile, matches / similar / length with the score,
 * /root/reference/src/aligner/mod.rs:431-450, src/alignment/mod.rs:79-98) and 4 (`sg_trace_striped_16` + get_cigar,
 * src/alignment/mod.rs:390-419) had only the SCALAR oracle as their CPU baseline (round-3 review: "a stated baseline that is known
 * to be ~10x low is not a baseline").  These are vectorised CPU restatements of the same recurrences, 16 int16 lanes of AVX2 --
 * INTER-sequence: lane k of a vector works on pair k of a group of 16 (the batch shapes of both configs hold >= 100 000 independent
 * pairs; against parasail's intra-sequence striped layout this form needs no lazy-F pass, so it is at least as fast per core on
 * these batches, and every lane executes the scalar oracle's comparisons in the oracle's order: ties break identically by
 * construction).  Checked bit for bit against pmx_oracle.c in tests/test_oracle.py.  The reference's real kernels live in
 * libparasail-sys 0.2.1 and are not in this image: `kind: "port"` wherever this is timed.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#include "../include/pmx_conventions.h"

#define TGT __attribute__((target("avx2")))
#define L16 16
typedef __m256i V;
#define NEG16 (-30000)

#define ORC_INS 1
#define ORC_DEL 2
#define ORC_DIAG 4
#define ORC_DIAG_E 8
#define ORC_INS_E 16
#define ORC_DIAG_F 32
#define ORC_DEL_F 64
#define ORC_CIGAR_FOR_INS_STATE PMX_CIGAR_LETTER_FOR_INS_STATE
#define ORC_CIGAR_FOR_DEL_STATE PMX_CIGAR_LETTER_FOR_DEL_STATE

int orc_cigar_text(const char *ops, int n, char *out, int cap);

TGT static inline V blend(V a, V b, V m) { return _mm256_blendv_epi8(a, b, m); }          /* m ? b : a */
TGT static inline V vnot(V a) { return _mm256_xor_si256(a, _mm256_set1_epi16(-1)); }

/* ---- config 3: global alignment with statistics, ONE shared query against n references ------------------------------ */
/* out[6 k ..]: score, end_query, end_ref, matches, similar, length.  Returns threads used, < 0 on error. */
TGT int pmx_cpu_nw_stats_inter16(long n, const uint8_t *q, int qlen, const uint8_t *rbuf, const int64_t *roff,
                                 int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                                 int32_t *out, int threads)
{
    int used = 1, fail = 0;
    const long groups = (n + L16 - 1) / L16;
    if (qlen <= 0 || msize > 64) return -1;
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    extern int omp_get_num_threads(void);
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads) reduction(|:fail)
    {
        /* per query row: H, HM, HS, HL of the previous column and E, EM, ES, EL */
        V *st = aligned_alloc(32, sizeof(V) * 8 * (size_t)qlen);
        V *prof = aligned_alloc(32, sizeof(V) * 3 * (size_t)msize);      /* per query symbol: score, equal (0/1), positive (0/1) */
        int *qs = malloc(sizeof(int) * (size_t)qlen);
        long g;
        if (!st || !prof || !qs) fail |= 1;
        else {
            int i;
            for (i = 0; i < qlen; ++i) qs[i] = mapper[q[i]];
#ifdef _OPENMP
#pragma omp single
            used = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 4)
            for (g = 0; g < groups; ++g) {
                const uint8_t *rp[L16]; int rl[L16]; int maxr = 0, k, j, a;
                int16_t tmp[L16] __attribute__((aligned(32)));
                V vrl, resH, resM, resS, resL;
                const V vOpen = _mm256_set1_epi16((int16_t)open), vExt = _mm256_set1_epi16((int16_t)ext), vOne = _mm256_set1_epi16(1);
                for (k = 0; k < L16; ++k) {
                    const long p = g * L16 + k < n ? g * L16 + k : n - 1;
                    rp[k] = rbuf + roff[p]; rl[k] = (int)(roff[p + 1] - roff[p]);
                    if (rl[k] > maxr) maxr = rl[k];
                    tmp[k] = (int16_t)(rl[k] - 1);
                }
                vrl = _mm256_load_si256((const V *)tmp);
                resH = resM = resS = resL = _mm256_setzero_si256();
                for (i = 0; i < qlen; ++i) {                                  /* column -1 */
                    V *s = st + 8 * (size_t)i;
                    s[0] = _mm256_set1_epi16((int16_t)(-(open + i * ext))); s[1] = s[2] = _mm256_setzero_si256();
                    s[3] = _mm256_set1_epi16((int16_t)(i + 1));
                    s[4] = _mm256_set1_epi16(NEG16); s[5] = s[6] = s[7] = _mm256_setzero_si256();
                }
                for (j = 0; j < maxr; ++j) {
                    int16_t sym[L16] __attribute__((aligned(32)));
                    V HN, HNM, HNS, HNL, F, FM, FS, FL, D, DM, DS, DL, vsym;
                    for (k = 0; k < L16; ++k) sym[k] = (int16_t)(j < rl[k] ? mapper[rp[k][j]] : 0);
                    vsym = _mm256_load_si256((const V *)sym);
                    for (a = 0; a < msize; ++a) {                             /* this column's profile: one vector per query symbol */
                        int16_t sc[L16] __attribute__((aligned(32)));
                        V vs;
                        for (k = 0; k < L16; ++k) sc[k] = (int16_t)matrix[(size_t)msize * a + sym[k]];
                        vs = _mm256_load_si256((const V *)sc);
                        prof[3 * a] = vs;
                        prof[3 * a + 1] = _mm256_and_si256(_mm256_cmpeq_epi16(vsym, _mm256_set1_epi16((int16_t)a)), vOne);
                        prof[3 * a + 2] = _mm256_and_si256(_mm256_cmpgt_epi16(vs, _mm256_setzero_si256()), vOne);
                    }
                    HN = _mm256_set1_epi16((int16_t)(-(open + j * ext))); HNM = HNS = _mm256_setzero_si256();
                    HNL = _mm256_set1_epi16((int16_t)(j + 1));
                    F = _mm256_set1_epi16(NEG16); FM = FS = FL = _mm256_setzero_si256();
                    D = j ? _mm256_set1_epi16((int16_t)(-(open + (j - 1) * ext))) : _mm256_setzero_si256();
                    DM = DS = _mm256_setzero_si256(); DL = _mm256_set1_epi16((int16_t)j);
                    for (i = 0; i < qlen; ++i) {
                        V *s = st + 8 * (size_t)i;
                        const V *pr = prof + 3 * qs[i];
                        const V HW = s[0], HWM = s[1], HWS = s[2], HWL = s[3];
                        V E = s[4], EM = s[5], ES = s[6], EL = s[7];
                        V m, a0, a1, Hd, dge, fge, H, HM, HS, HL;
                        a0 = _mm256_subs_epi16(HN, vOpen); a1 = _mm256_subs_epi16(F, vExt); m = _mm256_cmpgt_epi16(a0, a1);
                        F = blend(a1, a0, m); FM = blend(FM, HNM, m); FS = blend(FS, HNS, m); FL = _mm256_add_epi16(blend(FL, HNL, m), vOne);
                        a0 = _mm256_subs_epi16(HW, vOpen); a1 = _mm256_subs_epi16(E, vExt); m = _mm256_cmpgt_epi16(a0, a1);
                        E = blend(a1, a0, m); EM = blend(EM, HWM, m); ES = blend(ES, HWS, m); EL = _mm256_add_epi16(blend(EL, HWL, m), vOne);
                        Hd = _mm256_adds_epi16(D, pr[0]);
                        dge = _mm256_andnot_si256(_mm256_or_si256(_mm256_cmpgt_epi16(E, Hd), _mm256_cmpgt_epi16(F, Hd)), _mm256_set1_epi16(-1));
                        fge = vnot(_mm256_cmpgt_epi16(E, F));
                        H = blend(blend(E, F, fge), Hd, dge);
                        HM = blend(blend(EM, FM, fge), _mm256_add_epi16(DM, pr[1]), dge);
                        HS = blend(blend(ES, FS, fge), _mm256_add_epi16(DS, pr[2]), dge);
                        HL = blend(blend(EL, FL, fge), _mm256_add_epi16(DL, vOne), dge);
                        s[0] = H; s[1] = HM; s[2] = HS; s[3] = HL; s[4] = E; s[5] = EM; s[6] = ES; s[7] = EL;
                        D = HW; DM = HWM; DS = HWS; DL = HWL;
                        HN = H; HNM = HM; HNS = HS; HNL = HL;
                    }
                    {   /* the corner of the lanes whose reference ends here */
                        const V m = _mm256_cmpeq_epi16(vrl, _mm256_set1_epi16((int16_t)j));
                        resH = blend(resH, HN, m); resM = blend(resM, HNM, m); resS = blend(resS, HNS, m); resL = blend(resL, HNL, m);
                    }
                }
                {
                    int16_t h[L16] __attribute__((aligned(32))), mm[L16] __attribute__((aligned(32))), ss[L16] __attribute__((aligned(32))), ll[L16] __attribute__((aligned(32)));
                    _mm256_store_si256((V *)h, resH); _mm256_store_si256((V *)mm, resM); _mm256_store_si256((V *)ss, resS); _mm256_store_si256((V *)ll, resL);
                    for (k = 0; k < L16 && g * L16 + k < n; ++k) {
                        int32_t *o = out + 6 * (g * L16 + k);
                        o[0] = h[k]; o[1] = qlen - 1; o[2] = rl[k] - 1; o[3] = mm[k]; o[4] = ss[k]; o[5] = ll[k];
                    }
                }
            }
        }
        free(st); free(prof); free(qs);
    }
    return fail ? -2 : used;
}

/* ---- config 4: semi-global (all four ends free) or global alignment with the byte trace table, walk and CIGAR text ------------ */
/* Scoring restricted to what the lanes can look up without a gather: score(a, b) = match if a == b, mismatch otherwise, `wild` when
 * either symbol is the matrix's last one (Matrix::create's layout); the entry checks the matrix has that form (returns -3 if not).
 * text: n slots of `stride` bytes (NUL-terminated); rec[5 k ..]: score, end_query, end_ref, beg_query, beg_ref. */
TGT int pmx_cpu_trace_cigar_inter16(int mode /* 0 nw, 1 sg with all ends free */, long n, const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                                    int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                                    char *text, int stride, int32_t *rec, int threads)
{
    int used = 1, fail = 0, a, b;
    const long groups = (n + L16 - 1) / L16;
    const int wsym = msize - 1;
    int match, mismatch, wild;
    if (msize < 3) return -3;
    match = matrix[0]; mismatch = matrix[1]; wild = matrix[(size_t)msize * wsym];
    for (a = 0; a < msize; ++a) for (b = 0; b < msize; ++b) {
        const int want = (a == wsym || b == wsym) ? wild : (a == b ? match : mismatch);
        if (matrix[(size_t)msize * a + b] != want) return -3;
    }
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    extern int omp_get_num_threads(void);
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads) reduction(|:fail)
    {
        long g; size_t cap_cells = 0, cap_rows = 0;
        uint8_t *tr = NULL;          /* [column][row][lane] trace bytes */
        V *st = NULL;                /* per row: H of the previous column, E; then the query symbols of the 16 lanes */
        char *ops = NULL, *rev = NULL;
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 4)
        for (g = 0; g < groups; ++g) {
            const uint8_t *qp[L16], *rp[L16]; int ql[L16], rl[L16]; int maxq = 0, maxr = 0, k, i, j;
            int16_t t16[L16] __attribute__((aligned(32)));
            V vql1, vrl1, rbest, rcol, cbest, crow, corner;
            const V vOpen = _mm256_set1_epi16((int16_t)open), vExt = _mm256_set1_epi16((int16_t)ext);
            const V vMatch = _mm256_set1_epi16((int16_t)match), vMis = _mm256_set1_epi16((int16_t)mismatch), vWild = _mm256_set1_epi16((int16_t)wild);
            const V vW = _mm256_set1_epi16((int16_t)wsym);
            for (k = 0; k < L16; ++k) {
                const long p = g * L16 + k < n ? g * L16 + k : n - 1;
                qp[k] = qbuf + qoff[p]; ql[k] = (int)(qoff[p + 1] - qoff[p]);
                rp[k] = rbuf + roff[p]; rl[k] = (int)(roff[p + 1] - roff[p]);
                if (ql[k] > maxq) maxq = ql[k];
                if (rl[k] > maxr) maxr = rl[k];
            }
            if ((size_t)maxq * maxr > cap_cells || (size_t)maxq > cap_rows) {
                free(tr); free(st); free(ops); free(rev);
                cap_cells = (size_t)maxq * maxr; cap_rows = (size_t)maxq;
                tr = aligned_alloc(32, ((cap_cells * L16 + 31) & ~(size_t)31));
                st = aligned_alloc(32, sizeof(V) * 3 * cap_rows);
                ops = malloc((size_t)maxq + maxr + 2 + 64); rev = malloc((size_t)maxq + maxr + 2 + 64);
                if (!tr || !st || !ops || !rev) { fail |= 1; cap_cells = 0; cap_rows = 0; continue; }
            }
            for (k = 0; k < L16; ++k) t16[k] = (int16_t)(ql[k] - 1);
            vql1 = _mm256_load_si256((const V *)t16);
            for (k = 0; k < L16; ++k) t16[k] = (int16_t)(rl[k] - 1);
            vrl1 = _mm256_load_si256((const V *)t16);
            rbest = cbest = corner = _mm256_set1_epi16(NEG16); rcol = crow = _mm256_setzero_si256();
            for (i = 0; i < maxq; ++i) {
                for (k = 0; k < L16; ++k) t16[k] = (int16_t)(i < ql[k] ? mapper[qp[k][i]] : wsym);
                st[3 * (size_t)i + 2] = _mm256_load_si256((const V *)t16);
                st[3 * (size_t)i] = mode ? _mm256_setzero_si256() : _mm256_set1_epi16((int16_t)(-(open + i * ext)));      /* H(i, -1) */
                st[3 * (size_t)i + 1] = _mm256_set1_epi16(NEG16);                                                           /* E */
            }
            for (j = 0; j < maxr; ++j) {
                V HN, F, D, vr, rw;
                const V vj = _mm256_set1_epi16((int16_t)j);
                const V lastcol = _mm256_cmpeq_epi16(vrl1, vj);
                const V colok = vnot(_mm256_cmpgt_epi16(vj, vrl1));                  /* j <= rlen - 1 */
                for (k = 0; k < L16; ++k) t16[k] = (int16_t)(j < rl[k] ? mapper[rp[k][j]] : wsym);
                vr = _mm256_load_si256((const V *)t16);
                rw = _mm256_cmpeq_epi16(vr, vW);
                HN = mode ? _mm256_setzero_si256() : _mm256_set1_epi16((int16_t)(-(open + j * ext)));
                F = _mm256_set1_epi16(NEG16);
                D = (mode || !j) ? _mm256_setzero_si256() : _mm256_set1_epi16((int16_t)(-(open + (j - 1) * ext)));
                for (i = 0; i < maxq; ++i) {
                    V *s = st + 3 * (size_t)i;
                    const V HW = s[0], vq = s[2];
                    V E = s[1], m, a0, a1, Hd, dge, fge, H, T, sc;
                    const V vi = _mm256_set1_epi16((int16_t)i);
                    a0 = _mm256_subs_epi16(HN, vOpen); a1 = _mm256_subs_epi16(F, vExt); m = _mm256_cmpgt_epi16(a0, a1);
                    F = blend(a1, a0, m);
                    T = blend(_mm256_set1_epi16(ORC_DEL_F), _mm256_set1_epi16(ORC_DIAG_F), m);
                    a0 = _mm256_subs_epi16(HW, vOpen); a1 = _mm256_subs_epi16(E, vExt); m = _mm256_cmpgt_epi16(a0, a1);
                    E = blend(a1, a0, m);
                    T = _mm256_or_si256(T, blend(_mm256_set1_epi16(ORC_INS_E), _mm256_set1_epi16(ORC_DIAG_E), m));
                    sc = blend(vMis, vMatch, _mm256_cmpeq_epi16(vq, vr));
                    sc = blend(sc, vWild, _mm256_or_si256(rw, _mm256_cmpeq_epi16(vq, vW)));
                    Hd = _mm256_adds_epi16(D, sc);
                    dge = _mm256_andnot_si256(_mm256_or_si256(_mm256_cmpgt_epi16(E, Hd), _mm256_cmpgt_epi16(F, Hd)), _mm256_set1_epi16(-1));
                    fge = vnot(_mm256_cmpgt_epi16(E, F));
                    H = blend(blend(E, F, fge), Hd, dge);
                    T = _mm256_or_si256(T, blend(blend(_mm256_set1_epi16(ORC_INS), _mm256_set1_epi16(ORC_DEL), fge), _mm256_set1_epi16(ORC_DIAG), dge));
                    s[0] = H; s[1] = E;
                    _mm_storeu_si128((__m128i *)(tr + ((size_t)j * maxq + i) * L16),
                                     _mm256_castsi256_si128(_mm256_permute4x64_epi64(_mm256_packus_epi16(T, T), 0x08)));
                    D = HW; HN = H;
                    {   /* captures: last row (first maximum by ascending column), last column (smallest row), the corner */
                        const V lastrow = _mm256_cmpeq_epi16(vql1, vi);
                        const V rowok = vnot(_mm256_cmpgt_epi16(vi, vql1));
                        V up = _mm256_and_si256(_mm256_and_si256(lastrow, colok), _mm256_cmpgt_epi16(H, rbest));
                        rbest = blend(rbest, H, up); rcol = blend(rcol, vj, up);
                        up = _mm256_and_si256(_mm256_and_si256(lastcol, rowok), _mm256_cmpgt_epi16(H, cbest));
                        cbest = blend(cbest, H, up); crow = blend(crow, vi, up);
                        corner = blend(corner, H, _mm256_and_si256(lastrow, lastcol));
                    }
                }
            }
            {
                int16_t rb[L16] __attribute__((aligned(32))), rc[L16] __attribute__((aligned(32))), cb[L16] __attribute__((aligned(32))), cr[L16] __attribute__((aligned(32))), co[L16] __attribute__((aligned(32)));
                _mm256_store_si256((V *)rb, rbest); _mm256_store_si256((V *)rc, rcol); _mm256_store_si256((V *)cb, cbest); _mm256_store_si256((V *)cr, crow);
                _mm256_store_si256((V *)co, corner);
                for (k = 0; k < L16 && g * L16 + k < n; ++k) {
                    const long p = g * L16 + k;
                    int score, eq, er, nn = 0, where = ORC_DIAG, x;
                    if (mode == 0) { score = co[k]; eq = ql[k] - 1; er = rl[k] - 1; }
                    else {
                        score = rb[k]; eq = ql[k] - 1; er = rc[k];
                        if (cb[k] > score) { score = cb[k]; eq = cr[k]; er = rl[k] - 1; }
                    }
                    i = eq; j = er;
                    if (mode == 1) {
                        if (eq + 1 == ql[k]) { for (x = rl[k] - 1; x > j; --x) rev[nn++] = ORC_CIGAR_FOR_INS_STATE; }
                        else if (er + 1 == rl[k]) { for (x = ql[k] - 1; x > i; --x) rev[nn++] = ORC_CIGAR_FOR_DEL_STATE; }
                    }
                    while (i >= 0 || j >= 0) {
                        int t;
                        if (i < 0) { rev[nn++] = ORC_CIGAR_FOR_INS_STATE; --j; continue; }
                        if (j < 0) { rev[nn++] = ORC_CIGAR_FOR_DEL_STATE; --i; continue; }
                        t = tr[((size_t)j * maxq + i) * L16 + k];
                        if (where == ORC_DIAG) {
                            if (t & ORC_DIAG) { rev[nn++] = (mapper[qp[k][i]] == mapper[rp[k][j]]) ? '=' : 'X'; --i; --j; }
                            else if (t & ORC_INS) where = ORC_INS;
                            else if (t & ORC_DEL) where = ORC_DEL;
                            else break;
                        } else if (where == ORC_INS) {
                            rev[nn++] = ORC_CIGAR_FOR_INS_STATE;
                            if (t & ORC_DIAG_E) where = ORC_DIAG;
                            --j;
                        } else {
                            rev[nn++] = ORC_CIGAR_FOR_DEL_STATE;
                            if (t & ORC_DIAG_F) where = ORC_DIAG;
                            --i;
                        }
                    }
                    for (x = 0; x < nn; ++x) ops[x] = rev[nn - 1 - x];
                    ops[nn] = 0;
                    text[(size_t)p * stride] = 0;
                    if (orc_cigar_text(ops, nn, text + (size_t)p * stride, stride) < 0) fail |= 2;
                    rec[5 * p] = score; rec[5 * p + 1] = eq; rec[5 * p + 2] = er; rec[5 * p + 3] = i + 1; rec[5 * p + 4] = j + 1;
                }
            }
        }
        free(tr); free(st); free(ops); free(rev);
    }
    return fail ? -2 : used;
}
