/*
 * pmx_striped_cpu.c -- TEST INFRASTRUCTURE ONLY (CPU timing baseline + second checker).
 *
 * A CPU restatement of the algorithm class the reference dispatches to for
 * BASELINE config 2, `sw_striped_16` (name grammar src/aligner/mod.rs:319-329;
 * call site src/aligner/mod.rs:411-422): Farrar's striped Smith-Waterman with
 * a lazy-F correction loop, 16 saturating int16 lanes (AVX2), query profile
 * built per pair exactly as a one-off `Aligner::align(Some(q), r)` would.
 * The reference's real kernel is in libparasail-sys 0.2.1 (Cargo.lock:149-158)
 * and is NOT available in this image, so this is labelled a "port" everywhere
 * it is timed (bench.py cpu_baseline.kind = "port").
 *
 * It is checked bit-for-bit (score, end_query, end_ref) against the scalar
 * oracle pmx_oracle.c in tests/test_oracle.py.  End position rule: first
 * maximum in column-major order, as in pmx_oracle.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>

#define LANES 16

static inline __m256i shift_lanes_up1(__m256i v, int16_t fill)
{
    /* lane k <- lane k-1, lane 0 <- fill */
    __m256i t = _mm256_permute2x128_si256(v, v, _MM_SHUFFLE(0, 0, 2, 0));
    __m256i s = _mm256_alignr_epi8(v, t, 16 - 2);
    return _mm256_insert_epi16(s, fill, 0);
}

static inline int16_t hmax16(__m256i v)
{
    __m128i a = _mm_max_epi16(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    a = _mm_max_epi16(a, _mm_srli_si128(a, 8));
    a = _mm_max_epi16(a, _mm_srli_si128(a, 4));
    a = _mm_max_epi16(a, _mm_srli_si128(a, 2));
    return (int16_t)_mm_extract_epi16(a, 0);
}

typedef struct {
    int cap_seg, cap_sym;
    __m256i *profile, *H0, *H1, *E, *Hmax;
} sw_ws_t;

static int ws_reserve(sw_ws_t *ws, int segLen, int msize)
{
    if (segLen <= ws->cap_seg && msize <= ws->cap_sym) return 0;
    free(ws->profile); free(ws->H0);
    if (segLen > ws->cap_seg) ws->cap_seg = segLen + 8;
    if (msize > ws->cap_sym) ws->cap_sym = msize;
    ws->profile = aligned_alloc(32, sizeof(__m256i) * (size_t)ws->cap_seg * ws->cap_sym);
    ws->H0 = aligned_alloc(32, sizeof(__m256i) * (size_t)ws->cap_seg * 4);
    if (!ws->profile || !ws->H0) return -1;
    ws->H1 = ws->H0 + ws->cap_seg; ws->E = ws->H1 + ws->cap_seg; ws->Hmax = ws->E + ws->cap_seg;
    return 0;
}

/* one pair; returns score, fills ends; *sat set if the int16 range was hit */
static int sw_striped16_pair(sw_ws_t *ws, const uint8_t *q, int qlen, const uint8_t *r, int rlen,
                             int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                             int *end_query, int *end_ref, int *sat)
{
    const int segLen = (qlen + LANES - 1) / LANES;
    int a, i, k, j;
    __m256i *pvHLoad, *pvHStore, *pvE, *pvHMax;
    const __m256i vGapO = _mm256_set1_epi16((int16_t)open), vGapE = _mm256_set1_epi16((int16_t)ext);
    const __m256i vZero = _mm256_setzero_si256();
    int score = 0, eref = 0, equery = 0;
    int16_t *prof16;

    if (ws_reserve(ws, segLen, msize)) return 0;
    prof16 = (int16_t *)ws->profile;
    for (a = 0; a < msize; ++a)
        for (i = 0; i < segLen; ++i)
            for (k = 0; k < LANES; ++k) {
                const int idx = i + k * segLen;
                prof16[((size_t)a * segLen + i) * LANES + k] =
                    (idx < qlen) ? (int16_t)matrix[(size_t)msize * mapper[q[idx]] + a] : 0;
            }
    pvHLoad = ws->H0; pvHStore = ws->H1; pvE = ws->E; pvHMax = ws->Hmax;
    for (i = 0; i < segLen; ++i) { pvHStore[i] = vZero; pvE[i] = vZero; pvHMax[i] = vZero; }

    for (j = 0; j < rlen; ++j) {
        const __m256i *vP = ws->profile + (size_t)mapper[r[j]] * segLen;
        __m256i vF = vZero, vColMax = vZero, vH, vE, vHo;
        __m256i *tmp;
        int16_t cm;
        vH = shift_lanes_up1(pvHStore[segLen - 1], 0);
        tmp = pvHLoad; pvHLoad = pvHStore; pvHStore = tmp;
        for (i = 0; i < segLen; ++i) {
            vH = _mm256_adds_epi16(vH, vP[i]);
            vE = pvE[i];
            vH = _mm256_max_epi16(vH, vE);
            vH = _mm256_max_epi16(vH, vF);
            vH = _mm256_max_epi16(vH, vZero);
            pvHStore[i] = vH;
            vHo = _mm256_subs_epi16(vH, vGapO);
            vE = _mm256_max_epi16(_mm256_subs_epi16(vE, vGapE), vHo);
            vF = _mm256_max_epi16(_mm256_subs_epi16(vF, vGapE), vHo);
            pvE[i] = vE;
            vH = pvHLoad[i];
        }
        /* lazy F: carry F across stripe boundaries until it can no longer raise anything */
        for (k = 0; k < LANES; ++k) {
            vF = shift_lanes_up1(vF, 0);
            for (i = 0; i < segLen; ++i) {
                const __m256i vHoOld = _mm256_subs_epi16(pvHStore[i], vGapO);
                vH = _mm256_max_epi16(pvHStore[i], vF);
                pvHStore[i] = vH;
                vHo = _mm256_subs_epi16(vH, vGapO);
                pvE[i] = _mm256_max_epi16(pvE[i], vHo);   /* keep E exact after an F-raised H */
                vF = _mm256_max_epi16(_mm256_subs_epi16(vF, vGapE), vHo);
                /* the first pass already carried (old H - open) downwards: stop once F adds nothing */
                if (!_mm256_movemask_epi8(_mm256_cmpgt_epi16(vF, vHoOld))) goto lazy_done;
            }
        }
lazy_done:
        for (i = 0; i < segLen; ++i) vColMax = _mm256_max_epi16(vColMax, pvHStore[i]);
        cm = hmax16(vColMax);
        if (cm > score) {
            score = cm; eref = j;
            memcpy(pvHMax, pvHStore, sizeof(__m256i) * (size_t)segLen);
        }
    }
    {
        const int16_t *t = (const int16_t *)pvHMax;
        equery = qlen;
        for (i = 0; i < segLen; ++i)
            for (k = 0; k < LANES; ++k) {
                const int idx = i + k * segLen;
                if (idx < qlen && t[i * LANES + k] == score && idx < equery) equery = idx;
            }
        if (equery == qlen) equery = 0;
    }
    *end_query = equery; *end_ref = eref;
    *sat = (score >= INT16_MAX);
    return score;
}

/* Batch entry with the same packed layout as orc_align_batch / the GPU batch API.
 * threads <= 0 -> use all OpenMP threads. Returns threads used. */
int pmx_cpu_sw_striped16_batch(long n, const uint8_t *qbuf, const int64_t *qoff,
                               const uint8_t *rbuf, const int64_t *roff,
                               int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                               int32_t *out /* n*3 */, int threads)
{
    int used = 1;
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    extern int omp_get_num_threads(void);
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        sw_ws_t ws; long k;
        memset(&ws, 0, sizeof ws);
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
#pragma omp for schedule(static)
        for (k = 0; k < n; ++k) {
            int eq = 0, er = 0, sat = 0;
            int s = sw_striped16_pair(&ws, qbuf + qoff[k], (int)(qoff[k + 1] - qoff[k]),
                                      rbuf + roff[k], (int)(roff[k + 1] - roff[k]),
                                      open, ext, matrix, msize, mapper, &eq, &er, &sat);
            out[3 * k] = s; out[3 * k + 1] = eq; out[3 * k + 2] = er;
        }
        free(ws.profile); free(ws.H0);
    }
    return used;
}
