/* pmx_striped_body.h -- TEST INFRASTRUCTURE ONLY.  Body of the striped int16 Smith-Waterman port, instantiated by
 * pmx_striped_cpu.c once per vector width (SFX, TGT, LANES, VEC and the V_* operations are defined there). */
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define WS_T CAT(sw_ws_t, SFX)

typedef struct {
    int cap_seg, cap_sym;
    int prof_qlen;               /* > 0: `profile` holds the striped profile of a shared query of that length */
    VEC *profile, *H0, *H1, *E, *Hmax;
} WS_T;

TGT static int CAT(ws_reserve, SFX)(WS_T *ws, int segLen, int msize)
{
    if (segLen <= ws->cap_seg && msize <= ws->cap_sym) return 0;
    free(ws->profile); free(ws->H0);
    if (segLen > ws->cap_seg) ws->cap_seg = segLen + 8;
    if (msize > ws->cap_sym) ws->cap_sym = msize;
    ws->profile = aligned_alloc(64, sizeof(VEC) * (size_t)ws->cap_seg * ws->cap_sym);
    ws->H0 = aligned_alloc(64, sizeof(VEC) * (size_t)ws->cap_seg * 4);
    if (!ws->profile || !ws->H0) return -1;
    ws->H1 = ws->H0 + ws->cap_seg; ws->E = ws->H1 + ws->cap_seg; ws->Hmax = ws->E + ws->cap_seg;
    ws->prof_qlen = 0;
    return 0;
}

TGT static void CAT(build_profile, SFX)(WS_T *ws, const uint8_t *q, int qlen, int segLen,
                                        const int32_t *matrix, int msize, const int32_t *mapper)
{
    int16_t *prof16 = (int16_t *)ws->profile;
    int a, i, k;
    for (a = 0; a < msize; ++a)
        for (i = 0; i < segLen; ++i)
            for (k = 0; k < LANES; ++k) {
                const int idx = i + k * segLen;
                prof16[((size_t)a * segLen + i) * LANES + k] =
                    (idx < qlen) ? (int16_t)matrix[(size_t)msize * mapper[q[idx]] + a] : 0;
            }
}

/* one pair; returns score, fills ends; *sat set if the int16 range was hit */
TGT static int CAT(sw_pair, SFX)(WS_T *ws, const uint8_t *q, int qlen, const uint8_t *r, int rlen,
                                 int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                                 int shared, int *end_query, int *end_ref, int *sat)
{
    const int segLen = (qlen + LANES - 1) / LANES;
    int i, k, j;
    VEC *pvHLoad, *pvHStore, *pvE, *pvHMax;
    const VEC vGapO = V_SET1(open), vGapE = V_SET1(ext);
    const VEC vZero = V_ZERO();
    int score = 0, eref = 0, equery = 0;

    if (CAT(ws_reserve, SFX)(ws, segLen, msize)) return 0;
    if (!shared || ws->prof_qlen != qlen) {
        CAT(build_profile, SFX)(ws, q, qlen, segLen, matrix, msize, mapper);
        ws->prof_qlen = shared ? qlen : 0;
    }
    pvHLoad = ws->H0; pvHStore = ws->H1; pvE = ws->E; pvHMax = ws->Hmax;
    for (i = 0; i < segLen; ++i) { pvHStore[i] = vZero; pvE[i] = vZero; pvHMax[i] = vZero; }

    for (j = 0; j < rlen; ++j) {
        const VEC *vP = ws->profile + (size_t)mapper[r[j]] * segLen;
        VEC vF = vZero, vColMax = vZero, vH, vE, vHo;
        VEC *tmp;
        vH = V_SHIFTUP(pvHStore[segLen - 1]);
        tmp = pvHLoad; pvHLoad = pvHStore; pvHStore = tmp;
        for (i = 0; i < segLen; ++i) {
            vH = V_ADDS(vH, vP[i]);
            vE = pvE[i];
            vH = V_MAX(vH, vE);
            vH = V_MAX(vH, vF);
            vH = V_MAX(vH, vZero);
            pvHStore[i] = vH;
            vColMax = V_MAX(vColMax, vH);
            vHo = V_SUBS(vH, vGapO);
            vE = V_MAX(V_SUBS(vE, vGapE), vHo);
            vF = V_MAX(V_SUBS(vF, vGapE), vHo);
            pvE[i] = vE;
            vH = pvHLoad[i];
        }
        /* lazy F: carry F across stripe boundaries until it can no longer raise anything */
        for (k = 0; k < LANES; ++k) {
            vF = V_SHIFTUP(vF);
            for (i = 0; i < segLen; ++i) {
                const VEC vHoOld = V_SUBS(pvHStore[i], vGapO);
                vH = V_MAX(pvHStore[i], vF);
                pvHStore[i] = vH;
                vColMax = V_MAX(vColMax, vH);
                vHo = V_SUBS(vH, vGapO);
                pvE[i] = V_MAX(pvE[i], vHo);   /* keep E exact after an F-raised H */
                vF = V_MAX(V_SUBS(vF, vGapE), vHo);
                /* the first pass already carried (old H - open) downwards: stop once F adds nothing */
                if (!V_ANYGT(vF, vHoOld)) goto lazy_done;
            }
        }
lazy_done:
        if (V_ANYGT(vColMax, V_SET1(score))) {
            int16_t lanes[LANES] __attribute__((aligned(64)));
            int16_t cm = 0;
            memcpy(lanes, &vColMax, sizeof lanes);
            for (k = 0; k < LANES; ++k) if (lanes[k] > cm) cm = lanes[k];
            score = cm; eref = j;
            memcpy(pvHMax, pvHStore, sizeof(VEC) * (size_t)segLen);
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

TGT static int CAT(sw_batch, SFX)(long n, const uint8_t *qbuf, const int64_t *qoff, int qshared,
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
        WS_T ws; long k;
        memset(&ws, 0, sizeof ws);
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 64)
        for (k = 0; k < n; ++k) {
            int eq = 0, er = 0, sat = 0;
            const uint8_t *q = qoff ? qbuf + qoff[k] : qbuf;
            const int qlen = qoff ? (int)(qoff[k + 1] - qoff[k]) : qshared;
            int s = CAT(sw_pair, SFX)(&ws, q, qlen, rbuf + roff[k], (int)(roff[k + 1] - roff[k]),
                                      open, ext, matrix, msize, mapper, qoff == NULL, &eq, &er, &sat);
            out[3 * k] = s; out[3 * k + 1] = eq; out[3 * k + 2] = er;
        }
        free(ws.profile); free(ws.H0);
    }
    return used;
}
#undef WS_T
#undef CAT
#undef CAT2
