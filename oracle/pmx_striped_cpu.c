/*
 * pmx_striped_cpu.c -- TEST INFRASTRUCTURE ONLY (CPU timing baseline + second checker).
 *
 * A CPU restatement of the algorithm class the reference dispatches to for
 * BASELINE config 2, `sw_striped_16` (name grammar src/aligner/mod.rs:319-329;
 * call site src/aligner/mod.rs:411-422): Farrar's striped Smith-Waterman with
 * a lazy-F correction loop, saturating int16 lanes, query profile built per
 * pair exactly as a one-off `Aligner::align(Some(q), r)` would -- or once per
 * batch for the profile arm (`sw_striped_profile_*`, src/aligner/mod.rs:431-450;
 * BASELINE config 5).  Two instantiations of one body (pmx_striped_body.h):
 * 16 lanes (AVX2) and 32 lanes (AVX-512BW), picked at run time from the CPU's
 * feature bits, so the same shared object runs on the build container and on
 * the GPU box's host.
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

/* ---- 16 lanes, AVX2 ---------------------------------------------------------------- */
#define SFX _avx2
#define TGT __attribute__((target("avx2")))
#define LANES 16
#define VEC __m256i
#define V_ZERO() _mm256_setzero_si256()
#define V_SET1(x) _mm256_set1_epi16((int16_t)(x))
#define V_ADDS(a, b) _mm256_adds_epi16(a, b)
#define V_SUBS(a, b) _mm256_subs_epi16(a, b)
#define V_MAX(a, b) _mm256_max_epi16(a, b)
#define V_ANYGT(a, b) (_mm256_movemask_epi8(_mm256_cmpgt_epi16(a, b)) != 0)
TGT static inline __m256i shift_up1_avx2(__m256i v)
{
    /* lane k <- lane k-1, lane 0 <- 0 */
    __m256i t = _mm256_permute2x128_si256(v, v, _MM_SHUFFLE(0, 0, 2, 0));
    return _mm256_alignr_epi8(v, t, 16 - 2);
}
#define V_SHIFTUP(v) shift_up1_avx2(v)
#include "pmx_striped_body.h"
#undef SFX
#undef TGT
#undef LANES
#undef VEC
#undef V_ZERO
#undef V_SET1
#undef V_ADDS
#undef V_SUBS
#undef V_MAX
#undef V_ANYGT
#undef V_SHIFTUP

/* ---- 32 lanes, AVX-512BW ----------------------------------------------------------- */
#define SFX _avx512
#define TGT __attribute__((target("avx512f,avx512bw")))
#define LANES 32
#define VEC __m512i
#define V_ZERO() _mm512_setzero_si512()
#define V_SET1(x) _mm512_set1_epi16((int16_t)(x))
#define V_ADDS(a, b) _mm512_adds_epi16(a, b)
#define V_SUBS(a, b) _mm512_subs_epi16(a, b)
#define V_MAX(a, b) _mm512_max_epi16(a, b)
#define V_ANYGT(a, b) (_mm512_cmpgt_epi16_mask(a, b) != 0)
TGT static inline __m512i shift_up1_avx512(__m512i v)
{
    const __m512i idx = _mm512_set_epi16(30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15,
                                         14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 0);
    return _mm512_maskz_permutexvar_epi16(0xFFFFFFFEu, idx, v);
}
#define V_SHIFTUP(v) shift_up1_avx512(v)
#include "pmx_striped_body.h"

/* 32 or 16: the vector width the batch entries will use on this CPU (0 forces the choice back to auto) */
static int g_force_lanes = 0;
int pmx_cpu_striped_lanes(void)
{
    if (g_force_lanes) return g_force_lanes;
    __builtin_cpu_init();
    return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) ? 32 : 16;
}
void pmx_cpu_striped_force_lanes(int lanes) { g_force_lanes = (lanes == 16 || lanes == 32) ? lanes : 0; }

/* Batch entry with the same packed layout as orc_align_batch / the GPU batch API.
 * qoff == NULL: one shared query of qshared bytes at qbuf whose profile is built once per thread
 * (the profile arm).  threads <= 0 -> use all OpenMP threads.  Returns threads used. */
int pmx_cpu_sw_striped16_batch2(long n, const uint8_t *qbuf, const int64_t *qoff, int qshared,
                                const uint8_t *rbuf, const int64_t *roff,
                                int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                                int32_t *out /* n*3 */, int threads)
{
    if (pmx_cpu_striped_lanes() == 32)
        return sw_batch_avx512(n, qbuf, qoff, qshared, rbuf, roff, open, ext, matrix, msize, mapper, out, threads);
    return sw_batch_avx2(n, qbuf, qoff, qshared, rbuf, roff, open, ext, matrix, msize, mapper, out, threads);
}

int pmx_cpu_sw_striped16_batch(long n, const uint8_t *qbuf, const int64_t *qoff,
                               const uint8_t *rbuf, const int64_t *roff,
                               int open, int ext, const int32_t *matrix, int msize, const int32_t *mapper,
                               int32_t *out /* n*3 */, int threads)
{
    return pmx_cpu_sw_striped16_batch2(n, qbuf, qoff, 0, rbuf, roff, open, ext, matrix, msize, mapper, out, threads);
}
